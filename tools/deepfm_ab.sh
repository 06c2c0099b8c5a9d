#!/bin/bash
# DeepFM step (bench.py --workload deepfm) with the current library and with another build (FMX_LIB_PATH, e.g. an earlier
# commit's library built by hand into tools/micro/), alternating, on ONE box
other=${1:-tools/micro/libfmx_old.so}
run() { echo -n "${1:-current} : "; FMX_LIB_PATH=${1:+$(pwd)/$1} timeout -k 10 200 python bench.py --workload deepfm --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f M/s %.1f us/step' % (d['value']/1e6, d['ms_per_step']*1e3))"; }
for rep in 1 2 3; do
  [ -f "$other" ] && run "$other"
  run ""
done
