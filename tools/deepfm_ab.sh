#!/bin/bash
# DeepFM step (bench.py --workload deepfm) under the weight-gradient launch's switches and with an earlier library, on ONE box
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python bench.py --workload deepfm --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f M/s %.1f us/step' % (d['value']/1e6, d['ms_per_step']*1e3))"; }
for rep in 1 2; do
run FMX_LIB_PATH=$(pwd)/tools/micro/libfmx_old.so
run A=1
run FMX_WGRAD_REDUCE=0
run FMX_WGRAD_REDUCE=0 FMX_WGRAD_WGS=512
run FMX_WGRAD_WGS=512
run FMX_WGRAD_WGS=256
done
