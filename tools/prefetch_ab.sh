#!/bin/bash
# row prefetch from the sort stage on / off: fresh-process 20-step runs (the driver's protocol) and the steady state
run20() { for i in 1 2 3; do echo -n "$* 20-step: "; env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %.1f M/s' % (d['ms_per_step']*1e3, d['value']/1e6))"; done; }
runl() { echo -n "$* 2000-step loop: "; env "$@" timeout -k 10 120 python bench.py --steps 2000 --warmup 100 --loop-only --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %.1f M/s' % (d['ms_per_step']*1e3, d['value']/1e6))"; }
run20 FMX_SORT_PREFETCH=1
run20 FMX_SORT_PREFETCH=0
runl FMX_SORT_PREFETCH=1
runl FMX_SORT_PREFETCH=0
