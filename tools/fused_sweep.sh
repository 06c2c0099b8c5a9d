#!/bin/bash
# loop time of the fused step launch over its knobs (one process per setting: the knobs are read once)
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --steps 1000 --warmup 50 --loop-only --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
run FMX_FUSED_STEP=1
run FMX_FUSED_STEP=1 FMX_FUSED_DEBUG=1
run FMX_FUSED_STEP=1 FMX_FUSED_DEBUG=2
run FMX_FUSED_STEP=1 FMX_FUSED_DEBUG=4
run FMX_FUSED_STEP=1 FMX_FUSED_DEBUG=5
run FMX_FUSED_STEP=2 FMX_FUSED_DEBUG=4
run FMX_FUSED_STEP=2 FMX_FUSED_DEBUG=5
