#!/bin/bash
# what a fresh process's 20-step call costs under a few runtime settings (the driver's round-end run is --steps 20 --warmup 5)
run() { for i in 1 2 3; do echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %.1f M/s' % (d['ms_per_step']*1e3, d['value']/1e6))"; done; }
run A=1
run HSA_ENABLE_INTERRUPT=0
run FMX_BENCH_PROBE_FIRST=1
run HSA_ENABLE_INTERRUPT=0 FMX_BENCH_PROBE_FIRST=1
run GPU_MAX_HW_QUEUES=4
