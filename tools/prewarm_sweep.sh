#!/bin/bash
# fmx_set_option("table_prewarm", n): a fresh process's 20-step call (the driver's round-end run) and the 2,000-step loop
set -o pipefail
run() { for i in 1 2 3; do echo -n "$1 steps=$2 : "; env "$1" timeout -k 10 120 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --no-secondary --loop-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %.1f M/s' % (d['ms_per_step']*1e3, d['value']/1e6))" || return 1; done; }
for n in 0 128 256 512 2048; do run FMX_TABLE_PREWARM=$n 20 5 || exit 1; done
for n in 0 256 2048; do run FMX_TABLE_PREWARM=$n 2000 100 || exit 1; done
