#!/bin/bash
# Same-box A/B of the online loop under environment settings: steady state (2,000 steps, loop only) and three fresh 20-step runs each.
#   bash tools/ab_env.sh "FMX_X=0" "FMX_X=1" ...
for setting in "$@"; do
  for rep in 1 2; do
    env $setting python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary --loop-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$setting steady %.2f us/step' % (d['ms_per_step']*1e3))"
  done
  for rep in 1 2 3; do
    env $setting python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --loop-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$setting fresh-20 %.2f us/step' % (d['ms_per_step']*1e3))"
  done
done
