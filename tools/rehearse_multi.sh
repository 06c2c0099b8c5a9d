#!/bin/bash
# Rehearse the N > 1 bench path on a ONE-GPU box: every rank on GPU 0, collectives over gloo (host staged).  The numbers
# mean nothing; what is checked is that the path runs and prints its JSON line.
set -o pipefail
for mode in owner replicated; do
  for n in 2 4; do
    echo "== $mode x$n"
    FMX_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 10 --warmup 3 --mp-mode $mode 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], '%.1f M/s' % (d['value']/1e6), '%.1f us/step' % (d['ms_per_step']*1e3), d['final_loss'], d['config']['parallelism'][:60])" || exit 1
  done
done
