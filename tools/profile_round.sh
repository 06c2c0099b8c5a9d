#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile_round.sh <tag>   (e.g. r01_f)
# 1. bench.py as the driver runs it -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command (cpu baseline skipped; --steps 50: the measuring pass's 400 steps
#    dominate the trace) and of the online loop alone (--loop-only: the in-loop durations, every 8th step sharing the chip
#    with the next group's sort)
# 3. two --pmc passes (FETCH_SIZE, WRITE_SIZE), each with --kernel-trace only
# tools/summarise_profiles.py then writes profiles/<tag>_* from gpurun_out/.
set -e -o pipefail
tag=${1:-run}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
timeout -k 10 300 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $out/${tag}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_loop -o l -- python3 $root/bench.py --steps 2000 --warmup 100 --no-cpu-baseline --loop-only --no-secondary > $out/${tag}_loop.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o f -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $out/${tag}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -o w -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $out/${tag}_write.log 2>&1
cd $root
python3 tools/summarise_profiles.py $tag
