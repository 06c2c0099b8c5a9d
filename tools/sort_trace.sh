#!/bin/bash
# rocprofv3 kernel trace of tools/sort_times.py -> gpurun_out/<tag>_sorttrace_kernel_stats.csv
set -e -o pipefail
tag=${1:-run}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_sorttrace -o s -- python3 $root/tools/sort_times.py > $out/${tag}_sorttrace.log 2>&1
cd $root
find $out/${tag}_sorttrace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_sorttrace_kernel_stats.csv \;
cat $out/${tag}_sorttrace_kernel_stats.csv | cut -c1-200
