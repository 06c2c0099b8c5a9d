#!/bin/bash
# per-kernel durations of fmx_mlp_section (tools/mlp_section_times.py under rocprofv3 --kernel-trace --stats); extra env in "$@"
set -o pipefail
tag=${1:-mlp}; shift
root=$(pwd); out=$root/gpurun_out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_mlp -o m -- python3 $root/tools/mlp_section_times.py > $out/${tag}_mlp.log 2>&1
cd $root
f=$(find $out/${tag}_mlp -name "m_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_mlp" in r["Name"]:
        print("%-70s calls %5s  avg %8.2f us  min %8.2f  max %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
