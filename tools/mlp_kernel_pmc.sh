#!/bin/bash
# counters of the fmx_mlp_section kernels (tools/mlp_section_times.py): one rocprofv3 --pmc pass per counter group
set -o pipefail
tag=${1:-mlp}
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${tag}_pmc_$name -o p -- python3 $root/tools/mlp_section_times.py > $out/${tag}_pmc_$name.log 2>&1 || echo "pass $grp failed"
  f=$(find $out/${tag}_pmc_$name -name "p_counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        m = re.search(r"(k_mlp_\w+)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no counters:", e)
for k, d in acc.items():
    for c, v in d.items():
        print("%-20s %-30s launches %5d  mean %14.1f" % (k, c, len(v), sum(v) / len(v)))
PY
done
