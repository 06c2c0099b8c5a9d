#!/bin/bash
# kernel timeline of the DeepFM trainer step (tools/deepfm_host_time.py under rocprofv3 --kernel-trace): per-kernel mean
# durations and the gaps between consecutive kernels of the main stream over the last 100 steps
set -o pipefail
tag=${1:-deepfm}
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$2" = "stream" ]; then   # the native loop (fmx_deepfm_stream) as bench.py runs it; FMX_DEEPFM_STREAM=0 there: the trainer's step
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_trace -o t -- python3 $root/bench.py --workload deepfm --steps 100 --warmup 10 > $out/${tag}_trace.log 2>&1
else
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_trace -o t -- python3 $root/tools/deepfm_host_time.py > $out/${tag}_trace.log 2>&1
fi
cd $root
grep "steps:" $out/${tag}_trace.log
f=$(find $out/${tag}_trace -name "t_kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("k_fm_forward", "k_fm_update", "k_fm_fixup", "k_sort_occ", "k_mlp_chain", "k_mlp_wgrad_stream", "k_mlp_reduce"):
        if k in n: return k
    return n[:30]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
upd = [i for i, k in enumerate(ks) if k[0] == "k_fm_update"]
import os
if len(upd) >= 220:                  # bench.py --workload deepfm: 10 + 100 steps of the native loop, then 10 + 100 through the trainer
    lo, hi = upd[15], upd[105]
else:
    lo, hi = upd[-91], upd[-1]       # 90 steps: from the end of one update to the end of the last
seg = ks[lo + 1:hi + 1]
span = (ks[hi][2] - ks[lo][2]) / 90 / 1e3
dur = collections.defaultdict(list)
for k, s, e in seg: dur[k].append((e - s) / 1e3)
print("per step %.1f us on the device; kernels: %s" % (span, ", ".join("%s %.1f (x%.0f)" % (k, sum(v) / len(v), len(v) / 90) for k, v in dur.items())))
main = [x for x in seg if x[0] != "k_sort_occ"]
gaps = collections.defaultdict(list)
prev = ks[lo]
for x in main:
    gaps[prev[0] + " -> " + x[0]].append((x[1] - prev[2]) / 1e3)
    prev = x
for k, v in gaps.items(): print("gap %-42s mean %.2f us" % (k, sum(v) / len(v)))
print("sum of main-stream kernels %.1f us, of gaps %.1f us per step" % (sum(sum(v) for k, v in dur.items() if k != "k_sort_occ") / 90, sum(sum(v) for v in gaps.values()) / 90))
PY
