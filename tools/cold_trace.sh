#!/bin/bash
# kernel trace of the driver-style cold run (--steps 20 --warmup 5 in a fresh process): per-kernel durations and gaps of the
# 20 timed steps against the same kernels late in a long run
set -o pipefail
tag=${1:-run}
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_cold -o c -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/${tag}_cold.log 2>&1
cd $root
python3 - <<PY
import csv, glob
f = glob.glob("$out/${tag}_cold/**/c_kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if "k_fm_" in r["Kernel_Name"] or "k_sort" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "k_fm_" in r["Kernel_Name"]]
# the production loop's launches come first: 5 warm-up steps, then 20 timed steps, each forward + update
loop = main[:50]
def name(r): return "fwd" if "forward" in r["Kernel_Name"] else "upd"
for lo, hi, what in ((0, 10, "warm-up steps 0-4"), (10, 30, "timed steps 0-9"), (30, 50, "timed steps 10-19")):
    seg = loop[lo:hi]
    d = {}
    for r in seg: d.setdefault(name(r), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(seg, seg[1:])]
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / (len(seg) / 2)
    print(what, {k: round(sum(v) / len(v), 2) for k, v in d.items()}, "mean gap %.2f us" % (sum(gaps) / len(gaps)), "max gap %.1f" % max(gaps), "-> %.2f us/step" % span)
late = main[-400:]
d = {}
for r in late: d.setdefault(name(r), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("measuring pass (late, back to back)", {k: round(sum(v) / len(v), 2) for k, v in d.items()})
PY
