#!/bin/bash
# Kernel durations inside the steady online loop by POSITION in the 8-step sort group (rocprofv3 --kernel-trace of
# bench.py --steps 2000 --loop-only): the next group's sort runs beside the first steps of a group, the later steps have the chip
# to themselves -- the difference is what the side-stream sort costs the step's own kernels.
#   bash tools/loop_positions.sh <tag>
set -o pipefail
tag=${1:-run}
root=$(pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out/${tag}_pos -o p -- python3 $root/bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-secondary --loop-only > $out/${tag}_pos.log 2>&1
cd $root
python3 - <<PY | tee $out/${tag}_loop_positions.txt
import csv, glob
f = glob.glob("$out/${tag}_pos/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "k_fm_forward" in r["Kernel_Name"] or "k_fm_update" in r["Kernel_Name"]]
sorts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_sort" in r["Kernel_Name"]]
main = main[-3600:]          # the last 1,800 steps of the timed call
import bisect
starts = [s for s, e in sorts]
def overlap(s, e):
    i = bisect.bisect_right(starts, e) - 1
    tot = 0
    while i >= 0 and sorts[i][1] > s - 100000:
        tot += max(0, min(e, sorts[i][1]) - max(s, sorts[i][0]))
        i -= 1
    return tot
acc = {}
for r in main:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = "fwd" if "forward" in r["Kernel_Name"] else "upd"
    beside = overlap(s, e) > 0.5 * (e - s)
    acc.setdefault((k, beside), []).append((e - s) / 1e3)
for k in ("fwd", "upd"):
    for b in (False, True):
        v = acc.get((k, b), [])
        if v: print("%s %-22s n=%5d mean %.2f us  median %.2f" % (k, "beside a sort" if b else "chip to itself", len(v), sum(v) / len(v), sorted(v)[len(v) // 2]))
span = (int(main[-1]["End_Timestamp"]) - int(main[0]["Start_Timestamp"])) / 1e3 / (len(main) / 2)
sd = [(e - s) / 1e3 for s, e in sorts[-200:]]
print("step %.2f us; sort launches: mean %.2f us" % (span, sum(sd) / len(sd)))
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(main, main[1:])]
print("mean gap between the loop's kernels %.2f us" % (sum(gaps) / len(gaps)))
PY
