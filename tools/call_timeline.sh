#!/bin/bash
# Host API calls and kernels of the driver-style run (--steps 20 --warmup 5, fresh process) on ONE clock: where the wall time of
# the timed 20-step call goes besides the kernels (rocprofv3 --kernel-trace --hip-trace; no counters).
#   bash tools/call_timeline.sh <tag> [extra bench.py args]
set -o pipefail
tag=${1:-run}; shift
root=$(pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $out/${tag}_tl -o t -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --loop-only "$@" > $out/${tag}_tl.log 2>&1
cd $root
python3 - <<PY > $out/${tag}_timeline.txt
import csv, glob
kf = glob.glob("$out/${tag}_tl/**/t_kernel_trace.csv", recursive=True)[0]
af = glob.glob("$out/${tag}_tl/**/t_hip_api_trace.csv", recursive=True)[0]
ks = sorted(csv.DictReader(open(kf)), key=lambda r: int(r["Start_Timestamp"]))
aps = sorted(csv.DictReader(open(af)), key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in ks if "k_fm_forward" in r["Kernel_Name"] or "k_fm_update" in r["Kernel_Name"] or "k_fm_small" in r["Kernel_Name"]]
fw = [r for r in main if "k_fm_forward" in r["Kernel_Name"]]
first = fw[5]      # the first forward of the timed call (5 warm-up steps before it)
last = main[-1]
t0 = int(first["Start_Timestamp"]); t1 = int(last["End_Timestamp"])
lo, hi = t0 - 400_000, t1 + 300_000
ev = []
for r in ks:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s <= hi: ev.append((s, "K", r["Kernel_Name"].split("(")[0][:60], (e - s) / 1e3))
for r in aps:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s <= hi: ev.append((s, "A", r["Function"], (e - s) / 1e3))
ev.sort()
print("timed call: first forward starts at 0; last update ends at %.1f us -> %.2f us/step on the device" % ((t1 - t0) / 1e3, (t1 - t0) / 1e3 / 20))
for s, kind, name, d in ev:
    print("%9.1f %s %-62s %7.2f" % ((s - t0) / 1e3, kind, name, d))
PY
head -3 $out/${tag}_timeline.txt
