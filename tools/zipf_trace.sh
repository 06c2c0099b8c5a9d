#!/bin/bash
# kernel trace of tools/zipf_bisect.py: the loop's step period and the gaps per probe run
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/zipf_trace -o t -- python3 $root/tools/zipf_bisect.py > $out/zipf_trace.log 2>&1
cd $root
grep "us/step" $out/zipf_trace.log
python3 - "$(find $out/zipf_trace -name t_kernel_trace.csv | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = []
for r in rows:
    n = r["Kernel_Name"]
    k = "fwd" if "k_fm_forward" in n else "upd" if "k_fm_update" in n else "sort" if "k_sort_occ" in n else None
    if k: ks.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
# split into runs by idle gaps > 2 ms
runs, cur = [], [ks[0]]
for a, b in zip(ks, ks[1:]):
    if b[1] - a[2] > 2_000_000: runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
for i, run in enumerate(runs):
    upd = [k for k in run if k[0] == "upd"]
    if len(upd) < 300: continue
    period = (upd[-1][2] - upd[100][2]) / (len(upd) - 101) / 1e3
    main = [k for k in run if k[0] != "sort"]
    gaps = [(b[1] - a[2]) / 1e3 for a, b in zip(main, main[1:])]
    big = sum(1 for g in gaps if g > 5)
    qs = {k[0]: set() for k in run}
    for k in run: qs[k[0]].add((k[3], k[4]))
    print("run %d: %d updates, period %.1f us, mean gap %.2f us, gaps > 5 us: %d of %d; queues/streams %s" % (i, len(upd), period, sum(gaps) / len(gaps), big, len(gaps), qs))
    if period > 40:
        j = len(run) // 2
        t0 = run[j][1]
        print("   middle of the slow run:", [(k[0], round((k[1] - t0) / 1e3, 1), round((k[2] - t0) / 1e3, 1)) for k in run[j:j + 14]])
PY
