"""Turn gpurun_out/<tag>_{bench.json,stats,fetch,write} (tools/profile_round.sh) into profiles/<tag>_* and
profiles/traffic_latest.json.  Works on the GPU box (writes into gpurun_out/profiles_<tag>/ as well, which travels back)
and here."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
dst = [os.path.join(ROOT, "profiles"), os.path.join(G, "profiles_" + tag)]
for d in dst:
    os.makedirs(d, exist_ok=True)


def put(name, text=None, src=None):
    for d in dst:
        if src:
            shutil.copyfile(src, os.path.join(d, name))
        else:
            open(os.path.join(d, name), "w").write(text)


def find(sub, pat):
    hits = glob.glob(os.path.join(G, f"{tag}_{sub}", "**", pat), recursive=True)
    return hits[0] if hits else None


bench = os.path.join(G, f"{tag}_bench.json")
if os.path.exists(bench):
    put(f"{tag}_bench.json", src=bench)
stats = find("stats", "*kernel_stats.csv")
if stats:
    put(f"{tag}_kernel_stats.csv", src=stats)
loop = find("loop", "*kernel_stats.csv")
if loop:
    put(f"{tag}_loop_kernel_stats.csv", src=loop)


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def counters(sub, counter):
    path = find(sub, "*counter_collection.csv")
    acc = {}
    if not path:
        return acc
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = short(row["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        a = acc.setdefault(k, {"sum": 0.0, "ids": set()})
        a["sum"] += float(row["Counter_Value"])
        a["ids"].add(row["Dispatch_Id"])
    return {k: (v["sum"], len(v["ids"])) for k, v in acc.items()}


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
if fetch and write:
    per = {}
    for k in fetch:
        n = fetch[k][1]
        per[k] = {"FETCH_SIZE_KB": fetch[k][0] / n, "WRITE_SIZE_KB": write.get(k, (0.0, 1))[0] / max(write.get(k, (0, 1))[1], 1),
                  "launches": n}
    upd = [k for k in per if k.startswith("k_fm_update")]
    alg = None
    try:
        for line in open(os.path.join(ROOT, "bench.py")):
            if line.startswith("BYTES_K_UPDATE"):
                alg = eval(line.split("=", 1)[1].split("#")[0], {"F": 39}) * 4096
    except Exception:
        pass
    u = per[upd[0]] if upd else None
    doc = {
        "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 50 --warmup 10` "
                  f"(tools/profile_round.sh {tag}; the measuring pass launches every kernel 8x, counters are per-launch means)",
        "correction": "gfx950: FETCH_SIZE tallies one 64 B unit per request, so a 128-byte-line request reads as half its "
                      "bytes (MI355X_MICROARCH.md, HBM section; re-measured with tools/gather_bench: 64 B rows read 1.00x, "
                      "128 B rows 0.50x, writes exact).  k_fm_update mixes 64 B + 16 B pieces of line 0 with full 128 B "
                      "(z, n) lines: hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE is the upper bound reported here.",
        "per_kernel": per,
    }
    if u:
        doc["k_fm_update_hbm_bytes_per_launch"] = (2 * u["FETCH_SIZE_KB"] + u["WRITE_SIZE_KB"]) * 1024
        doc["k_fm_update_hbm_bytes_per_launch_uncorrected"] = (u["FETCH_SIZE_KB"] + u["WRITE_SIZE_KB"]) * 1024
        doc["algorithmic_bytes_per_launch"] = alg
    put(f"{tag}_traffic_pmc.json", json.dumps(doc, indent=1))
    put("traffic_latest.json", json.dumps(doc, indent=1))
print("profiles written for", tag, ":", sorted(os.listdir(dst[1])))
