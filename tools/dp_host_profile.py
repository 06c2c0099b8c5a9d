"""Host-side cost of one DataParallelFM step (world size 1, so no collectives): cProfile of 300 steps."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
B, n_pool = 4096, 8
idx_np, y_np = bench.synth_pool(n_pool, B, bench.CRITEO_SIZES, 3)
idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
t = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
eng = fmx.FMEngine(t, max_batch=B)
dp = fmx.DataParallelFM(fmx.HipBackend(eng, fmx.Hyper(**bench.HYPER), "ftrl", "logits"))
work = torch.cuda.Stream()
def run(n):
    tok = dp.prefetch(idx_pool[0])
    for s in range(n):
        nxt = dp.prefetch(idx_pool[(s + 1) % n_pool]) if s + 1 < n else None
        dp.step(idx_pool[s % n_pool], y_pool[s % n_pool], tok)
        tok = nxt
with torch.cuda.stream(work):
    run(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(300); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e6*(t1-t0)/300:.1f} us/step, until GPU done {1e6*(t2-t0)/300:.1f} us/step")
    pr = cProfile.Profile(); pr.enable(); run(300); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
