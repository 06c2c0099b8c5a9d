"""Is the global sort hidden behind the steps when it is prefetched?  One process (world size 1) stepping 16,384-sample
batches through DataParallelFM with prefetch depth 0 / 1 / 2: the kernels of a 4-GPU step minus the collectives."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
B, n_pool = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8
idx_np, y_np = bench.synth_pool(n_pool, B, bench.CRITEO_SIZES, 3)
idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
for depth in (0, 1, 2):
    t = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
    t.rows[:, :16].normal_(0, 0.01)
    eng = fmx.FMEngine(t, max_batch=B)
    dp = fmx.DataParallelFM(fmx.HipBackend(eng, fmx.Hyper(**bench.HYPER), "ftrl", "logits"))
    work = torch.cuda.Stream()
    with torch.cuda.stream(work):
        def run(n):
            tokens = {d: dp.prefetch(idx_pool[d % n_pool]) for d in range(min(depth, n))}
            for s in range(n):
                if depth and s + depth < n:
                    tokens[s + depth] = dp.prefetch(idx_pool[(s + depth) % n_pool])
                dp.step(idx_pool[s % n_pool], y_pool[s % n_pool], tokens.pop(s, None))
        run(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(200)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
    print(f"B={B} prefetch depth {depth}: {dt*1e6:.1f} us/step  {B/dt/1e6:.1f} M samples/s", flush=True)
