"""Online-loop time per step against the number of batches sorted per side-stream launch (fmx_set_option sort_ahead)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
lib = fmx._lib.load()
sizes, k, B, n_pool = bench.CRITEO_SIZES, 16, 4096, 16
idx_np, y_np = bench.synth_pool(n_pool, B, sizes, 7)
idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=dict(alpha=0.05, beta=1.0, l1=0.0, l2=1e-4))
t.rows[:, :k].normal_(0, 0.01)
eng = fmx.FMEngine(t, max_batch=B)
hyp = fmx.Hyper(lr=0.01, alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
loss = torch.zeros(4096, device="cuda")
for ahead in [int(a) for a in (sys.argv[1:] or ["4", "8", "12", "16"])]:
    lib.fmx_set_option(b"sort_ahead", ahead)
    best = 1e9
    for rep in range(4):
        eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 100, loss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 1920, loss)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1920)
    print(f"sort_ahead={ahead:2d}: {best*1e6:.2f} us/step  {B/best/1e6:.1f} M samples/s", flush=True)
