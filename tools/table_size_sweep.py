"""Online-loop time per step against the table size: the Criteo vocabulary list scaled by 1x / 4x / 16x / 64x
(258 MB -> 16.5 GB of FTRL rows at k = 16), i.e. from Infinity-Cache-sized to HBM-resident."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
k, B, n_pool = 16, 4096, 16
for scale in [int(a) for a in (sys.argv[1:] or ["1", "4", "16", "64"])]:
    sizes = [min(s * scale, (1 << 20) - 1) for s in bench.CRITEO_SIZES]          # 20 index bits + 12 sample bits = 32
    idx_np, y_np = bench.synth_pool(n_pool, B, sizes, 7)
    idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
    t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=ftrl)
    t.rows[:, :k].normal_(0, 0.01)
    eng = fmx.FMEngine(t, max_batch=B)
    hyp = fmx.Hyper(lr=0.01, **ftrl)
    loss = torch.zeros(4096, device="cuda")
    best = 1e9
    for rep in range(3):
        eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 100, loss)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 1600, loss)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1600)
    eng.check_error_flag()
    ms = eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 50, loss, timed=True)
    R = sum(sizes)
    print(f"x{scale:<3d} R={R:>11,d} rows  table {R * 256 / 1e9:6.2f} GB: {best*1e6:6.2f} us/step  {B/best/1e6:6.1f} M samples/s   "
          f"fwd {ms[1]/50*1e3:.1f} upd {ms[2]/50*1e3:.1f} us", flush=True)
    del t, eng
    torch.cuda.empty_cache()
