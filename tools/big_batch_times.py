"""Kernel times of one step at the global batch sizes the exact data-parallel mode reaches (G x 4096)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
table.rows[:, :16] = torch.randn((table.n_rows, 16), device=dev) * 0.01
hyper = fmx.Hyper(**bench.HYPER)
for B in (4096, 8192, 16384):
    eng = fmx.FMEngine(table, max_batch=B)
    idx_np, y_np = bench.synth_pool(2, B, bench.CRITEO_SIZES, 1)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
    loss = torch.zeros(64, device=dev)
    eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 4, loss)
    ms = eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 20, loss, timed=True)
    torch.cuda.synchronize()
    print(B, {k: round(v / 20 * 1e3, 1) for k, v in zip(("sort", "fwd", "upd", "empty event pair"), ms)}, "us")
