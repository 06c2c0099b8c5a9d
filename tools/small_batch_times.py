"""Per-launch kernel times against the batch size (the launch + dependent-round-trip floor shows at B = 64)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
table.rows[:, :16] = torch.randn((table.n_rows, 16), device=dev) * 0.01
hyper = fmx.Hyper(**bench.HYPER)
for B in (64, 256, 1024, 4096):
    eng = fmx.FMEngine(table, max_batch=B)
    idx_np, y_np = bench.synth_pool(8, B, bench.CRITEO_SIZES, 1)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
    loss = torch.zeros(512, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 16, loss)
        ms = eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 400, loss, timed=True)
        torch.cuda.synchronize()
    print(B, {k: round(v / 400 * 1e3, 2) for k, v in zip(("sort(8 batches)", "fwd", "upd", "empty event pair"), ms)}, "us")
