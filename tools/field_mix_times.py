"""Forward / update launch times for tables whose 39 fields are all large (no run crosses a tile), all tiny (every run
crosses many tiles) or the Criteo mix: where the update's time goes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
k, B, n_pool = 16, 4096, 16
for name, sizes in (("criteo", bench.CRITEO_SIZES), ("39 x 100000 rows", [100000] * 39), ("39 x 1000 rows", [1000] * 39),
                    ("39 x 50 rows", [50] * 39), ("39 x 3 rows", [3] * 39)):
    idx_np, y_np = bench.synth_pool(n_pool, B, sizes, 7)
    idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
    t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=ftrl)
    t.rows[:, :k].normal_(0, 0.01)
    eng = fmx.FMEngine(t, max_batch=B)
    hyp = fmx.Hyper(lr=0.01, **ftrl)
    loss = torch.zeros(512, device="cuda")
    eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 64, loss)
    ms = eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 400, loss, timed=True)
    torch.cuda.synchronize()
    eng.check_error_flag()
    print(f"{name:20s} sort {ms[0]/400*1e3:5.1f} (8 batches)  fwd {ms[1]/400*1e3:5.2f}  upd {ms[2]/400*1e3:5.2f} us", flush=True)
