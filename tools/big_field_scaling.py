"""Is the update of the LARGE fields bound by its dependent chain or by throughput?  n copies of a 150,000-row field, n = 1 ... 39:
a chain-bound launch takes the same time at every n, a throughput-bound one scales with n."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
k, B, n_pool = 16, 4096, 16
for n in (1, 3, 6, 12, 24, 39):
    sizes = [150000] * n
    idx_np, y_np = bench.synth_pool(n_pool, B, sizes, 7)
    idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
    t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=ftrl)
    t.rows[:, :k].normal_(0, 0.01)
    eng = fmx.FMEngine(t, max_batch=B)
    hyp = fmx.Hyper(lr=0.01, **ftrl)
    loss = torch.zeros(2048, device="cuda")
    eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 64, loss)
    ms = eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 400, loss, timed=True)
    torch.cuda.synchronize()
    eng.check_error_flag()
    mb = n * B * 256 * 2 / 1e6
    print(f"{n:2d} fields x 150,000 rows ({n * 150000 * 256 / 1e6:6.0f} MB table): fwd {ms[1]/400*1e3:5.2f}  upd {ms[2]/400*1e3:5.2f} us  ({mb:5.1f} MB of rows read + written per launch)", flush=True)
