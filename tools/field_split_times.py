"""Which fields set the update's time?  The Criteo vocabulary list cut by field size: all 39 fields, the 20 fields above 128 rows,
the 19 fields of at most 128 rows (every row hit more than 32 times per step: hand-off chains), the 12 fields above 4,096 rows."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
k, B, n_pool = 16, 4096, 16
C = bench.CRITEO_SIZES
for name, sizes in (("criteo (39)", C), ("> 128 rows (%d)" % sum(s > 128 for s in C), [s for s in C if s > 128]),
                    ("<= 128 rows (%d)" % sum(s <= 128 for s in C), [s for s in C if s <= 128]),
                    ("> 4096 rows (%d)" % sum(s > 4096 for s in C), [s for s in C if s > 4096]),
                    ("129..4096 rows (%d)" % sum(128 < s <= 4096 for s in C), [s for s in C if 128 < s <= 4096])):
    idx_np, y_np = bench.synth_pool(n_pool, B, sizes, 7)
    idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
    t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=ftrl)
    t.rows[:, :k].normal_(0, 0.01)
    eng = fmx.FMEngine(t, max_batch=B)
    hyp = fmx.Hyper(lr=0.01, **ftrl)
    loss = torch.zeros(2048, device="cuda")
    work = torch.cuda.Stream()
    torch.cuda.synchronize()
    run = eng.prepare_stream(hyp, "ftrl", "logits", idx_pool, y_pool, loss, stream=work)
    run(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(2000)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 2000 * 1e6
    ms = eng.stream(hyp, "ftrl", "logits", idx_pool, y_pool, 400, loss, timed=True)
    torch.cuda.synchronize()
    eng.check_error_flag()
    print(f"{name:22s} step {step:5.2f} us   back to back: sort {ms[0]/400*1e3:5.1f} (8 batches)  fwd {ms[1]/400*1e3:5.2f}  upd {ms[2]/400*1e3:5.2f} us", flush=True)
