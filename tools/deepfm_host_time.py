"""Is the DeepFM trainer step (bench.py --workload deepfm) bound by its host side?  Time to ENQUEUE n steps against the time
until they have run."""
import os, sys, time
import torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd")); sys.path.insert(0, ROOT)
import fmx, bench as B
dev = torch.device("cuda", 0)
table = fmx.FlatTable(B.CRITEO_SIZES, B.K_EMB, layout="weights", device=dev)
table.rows[:, :B.K_EMB] = torch.randn((table.n_rows, B.K_EMB), device=dev) * 0.01
layers = [nn.Linear(B.K_EMB if j == 0 else 256, 256).to(dev) for j in range(3)]
eng = fmx.FMEngine(table, max_batch=B.BATCH)
tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, fmx.Hyper(lr=1e-3), "sgd"), layers, B.K_EMB, table.kp, mlp_lr=1e-3)
idx_np, y_np = B.synth_pool(B.N_POOL, B.BATCH, B.CRITEO_SIZES, 5)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
for s in range(20): tr.step(idx_pool[s % B.N_POOL], y_pool[s % B.N_POOL])
torch.cuda.synchronize()
for n in (50, 200):
    t0 = time.perf_counter()
    for s in range(n): tr.step(idx_pool[s % B.N_POOL], y_pool[s % B.N_POOL])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueued in {(t1 - t0) / n * 1e6:.1f} us/step, run in {(t2 - t0) / n * 1e6:.1f} us/step", flush=True)
