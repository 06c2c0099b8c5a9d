"""Host enqueue time per step of fmx_fm_stream against the time until the GPU is done (is the loop host-bound?)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
table.rows[:, :16] = torch.randn((table.n_rows, 16), device=dev) * 0.01
hyper = fmx.Hyper(**bench.HYPER)
eng = fmx.FMEngine(table, max_batch=4096)
idx_np, y_np = bench.synth_pool(16, 4096, bench.CRITEO_SIZES, 1)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
loss = torch.zeros(2000, device=dev)
eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 50, loss); torch.cuda.synchronize()
for n in (200, 1000):
    t0 = time.perf_counter(); eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, n, loss); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e6*(t1-t0)/n:.1f} us/step, until GPU done {1e6*(t2-t0)/n:.1f} us/step")
