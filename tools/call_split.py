import os, sys, time
import numpy as np, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd")); sys.path.insert(0, ROOT)
import fmx, bench
dev = torch.device("cuda", 0)
hyper = fmx.Hyper(**bench.HYPER)
table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", device=dev, ftrl=bench.HYPER)
w0 = torch.randn((table.n_rows, 16), device=dev) * 0.01
table.rows[:, :16] = w0
table.rows[:, table.z_offset:table.z_offset + 16] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
eng = fmx.FMEngine(table, max_batch=4096)
idx_np, y_np = bench.synth_pool(16, 4096, bench.CRITEO_SIZES, 1)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
loss = torch.zeros(100, device=dev)
work = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
run = eng.prepare_stream(hyper, "ftrl", "logits", idx_pool, y_pool, loss, stream=work)
run(5); torch.cuda.synchronize()
for rep in range(6):
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(work)
    t1 = time.perf_counter()
    run(n)
    t2 = time.perf_counter()
    e1.record(work)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"rep {rep}: wall {1e6*(t3-t0):.0f} us = {1e6*(t3-t0)/n:.2f}/step | record {1e6*(t1-t0):.1f}, host in call {1e6*(t2-t1):.0f}, after call -> synced {1e6*(t3-t2):.0f} | device span (events) {e0.elapsed_time(e1)*1e3:.0f} us", flush=True)
