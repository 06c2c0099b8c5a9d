"""Do two global sorts on two streams run side by side?  Time n sorts of a 16,384-sample batch on 1, 2 and 4 streams."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
idx_np, _ = bench.synth_pool(4, B, bench.CRITEO_SIZES, 3)
idx_pool = torch.from_numpy(idx_np).cuda()
t = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
eng = fmx.FMEngine(t, max_batch=B)
wss = [eng.new_workspace(B) for _ in range(4)]
for n_streams in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            eng.sort(idx_pool[i % 4], workspace=wss[i % n_streams], stream=streams[i % n_streams])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
    print(f"B={B}: {n_streams} stream(s): {dt*1e6:.1f} us per sort", flush=True)
