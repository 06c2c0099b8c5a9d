"""Occurrence-sort times: one batch per launch and 8 batches per launch, chunked (k_sort_chunk + k_sort_merge) vs one
workgroup per field (k_sort_occ), Criteo vocabulary."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
lib = fmx._lib.load()
table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
hyper = fmx.Hyper(**bench.HYPER)
work = torch.cuda.Stream()
for B in (4096, 8192, 16384):
    eng = fmx.FMEngine(table, max_batch=B)
    idx_np, y_np = bench.synth_pool(16, B, bench.CRITEO_SIZES, 1)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
    loss = torch.zeros(256, device=dev)
    for chunked in (2, 1, 0):
        old = lib.fmx_set_option(b"sort_chunked", chunked)
        with torch.cuda.stream(work):
            for _ in range(3):
                eng.sort(idx_pool[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 40
            e0.record()
            for i in range(n):
                eng.sort(idx_pool[i % 16])
            e1.record()
            torch.cuda.synchronize()
            one = e0.elapsed_time(e1) / n * 1e3
            ms = eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 64, loss, timed=True)
            torch.cuda.synchronize()
        lib.fmx_set_option(b"sort_chunked", old)
        print(f"B={B} chunked={chunked}: one batch per call {one:.1f} us (back to back, launch-bound floor included); "
              f"8 batches per launch {ms[0] / 64 * 1e3:.1f} us; fwd {ms[1] / 64 * 1e3:.1f} upd {ms[2] / 64 * 1e3:.1f}", flush=True)
