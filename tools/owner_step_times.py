"""Per-rank work of the field-owner mode at G = 1, 2, 4, 8, measured on ONE GPU: rank 0's shard of the Criteo table, the
global batch of G x 4096 samples, every kernel of a step timed back to back (HIP events, 40 launches each), plus the whole
step through FieldOwnerFM at G = 1 (host cost included).  The collectives are not part of this: DESIGN.md adds them."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
from fmx.owner import FieldOwnerFM, HipOwnerBackend
dev = torch.device("cuda")
hyper = fmx.Hyper(**bench.HYPER)
B = 4096
work = torch.cuda.Stream()

def timed(fn, n=40):
    with torch.cuda.stream(work):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for G in (1, 2, 4, 8):
    GB = G * B
    rows, tot = [], {}
    for rank in sorted({0, G - 1}):                 # the first and the last rank's shard (the plan balances them: fmx/plan.py)
        be = HipOwnerBackend(bench.CRITEO_SIZES, 16, hyper, "ftrl", "logits", rank, G, ftrl=bench.HYPER, max_local_batch=B)
        w0 = torch.randn((be.table.n_rows, 16), device=dev) * 0.01
        be.table.rows[:, :16] = w0
        be.table.rows[:, be.table.z_offset:be.table.z_offset + 16] = fmx.table.ftrl_z_for_weight_torch(w0, be.table.ftrl)
        idx_np, y_np = bench.synth_pool(4, GB, bench.CRITEO_SIZES, 7)
        idx_all = torch.from_numpy(idx_np).to(dev)
        y = torch.from_numpy(y_np[0][:B].copy()).to(dev)
        t_sort = timed(lambda: be.start_sort(idx_all[0], 0, stream=work))
        t_part = timed(lambda: be.partial_forward(idx_all[1], B, stream=work))
        parts = be.partial_forward(idx_all[1], B).view(G, be.nlb, B, -1)
        mine = torch.cat([parts[0]] * (be.nb // be.nlb)).contiguous()[:be.nb].contiguous()
        t_fin = timed(lambda: be.finish(mine, y, 1.0 / GB, stream=work))
        rec = be.finish(mine, y, 1.0 / GB)
        rec_g = torch.cat([rec] * G).contiguous()
        be.start_sort(idx_all[1], 1)
        t_upd = timed(lambda: be.update(idx_all[1], rec_g, 1.0 / GB, 1, stream=work))
        n_pieces = sum(1 for f in be.fields if f[2])
        print(f"G={G} rank {rank}: {n_pieces} pieces, {be.table.n_rows} rows; global batch {GB}: sort {t_sort:.1f} (ahead of time)  "
              f"partial fwd {t_part:.1f}  finish {t_fin:.1f}  update {t_upd:.1f} us -> on the critical path {t_part + t_fin + t_upd:.1f} us of kernels", flush=True)
        del be

be = HipOwnerBackend(bench.CRITEO_SIZES, 16, hyper, "ftrl", "logits", 0, 1, ftrl=bench.HYPER, max_local_batch=B)
fo = FieldOwnerFM(be)
idx_np, y_np = bench.synth_pool(8, B, bench.CRITEO_SIZES, 7)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
def run(n):
    tokens = {0: fo.prefetch(idx_pool[0]), 1: fo.prefetch(idx_pool[1])}
    for s in range(n):
        fo.step(idx_pool[s % 8], y_pool[s % 8], tokens.pop(s, None))
        if s + 2 < n:
            tokens[s + 2] = fo.prefetch(idx_pool[(s + 2) % 8])
with torch.cuda.stream(work):
    run(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(300)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"FieldOwnerFM.step at G = 1 (no collectives), host included: {dt / 300 * 1e6:.1f} us/step = {300 * B / dt / 1e6:.1f} M samples/s")

# the same loop through fmx_owner_prefetch / fmx_owner_step (one C call each per step), and what the HOST spends per step
from fmx.owner import NativeOwnerFM
nat = NativeOwnerFM(be, stream=work)
idx_l, y_l = [idx_pool[j] for j in range(8)], [y_pool[j] for j in range(8)]
def run_native(n):
    tokens = {0: nat.prefetch(idx_l[0]), 1: nat.prefetch(idx_l[1])}
    for s in range(n):
        nat.step(idx_l[s % 8], y_l[s % 8], tokens.pop(s, None))
        if s + 2 < n:
            tokens[s + 2] = nat.prefetch(idx_l[(s + 2) % 8])
with torch.cuda.stream(work):
    run_native(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_native(300)
    t_host = time.perf_counter() - t0          # the host is done issuing; the device still runs
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"NativeOwnerFM.step at G = 1 (fmx_owner_step, no collectives): {dt / 300 * 1e6:.1f} us/step = {300 * B / dt / 1e6:.1f} M samples/s; "
      f"host time issuing a step (prefetch + step calls) {t_host / 300 * 1e6:.1f} us")
