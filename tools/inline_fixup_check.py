"""FMX_INLINE_FIXUP=1 must give the same bits as the two-launch path (same record order).  Run once per mode and compare
checksums: python tools/inline_fixup_check.py > out ; the driver script below diffs the two outputs."""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
for zipf in (False, True):
    table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
    g = torch.Generator(device=dev).manual_seed(1)
    w0 = torch.randn((table.n_rows, 16), generator=g, device=dev) * 0.01
    table.rows[:, :16] = w0
    table.rows[:, table.z_offset:table.z_offset + 16] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
    eng = fmx.FMEngine(table, max_batch=4096)
    idx_np, y_np = bench.synth_pool(16, 4096, bench.CRITEO_SIZES, 7, zipf=zipf)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
    loss = torch.zeros(2000, device=dev)
    eng.stream(fmx.Hyper(**bench.HYPER), "ftrl", "logits", idx_pool, y_pool, 1500, loss)
    torch.cuda.synchronize()
    eng.check_error_flag()
    h = hashlib.sha256(table.rows.cpu().numpy().tobytes()).hexdigest()[:16]
    print("zipf" if zipf else "uniform", h, hashlib.sha256(loss.cpu().numpy().tobytes()).hexdigest()[:16], float(loss[1499]))
