"""Online-loop time per step with the tile-crossing runs finished in-launch or by k_fm_fixup (fmx_set_option inline_fixup)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
import bench
lib = fmx._lib.load()
sizes, k, B, n_pool = bench.CRITEO_SIZES, 16, 4096, 16
rule = sys.argv[1] if len(sys.argv) > 1 else "ftrl"
for zipf in (False, True):
    idx_np, y_np = bench.synth_pool(n_pool, B, sizes, 7, zipf=zipf)
    idx_pool, y_pool = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    for inline in (1, 0):
        lib.fmx_set_option(b"inline_fixup", inline)
        ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
        t = fmx.FlatTable(sizes, k, layout="ftrl" if rule == "ftrl" else "weights", ftrl=ftrl)
        t.rows[:, :k].normal_(0, 0.01)
        eng = fmx.FMEngine(t, max_batch=B)
        hyp = fmx.Hyper(lr=0.01, **ftrl)
        loss = torch.zeros(4096, device="cuda")
        best = 1e9
        for rep in range(4):
            eng.stream(hyp, rule, "logits", idx_pool, y_pool, 100, loss)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.stream(hyp, rule, "logits", idx_pool, y_pool, 1600, loss)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 1600)
        eng.check_error_flag()
        ms = eng.stream(hyp, rule, "logits", idx_pool, y_pool, 50, loss, timed=True)
        print(f"{rule} zipf={int(zipf)} inline={inline}: {best*1e6:.2f} us/step  {B/best/1e6:.1f} M samples/s   "
              f"per launch us: sort {ms[0]/50*1e3:.1f} fwd {ms[1]/50*1e3:.1f} upd/step {ms[2]/50*1e3:.1f} pair {ms[3]/50*1e3:.1f}", flush=True)
