"""Is the one-launch step (k_fm_fused) bit-identical to separate launches, and to itself from run to run?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
lib = fmx._lib.load()
hyper = fmx.Hyper(**bench.HYPER)
idx_np, y_np = bench.synth_pool(5, 4096, bench.CRITEO_SIZES, 3)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
w0 = torch.randn((sum(bench.CRITEO_SIZES), 16), generator=g, device=dev) * 0.01
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 41
def run(mode):
    table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
    table.rows[:, :16] = w0
    table.rows[:, table.z_offset:table.z_offset + 16] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
    eng = fmx.FMEngine(table, max_batch=4096)
    loss = torch.zeros(n_steps, device=dev)
    old = lib.fmx_set_option(b"fused_step", mode)
    torch.cuda.synchronize()
    eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, n_steps, loss)
    torch.cuda.synchronize()
    lib.fmx_set_option(b"fused_step", old)
    err = int(eng.error.item())
    return table.rows.cpu().numpy(), loss.cpu().numpy(), err
ref = run(0)
ref2 = run(0)
print("separate vs separate: rows differing", int((ref[0] != ref2[0]).sum()), flush=True)
for mode in (1, 1, 2, 2):
    r = run(mode)
    d = r[0] != ref[0]
    first_bad = int(np.argmax(r[1] != ref[1])) if (r[1] != ref[1]).any() else -1
    print(f"fused_step={mode}: error flag {r[2]}, elements differing {int(d.sum())} (rows {int(d.any(axis=1).sum())}), "
          f"first differing loss at step {first_bad}, max |d| {float(np.abs(r[0] - ref[0]).max()):.3e}", flush=True)
