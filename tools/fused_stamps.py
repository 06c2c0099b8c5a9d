"""Inside ONE k_fm_fused launch: s_memrealtime stamps (100 MHz) of every workgroup -- start, arrival (update workgroups) or
poll match (forward workgroups), end.  FMX_FUSED_DEBUG=6 must be set."""
import sys, os
os.environ["FMX_FUSED_DEBUG"] = "6"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import numpy as np, torch, fmx, bench
dev = torch.device("cuda")
hyper = fmx.Hyper(**bench.HYPER)
idx_np, y_np = bench.synth_pool(16, 4096, bench.CRITEO_SIZES, 3)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
table = fmx.FlatTable(bench.CRITEO_SIZES, 16, layout="ftrl", ftrl=bench.HYPER)
w0 = torch.randn((table.n_rows, 16), device=dev) * 0.01
table.rows[:, :16] = w0
table.rows[:, table.z_offset:table.z_offset + 16] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
eng = fmx.FMEngine(table, max_batch=4096)
n_steps = 64
NB = 1024
loss = torch.zeros(n_steps + 64 + NB * 8, device=dev)
for _ in range(3):
    eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, n_steps, loss)
torch.cuda.synchronize()
off = (n_steps + 63) // 64 * 64
st = loss[off:off + NB * 8].cpu().numpy().view(np.uint64).reshape(NB, 4).astype(np.int64)
n_upd = 1 + (39 * 64 + 3) // 4
n_all = n_upd + 256
t0 = st[:n_all, 0].min()
us = lambda x: (x - t0) / 100.0
u, f = st[:n_upd], st[n_upd:n_all]
print("update workgroups: start  min %.2f median %.2f max %.2f us" % (us(u[:, 0].min()), us(np.median(u[:, 0])), us(u[:, 0].max())))
print("update workgroups: arrive min %.2f median %.2f p90 %.2f max %.2f us" % (us(u[:, 1].min()), us(np.median(u[:, 1])), us(np.percentile(u[:, 1], 90)), us(u[:, 1].max())))
late = np.argsort(u[:, 1])[-8:]
print("  latest update workgroups (index: start -> arrive):", [(int(i), round(float(us(u[i, 0])), 2), round(float(us(u[i, 1])), 2)) for i in late])
print("forward workgroups: start  min %.2f median %.2f max %.2f us" % (us(f[:, 0].min()), us(np.median(f[:, 0])), us(f[:, 0].max())))
print("forward workgroups: poll ok min %.2f median %.2f max %.2f us" % (us(f[:, 1].min()), us(np.median(f[:, 1])), us(f[:, 1].max())))
print("forward workgroups: end    min %.2f median %.2f max %.2f us" % (us(f[:, 2].min()), us(np.median(f[:, 2])), us(f[:, 2].max())))
