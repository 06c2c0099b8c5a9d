"""Phases of k_mlp_wgrad's workgroups (3 x 256, B = 4096): s_memrealtime stamps (100 MHz), FMX_MLP_CHAIN=3 routes them into the
Hedge-only part of the section's workspace (8 per workgroup: start, first tile staged, tiles 0 / 1 / 3 / last done, end, XCC id)."""
import os, sys, ctypes as C
os.environ["FMX_MLP_CHAIN"] = "3"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import fmx
lib = fmx._lib.load()
B, k, H, L = 4096, 16, 256, 3
n_par = sum(H * (k if l == 0 else H) + H for l in range(L))
params = (torch.randn(n_par) / 16).cuda(); grads = torch.zeros_like(params)
bi = torch.randn(B, k).cuda(); base = torch.randn(B).cuda(); y = (torch.rand(B) < 0.3).float().cuda()
m = fmx._lib.Mlp(params.data_ptr(), L, k, H, 0)
ws = torch.zeros(int(lib.fmx_mlp_section_workspace_bytes(C.byref(m), B)) // 4, device="cuda")
dz = torch.empty(B, device="cuda"); gbi = torch.empty(B, k, device="cuda"); loss = torch.zeros(1, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    fmx._lib.check(lib.fmx_mlp_section(C.byref(m), 1, bi.data_ptr(), k, base.data_ptr(), y.data_ptr(), B, 1.0 / B, ws.data_ptr(), None, dz.data_ptr(),
                                       gbi.data_ptr(), k, grads.data_ptr(), 0.0, loss.data_ptr(), st))
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
off = 2 * L * al(B * H * 4) + al(B * 4)          # bytes: acts, dH, loss_b -> dzl
n_wg = 768
s = ws.view(torch.uint8)[off:off + n_wg * 8 * 8].cpu().numpy().view(np.uint64).reshape(n_wg, 8).astype(np.int64)
live = s[:, 0] > 0
t0 = s[live, 0].min()
dur = (s[:, 4] - s[:, 0]) / 100.0
heavy = live & (dur > np.median(dur[live]) * 0.6)
print("workgroups: %d live, %d heavy; start spread %.2f us; last end %.2f us" % (live.sum(), heavy.sum(), (s[live, 0].max() - t0) / 100.0, (s[live, 4].max() - t0) / 100.0))
h = s[heavy]
names = ["start -> first 8 steps done", "the other steps", "LDS hand-over + barrier", "sum + stores"]
for i in range(1, 5):
    d = (h[:, i] - h[:, i - 1]) / 100.0
    print("%-28s median %.2f us  p10 %.2f  p90 %.2f  max %.2f" % (names[i - 1], np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max()))
print("heavy workgroup, start -> end: median %.2f us, max %.2f; light: median %.2f" % (np.median(dur[heavy]), dur[heavy].max(), np.median(dur[live & ~heavy]) if (live & ~heavy).any() else 0))
hw = s[:, 7] >> 8
cu = ((s[:, 7] & 0xF) << 8) | (((hw >> 13) & 0x7) << 4) | ((hw >> 8) & 0xF)     # (XCC, SE, CU)
import collections
cnt = collections.Counter(cu[heavy].tolist())
print("heavy workgroups per CU: %s over %d CUs" % (dict(collections.Counter(cnt.values())), len(cnt)))
cnt2 = collections.Counter(cu[live].tolist())
print("all workgroups per CU: %s over %d CUs" % (dict(collections.Counter(cnt2.values())), len(cnt2)))
