"""Phases of k_mlp_wgrad's workgroups (3 x 256, B = 4096): s_memrealtime stamps (100 MHz).  FMX_WGRAD_REDUCE=2 routes the
stamps into the Hedge-only part of the section's workspace."""
import os, sys, ctypes as C
os.environ["FMX_WGRAD_REDUCE"] = "2"
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
off = 2 * L * al(B * H * 4) + al(B * 4)                          # bytes: acts, dH, loss_b -> dzl, loss_lb
n_wg = 750
s = ws.view(torch.uint8)[off:off + n_wg * 16 * 8].cpu().numpy().view(np.uint64).reshape(n_wg, 16).astype(np.int64)
s = s[(s[:, 0] > 0) & (s[:, 6] > 0)]                              # the GEMM workgroups
t0 = s[:, 0].min()
print("%d GEMM workgroups; start spread: median %.2f us, max %.2f us after the first" % (len(s), np.median(s[:, 0] - t0) / 100.0, (s[:, 0].max() - t0) / 100.0))
names = ["", "first fetch + stage + barrier", "k tile 0 (MFMAs + stage + barrier)", "k tiles 1..n-1", "epilogue (LDS transpose + sc1 stores)",
         "waitcnt + ticket", "last arriver's reduction (others: nothing)"]
for i in range(1, 7):
    d = (s[:, i] - s[:, i - 1]) / 100.0
    print("%-45s median %.2f us  p90 %.2f  max %.2f" % (names[i], np.median(d), np.percentile(d, 90), d.max()))
last = s[s[:, 7] == 1]
print("last arrivers: %d, their reduction median %.2f us" % (len(last), np.median(last[:, 6] - last[:, 5]) / 100.0))
print("whole kernel: first start -> last end %.2f us; median workgroup %.2f us" % ((s[:, 6].max() - t0) / 100.0, np.median(s[:, 6] - s[:, 0]) / 100.0))
q = np.sort(s[:, 6] - t0) / 100.0
print("workgroup end times (us after first start): p10 %.1f p50 %.1f p90 %.1f max %.1f" % (q[len(q) // 10], q[len(q) // 2], q[len(q) * 9 // 10], q[-1]))
q = np.sort(s[:, 0] - t0) / 100.0
print("workgroup start times: p10 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f" % (q[len(q) // 10], q[len(q) // 2], q[len(q) * 3 // 4], q[len(q) * 9 // 10], q[-1]))
d = s[s[:, 8] > 0]
print("inside k tile 2 (wave 0): fetch issue %.2f us, MFMA loop %.2f, wait for tile 3's loads + LDS stores %.2f, barrier %.2f" % tuple(
    np.median(d[:, i] - d[:, j]) / 100.0 for i, j in ((8, 2), (9, 8), (10, 9), (11, 10))))
