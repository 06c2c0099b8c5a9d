"""Phases of k_mlp_chain (3 x 256, B = 4096): s_memrealtime stamps (100 MHz) after every phase, median over the workgroups.
FMX_MLP_CHAIN=2 routes the stamps into the Hedge-only part of the section's workspace."""
import os, sys, ctypes as C
os.environ["FMX_MLP_CHAIN"] = "2"
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
off = 2 * L * al(B * H * 4) + al(B * 4) + al(L * B * 4)          # bytes: acts, dH, loss_b, dzl -> loss_lb
s = ws.view(torch.uint8)[off:off + (B // 16) * 16 * 8].cpu().numpy().view(np.uint64).reshape(B // 16, 16).astype(np.int64)
names = ["start", "fwd 16->256", "fwd 256->256", "fwd 256->256", "loss + dH_L", "dgrad 256->256", "dgrad 256->256", "dgrad 256->16 + end"]
t0 = s[:, 0].min()
print("workgroup start spread: %.2f us" % ((s[:, 0].max() - t0) / 100.0))
for i in range(1, 8):
    d = (s[:, i] - s[:, i - 1]) / 100.0
    print("%-22s median %.2f us  (p90 %.2f)" % (names[i], np.median(d), np.percentile(d, 90)))
print("whole kernel: first start -> last end %.2f us; median workgroup %.2f us" % ((s[:, 7].max() - t0) / 100.0, np.median(s[:, 7] - s[:, 0]) / 100.0))
