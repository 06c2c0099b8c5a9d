"""Per-kernel times of one fmx_mlp_section call (3 x 256, B = 4096) from torch events around repeated calls."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
import fmx
lib = fmx._lib.load()
B, k, H, L = 4096, 16, 256, 3
n_par = sum(H * (k if l == 0 else H) + H for l in range(L))
params = (torch.randn(n_par) / 16).cuda()
grads = torch.zeros_like(params)
bi = torch.randn(B, k).cuda(); base = torch.randn(B).cuda(); y = (torch.rand(B) < 0.3).float().cuda()
m = fmx._lib.Mlp(params.data_ptr(), L, k, H, 0)
ws = torch.empty(int(lib.fmx_mlp_section_workspace_bytes(C.byref(m), B)) // 4, device="cuda")
dz = torch.empty(B, device="cuda"); gbi = torch.empty(B, k, device="cuda"); loss = torch.zeros(1, device="cuda")
st = torch.cuda.Stream()
for chain in (1, 0):
    old = lib.fmx_set_option(b"mlp_chain", chain)
    with torch.cuda.stream(st):
        def call():
            fmx._lib.check(lib.fmx_mlp_section(C.byref(m), 1, bi.data_ptr(), k, base.data_ptr(), y.data_ptr(), B, 1.0 / B, ws.data_ptr(),
                                               None, dz.data_ptr(), gbi.data_ptr(), k, grads.data_ptr(), 0.0, loss.data_ptr(), st.cuda_stream))
        for _ in range(20): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): call()
        e1.record()
        torch.cuda.synchronize()
    lib.fmx_set_option(b"mlp_chain", old)
    us = e0.elapsed_time(e1) / 200 * 1e3
    flop = 6 * B * (k * H + (L - 1) * H * H)
    print(f"mlp_chain={chain}: {us:.1f} us per fmx_mlp_section call (3 x 256, B = 4096; {3 if chain else 10} launches) = "
          f"{flop / us / 1e6:.1f} TFLOP/s = {flop / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak", flush=True)
