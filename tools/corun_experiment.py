"""Timing experiment: k_mlp_wgrad_stream beside k_fm_update on two unordered streams against the two in sequence on one stream.
Needs a DIAGNOSTIC build of the library in which fmx_mlp_section can launch the gradient kernel alone:
    cd fm-for-online-recommendation_amd/csrc && make FLAGS_EXTRA=-DFMX_MLP_EXPERIMENTS   (or add the define to FLAGS by hand)
with the product library the "wgrad" lines time the whole section."""
import os, sys, ctypes as C, time
os.environ["FMX_EXP_ONLY_WGRAD"] = "1"
sys.path.insert(0, "/root/repo/fm-for-online-recommendation_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch, fmx, bench as Bn
lib = fmx._lib.load()
dev = torch.device("cuda", 0)
B, k, H, L = 4096, 16, 256, 3
n_par = sum(H * (k if l == 0 else H) + H for l in range(L))
params = (torch.randn(n_par) / 16).cuda(); grads = torch.zeros_like(params)
bi = torch.randn(B, k).cuda(); base = torch.randn(B).cuda(); y = (torch.rand(B) < 0.3).float().cuda()
m = fmx._lib.Mlp(params.data_ptr(), L, k, H, 0)
ws = torch.zeros(int(lib.fmx_mlp_section_workspace_bytes(C.byref(m), B)) // 4, device="cuda")
dz = torch.empty(B, device="cuda"); gbi = torch.zeros(B, k, device="cuda"); loss = torch.zeros(1, device="cuda")
table = fmx.FlatTable(Bn.CRITEO_SIZES, k, layout="weights", device=dev)
table.rows[:, :k] = torch.randn((table.n_rows, k), device=dev) * 0.01
eng = fmx.FMEngine(table, max_batch=B)
idx_np, y_np = Bn.synth_pool(1, B, Bn.CRITEO_SIZES, 3)
idx_d = torch.from_numpy(idx_np[0]).to(dev)
hyper = fmx.Hyper(lr=1e-3)
eng.sort(idx_d); eng.forward(hyper, idx_d, None, want_first=False, want_bi=True)
dzz = torch.randn(B, device=dev) * 1e-4
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
def upd(st):
    eng.update(hyper, "sgd", B, None, dzz, dzz, gbi, inv_b=1.0 / B, with_loss=False, stream=st)
def wgrad(st):
    fmx._lib.check(lib.fmx_mlp_section(C.byref(m), 1, bi.data_ptr(), k, base.data_ptr(), y.data_ptr(), B, 1.0 / B, ws.data_ptr(),
                                       None, dz.data_ptr(), gbi.data_ptr(), k, grads.data_ptr(), 0.0, loss.data_ptr(), st.cuda_stream))
def timed(f, n=300):
    for _ in range(30): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("update alone           %.1f us" % timed(lambda: upd(sa)))
print("wgrad alone            %.1f us" % timed(lambda: wgrad(sa)))
print("both, one stream       %.1f us" % timed(lambda: (upd(sa), wgrad(sa))))
print("both, two streams      %.1f us" % timed(lambda: (upd(sa), wgrad(sb))))
