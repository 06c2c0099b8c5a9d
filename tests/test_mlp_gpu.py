"""fmx_mlp_section (the relu MLP at mini-batch sizes: fp32 MFMA GEMMs for forward, loss, backward) against a float64
PyTorch autograd reference of the same network (reference deepfm_adam.py:79-89,106-119: nn.Linear + relu + autograd).
Tolerance: fp32 sums over up to 4096 samples in a different order than the reference -> 2e-5 relative to the largest
magnitude of each output."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fmx():
    import fmx as _fmx
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _fmx


CASES = [  # B, k, kp, hidden, layers, loss
    (4096, 16, 16, 256, 3, "logits"),       # BASELINE configs[3]
    (300, 10, 12, 40, 2, "sigmoid"),        # reference embedding size, ragged tiles, the double-sigmoid loss
    (37, 4, 4, 33, 1, "logits"),            # odd hidden: the scalar-load path; one layer
    (1000, 16, 16, 64, 5, "logits"),        # the reference's depth
    (4096, 16, 20, 256, 3, "logits"),       # bi rows strided (records), gbi padded
    # the chain kernel (hidden = 256) and the streaming weight gradients away from configs[3]: a ragged last slab of 16 rows,
    # every k the chain takes, one layer (no 256 x 256 GEMM at all), the deepest network, few rows per batch split
    (1000, 32, 32, 256, 2, "sigmoid"),
    (37, 48, 48, 256, 1, "logits"),
    (530, 64, 64, 256, 4, "logits"),
    (100, 16, 16, 256, 8, "logits"),
    (2048, 16, 16, 128, 3, "logits"),       # GEMM launches for the chain's part, streaming weight gradients (128-wide layers)
    (700, 24, 24, 72, 3, "logits"),         # narrow first layer of 24 columns (two natural tiles), 72-wide layers (two blocks, ragged)
]


def reference(params, k, H, L, loss, bi, base, y, inv_b):
    p = params.double().cpu()
    bi = bi.double().cpu().requires_grad_(True)
    base = base.double().cpu().requires_grad_(True)
    Ws, bs, off = [], [], 0
    for l in range(L):
        i = k if l == 0 else H
        Ws.append(p[off:off + H * i].view(H, i).clone().requires_grad_(True)); off += H * i
        bs.append(p[off:off + H].clone().requires_grad_(True)); off += H
    x = bi
    for W, b in zip(Ws, bs):
        x = F.relu(x @ W.t() + b)
    out = base + x.sum(1)
    z = torch.sigmoid(out) if loss == "sigmoid" else out
    ls = F.binary_cross_entropy_with_logits(z, y.double().cpu(), reduction="sum") * inv_b
    ls.backward()
    flat = torch.cat([t.grad.reshape(-1) for pair in zip(Ws, bs) for t in pair])
    return float(ls.detach()), base.grad.numpy(), bi.grad.numpy(), flat.numpy(), out.detach().numpy()


def close(a, b, what, rel=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B,k,kp,H,L,loss", CASES)
def test_mlp_section_vs_autograd(fmx, B, k, kp, H, L, loss):
    torch.manual_seed(B + H + L)
    n_par = sum(H * (k if l == 0 else H) + H for l in range(L))
    # the double-sigmoid loss multiplies by p (1 - p), p = sigmoid(z): with 256 relu outputs summed into z, N(0, 1 / H) weights put
    # p within 1e-8 of 1 where fp32 cancels (the reference's fp32 does too; float64 is then a different function): smaller weights
    # there keep the comparison about the kernels
    wscale = 0.25 if (loss == "sigmoid" and H >= 256) else 1.0
    params = (torch.randn(n_par) * (wscale / np.sqrt(H))).cuda()
    bi_full = torch.zeros(B, kp)
    bi_full[:, :k] = torch.randn(B, k) * 0.5
    bi_d = bi_full.cuda()
    base = (torch.randn(B) * 0.3).cuda()
    y = (torch.rand(B) < 0.3).float().cuda()
    inv_b = 1.0 / B
    grads = torch.zeros_like(params)
    import ctypes as C
    lib = fmx._lib.load()
    m = fmx._lib.Mlp(params.data_ptr(), L, k, H, 0)
    ws = torch.empty(int(lib.fmx_mlp_section_workspace_bytes(C.byref(m), B)) // 4, device="cuda")
    dz = torch.empty(B, device="cuda")
    gbi = torch.full((B, kp), 7.0, device="cuda")
    logit = torch.empty(B, device="cuda")
    loss_out = torch.zeros(1, device="cuda")
    p0 = params.clone()
    fmx._lib.check(lib.fmx_mlp_section(C.byref(m), fmx._lib.LOSSES[loss], bi_d.data_ptr(), kp, base.data_ptr(), y.data_ptr(), B,
                                       inv_b, ws.data_ptr(), logit.data_ptr(), dz.data_ptr(), gbi.data_ptr(), kp,
                                       grads.data_ptr(), 0.0, loss_out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    r_loss, r_dz, r_gbi, r_flat, r_out = reference(params, k, H, L, loss, bi_d[:, :k], base, y, inv_b)
    close(logit.cpu().numpy(), r_out, "logit")
    close(loss_out.item(), r_loss, "loss")
    close(dz.cpu().numpy(), r_dz, "dz")
    close(gbi.cpu().numpy()[:, :k], r_gbi, "gbi")
    assert (gbi.cpu().numpy()[:, k:] == 0).all(), "padding columns of gbi must be zeroed"
    close(grads.cpu().numpy(), r_flat, "flat gradients")
    assert torch.equal(params, p0), "lr_apply = 0 must leave the parameters alone"
    # determinism + the fused SGD application
    grads2 = torch.zeros_like(params)
    fmx._lib.check(lib.fmx_mlp_section(C.byref(m), fmx._lib.LOSSES[loss], bi_d.data_ptr(), kp, base.data_ptr(), y.data_ptr(), B,
                                       inv_b, ws.data_ptr(), None, dz.data_ptr(), gbi.data_ptr(), kp,
                                       grads2.data_ptr(), 0.25, loss_out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(grads, grads2), "two runs must give identical bits"
    np.testing.assert_array_equal((p0 - 0.25 * grads).cpu().numpy(), params.cpu().numpy())


def test_mlp_section_rejects_bad_arguments(fmx):
    import ctypes as C
    lib = fmx._lib.load()
    params = torch.zeros(100, device="cuda")
    m = fmx._lib.Mlp(params.data_ptr(), 9, 4, 8, 0)           # too many layers
    assert lib.fmx_mlp_section_workspace_bytes(C.byref(m), 16) < 0
    m = fmx._lib.Mlp(params.data_ptr(), 1, 4, 8, 0)
    x = torch.zeros(64, device="cuda")
    rc = lib.fmx_mlp_section(C.byref(m), fmx._lib.LOSSES["logits"], x.data_ptr(), 2, x.data_ptr(), x.data_ptr(), 4, 0.25,
                             x.data_ptr(), None, x.data_ptr(), x.data_ptr(), 4, x.data_ptr(), 0.0, None, None)
    assert rc == fmx._lib.ERR_SHAPE                                # ld_bi < k


@pytest.mark.parametrize("B,k,kp,H,L", [(4096, 16, 16, 256, 3), (300, 10, 12, 40, 2), (50, 4, 4, 33, 4)])
def test_mlp_hedge_section_vs_autograd(fmx, B, k, kp, H, L):
    """fmx_mlp_hedge_section against a float64 autograd statement of Hedge backprop (reference deepfm_onn.py:109-154)."""
    import ctypes as C
    torch.manual_seed(B + H)
    n_par = sum(H * (k if l == 0 else H) + H for l in range(L))
    # small enough that sigmoid(base + sum of H activations) stays away from 1.0f: where fp32 saturates, BCELoss clamps
    # log(1 - p) at -100 and a float64 reference (which does not saturate there) stops being a reference
    params = (torch.randn(n_par) * min(1.0 / np.sqrt(H), 2.0 / H)).cuda()
    bi = torch.zeros(B, kp)
    bi[:, :k] = torch.randn(B, k) * 0.5
    bi_d, base = bi.cuda(), (torch.randn(B) * 0.3).cuda()
    y = (torch.rand(B) < 0.3).float().cuda()
    alpha = torch.full((L,), 1.0 / (L + 1)).cuda()
    lr, hb, hs = 0.05, 0.99, 0.2
    # float64 reference
    p = params.double().cpu()
    Ws, bs, off = [], [], 0
    for l in range(L):
        i = k if l == 0 else H
        Ws.append(p[off:off + H * i].view(H, i).clone().requires_grad_(True)); off += H * i
        bs.append(p[off:off + H].clone().requires_grad_(True)); off += H
    x, losses = bi_d[:, :k].double().cpu(), []
    for W, b in zip(Ws, bs):
        x = F.relu(x @ W.t() + b)
        losses.append(F.binary_cross_entropy(torch.sigmoid(base.double().cpu() + x.sum(1)), y.double().cpu()))
    a0 = alpha.double().cpu()
    (a0 * torch.stack(losses)).sum().backward()
    new = torch.cat([(t - lr * t.grad).reshape(-1) for pair in zip(Ws, bs) for t in pair]).detach().numpy()
    a1 = torch.maximum(a0 * torch.pow(torch.tensor(hb, dtype=torch.float64), torch.stack(losses).detach()),
                       torch.tensor(hs / L, dtype=torch.float64))
    a1 = (a1 / a1.sum()).numpy()
    lib = fmx._lib.load()
    m = fmx._lib.Mlp(params.data_ptr(), L, k, H, 0)
    ws = torch.empty(int(lib.fmx_mlp_section_workspace_bytes(C.byref(m), B)) // 4, device="cuda")
    grads, lout, p0 = torch.zeros_like(params), torch.zeros(L, device="cuda"), params.clone()
    fmx._lib.check(lib.fmx_mlp_hedge_section(C.byref(m), lr, hb, hs, alpha.data_ptr(), bi_d.data_ptr(), kp, base.data_ptr(),
                                             y.data_ptr(), B, ws.data_ptr(), grads.data_ptr(), lout.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    close(lout.cpu().numpy(), torch.stack(losses).detach().numpy(), "per-layer losses")
    close(alpha.cpu().numpy(), a1, "alpha")
    close((params - p0).cpu().numpy(), new - p.numpy(), "parameter deltas", rel=5e-5)
    np.testing.assert_array_equal((p0 - lr * grads).cpu().numpy(), params.cpu().numpy())

