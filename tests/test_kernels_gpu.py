"""Kernel-level parity: libfmx.so through the C ABI (ctypes) vs the numpy oracle on the same seeded inputs.
Bit-exact for the integer work (sorted occurrence lists), fp32 within the tolerances written at each check."""
import numpy as np
import pytest
import torch

from oracle import fm_oracle as orc
from helpers import assert_ftrl_step_within_f64, assert_within_f64

pytestmark = pytest.mark.gpu

CRITEO_SIZES = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887,
                632, 3, 41738, 5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86,
                44549]
MIXED_SIZES = [3, 9, 1000, 50000, 4, 17, 200, 31, 7, 2, 1]
HYP = dict(lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.001, l2=0.01)


@pytest.fixture(scope="module")
def fmx():
    import fmx as _fmx
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _fmx


def make_problem(sizes, k, B, seed, real_x=False, zipf=False):
    rng = np.random.default_rng(seed)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    R = int(offs[-1])
    if zipf:
        idx = np.stack([np.minimum(rng.zipf(1.3, size=B) - 1, s - 1) for s in sizes], axis=1).astype(np.int32)
    else:
        idx = np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1).astype(np.int32)
    x = rng.uniform(-1, 1, size=(B, len(sizes))).astype(np.float32) if real_x else None
    y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    V = (rng.normal(size=(R, k)) * 0.3).astype(np.float32)
    w = (rng.normal(size=R) * 0.3).astype(np.float32)
    bias = np.float32(0.37)
    return dict(offs=offs, R=R, idx=idx, x=x, y=y, V=V, w=w, bias=bias, rows=idx.astype(np.int64) + offs[:-1][None, :])


def weights_table(fmx, sizes, k, pr, stride=None):
    t = fmx.FlatTable(sizes, k, layout="weights", row_stride=stride)
    t.rows[:, :k] = torch.from_numpy(pr["V"]).cuda()
    t.rows[:, t.kp] = torch.from_numpy(pr["w"]).cuda()
    t.bias[0] = float(pr["bias"])
    return t


def ftrl_state(pr, hyp):
    h = dict(alpha=hyp["alpha"], beta=hyp["beta"], l1=hyp["l1"], l2=hyp["l2"])
    rng = np.random.default_rng(5)
    st = dict(zV=orc.ftrl_z_for_weight(pr["V"], **h), nV=(rng.uniform(size=pr["V"].shape) * 0.5).astype(np.float32),
              zw=orc.ftrl_z_for_weight(pr["w"], **h), nw=(rng.uniform(size=pr["w"].shape) * 0.5).astype(np.float32),
              zb=np.float32(-0.4), nb=np.float32(0.2))
    # a band of coordinates inside the L1 dead zone
    st["zV"][::7] *= 1e-4
    return st


def ftrl_table(fmx, sizes, k, st, stride=None):
    t = fmx.FlatTable(sizes, k, layout="ftrl", row_stride=stride, ftrl=HYP)
    t.load_ftrl_state(st["zV"], st["nV"], st["zw"], st["nw"])
    t.bias[0], t.bias[1] = float(st["zb"]), float(st["nb"])
    return t


def close(a, b, rtol, floor, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    err = np.abs(a - b)
    tol = rtol * np.abs(b) + floor
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (tol there {tol.flat[err.argmax()]:.3e}), n_bad {(err > tol).sum()}"


# ---------------------------------------------------------------------------------------------------------
# sort
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 5, 64, 100, 1000, 1025, 2048, 3000, 4096, 5000, 16384, 20000, 32768])
def test_sort_bit_exact(fmx, B):
    pr = make_problem(MIXED_SIZES, 4, B, seed=B)
    t = weights_table(fmx, MIXED_SIZES, 4, pr)
    eng = fmx.FMEngine(t, max_batch=B)
    idx_d, _, _ = eng.to_device(pr["idx"])
    eng.sort(idx_d)
    torch.cuda.synchronize()
    got = eng.sorted.cpu().numpy().view(np.uint32)
    Bp, bbits = eng.lib.fmx_sorted_width(B), eng.lib.fmx_sorted_bbits(B)
    assert got.shape == (len(MIXED_SIZES), Bp)
    for f in range(len(MIXED_SIZES)):
        comp = (pr["idx"][:, f].astype(np.uint32) << np.uint32(bbits)) | np.arange(B, dtype=np.uint32)
        want = np.full(Bp, 0xFFFFFFFF, dtype=np.uint32)
        want[:B] = np.sort(comp)
        np.testing.assert_array_equal(got[f], want)
    assert int(eng.error.item()) == 0


def test_chunked_sort_equals_one_workgroup_sort(fmx):
    """Wide sorts are k_sort_chunk + k_sort_merge (1,024-composite chunks spread over the chip, stable rank merge): by
    default from 8,192 composites per field on, with fmx_set_option("sort_chunked", 2) from 2,048 on; 0 keeps one workgroup
    per field at every width.  Same lists, bit for bit, including the 0xFFFFFFFF padding and an out-of-range index (which
    becomes padding and raises the flag)."""
    lib = fmx._lib.load()
    for B in (2048, 4000, 4096, 9000):
        pr = make_problem(MIXED_SIZES, 4, B, seed=B + 1, zipf=(B == 4000))
        idx = pr["idx"].copy()
        if B == 9000:
            idx[17, 2] = MIXED_SIZES[2] + 5
        t = weights_table(fmx, MIXED_SIZES, 4, pr)
        eng = fmx.FMEngine(t, max_batch=B)
        idx_d, _, _ = eng.to_device(idx)
        got = []
        for chunked in (2, 0):
            old = lib.fmx_set_option(b"sort_chunked", chunked)
            try:
                eng.sorted.fill_(7)
                eng.sort(idx_d)
                torch.cuda.synchronize()
                got.append(eng.sorted.cpu().numpy().view(np.uint32).copy())
                assert int(eng.error.item()) == (1 if B == 9000 else 0)
                eng.error.zero_()
            finally:
                lib.fmx_set_option(b"sort_chunked", old)
        np.testing.assert_array_equal(got[0], got[1])
        assert (np.diff(got[0].astype(np.int64), axis=1) >= 0).all()


def test_out_of_range_index_is_flagged(fmx):
    pr = make_problem(MIXED_SIZES, 4, 8, seed=1)
    t = weights_table(fmx, MIXED_SIZES, 4, pr)
    eng = fmx.FMEngine(t, max_batch=8)
    bad = pr["idx"].copy()
    bad[3, 0] = MIXED_SIZES[0]            # one past the end of field 0
    idx_d, _, _ = eng.to_device(bad)
    eng.forward(fmx.Hyper(**HYP), idx_d)
    with pytest.raises(IndexError):
        eng.check_error_flag()
    eng.sort(idx_d)
    with pytest.raises(IndexError):
        eng.check_error_flag()
    with pytest.raises(IndexError):
        fmx.normalize_inputs(bad, np.ones_like(bad), len(MIXED_SIZES), MIXED_SIZES)


# ---------------------------------------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [4, 10, 16, 24, 64])
@pytest.mark.parametrize("real_x", [False, True])
def test_forward_weights(fmx, k, real_x):
    B = 37
    pr = make_problem(MIXED_SIZES, k, B, seed=10 + k, real_x=real_x)
    t = weights_table(fmx, MIXED_SIZES, k, pr)
    eng = fmx.FMEngine(t, max_batch=B)
    idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
    x = pr["x"] if real_x else np.ones((B, len(MIXED_SIZES)), dtype=np.float32)
    ref = orc.flat_forward(pr["V"], pr["w"], pr["bias"], pr["rows"], x)
    for loss in ("logits", "sigmoid"):
        eng.forward(fmx.Hyper(**HYP), idx_d, xv_d, y_d, loss=loss)
        torch.cuda.synchronize()
        S = eng.S[:B, :k].cpu().numpy()
        smax = np.abs(ref["S"]).max()
        close(S, ref["S"], 1e-5, 1e-6 * smax, "S")
        assert (eng.S[:B, k:].cpu().numpy() == 0).all()
        close(eng.bi[:B, :k].cpu().numpy(), ref["bi"], 1e-5, 2e-6 * smax * smax, "bi")
        close(eng.first[:B].cpu().numpy(), ref["first"], 1e-6, 1e-8, "first")
        close(eng.sfirst[:B].cpu().numpy(), ref["sfirst"], 1e-5, 1e-6, "sfirst")
        close(eng.sbi[:B].cpu().numpy(), ref["sbi"], 1e-5, 4e-6 * smax * smax * k, "sbi")
        z = eng.logit[:B].cpu().numpy()
        close(z, ref["logit"], 1e-5, 4e-6 * smax * smax * k, "logit")
        # loss / dz are checked against the oracle evaluated at the kernel's own logit (the epilogue itself)
        close(eng.loss_b[:B].cpu().numpy(), orc.loss_value(z, pr["y"], loss), 1e-5, 1e-7, "loss")
        close(eng.dz[:B].cpu().numpy(), orc.dloss_dlogit(z, pr["y"], loss, 1.0 / B), 1e-5, 1e-9, "dz")


@pytest.mark.parametrize("k", [4, 16])
def test_forward_ftrl_layout(fmx, k):
    B = 50
    pr = make_problem(MIXED_SIZES, k, B, seed=3, real_x=True)
    st = ftrl_state(pr, HYP)
    t = ftrl_table(fmx, MIXED_SIZES, k, st)
    eng = fmx.FMEngine(t, max_batch=B)
    idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
    eng.forward(fmx.Hyper(**HYP), idx_d, xv_d, y_d, loss="logits")
    torch.cuda.synchronize()
    h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
    V = orc.ftrl_weight(st["zV"], st["nV"], **h)
    w = orc.ftrl_weight(st["zw"], st["nw"], **h)
    b = orc.ftrl_weight(st["zb"], st["nb"], **h)
    assert (V == 0).any(), "the L1 dead zone must be exercised"
    ref = orc.flat_forward(V, w, b, pr["rows"], pr["x"])
    smax = np.abs(ref["S"]).max()
    close(eng.S[:B, :k].cpu().numpy(), ref["S"], 1e-5, 2e-6 * smax, "S")
    close(eng.logit[:B].cpu().numpy(), ref["logit"], 1e-5, 4e-6 * smax * smax * k + 1e-6, "logit")


# ---------------------------------------------------------------------------------------------------------
# full step (sort + forward + update) vs the oracle's flat step
# ---------------------------------------------------------------------------------------------------------
def run_step_weights(fmx, sizes, k, B, rule, loss, seed, real_x=False, zipf=False, stride=None, n_steps=1):
    pr = make_problem(sizes, k, B, seed, real_x=real_x, zipf=zipf)
    t = weights_table(fmx, sizes, k, pr, stride)
    eng = fmx.FMEngine(t, max_batch=B)
    hyp = fmx.Hyper(**HYP)
    idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
    state = dict(V=pr["V"].copy(), w=pr["w"].copy(), bias=np.float32(pr["bias"]))
    x = pr["x"] if real_x else np.ones((B, len(sizes)), dtype=np.float32)
    outs = []
    for _ in range(n_steps):
        eng.step(hyp, rule, loss, idx_d, xv_d, y_d)
        outs.append((orc.flat_fm_step(state, pr["rows"], x, pr["y"], loss, rule, dict(lr=HYP["lr"])),
                     float(eng.loss_out.item())))
    torch.cuda.synchronize()
    eng.check_error_flag()
    return pr, t, state, outs


def check_weights_after(pr, t, state, out, rule, k):
    V_new = t.rows[:, :k].cpu().numpy()
    w_new = t.rows[:, t.kp].cpu().numpy()
    touched = np.zeros(pr["R"], dtype=bool)
    touched[out["urows"]] = True
    # rows no sample touched are bit-identical
    np.testing.assert_array_equal(V_new[~touched], pr["V"][~touched])
    np.testing.assert_array_equal(w_new[~touched], pr["w"][~touched])
    assert (t.rows[:, k:t.kp].cpu().numpy() == 0).all() and (t.rows[:, t.kp + 1:].cpu().numpy() == 0).all()
    u = out["urows"]
    gV, gw = out["dV"], out["dw"]
    dV_hip, dw_hip = V_new[u] - pr["V"][u], w_new[u] - pr["w"][u]
    dV_ref, dw_ref = state["V"][u] - pr["V"][u], state["w"][u] - pr["w"][u]
    gmax = max(np.abs(gV).max(), 1e-30)
    if rule == "sgd":
        # delta = -lr * g: 1e-5 relative plus the cancellation floor of the fp32 subtraction (after - before)
        close(dV_hip, dV_ref, 1e-5, 2e-6 * HYP["lr"] * gmax + 1.2e-7 * np.abs(pr["V"][u]), "dV")
        close(dw_hip, dw_ref, 1e-5, 2e-6 * HYP["lr"] * np.abs(gw).max() + 1.2e-7 * np.abs(pr["w"][u]), "dw")
    else:
        # p -= lr g / (|g| + eps) is sign-like, so the step of a coordinate with |g| ~ eps amplifies fp32 noise in g.
        # Any fp32 evaluation (the oracle's included) carries
        #   summation noise   ~ eps32 * sqrt(run length) * aV        (aV = sum |x G| (|S| + |e|), the rounding scale)
        #   coefficient noise ~ eps32 / B * sV                       (sigmoid(z) and 1 - p are only known to an absolute
        #                                                             6e-8, whatever the implementation; sV = sum |x|(|S|+|e|))
        # so the step is checked to be the rule applied to SOME gradient inside that noise ball around the oracle's
        # (the rule is monotone in g), and the ball is checked to be negligible for almost every coordinate.
        lr, eps = HYP["lr"], HYP["eps"]
        B = pr["idx"].shape[0]
        cnt = np.bincount(np.searchsorted(u, pr["rows"].reshape(-1)), minlength=len(u)).astype(np.float64)
        rule_f = lambda g: -lr * g / (np.abs(g) + eps)
        for d_hip, g, a, sc, p0, what in (
                (dV_hip, gV.astype(np.float64), out["aV"] * np.sqrt(cnt)[:, None], out["sV"], pr["V"][u], "dV"),
                (dw_hip, gw.astype(np.float64), out["aw"] * np.sqrt(cnt), out["sw"], pr["w"][u], "dw")):
            noise = 4e-7 * a + 2.4e-7 / B * sc
            lo, hi = rule_f(g + noise), rule_f(g - noise)          # rule_f is decreasing in g
            ulp = 2.4e-7 * np.abs(p0) + 1e-6 * lr
            inside = (d_hip >= lo - ulp) & (d_hip <= hi + ulp)
            assert inside.all(), f"{what}: {(~inside).sum()} steps outside the rule's noise band"
            assert (np.abs(g) > 4 * noise).mean() > 0.97        # ... and the band is tight almost everywhere
    close(t.bias[0].item(), state["bias"], 1e-5, 1e-7, "bias")


@pytest.mark.parametrize("rule", ["signadam", "sgd"])
@pytest.mark.parametrize("B,k,real_x", [(1, 4, False), (7, 10, True), (64, 16, False), (300, 16, True),
                                        (4096, 16, False), (2500, 10, False)])
def test_step_weights_rules(fmx, rule, B, k, real_x):
    loss = "logits" if B % 2 else "sigmoid"
    pr, t, state, outs = run_step_weights(fmx, MIXED_SIZES, k, B, rule, loss, seed=B + k, real_x=real_x)
    out, loss_hip = outs[0]
    close(loss_hip, out["loss"], 1e-5, 1e-7, "mean loss")
    check_weights_after(pr, t, state, out, rule, k)


def test_step_zipf_and_custom_stride(fmx):
    pr, t, state, outs = run_step_weights(fmx, MIXED_SIZES, 16, 1000, "sgd", "logits", seed=9, zipf=True, stride=20)
    check_weights_after(pr, t, state, outs[0][0], "sgd", 16)


@pytest.mark.parametrize("B,k", [(1, 4), (33, 16), (4096, 16)])
def test_step_ftrl(fmx, B, k):
    sizes = MIXED_SIZES
    pr = make_problem(sizes, k, B, seed=77 + B, real_x=(B == 33))
    st = ftrl_state(pr, HYP)
    st0 = {kk: np.array(v, copy=True) for kk, v in st.items()}
    t = ftrl_table(fmx, sizes, k, st)
    eng = fmx.FMEngine(t, max_batch=B)
    idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
    eng.step(fmx.Hyper(**HYP), "ftrl", "logits", idx_d, xv_d, y_d)
    torch.cuda.synchronize()
    x = pr["x"] if pr["x"] is not None else np.ones((B, len(sizes)), dtype=np.float32)
    h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
    # FTRL-proximal is not in the reference (parity unpinned by it): the check is against the float64 evaluation of the
    # same step (order-free; pinned to the paper's Algorithm 1 by tests/test_ftrl_pin.py) at north_star's 1e-5 relative,
    # plus the fp32 rounding floor of each quantity, element by element -- 1-ulp v_rcp_f32 / v_sqrt_f32 included
    ref = orc.flat_fm_step_f64(st0, pr["rows"], x, pr["y"], "logits", "ftrl", h)
    assert_within_f64(float(eng.loss_out.item()), ref["loss"], ref["floor"]["loss"], "loss")
    zV, nV, zw, nw = [a.numpy() for a in t.export_ftrl_state()]
    assert_ftrl_step_within_f64(dict(zV=zV, nV=nV, zw=zw, nw=nw, zb=t.bias[0].item(), nb=t.bias[1].item()), ref, before=st0)
    # and the fp32 oracle step (what bench.py's cpu_baseline times) sits inside the same band
    out = orc.flat_fm_step(st, pr["rows"], x, pr["y"], "logits", "ftrl", h)
    assert_ftrl_step_within_f64(st, ref, "fp32 oracle ", before=st0)
    np.testing.assert_array_equal(out["urows"], ref["urows"])


def test_update_with_network_gradient(fmx):
    """General form: G[b,d] = dz_bi[b] + gbi[b,d] with separate first-order coefficient (DeepFM / NFM callers)."""
    sizes, k, B = MIXED_SIZES, 10, 257
    pr = make_problem(sizes, k, B, seed=21, real_x=True)
    t = weights_table(fmx, sizes, k, pr)
    eng = fmx.FMEngine(t, max_batch=B)
    hyp = fmx.Hyper(**HYP)
    rng = np.random.default_rng(4)
    dz_first = (rng.normal(size=B) * 0.1).astype(np.float32)
    dz_bi = (rng.normal(size=B) * 0.1).astype(np.float32)
    gbi = np.zeros((B, t.kp), dtype=np.float32)
    gbi[:, :k] = rng.normal(size=(B, k)) * 0.1
    idx_d, xv_d, _ = eng.to_device(pr["idx"], pr["x"])
    eng.sort(idx_d)
    eng.forward(hyp, idx_d, xv_d)
    for use_dzbi in (True, False):
        t.rows[:, :k] = torch.from_numpy(pr["V"]).cuda()
        t.rows[:, t.kp] = torch.from_numpy(pr["w"]).cuda()
        t.bias[0] = float(pr["bias"])
        eng.update(hyp, "sgd", B, xv_d, torch.from_numpy(dz_first).cuda(),
                   torch.from_numpy(dz_bi).cuda() if use_dzbi else None, torch.from_numpy(gbi).cuda(), with_loss=False)
        torch.cuda.synchronize()
        fw = orc.flat_forward(pr["V"], pr["w"], pr["bias"], pr["rows"], pr["x"])
        G = gbi[:, :k] + (dz_bi[:, None] if use_dzbi else 0)
        urows, dV, dw = orc.flat_row_gradients(pr["V"], pr["rows"], pr["x"], fw["S"], dz_first, G.astype(np.float32))
        V_new = t.rows[:, :k].cpu().numpy()
        w_new = t.rows[:, t.kp].cpu().numpy()
        close(V_new[urows] - pr["V"][urows], -HYP["lr"] * dV, 1e-5,
              2e-6 * HYP["lr"] * np.abs(dV).max() + 1.2e-7 * np.abs(pr["V"][urows]), "dV")
        close(w_new[urows] - pr["w"][urows], -HYP["lr"] * dw, 1e-5,
              2e-6 * HYP["lr"] * np.abs(dw).max() + 1.2e-7 * np.abs(pr["w"][urows]), "dw")
        close(t.bias[0].item(), pr["bias"] - HYP["lr"] * dz_first.sum(dtype=np.float32), 1e-5, 1e-7, "bias")


def test_step_is_deterministic(fmx):
    res = []
    for _ in range(2):
        pr, t, state, outs = run_step_weights(fmx, MIXED_SIZES, 16, 4096, "signadam", "sigmoid", seed=5, zipf=True,
                                              n_steps=3)
        res.append((t.rows.cpu().numpy().copy(), t.bias.cpu().numpy().copy(), outs[-1][1]))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json full size: Criteo-39 vocabulary, k = 16, B = 4096
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rule", ["sgd", "ftrl"])
def test_full_size_criteo(fmx, rule):
    sizes, k, B = CRITEO_SIZES, 16, 4096
    pr = make_problem(sizes, k, B, seed=2024)
    hyp = fmx.Hyper(**HYP)
    x = np.ones((B, len(sizes)), dtype=np.float32)
    if rule == "sgd":
        pr, t, state, outs = run_step_weights(fmx, sizes, k, B, "sgd", "logits", seed=2024)
        close(outs[0][1], outs[0][0]["loss"], 1e-5, 1e-7, "loss")
        check_weights_after(pr, t, state, outs[0][0], "sgd", k)
        return
    st = ftrl_state(pr, HYP)
    st0 = {kk: np.array(v, copy=True) for kk, v in st.items()}
    t = ftrl_table(fmx, sizes, k, st)
    eng = fmx.FMEngine(t, max_batch=B)
    idx_d, _, y_d = eng.to_device(pr["idx"], None, pr["y"])
    eng.step(hyp, "ftrl", "logits", idx_d, None, y_d)
    torch.cuda.synchronize()
    h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
    # the headline rule at the headline size, against the float64 step at 1e-5 relative + the fp32 floor (see test_step_ftrl)
    ref = orc.flat_fm_step_f64(st0, pr["rows"], x, pr["y"], "logits", "ftrl", h)
    assert_within_f64(float(eng.loss_out.item()), ref["loss"], ref["floor"]["loss"], "loss")
    zV, nV, zw, nw = [a.numpy() for a in t.export_ftrl_state()]
    assert_ftrl_step_within_f64(dict(zV=zV, nV=nV, zw=zw, nw=nw, zb=t.bias[0].item(), nb=t.bias[1].item()), ref, before=st0)
    # size-independent properties: n never decreases, sortedness of every list, the cached weights are the derived ones
    assert (nV >= st0["nV"]).all()
    srt = eng.sorted.cpu().numpy().view(np.uint32)
    assert (np.diff(srt.astype(np.int64), axis=1) >= 0).all()
    u = ref["urows"]
    Vc = t.rows[:, :k].cpu().numpy()
    np.testing.assert_allclose(Vc[u], orc.ftrl_weight(zV[u], nV[u], **h), rtol=2e-6, atol=1e-7)


def test_stream_matches_repeated_steps(fmx):
    """fmx_fm_stream over a pool == the same steps issued one by one; the measuring variant (which repeats every launch,
    so its table is not compared) returns kernel times."""
    sizes, k, B, n_pool, n_steps = MIXED_SIZES, 16, 512, 3, 7
    prs = [make_problem(sizes, k, B, seed=40 + j) for j in range(n_pool)]
    hyp = fmx.Hyper(**HYP)
    t1 = weights_table(fmx, sizes, k, prs[0])
    e1 = fmx.FMEngine(t1, max_batch=B)
    losses1 = []
    for s in range(n_steps):
        pr = prs[s % n_pool]
        idx_d, _, y_d = e1.to_device(pr["idx"], None, pr["y"])
        e1.step(hyp, "signadam", "logits", idx_d, None, y_d)
        losses1.append(float(e1.loss_out.item()))
    for timed in (False, True):
        t2 = weights_table(fmx, sizes, k, prs[0])
        e2 = fmx.FMEngine(t2, max_batch=B)
        idx_pool = torch.from_numpy(np.stack([p["idx"] for p in prs])).cuda()
        y_pool = torch.from_numpy(np.stack([p["y"] for p in prs])).cuda()
        loss_out = torch.zeros(n_steps, device="cuda")
        ms = e2.stream(hyp, "signadam", "logits", idx_pool, y_pool, n_steps, loss_out, timed=timed)
        torch.cuda.synchronize()
        e2.check_error_flag()
        if timed:
            assert len(ms) == 4 and all(v > 0 for v in ms)
            assert np.isfinite(t2.rows.cpu().numpy()).all()
            continue
        np.testing.assert_array_equal(t1.rows.cpu().numpy(), t2.rows.cpu().numpy())
        np.testing.assert_array_equal(np.asarray(losses1, dtype=np.float32), loss_out.cpu().numpy())


@pytest.mark.parametrize("rule,real_x", [("ftrl", False), ("sgd", True)])
def test_split_sort_fields_equal_whole_fields(fmx, rule, real_x):
    """Large fields cut into sort pieces (fmx_table_t.sort_offsets; needed when (index, sample) would not fit 32 bits) are
    sorted and updated piece by piece: the same rows get the same sums.  A run of equal rows then sits at another offset of
    its list, so the 64-occurrence tiles group its terms differently: equal within fp32 rounding, not bit for bit (the
    float64 check of the split is test_one_exact_step_of_32768_samples_with_an_18_bit_field).  Forced here at a small batch
    (pieces of at most 700 rows), including real-valued x (the piece -> field column map)."""
    sizes, k, B = [3, 5000, 17, 50000, 2, 901], 16, 1500
    pr = make_problem(sizes, k, B, seed=33, real_x=real_x)
    hyp = fmx.Hyper(**HYP)
    res = []
    for cap in (None, 700):
        t = ftrl_table(fmx, sizes, k, ftrl_state(pr, HYP)) if rule == "ftrl" else weights_table(fmx, sizes, k, pr)
        t.sort_cap_override = cap
        eng = fmx.FMEngine(t, max_batch=B)
        assert (t._sort_split is None) == (cap is None)
        if cap is not None:
            assert t._sort_split[1].numel() == 1 + 8 + 1 + 72 + 1 + 2 and t._sort_split[2] <= 700
        idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
        losses = []
        for _ in range(3):
            eng.step(hyp, rule, "logits", idx_d, xv_d, y_d)
            losses.append(float(eng.loss_out.item()))
        eng.check_error_flag()
        res.append((t.rows.cpu().numpy(), t.bias.cpu().numpy(), losses))
    # three lr-sized steps on O(0.3) weights / O(10) z: 1e-5 of the values, and the untouched rows identical
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-6)
    assert (res[0][0] == res[1][0]).mean() > 0.9


def test_one_exact_step_of_32768_samples_with_an_18_bit_field(fmx):
    """8 GPUs x 4,096 samples is one exact step of 32,768: 15 sample bits leave 17 index bits, the Criteo list's largest
    fields need 18 -- they are split into two sort pieces automatically.  Against the float64 step."""
    sizes, k, B = [3, 176373, 19, 129683, 44549], 16, 32768
    pr = make_problem(sizes, k, B, seed=8)
    st = ftrl_state(pr, HYP)
    st0 = {kk: np.array(v, copy=True) for kk, v in st.items()}
    t = ftrl_table(fmx, sizes, k, st)
    eng = fmx.FMEngine(t, max_batch=B)
    assert t._sort_split is not None and t._sort_split[1].tolist() == [0, 1, 1, 2, 3, 4]
    idx_d, _, y_d = eng.to_device(pr["idx"], None, pr["y"])
    eng.step(fmx.Hyper(**HYP), "ftrl", "logits", idx_d, None, y_d)
    torch.cuda.synchronize()
    eng.check_error_flag()
    h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
    ref = orc.flat_fm_step_f64(st0, pr["rows"], np.ones((B, len(sizes)), np.float32), pr["y"], "logits", "ftrl", h)
    assert_within_f64(float(eng.loss_out.item()), ref["loss"], ref["floor"]["loss"], "loss")
    zV, nV, zw, nw = [a.numpy() for a in t.export_ftrl_state()]
    assert_ftrl_step_within_f64(dict(zV=zV, nV=nV, zw=zw, nw=nw, zb=t.bias[0].item(), nb=t.bias[1].item()), ref, before=st0)
    srt = eng.sorted.cpu().numpy().view(np.uint32)
    assert srt.shape[0] == 6 and (np.diff(srt.astype(np.int64), axis=1) >= 0).all()
    assert ((srt != 0xFFFFFFFF).sum(axis=1)[[1, 2]].sum()) == B          # every sample is in exactly one piece of field 1


def test_abi_rejects_bad_arguments(fmx):
    pr = make_problem(MIXED_SIZES, 4, 8, seed=1)
    t = weights_table(fmx, MIXED_SIZES, 4, pr)
    eng = fmx.FMEngine(t, max_batch=8)
    idx_d, _, y_d = eng.to_device(pr["idx"], None, pr["y"])
    with pytest.raises(fmx._lib.FmxError) as ei:
        eng.step(fmx.Hyper(**HYP), "ftrl", "logits", idx_d, None, y_d)     # FTRL rule on a weights-layout table
    assert ei.value.code == fmx._lib.ERR_ARG
    big = fmx.FlatTable([1 << 22, 5], 4)                                      # 22 index bits + 12 sample bits > 32
    e2 = fmx.FMEngine(big, max_batch=4096)                                     # ... so the engine cuts the field into sort pieces
    assert big._sort_split is not None and big._sort_split[1].tolist() == [0] * 5 + [1]
    idx = torch.zeros((4096, 2), dtype=torch.int32, device="cuda")
    e2.sort(idx)
    torch.cuda.synchronize()
    big._sort_split, big._cstruct = None, None                                 # the same table WITHOUT the split is refused
    with pytest.raises(fmx._lib.FmxError) as ei:
        e2.sort(idx)
    assert ei.value.code == fmx._lib.ERR_UNSUPPORTED


def test_data_parallel_wrapper_single_rank_equals_step(fmx):
    """fmx.DataParallelFM with the HIP backend at world size 1 is the plain step (the N > 1 sharding / gather logic is
    covered on CPU with gloo in tests/test_distributed_cpu.py)."""
    sizes, k, B = MIXED_SIZES, 16, 700
    pr = make_problem(sizes, k, B, seed=61)
    hyp = fmx.Hyper(**HYP)
    t1, t2 = weights_table(fmx, sizes, k, pr), weights_table(fmx, sizes, k, pr)
    e1, e2 = fmx.FMEngine(t1, max_batch=B), fmx.FMEngine(t2, max_batch=B)
    idx_d, _, y_d = e1.to_device(pr["idx"], None, pr["y"])
    dp = fmx.DataParallelFM(fmx.HipBackend(e2, hyp, "signadam", "logits"))
    for _ in range(3):
        e1.step(hyp, "signadam", "logits", idx_d, None, y_d)
        l1 = float(e1.loss_out.item())
        l2 = float(dp.step(idx_d, y_d)[0])
        assert l1 == l2
    np.testing.assert_array_equal(t1.rows.cpu().numpy(), t2.rows.cpu().numpy())
    np.testing.assert_array_equal(t1.bias.cpu().numpy(), t2.bias.cpu().numpy())


@pytest.mark.parametrize("F,k,B", [(1, 4, 65), (2, 8, 129), (70, 16, 257), (5, 24, 300), (3, 64, 200), (39, 7, 64), (17, 16, 8192),
                                   (9, 16, 16384), (5, 8, 32768)])
def test_step_shape_sweep(fmx, F, k, B):
    """Template paths the Criteo shape does not reach: 1 / 2 / 8 / 16 lanes per row, the generic field loop (F > 64),
    batch sizes straddling the 64-entry tiles, and the widest sorts (16 and 32 composites per thread; 32,768 is the
    maximum batch of one step)."""
    rng = np.random.default_rng(F * 1000 + k)
    sizes = [int(s) for s in rng.choice([1, 2, 3, 5, 40, 700, 20000], size=F)]
    for rule in ("sgd", "ftrl"):
        pr = make_problem(sizes, k, B, seed=F + k + B, real_x=(F % 2 == 1))
        hyp = fmx.Hyper(**HYP)
        x = pr["x"] if pr["x"] is not None else np.ones((B, F), dtype=np.float32)
        if rule == "sgd":
            pr, t, state, outs = run_step_weights(fmx, sizes, k, B, "sgd", "logits", seed=F + k + B, real_x=(F % 2 == 1))
            close(outs[0][1], outs[0][0]["loss"], 1e-5, 1e-7, "loss")
            check_weights_after(pr, t, state, outs[0][0], "sgd", k)
            continue
        st = ftrl_state(pr, HYP)
        st0 = {kk: np.array(v, copy=True) for kk, v in st.items()}
        t = ftrl_table(fmx, sizes, k, st)
        eng = fmx.FMEngine(t, max_batch=B)
        idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
        eng.step(hyp, "ftrl", "sigmoid", idx_d, xv_d, y_d)
        torch.cuda.synchronize()
        eng.check_error_flag()
        h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
        ref = orc.flat_fm_step_f64(st0, pr["rows"], x, pr["y"], "sigmoid", "ftrl", h)
        assert_within_f64(float(eng.loss_out.item()), ref["loss"], ref["floor"]["loss"], "loss")
        zV, nV, zw, nw = [a.numpy() for a in t.export_ftrl_state()]
        assert_ftrl_step_within_f64(dict(zV=zV, nV=nV, zw=zw, nw=nw, zb=t.bias[0].item(), nb=t.bias[1].item()), ref, before=st0)
        # the cached weights equal the weights derived from the stored (z, n)
        Vc = t.rows[:, :k].cpu().numpy()
        np.testing.assert_allclose(Vc, orc.ftrl_weight(zV, nV, **h), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("zipf", [False, True])
def test_inline_fixup_and_second_launch_give_identical_bits(fmx, zipf):
    """Runs that cross tiles are finished either by an in-launch hand-off (default) or by k_fm_fixup: same record order,
    so the tables must agree bit for bit over a few hundred full-size steps (and no hand-off wait may hit its bound)."""
    sizes, k, B, n_pool, n_steps = CRITEO_SIZES, 16, 4096, 4, 200
    rng = np.random.default_rng(11)
    if zipf:
        idx = np.stack([np.stack([np.minimum(rng.zipf(1.1, size=B) - 1, s - 1) for s in sizes], axis=1)
                        for _ in range(n_pool)]).astype(np.int32)
    else:
        idx = np.stack([np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1) for _ in range(n_pool)]).astype(np.int32)
    y = (rng.uniform(size=(n_pool, B)) < 0.3).astype(np.float32)
    idx_pool, y_pool = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    R = sum(sizes)
    V0 = torch.from_numpy((rng.normal(size=(R, k)) * 0.05).astype(np.float32)).cuda()
    res = []
    lib = fmx._lib.load()
    for inline in (1, 0):
        prev = lib.fmx_set_option(b"inline_fixup", inline)
        try:
            t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=HYP)
            t.rows[:, :k] = V0
            t.rows[:, t.z_offset:t.z_offset + k] = fmx.table.ftrl_z_for_weight_torch(V0, t.ftrl)
            eng = fmx.FMEngine(t, max_batch=B)
            loss = torch.zeros(n_steps, device="cuda")
            eng.stream(fmx.Hyper(**HYP), "ftrl", "logits", idx_pool, y_pool, n_steps, loss)
            torch.cuda.synchronize()
            eng.check_error_flag()
            res.append((t.rows.cpu().numpy(), t.bias.cpu().numpy(), loss.cpu().numpy()))
        finally:
            lib.fmx_set_option(b"inline_fixup", prev)
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
    assert np.isfinite(res[0][2]).all()


@pytest.mark.parametrize("k,B,rule", [(4, 37, "sgd"), (8, 300, "signadam"), (16, 1000, "ftrl"), (32, 129, "ftrl"), (64, 64, "sgd")])
def test_inline_fixup_matches_second_launch_small_shapes(fmx, k, B, rule):
    """The same identity for every lane layout (kp = 4..64), ragged batch sizes and feature values != 1, through fmx_fm_step."""
    sizes = MIXED_SIZES
    lib = fmx._lib.load()
    res = []
    for inline in (1, 0):
        prev = lib.fmx_set_option(b"inline_fixup", inline)
        try:
            pr = make_problem(sizes, k, B, seed=5, real_x=True)
            if rule == "ftrl":
                t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=HYP)
                V0 = torch.from_numpy(pr["V"]).cuda()
                t.rows[:, :k] = V0
                t.rows[:, t.z_offset:t.z_offset + k] = fmx.table.ftrl_z_for_weight_torch(V0, t.ftrl)
            else:
                t = weights_table(fmx, sizes, k, pr)
            eng = fmx.FMEngine(t, max_batch=B)
            idx_d, xv_d, y_d = eng.to_device(pr["idx"], pr["x"], pr["y"])
            losses = []
            for _ in range(5):
                eng.step(fmx.Hyper(**HYP), rule, "logits", idx_d, xv_d, y_d)
                losses.append(float(eng.loss_out.item()))
            eng.check_error_flag()
            res.append((t.rows.cpu().numpy(), t.bias.cpu().numpy(), np.asarray(losses)))
        finally:
            lib.fmx_set_option(b"inline_fixup", prev)
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("rule,k,real_x", [("signadam", 10, False), ("sgd", 4, True), ("ftrl", 16, False), ("ftrl", 8, True)])
def test_online_run_equals_single_sample_steps(fmx, rule, k, real_x):
    """fmx_fm_online_run (one wavefront walking the stream: predict, then fit, per sample) leaves the table bit-identical
    to N calls of fmx_fm_step with B = 1, and its predictions are sigmoid(logit) > 0.5 of the forward before each step."""
    sizes, N = MIXED_SIZES, 300
    pr = make_problem(sizes, k, N, seed=77, real_x=real_x, zipf=True)        # skewed: consecutive samples share rows
    loss_kind = "sigmoid"

    def table():
        if rule == "ftrl":
            t = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=HYP)
            V0 = torch.from_numpy(pr["V"]).cuda()
            t.rows[:, :k] = V0
            t.rows[:, t.z_offset:t.z_offset + k] = fmx.table.ftrl_z_for_weight_torch(V0, t.ftrl)
            return t
        return weights_table(fmx, sizes, k, pr)
    hyp = fmx.Hyper(**HYP)
    t1 = table()
    e1 = fmx.FMEngine(t1, max_batch=N)
    idx_d, xv_d, y_d = e1.to_device(pr["idx"], pr["x"] if real_x else None, pr["y"])
    pred, loss_b = e1.online_run(hyp, rule, loss_kind, idx_d, xv_d, y_d, want_loss=True)
    torch.cuda.synchronize()
    e1.check_error_flag()
    t2 = table()
    e2 = fmx.FMEngine(t2, max_batch=8)
    preds2, losses2 = [], []
    for i in range(N):
        xi = xv_d[i:i + 1] if xv_d is not None else None
        e2.forward(hyp, idx_d[i:i + 1], xi, want_first=False, want_bi=False)
        preds2.append(bool(torch.sigmoid(e2.logit[0]) > 0.5))
        e2.step(hyp, rule, loss_kind, idx_d[i:i + 1], xi, y_d[i:i + 1])
        losses2.append(float(e2.loss_out.item()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t1.rows.cpu().numpy(), t2.rows.cpu().numpy())
    np.testing.assert_array_equal(t1.bias.cpu().numpy(), t2.bias.cpu().numpy())
    np.testing.assert_array_equal(loss_b.cpu().numpy(), np.asarray(losses2, dtype=np.float32))
    assert pred.cpu().numpy().astype(bool).tolist() == preds2


def test_online_run_edge_cases(fmx):
    """N = 0 is a no-op; more fields than one wavefront holds is refused; an out-of-range index raises the error flag and
    contributes nothing; the loss-free call works."""
    k = 16
    pr = make_problem(MIXED_SIZES, k, 8, seed=2)
    t = weights_table(fmx, MIXED_SIZES, k, pr)
    eng = fmx.FMEngine(t, max_batch=8)
    hyp = fmx.Hyper(**HYP)
    idx_d, _, y_d = eng.to_device(pr["idx"], None, pr["y"])
    before = t.rows.clone()
    pred, loss = eng.online_run(hyp, "sgd", "logits", idx_d[:0], None, y_d[:0])
    torch.cuda.synchronize()
    assert pred.numel() == 0 and loss is None and torch.equal(before, t.rows)
    bad = idx_d.clone()
    bad[3, 2] = MIXED_SIZES[2] + 5
    eng.online_run(hyp, "sgd", "logits", bad, None, y_d)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        eng.check_error_flag()
    wide = fmx.FlatTable([7] * 70, k)                                  # 70 fields > 64 at kp = 16
    e2 = fmx.FMEngine(wide, max_batch=4)
    assert not e2.online_run_fits(70, wide.kp)
    idx_w = torch.zeros((2, 70), dtype=torch.int32, device="cuda")
    with pytest.raises(fmx._lib.FmxError) as ei:
        e2.online_run(hyp, "sgd", "logits", idx_w, None, torch.zeros(2, device="cuda"))
    assert ei.value.code == fmx._lib.ERR_UNSUPPORTED
