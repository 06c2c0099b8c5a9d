"""Pins for the two update rules the reference does not contain: FTRL-proximal and plain SGD.

No reference fixture can exist for them, so the oracle's statements (oracle.fm_oracle.ftrl_weight / ftrl_step / sgd_step,
flat_fm_step) are checked here against things that do not share code with them:

  1. a scalar, per-coordinate float statement of Algorithm 1 of McMahan et al., "Ad Click Prediction: a View from the
     Trenches" (KDD 2013), written from the paper (pure Python floats, no numpy);
  2. the identity the paper gives in section 3: with lambda1 = lambda2 = 0 FTRL-proximal IS per-coordinate online gradient
     descent  w_{t+1} = w_t - eta_t g_t  with  eta_t = alpha / (beta + sqrt(sum_{s<=t} g_s^2))  -- from any start;
  3. the closed form of the L1 dead zone under a constant gradient;
  4. the whole mini-batch step in float64 (flat_fm_step_f64, order-free) against the fp32 oracle step at north_star's
     1e-5 relative tolerance plus the fp32 rounding floor of each quantity (the same comparison the -m gpu tests make
     for the HIP kernels).
"""
import math

import numpy as np
import pytest

from oracle import fm_oracle as orc


# ---- 1. Algorithm 1, per coordinate, from the paper ----
class PaperFTRL:
    """One coordinate.  State (z_i, n_i), hyper-parameters alpha, beta, lambda1, lambda2 (paper's names)."""

    def __init__(self, alpha, beta, lam1, lam2, z=0.0, n=0.0):
        self.alpha, self.beta, self.lam1, self.lam2, self.z, self.n = alpha, beta, lam1, lam2, z, n

    def weight(self):
        if abs(self.z) <= self.lam1:
            return 0.0
        sgn = 1.0 if self.z > 0 else -1.0
        return -(self.z - sgn * self.lam1) / ((self.beta + math.sqrt(self.n)) / self.alpha + self.lam2)

    def update(self, g):
        w = self.weight()
        sigma = (math.sqrt(self.n + g * g) - math.sqrt(self.n)) / self.alpha
        self.z += g - sigma * w
        self.n += g * g


HYPERS = [dict(alpha=0.05, beta=1.0, l1=0.0, l2=0.0), dict(alpha=0.1, beta=0.5, l1=0.02, l2=0.3),
          dict(alpha=1.0, beta=2.0, l1=0.5, l2=0.0)]


@pytest.mark.parametrize("h", HYPERS)
def test_oracle_equals_the_papers_algorithm_1(h):
    rng = np.random.default_rng(5)
    C, T = 64, 200
    w0 = rng.normal(size=C) * 0.3
    w0[:8] = 0.0
    g = rng.normal(size=(T, C)) * np.exp(rng.normal(size=(T, C)))          # magnitudes over several decades
    g[:, 8:16] *= 1e-3                                                      # coordinates that live inside the L1 dead zone
    z = orc.ftrl_z_for_weight(w0, dtype=np.float64, **h)
    n = np.zeros(C)
    paper = [PaperFTRL(h["alpha"], h["beta"], h["l1"], h["l2"], z=float(z[c])) for c in range(C)]
    np.testing.assert_allclose([p.weight() for p in paper], w0, rtol=1e-13, atol=1e-15)   # the start reproduces w0
    dead = 0
    for t in range(T):
        w_or = orc.ftrl_weight(z, n, dtype=np.float64, **h)
        w_pp = np.array([p.weight() for p in paper])
        # z - sgn(z) lambda1 is a cancelled difference just outside the dead zone: float64 noise of ~1e-16 lambda1 there
        np.testing.assert_allclose(w_or, w_pp, rtol=1e-10, atol=1e-13)
        dead += int((w_pp == 0).sum())
        assert ((w_or == 0) == (w_pp == 0)).all() or np.abs(np.abs(z) - h["l1"]).min() < 1e-12   # the dead zone is exact
        z, n = orc.ftrl_step(z, n, g[t], dtype=np.float64, **h)
        for c in range(C):
            paper[c].update(float(g[t, c]))
    np.testing.assert_allclose(z, [p.z for p in paper], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(n, [p.n for p in paper], rtol=1e-12)
    if h["l1"] > 0:
        assert dead > 0                                                     # the stream did exercise the dead zone


@pytest.mark.parametrize("alpha,beta", [(0.05, 1.0), (0.5, 0.1)])
def test_ftrl_without_regularisation_is_per_coordinate_ogd(alpha, beta):
    """Known answer from the paper (section 3): lambda1 = lambda2 = 0  =>  w_{t+1} = w_t - alpha / (beta + sqrt(n_t)) g_t."""
    rng = np.random.default_rng(11)
    C, T = 32, 300
    w0 = rng.normal(size=C)
    g = rng.normal(size=(T, C)) * np.exp(rng.normal(size=(T, C)))
    h = dict(alpha=alpha, beta=beta, l1=0.0, l2=0.0)
    # float64: the identity holds to rounding
    z, n = orc.ftrl_z_for_weight(w0, dtype=np.float64, **h), np.zeros(C)
    w_ogd, n_ogd = w0.copy(), np.zeros(C)
    for t in range(T):
        z, n = orc.ftrl_step(z, n, g[t], dtype=np.float64, **h)
        n_ogd += g[t] ** 2
        w_ogd -= alpha / (beta + np.sqrt(n_ogd)) * g[t]
        np.testing.assert_allclose(orc.ftrl_weight(z, n, dtype=np.float64, **h), w_ogd, rtol=1e-10, atol=1e-12)
    # fp32 (what the kernels and the timed C port compute in): the same trajectory within 1e-5 of its scale.  z carries
    # |z| ~ |w| (beta + sqrt(n)) / alpha, so one fp32 rounding of z is eps |w|: T roundings random-walk to ~sqrt(T) eps |w|.
    z32, n32 = orc.ftrl_z_for_weight(w0.astype(np.float32), **h), np.zeros(C, np.float32)
    for t in range(T):
        z32, n32 = orc.ftrl_step(z32, n32, g[t].astype(np.float32), **h)
    w32 = orc.ftrl_weight(z32, n32, **h)
    assert w32.dtype == np.float32
    scale = np.maximum(np.abs(w_ogd), np.abs(w0))
    assert (np.abs(w32 - w_ogd) <= 1e-5 * scale + 4 * math.sqrt(T) * orc.EPS32 * scale).all()


def test_l1_dead_zone_closed_form():
    """Constant gradient g from (z, n) = (0, 0): while t |g| <= lambda1 the weight is exactly 0, so sigma w = 0 and z = t g,
    n = t g^2; the first non-zero weight is -(z - sgn(z) lambda1) / ((beta + |g| sqrt(t)) / alpha + lambda2)."""
    h = dict(alpha=0.1, beta=1.0, l1=1.0, l2=0.25)
    for g in (0.3, -0.07, 0.999):
        for dtype in (np.float64, np.float32):
            z, n = dtype(0), dtype(0)
            t_first = math.floor(h["l1"] / abs(g)) + 1
            for t in range(1, t_first + 1):
                assert orc.ftrl_weight(z, n, dtype=dtype, **h) == 0          # exactly zero inside the zone
                z, n = orc.ftrl_step(z, n, dtype(g), dtype=dtype, **h)
            tol = 1e-12 if dtype is np.float64 else 1e-5
            assert abs(float(z) - t_first * g) <= tol * abs(t_first * g)
            assert abs(float(n) - t_first * g * g) <= tol * t_first * g * g
            want = -(t_first * g - math.copysign(h["l1"], g)) / ((h["beta"] + abs(g) * math.sqrt(t_first)) / h["alpha"] + h["l2"])
            got = float(orc.ftrl_weight(z, n, dtype=dtype, **h))
            # the numerator z - sgn(z) l1 is a cancelled difference of O(l1) numbers: its fp32 floor is eps * l1
            assert want != 0 and abs(got - want) <= tol * abs(want) + (0 if dtype is np.float64 else 4 * orc.EPS32 * h["l1"] * abs(want) / abs(t_first * g - math.copysign(h["l1"], g)))


def test_sgd_is_the_textbook_rule():
    rng = np.random.default_rng(2)
    p, g = rng.normal(size=100).astype(np.float32), rng.normal(size=100).astype(np.float32)
    out = orc.sgd_step(p, g, 0.01)
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, p.astype(np.float64) - 0.01 * g.astype(np.float64), rtol=2e-7, atol=1e-9)


def _problem(sizes, k, B, seed, real_x):
    rng = np.random.default_rng(seed)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    R = int(offs[-1])
    idx = np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1)
    rows = idx + offs[:-1][None, :]
    x = (rng.uniform(0.2, 1.5, size=(B, len(sizes))) if real_x else np.ones((B, len(sizes)))).astype(np.float32)
    y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    V = (rng.normal(size=(R, k)) * 0.3).astype(np.float32)
    w = (rng.normal(size=R) * 0.3).astype(np.float32)
    return rows, x, y, V, w


def within(a, ref, floor, what):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    err = np.abs(a - ref)
    tol = 1e-5 * np.abs(ref) + floor
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{a.size} beyond 1e-5 rel + fp32 floor; worst err/tol {float((err / np.maximum(tol, 1e-300)).max()):.2f}"
    return float((err / np.maximum(tol, 1e-300)).max())


@pytest.mark.parametrize("rule,loss", [("ftrl", "logits"), ("ftrl", "sigmoid"), ("sgd", "logits"), ("signadam", "sigmoid")])
@pytest.mark.parametrize("B,real_x", [(1, False), (64, True), (1500, False)])
def test_fp32_step_vs_float64_step(rule, loss, B, real_x):
    """The fp32 oracle step (what the golden fixtures pin for the reference's rule, and what the C port times) against the
    order-free float64 step: 1e-5 relative + the rounding floor.  Small-vocabulary fields give rows hit hundreds of times
    (long sums), large ones give rows hit once."""
    sizes, k = [3, 9, 1000, 50000, 4, 17, 200, 31], 16
    rows, x, y, V, w = _problem(sizes, k, B, seed=B + len(rule), real_x=real_x)
    hyp = dict(lr=0.01, alpha=0.05, beta=1.0, l1=0.001, l2=0.01)
    h = {kk: hyp[kk] for kk in ("alpha", "beta", "l1", "l2")}
    if rule == "ftrl":
        st = dict(zV=orc.ftrl_z_for_weight(V, **h), nV=np.full_like(V, 0.1), zw=orc.ftrl_z_for_weight(w, **h),
                  nw=np.full_like(w, 0.1), zb=np.float32(0.3), nb=np.float32(0.2))
        hy = h
    else:
        st = dict(V=V.copy(), w=w.copy(), bias=np.float32(0.25))
        hy = dict(lr=hyp["lr"])
    ref = orc.flat_fm_step_f64(st, rows, x, y, loss, rule, hy)
    out = orc.flat_fm_step(st, rows, x, y, loss, rule, hy)        # mutates st: st is now the fp32 state after the step
    fl, new = ref["floor"], ref["new"]
    within(out["loss"], ref["loss"], fl["loss"], "loss")
    within(out["logit"], ref["logit"], fl["logit"], "logit")
    within(out["dz"], ref["dz"], fl["dz"], "dz")
    np.testing.assert_array_equal(out["urows"], ref["urows"])
    u = ref["urows"]
    within(out["dV"], ref["dV"], fl["dV"], "dV")
    within(out["dw"], ref["dw"], fl["dw"], "dw")
    if rule == "ftrl":
        for kk in ("zV", "nV", "zw", "nw"):
            within(st[kk][u], new[kk][u], fl[kk], kk)
        within(st["zb"], new["zb"], fl["zb"], "zb")
        within(st["nb"], new["nb"], fl["nb"], "nb")
    else:
        within(st["V"][u], new["V"][u], fl["V"], "V")
        within(st["w"][u], new["w"][u], fl["w"], "w")
        within(st["bias"], new["bias"], fl["bias"], "bias")
    # the floors are floors, not blank cheques: for the typical row-gradient coordinate the floor is below 1e-4 of the value
    # (it is the conditioning of the logit, sum_d (S_d^2 - SS_d) / 2, that an fp32 evaluation cannot beat)
    assert np.median(fl["dV"] / (np.abs(ref["dV"]) + 1e-30)) <= 1e-4
