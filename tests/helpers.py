"""Shared helpers for the tests (fixture loading, tolerant comparisons)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLASS_NAMES = ["FMAdam", "DeepFMAdam", "NFMAdam", "DeepFMOnn", "NFMOnn"]
TAGS = ["criteo39s", "tiny4"]


def load_model_fixture(name, tag):
    z = np.load(os.path.join(GOLDEN, f"{name}_{tag}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def sub(z, prefix):
    p = prefix + "/"
    return {k[len(p):]: z[k] for k in z.files if k.startswith(p)}


def assert_close(a, b, rtol=1e-5, atol=0.0, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.size == 0:
        return
    scale = max(float(np.max(np.abs(b))), 1e-30)
    err = np.abs(a - b)
    tol = atol + rtol * np.maximum(np.abs(b), 0.0)
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{a.size} off; max abs err {err.max():.3e}, "
                           f"max |ref| {scale:.3e}, worst rel {np.max(err / np.maximum(np.abs(b), 1e-30)):.3e}")


def assert_state_close(sd_a, sd_b, sd_prev=None, rtol=1e-5, what="", sign_rule=None):
    """Parameters after a step: compare the *deltas* (after - before) when the previous state is given, with an
    absolute floor of rtol * lr-sized steps, plus the values themselves.

    sign_rule = (lr, eps, g_noise): the step was p -= lr g / (|g| + eps) (the reference's fresh Adam).  Where |g| is of
    the order of eps the step is as sensitive as a sign function: a gradient that differs by fp32 cancellation noise
    g_noise moves the step by lr eps g_noise / (|g| + eps)^2.  |g| is recovered from the reference's own step
    (|delta| / lr = |g| / (|g| + eps)) and that much is added to the tolerance, element by element."""
    assert set(sd_a.keys()) == set(sd_b.keys()), f"{what}: keys differ: {set(sd_a) ^ set(sd_b)}"
    for k in sd_b:
        a = np.asarray(sd_a[k], dtype=np.float64)
        b = np.asarray(sd_b[k], dtype=np.float64)
        assert a.shape == b.shape, f"{what}/{k}: {a.shape} vs {b.shape}"
        extra = 0.0
        if sd_prev is not None and sign_rule is not None and a.size:
            lr, eps, g_noise = sign_rule
            r = np.minimum(np.abs(b - np.asarray(sd_prev[k], dtype=np.float64)) / lr, 1.0 - 1e-6)
            g = eps * r / (1.0 - r)
            extra = np.minimum(lr * eps * g_noise / (g + eps) ** 2, 2 * lr) * (r > 0)
        err = np.abs(a - b)
        tol = 1e-7 + rtol * np.abs(b) + extra
        assert not (err > tol).any(), (f"{what}/{k}: {int((err > tol).sum())}/{a.size} off; max abs err {err.max():.3e}")
        if sd_prev is not None:
            p = np.asarray(sd_prev[k], dtype=np.float64)
            da, db = a - p, b - p
            dscale = float(np.max(np.abs(db))) if db.size else 0.0
            # deltas are O(lr); fp32 cancellation in (after - before) leaves ~1e-7 * |param| of noise
            atol = max(1e-6 * dscale, 3e-7 * float(np.max(np.abs(b)) if b.size else 0))
            derr = np.abs(da - db)
            dtol = atol + 1e-4 * np.abs(db) + extra
            assert not (derr > dtol).any(), (f"{what}/{k} (delta): {int((derr > dtol).sum())}/{a.size} off; max abs err "
                                             f"{derr.max():.3e}, max |ref delta| {dscale:.3e}")



def assert_within_f64(a, ref, floor, what, rtol=1e-5):
    """|a - ref| <= 1e-5 |ref| + floor, element by element: north_star's tolerance against a float64 evaluation, with the
    fp32 rounding floor of the quantity (oracle.fm_oracle.flat_fm_step_f64 derives it term by term) where the value is a
    cancelled difference."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    err = np.abs(a - ref)
    tol = rtol * np.abs(ref) + floor
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{a.size} beyond {rtol:g} rel + fp32 floor; worst err/tol "
                           f"{float((err / np.maximum(tol, 1e-300)).max()):.2f}, max abs err {float(err.max()):.3e}")


def assert_ftrl_step_within_f64(hip, ref, what="", before=None):
    """hip: the (z, n) state after the HIP step as dict(zV [R,k], nV, zw [R], nw, zb, nb); ref: flat_fm_step_f64's result
    for the same step.  Touched rows within 1e-5 + floor of the float64 step, untouched rows bit-identical.
    before: the state the step started from -- then the STEP itself is checked too, |(hip - before) - (ref - before)| <=
    1e-5 |ref - before| + floor: at the headline batch a rarely hit row moves by ~1e-4 in z with |z| ~ 10, inside 1e-5 |z|, so
    the check on values alone would pass a skipped update; the check on the step does not (the floors are those of the
    step's own terms)."""
    u, new, fl = ref["urows"], ref["new"], ref["floor"]
    for kk in ("zV", "nV", "zw", "nw"):
        assert_within_f64(np.asarray(hip[kk])[u], new[kk][u], fl[kk], f"{what}{kk}")
        if before is not None:
            b0 = np.asarray(before[kk], np.float64)[u]
            assert_within_f64(np.asarray(hip[kk], np.float64)[u] - b0, new[kk][u] - b0, fl[kk], f"{what}{kk} (the step)")
        mask = np.ones(len(new[kk]), dtype=bool)
        mask[u] = False
        np.testing.assert_array_equal(np.asarray(hip[kk])[mask], new[kk][mask].astype(np.float32), err_msg=f"{what}{kk} untouched rows")
    assert_within_f64(hip["zb"], new["zb"], fl["zb"], f"{what}zb")
    assert_within_f64(hip["nb"], new["nb"], fl["nb"], f"{what}nb")


def oracle_float64():
    """The oracle's restatement evaluated in FLOAT64: the same source (oracle/fm_oracle.py) with its working type `f32` bound to
    numpy.float64, as a module of its own.  What an fp32 implementation approximates, free of any fp32 summation order -- the
    reference for "within 1e-5 of the float64 value + the fp32 rounding floor" checks of the paths the oracle only has in fp32
    (Hedge backprop).  Test infrastructure, like the oracle itself."""
    import types
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "fm_oracle.py")
    src = open(path).read()
    assert src.count("f32 = np.float32") == 1
    mod = types.ModuleType("fm_oracle_f64")
    mod.__file__ = path
    exec(compile(src.replace("f32 = np.float32", "f32 = np.float64"), path, "exec"), mod.__dict__)
    return mod
