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


def assert_state_close(sd_a, sd_b, sd_prev=None, rtol=1e-5, what=""):
    """Parameters after a step: compare the *deltas* (after - before) when the previous state is given, with an
    absolute floor of rtol * lr-sized steps, plus the values themselves."""
    assert set(sd_a.keys()) == set(sd_b.keys()), f"{what}: keys differ: {set(sd_a) ^ set(sd_b)}"
    for k in sd_b:
        a = np.asarray(sd_a[k], dtype=np.float64)
        b = np.asarray(sd_b[k], dtype=np.float64)
        assert a.shape == b.shape, f"{what}/{k}: {a.shape} vs {b.shape}"
        assert_close(a, b, rtol=rtol, atol=1e-7, what=f"{what}/{k}")
        if sd_prev is not None:
            p = np.asarray(sd_prev[k], dtype=np.float64)
            da, db = a - p, b - p
            dscale = float(np.max(np.abs(db))) if db.size else 0.0
            # deltas are O(lr); fp32 cancellation in (after - before) leaves ~1e-7 * |param| of noise
            assert_close(da, db, rtol=1e-4, atol=max(1e-6 * dscale, 3e-7 * float(np.max(np.abs(b)) if b.size else 0)),
                         what=f"{what}/{k} (delta)")
