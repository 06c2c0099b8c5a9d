"""ctypes wrapper of oracle/_build/liboracle.so (oracle/fm_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
RULES = {"signadam": 0, "sgd": 1, "ftrl": 2}
LOSSES = {"logits": 1, "sigmoid": 2}


class Hyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "eps", "alpha", "beta", "l1", "l2")]


_lib = None


def available():
    return os.path.exists(LIB_PATH)


def load():
    global _lib
    if _lib is None:
        if not available():
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle`")
        _lib = C.CDLL(LIB_PATH)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int64)
        _lib.fmo_fm_step.restype = C.c_double
        _lib.fmo_fm_step.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_int64, C.c_int, ip, fp, fp, C.c_int, C.c_int,
                                     C.POINTER(Hyper), C.c_float, fp, fp]
        _lib.fmo_fm_step_mt.restype = C.c_double
        _lib.fmo_fm_step_mt.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_int64, C.c_int, ip, fp, fp, C.c_int, C.c_int,
                                        C.POINTER(Hyper), C.c_float, C.c_int]
    return _lib


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def fm_step(state, rows, x, y, loss_kind, rule, hyper, inv_b=None, threads=1):
    """Same contract as fm_oracle.flat_fm_step (state arrays are updated in place); returns the mean loss.
    threads > 1: the OpenMP form (columns of `rows` must hold disjoint row ids, as field-partitioned global ids do)."""
    lib = load()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    B, F = rows.shape
    y = np.ascontiguousarray(y, dtype=np.float32)
    x = None if x is None else np.ascontiguousarray(x, dtype=np.float32)
    h = Hyper(hyper.get("lr", 0.0), hyper.get("eps", 1e-8), hyper.get("alpha", 1.0), hyper.get("beta", 1.0),
              hyper.get("l1", 0.0), hyper.get("l2", 0.0))
    inv_b = 1.0 / B if inv_b is None else inv_b
    if rule == "ftrl":
        P0, P1, P2, P3 = state["zV"], state["zw"], state["nV"], state["nw"]
        bias = np.array([state["zb"], state["nb"]], dtype=np.float32)
    else:
        P0, P1, P2, P3 = state["V"], state["w"], None, None
        bias = np.array([state["bias"], 0.0], dtype=np.float32)
    for a in (P0, P1, P2, P3):
        assert a is None or (a.dtype == np.float32 and a.flags["C_CONTIGUOUS"])
    k = P0.shape[1]
    assert k <= 256
    if threads > 1:
        loss = lib.fmo_fm_step_mt(RULES[rule], LOSSES[loss_kind], _f(P0), _f(P1), _f(P2), _f(P3), _f(bias), P0.shape[0], k,
                                  rows.ctypes.data_as(C.POINTER(C.c_int64)), _f(x), _f(y), B, F, C.byref(h), inv_b, threads)
    else:
        loss = lib.fmo_fm_step(RULES[rule], LOSSES[loss_kind], _f(P0), _f(P1), _f(P2), _f(P3), _f(bias), P0.shape[0], k,
                               rows.ctypes.data_as(C.POINTER(C.c_int64)), _f(x), _f(y), B, F, C.byref(h), inv_b, None, None)
    if rule == "ftrl":
        state["zb"], state["nb"] = np.float32(bias[0]), np.float32(bias[1])
    else:
        state["bias"] = np.float32(bias[0])
    return np.float32(loss)
