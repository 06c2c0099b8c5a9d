"""Host fp64 loop against fmx_sftrl_run on the same stream (sketched FTRL, d = 8 and d = 32): seconds per run and per sample."""
import contextlib, io, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
from models.models_online.SFTRL_CCFM import SFTRL_CCFM
for D, m, n in ((8, 4, 20000), (32, 16, 20000), (32, 64, 20000)):
    rng = np.random.default_rng(D)
    X = rng.standard_normal((n, D)) / np.sqrt(D)
    y = X @ rng.standard_normal(D) + X[:, 0] * X[:, 1] * D
    res = {}
    for dev in ("gpu", "gpu", "host"):
        mdl = SFTRL_CCFM(torch.DoubleTensor(X), torch.DoubleTensor(y), "reg", 0.05, m, device=dev)
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.time(); p, _, _ = mdl.online_learning(); res[dev] = (time.time() - t0, p)
    err = np.abs(res["gpu"][1] - res["host"][1]).max() / np.abs(res["host"][1]).max()
    print(f"d = {D}, m = {m}, {n} samples: host {res['host'][0]:.3f} s ({res['host'][0] / n * 1e6:.1f} us/sample), "
          f"gpu {res['gpu'][0]:.3f} s ({res['gpu'][0] / n * 1e6:.1f} us/sample), max prediction difference {err:.1e} of the largest", flush=True)
