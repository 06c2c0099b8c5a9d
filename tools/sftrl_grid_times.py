"""A grid of sketched-FTRL settings in one launch (fmx_sftrl_grid) against the same settings one host run after the other."""
import contextlib, io, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
from models.models_online.SFTRL_CCFM import SFTRL_CCFM
rng = np.random.default_rng(8)
n, D = 20000, 8
X = rng.standard_normal((n, D)) / np.sqrt(D)
y = X @ rng.standard_normal(D) + X[:, 0] * X[:, 1] * D
lrs = [0.001 * 1.5 ** i for i in range(16)]
ms = [2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 60, 64]
Xt, yt = torch.DoubleTensor(X), torch.DoubleTensor(y)
with contextlib.redirect_stdout(io.StringIO()):
    SFTRL_CCFM.grid(Xt, yt, "reg", lrs[:2], ms[:2])
    t0 = time.time(); res = SFTRL_CCFM.grid(Xt, yt, "reg", lrs, ms); tg = time.time() - t0
    t0 = time.time()
    for lr, m in [(lrs[0], ms[0]), (lrs[5], ms[5]), (lrs[10], ms[10]), (lrs[15], ms[15])]:
        SFTRL_CCFM(Xt, yt, "reg", lr, m).online_learning()
    th = (time.time() - t0) / 4
print(f"{len(res)} settings x {n} samples (d = {D}): one launch {tg:.2f} s = {len(res) * n / tg / 1e6:.2f} M sample-updates/s; "
      f"host {th:.2f} s per setting = {n / th / 1e3:.0f} K sample-updates/s -> the grid is {th * len(res) / tg:.0f} x one host core", flush=True)
