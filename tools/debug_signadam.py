import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fmx as F
import test_kernels_gpu as T
from oracle import fm_oracle as orc
B, k = 4096, 16
pr, t, state, outs = T.run_step_weights(F, T.MIXED_SIZES, k, B, "signadam", "sigmoid", seed=B + k)
out = outs[0][0]
u = out["urows"]; gV = out["dV"].astype(np.float64)
V_new = t.rows[:, :k].cpu().numpy()
d_hip = V_new[u] - pr["V"][u]
lr, eps = 0.01, 1e-8
cnt = np.bincount(np.searchsorted(u, pr["rows"].reshape(-1)), minlength=len(u)).astype(np.float64)
a = out["aV"] * np.sqrt(cnt)[:, None]
noise = 4e-7 * a
f = lambda g: -lr * g / (np.abs(g) + eps)
lo, hi = f(gV + noise), f(gV - noise)
ulp = 1.2e-7 * np.abs(pr["V"][u]) + 1e-6 * lr
bad = ~((d_hip >= lo - ulp) & (d_hip <= hi + ulp))
print("n bad", bad.sum())
# recompute g in float64 for the bad ones
x = np.ones((B, len(T.MIXED_SIZES)))
fw = orc.flat_forward(pr["V"], pr["w"], pr["bias"], pr["rows"], x.astype(np.float32))
for (i, d) in list(zip(*np.nonzero(bad)))[:12]:
    row = u[i]
    occ = np.nonzero(pr["rows"] == row)
    bs = occ[0]
    S = fw["S"][bs, d].astype(np.float64); dz = out["dz"][bs].astype(np.float64)
    g64 = np.sum(dz * (S - pr["V"][row, d]))
    print(f"row {row} d {d} cnt {cnt[i]:.0f} g32 {gV[i,d]:.4e} g64 {g64:.4e} aV {out['aV'][i,d]:.3e} noise {noise[i,d]:.3e} d_hip {d_hip[i,d]:.6e} f(g32) {f(gV[i,d]):.6e} f(g64) {f(g64):.6e} lo {lo[i,d]:.6e} hi {hi[i,d]:.6e} V {pr['V'][row,d]:.4f}")
