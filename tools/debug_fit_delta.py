"""Find the elements of the DeepFMAdam fit delta that differ from the golden fixture and show their gradient context."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fm-for-online-recommendation_amd"), os.path.join(ROOT, "tests")]
from helpers import load_model_fixture, sub
from test_models_gpu import build, sd_np
name = sys.argv[1] if len(sys.argv) > 1 else "DeepFMAdam"
z, meta = load_model_fixture(name, "criteo39s")
m = build(name, meta, meta["B2"])
m.load_state_dict(sub(z, "A/sd0"))
m.update_embedding(z["A/Xi1"].tolist(), z["A/Xv1"].tolist(), z["A/Y1"].tolist())
m.update_embedding(z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist())
sd2g = sub(z, "A/sd2")
sd2 = sd_np(m)
m.fit(z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist())
got, ref = sd_np(m), sub(z, "A/sd3")
Xi = z["A/Xi2"]
for k in ref:
    a, b, p = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64), np.asarray(sd2g[k], np.float64)
    da, db = a - p, b - p
    bad = np.argwhere(np.abs(da - db) > 5e-7)
    for idx in bad[:5]:
        idx = tuple(idx)
        f = int(k.split(".")[1]) if "embeddings" in k else -1
        rows_in_batch = Xi[:, f].reshape(-1).tolist() if f >= 0 else []
        print(k, idx, "ours delta", da[idx], "ref delta", db[idx], "value", b[idx], "row occurrences in batch", rows_in_batch.count(idx[0]),
              "our sd2 vs golden sd2", np.asarray(sd2[k], np.float64)[idx] - p[idx])
