import sys, os, time, cProfile, pstats
sys.path.insert(0, "/root/repo/fm-for-online-recommendation_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench as B
from models.models_online_deep.fm_adam import FMAdam
from utils.data_preprocess import PinnedBatchStager
rng = np.random.default_rng(5)
N = B.BATCH * 8
index = np.stack([rng.integers(0, s, size=N) for s in B.CRITEO_SIZES], axis=1).astype(np.int32)
label = (rng.uniform(size=N) < 0.3).astype(np.int64)
m = FMAdam(B.CRITEO_SIZES, embedding_size=B.K_EMB, n=1e-4); m.strict_index_check = False
for in_place in (True, True, True):
    st = PinnedBatchStager(index, label, B.BATCH, register_in_place=in_place)
    for a, b, c in st: m.update_embedding(a, b, c)
    torch.cuda.synchronize()
    def loop(n):
        k = 0
        while k < n:
            for a, b, c in st:
                m.update_embedding(a, b, c); k += 1
    t0 = time.perf_counter(); loop(160); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("pinned in place: %s  enqueue %.1f us/batch, run %.1f us/batch" % (in_place, (t1 - t0) / 160 * 1e6, (t2 - t0) / 160 * 1e6), flush=True)
    st.close()
st = PinnedBatchStager(index, label, B.BATCH)
pr = cProfile.Profile(); pr.enable(); loop(80); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
