import sys, time
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
import contextlib
work = torch.cuda.Stream()
for in_place, own in ((True, False), (True, True), (False, False), (True, False), (True, True)):
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    with (torch.cuda.stream(work) if own else contextlib.nullcontext()):
        st = PinnedBatchStager(index, label, B.BATCH, register_in_place=in_place)
        print("in place", in_place, "own stream", own, "construct %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        for p in range(5):
            t0 = time.perf_counter(); n = 0
            for a, b, c in st:
                m.update_embedding(a, b, c); n += 1
            torch.cuda.synchronize()
            print("  pass %d: %.1f us/batch" % (p, (time.perf_counter() - t0) / n * 1e6), flush=True)
        st.close()
