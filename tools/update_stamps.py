"""Inside ONE k_fm_update launch of the steady loop: s_memrealtime stamps (100 MHz) of every tile wave -- start, list arrived, loads arrived,
row updates issued, (closing tiles) crossing run done.  Needs the diagnostic build: tools/update_stamps.sh builds it with -DFMX_STAMPS."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd")); sys.path.insert(0, ROOT)
import fmx, bench
lib = fmx._lib.load()
dev = torch.device("cuda", 0)
sizes = bench.CRITEO_SIZES
hyper = fmx.Hyper(**bench.HYPER)
table = fmx.FlatTable(sizes, 16, layout="ftrl", device=dev, ftrl=bench.HYPER)
w0 = torch.randn((table.n_rows, 16), device=dev) * 0.01
table.rows[:, :16] = w0
table.rows[:, table.z_offset:table.z_offset + 16] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
eng = fmx.FMEngine(table, max_batch=4096)
idx_np, y_np = bench.synth_pool(16, 4096, sizes, 1)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
loss = torch.zeros(2048, device=dev)
work = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
run = eng.prepare_stream(hyper, "ftrl", "logits", idx_pool, y_pool, loss, stream=work)
tiles = 39 * 64
for n_steps in (600, 603, 605):          # the last launch of each run: different positions inside a sort group
    run(n_steps)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8192 * 6))()
    assert lib.fmx_debug_update_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 6)[:tiles].astype(np.float64) / 100.0      # us
    t0 = st[:, 0].min()
    rel = st - t0
    closing = st[:, 4] > st[:, 0]
    def q(a): return "p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(a, [10, 50, 90, 100]))
    print(f"--- last update of a {n_steps}-step run ({tiles} tile waves; us from the first wave's start) ---")
    print("wave start          ", q(rel[:, 0]))
    print("list arrived        ", q(rel[:, 1]), "  (list round trip p50 %.2f)" % np.median(rel[:, 1] - rel[:, 0]))
    print("loads arrived       ", q(rel[:, 2]), "  (S + rows after the list p50 %.2f)" % np.median(rel[:, 2] - rel[:, 1]))
    print("row updates issued  ", q(rel[:, 3]), "  (rule + stores p50 %.2f)" % np.median(rel[:, 3] - rel[:, 2]))
    if closing.any():
        print("closing tiles done  ", q(rel[closing, 4]), f"  ({int(closing.sum())} tiles; hand-off p50 %.2f)" % np.median(rel[closing, 4] - rel[closing, 3]))
    per_field = [(f, rel[f * 64:(f + 1) * 64, 3].max(), rel[f * 64:(f + 1) * 64, 4].max() if closing[f * 64:(f + 1) * 64].any() else 0.0) for f in range(39)]
    worst = sorted(per_field, key=lambda t: -max(t[1], t[2]))[:6]
    print("latest fields (rows, last update issued, last closing tile):", [(sizes[f], round(a, 2), round(b, 2)) for f, a, b in worst])
