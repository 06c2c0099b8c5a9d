"""Which part of bench.py's secondary section slows the Zipf probe that follows it?  (per-chunk times of the probe's loop)"""
import sys, argparse, time
sys.path.insert(0, "/root/repo/fm-for-online-recommendation_amd"); sys.path.insert(0, "/root/repo")
import torch, bench as B, fmx
import torch.distributed as dist
dev = torch.device("cuda", 0)
def zipf(tag, zf=True):
    torch.cuda.empty_cache()
    hyper = fmx.Hyper(**B.HYPER)
    table = fmx.FlatTable(B.CRITEO_SIZES, B.K_EMB, layout="ftrl", device=dev, ftrl=B.HYPER)
    w0 = torch.randn((table.n_rows, B.K_EMB), device=dev) * 0.01
    table.rows[:, :B.K_EMB] = w0
    table.rows[:, table.z_offset:table.z_offset + B.K_EMB] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
    eng = fmx.FMEngine(table, max_batch=B.BATCH)
    idx_np, y_np = B.synth_pool(B.N_POOL, B.BATCH, B.CRITEO_SIZES, B.SEED + 7, zipf=zf)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
    loss = torch.zeros(100, device=dev)
    work = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    run = eng.prepare_stream(hyper, "ftrl", "logits", idx_pool, y_pool, loss, stream=work)
    run(100); torch.cuda.synchronize()
    ch = []
    for _ in range(8):
        t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); ch.append((time.perf_counter() - t0) / 100 * 1e6)
    eng.check_error_flag()
    with torch.cuda.stream(work):
        ms = eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, 64, timed=True)
    torch.cuda.synchronize()
    tag = tag + " [sort %.1f fwd %.1f upd %.1f]" % tuple(m / 64 * 1e3 for m in ms[:3])
    print("%-62s %s us/step | rows @%x ws @%x S @%x" % (tag, " ".join("%.1f" % c for c in ch[:4]), table.rows.data_ptr(), eng.workspace.data_ptr(), eng.S.data_ptr()), flush=True)
big = [min(s_ * 64, (1 << 20) - 1) for s_ in B.CRITEO_SIZES]
zipf("fresh zipf")
B.fm_loop_probe(fmx, torch, dev, big, zipf=False)
zipf("after hbm: zipf")
zipf("zipf 3")
B.fm_loop_probe(fmx, torch, dev, B.CRITEO_SIZES, zipf=False)
zipf("after uniform probe: zipf")
zipf("zipf 5")
