"""Which of a process's streams share a hardware queue with the library's side stream?  The Zipf loop (fmx_fm_stream) on the k-th
stream the process creates, k = 1 .. 12: us per step (a loop on the side stream's queue runs behind the sorts)."""
import sys, time, os
sys.path.insert(0, "/root/repo/fm-for-online-recommendation_amd"); sys.path.insert(0, "/root/repo")
import bench as B
import torch, fmx
dev = torch.device("cuda", 0)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
hyper = fmx.Hyper(**B.HYPER)
table = fmx.FlatTable(B.CRITEO_SIZES, B.K_EMB, layout="ftrl", device=dev, ftrl=B.HYPER)
w0 = torch.randn((table.n_rows, B.K_EMB), device=dev) * 0.01
table.rows[:, :B.K_EMB] = w0
table.rows[:, table.z_offset:table.z_offset + B.K_EMB] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
eng = fmx.FMEngine(table, max_batch=B.BATCH)
idx_np, y_np = B.synth_pool(B.N_POOL, B.BATCH, B.CRITEO_SIZES, B.SEED + 7, zipf=True)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
loss = torch.zeros(200, device=dev)
out = []
for k in range(1, 13):
    work = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    run = eng.prepare_stream(hyper, "ftrl", "logits", idx_pool, y_pool, loss, stream=work)
    run(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200); torch.cuda.synchronize()
    out.append("%d: %.1f" % (k, (time.perf_counter() - t0) / 200 * 1e6))
print("stream number: us per step   " + "  ".join(out))
