"""The RCCL calls of a multi-GPU step on a ONE-GPU box: one rank, backend nccl, FMX_FORCE_COLLECTIVES=1 makes fmx.owner /
fmx.distributed issue their all-gathers and the all-to-all even though the world is one rank.  Checks that 20 steps through
FieldOwnerFM (prefetch tokens, the bench's loop) and through DataParallelFM end bit-identical to fmx_fm_stream on one table.
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/nccl_world1_check.py"""
import os, sys
os.environ["FMX_FORCE_COLLECTIVES"] = "1"
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)
import fmx
from fmx.owner import FieldOwnerFM, HipOwnerBackend
import bench as B
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
hyper = fmx.Hyper(**B.HYPER)
sizes, K, BATCH, NP, STEPS = B.CRITEO_SIZES, B.K_EMB, B.BATCH, 4, 20
idx_np, y_np = B.synth_pool(NP, BATCH, sizes, 77)
idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)

def init(t):
    g = torch.Generator(device=dev).manual_seed(5)
    w0 = torch.randn((t.n_rows, K), generator=g, device=dev) * 0.01
    t.rows[:, :K] = w0
    t.rows[:, t.z_offset:t.z_offset + K] = fmx.table.ftrl_z_for_weight_torch(w0, t.ftrl)

# reference: the one-GPU loop
ref = fmx.FlatTable(sizes, K, layout="ftrl", device=dev, ftrl=B.HYPER)
init(ref)
eng = fmx.FMEngine(ref, max_batch=BATCH)
loss_ref = torch.zeros(STEPS, device=dev)
eng.stream(hyper, "ftrl", "logits", idx_pool, y_pool, STEPS, loss_ref)
torch.cuda.synchronize()

# field owners, one rank, collectives forced
obe = HipOwnerBackend(sizes, K, hyper, "ftrl", "logits", 0, 1, ftrl=B.HYPER, device=dev, max_local_batch=BATCH)
init(obe.table)
fo = FieldOwnerFM(obe)
assert fo._force
work = torch.cuda.Stream(device=dev)
losses = []
with torch.cuda.stream(work):
    tokens = {d: fo.prefetch(idx_pool[d % NP]) for d in range(2)}
    for s in range(STEPS):
        out = fo.step(idx_pool[s % NP], y_pool[s % NP], tokens.pop(s, None))
        losses.append(out.clone())
        if s + 2 < STEPS:
            tokens[s + 2] = fo.prefetch(idx_pool[(s + 2) % NP])
torch.cuda.synchronize()
obe.e.check_error_flag()
lo = torch.cat([l.reshape(-1)[:1] for l in losses])
assert torch.equal(obe.table.rows, ref.rows), "field owners over RCCL (one rank): rows differ from the one-GPU loop"
assert torch.equal(lo, loss_ref), "losses differ"
print("field owners, nccl, 1 rank, forced collectives: %d steps bit-identical to fmx_fm_stream; last loss %.6f" % (STEPS, float(lo[-1])), flush=True)

# the same through the one-call-per-step entry points with the library's own RCCL communicator, collectives forced
from fmx.owner import NativeOwnerFM
obe2 = HipOwnerBackend(sizes, K, hyper, "ftrl", "logits", 0, 1, ftrl=B.HYPER, device=dev, max_local_batch=BATCH)
init(obe2.table)
nat = NativeOwnerFM(obe2, force_collectives=True, stream=work)
losses = []
with torch.cuda.stream(work):
    tokens = {d: nat.prefetch(idx_pool[d % NP]) for d in range(2)}
    for s in range(STEPS):
        out = nat.step(idx_pool[s % NP], y_pool[s % NP], tokens.pop(s, None))
        losses.append(out.clone())
        if s + 2 < STEPS:
            tokens[s + 2] = nat.prefetch(idx_pool[(s + 2) % NP])
torch.cuda.synchronize()
obe2.e.check_error_flag()
lo = torch.cat([l.reshape(-1)[:1] for l in losses])
assert torch.equal(obe2.table.rows, ref.rows), "fmx_owner_step over RCCL (one rank): rows differ from the one-GPU loop"
assert torch.equal(lo, loss_ref), "fmx_owner_step: losses differ"
print("fmx_owner_step, own RCCL communicator, 1 rank, forced collectives: %d steps bit-identical to fmx_fm_stream" % STEPS, flush=True)
del nat

# replicated mode
t2 = fmx.FlatTable(sizes, K, layout="ftrl", device=dev, ftrl=B.HYPER)
init(t2)
dp = fmx.DataParallelFM(fmx.HipBackend(fmx.FMEngine(t2, max_batch=BATCH), hyper, "ftrl", "logits"))
with torch.cuda.stream(work):
    tok = None
    for s in range(STEPS):
        nxt = dp.prefetch(idx_pool[(s + 1) % NP]) if s + 1 < STEPS else None
        out = dp.step(idx_pool[s % NP], y_pool[s % NP], tok)
        tok = nxt
torch.cuda.synchronize()
assert torch.equal(t2.rows, ref.rows), "replicated mode over RCCL (one rank): rows differ"
print("replicated mode, nccl, 1 rank, forced collectives: bit-identical; last loss %.6f" % float(out.reshape(-1)[0]), flush=True)
dist.barrier()
dist.destroy_process_group()
