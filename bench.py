#!/usr/bin/env python3
"""bench.py -- online FM + FTRL-proximal on synthetic Criteo-39 (BASELINE.json configs[1]/[2]).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one mini-batch: sort the batch's occurrences, gather + bi-interaction
forward with the fused BCE loss, row-reduced backward + fused FTRL-proximal update of every touched row.  Inputs (a
pool of batches) and the table are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# Hardware queues per process (must be set before the HIP runtime starts).  The runtime's default is 4; with it two of the streams
# of the multi-GPU step (main, prefetch, RCCL's) share a queue and a prefetched sort waits behind the step it should overlap (77 vs
# 110 us per 16,384-sample step, replicated mode rehearsed on one GPU), and the Python paths that keep a torch side stream beside
# the work stream read a quarter of their rate (trainer step 13.7 vs 51 M samples/s).  With 8 the FOURTH and ELEVENTH stream a
# process creates land on the queue of the library's side stream and a loop there runs behind the sorts (89 instead of 21 us per
# step; with 16 queues every fourth stream; tools/queue_alias.py) -- hence work_stream() below: ONE stream for every loop.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(ROOT, "fm-for-online-recommendation_amd"))
sys.path.insert(0, ROOT)

CRITEO_SIZES = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887,
                632, 3, 41738, 5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86,
                44549]   # reference main_experiment.py:56-58; sum = 1,006,628 rows
K_EMB, BATCH, N_POOL, SEED = 16, 4096, 16, 20240922
HYPER = dict(lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.0, l2=1e-4)
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per sample (DESIGN.md "Byte accounting"; fp32, int32 indices, 17 coordinates per row)
F = len(CRITEO_SIZES)
BYTES_STEP_FTRL = 4 * F + 2 * (2 * 68 * F) + 8                       # 10,772: SURVEY.md section 8(d)
BYTES_K_FORWARD = 4 * F + 2 * 68 * F + 4 + 68 + 4                    # idx + (z,n) rows + y + S,dz out + loss
BYTES_K_UPDATE = 4 * F + 2 * (2 * 68 * F) + 68                       # sorted list + rows read + rows written + S,dz in
BYTES_K_SORT = 4 * F + 4 * F                                         # idx in, sorted out


def synth_pool(n_pool, B, sizes, seed, zipf=False):
    """Counter-based (Philox) synthetic Criteo-shaped stream: batch j is a pure function of (seed, j)."""
    idx = np.empty((n_pool, B, len(sizes)), dtype=np.int32)
    y = np.empty((n_pool, B), dtype=np.float32)
    for j in range(n_pool):
        rng = np.random.Generator(np.random.Philox(key=seed + j))
        for f, s in enumerate(sizes):
            if zipf:
                idx[j, :, f] = np.minimum(rng.zipf(1.05, size=B) - 1, s - 1)
            else:
                idx[j, :, f] = rng.integers(0, s, size=B)
        y[j] = (rng.uniform(size=B) < 0.3)
    return idx, y


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


_WORK = {}


def work_stream(torch, dev):
    """ONE stream of its own for every loop of this process (not torch's legacy default stream, which is ordered against every
    blocking stream and slow to enqueue on).  One, not one per probe: HIP spreads the streams of a process over its hardware queues,
    and a loop whose stream lands on the queue of the library's side stream runs BEHIND the sorts it should run beside -- with
    GPU_MAX_HW_QUEUES = 8 the fourth and the eleventh stream a process created did: 32-89 instead of 21 us per step
    (tools/queue_alias.py, tools/zipf_trace.sh)."""
    key = (dev.type, dev.index)
    if key not in _WORK:
        _WORK[key] = torch.cuda.Stream(device=dev)
    return _WORK[key]


def mlp_section_probe(fmx, torch, dev, B=4096, k=16, H=256, L=3, reps=200):
    """fmx_mlp_section alone (BASELINE configs[3]'s network at the bench's batch): HIP events on the launch stream around `reps`
    back-to-back calls -> the `roofline` object of secondary.deepfm (bound: the fp32 MFMA peak, 157.3 TFLOP/s)."""
    import ctypes as C
    lib = fmx._lib.load()
    n_par = sum(H * (k if l == 0 else H) + H for l in range(L))
    params = (torch.randn(n_par, device=dev) / 16)
    grads = torch.zeros_like(params)
    bi, base = torch.randn(B, k, device=dev), torch.randn(B, device=dev)
    y = (torch.rand(B, device=dev) < 0.3).float()
    m = fmx._lib.Mlp(params.data_ptr(), L, k, H, 0)
    ws = torch.empty(int(lib.fmx_mlp_section_workspace_bytes(C.byref(m), B)) // 4, device=dev)
    dz, gbi, loss = torch.empty(B, device=dev), torch.empty(B, k, device=dev), torch.zeros(1, device=dev)
    st = work_stream(torch, dev)
    torch.cuda.synchronize()

    def call():
        fmx._lib.check(lib.fmx_mlp_section(C.byref(m), 1, bi.data_ptr(), k, base.data_ptr(), y.data_ptr(), B, 1.0 / B, ws.data_ptr(),
                                           None, dz.data_ptr(), gbi.data_ptr(), k, grads.data_ptr(), 0.0, loss.data_ptr(), st.cuda_stream))
    with torch.cuda.stream(st):
        for _ in range(20):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flop = 6 * B * (k * H + (L - 1) * H * H)
    tf = flop / us / 1e6
    return {"bound": "mfma", "scope": "fmx_mlp_section alone: forward + loss + dgrad chain, weight gradients, their reduction (3 launches), "
            f"B = {B}, 16-256-256-256", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "section_us": us,
            "flop_per_call": flop, "how": f"HIP events around {reps} back-to-back calls on the launch stream", "traffic": None}


def fm_loop_probe(fmx, torch, dev, sizes, zipf, steps=400, warm=100):
    """A secondary FM + FTRL stream through the same loop as the headline (fmx_fm_stream, prepared call, own stream): another
    vocabulary (the HBM-resident companion) or Zipf(1.05) indices.  -> dict(value, ms_per_step, ...)."""
    hyper = fmx.Hyper(**HYPER)
    table = fmx.FlatTable(sizes, K_EMB, layout="ftrl", device=dev, ftrl=HYPER)
    g = torch.Generator(device=dev).manual_seed(SEED)
    w0 = torch.randn((table.n_rows, K_EMB), generator=g, device=dev) * 0.01
    table.rows[:, :K_EMB] = w0
    table.rows[:, table.z_offset:table.z_offset + K_EMB] = fmx.table.ftrl_z_for_weight_torch(w0, table.ftrl)
    del w0
    eng = fmx.FMEngine(table, max_batch=BATCH)
    idx_np, y_np = synth_pool(N_POOL, BATCH, sizes, SEED + 7, zipf=zipf)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)
    loss = torch.zeros(max(steps, warm), device=dev)
    work = work_stream(torch, dev)
    torch.cuda.synchronize()
    run = eng.prepare_stream(hyper, "ftrl", "logits", idx_pool, y_pool, loss, stream=work)
    run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check_error_flag()
    R = int(sum(sizes))
    return {"value": steps * BATCH / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warm,
            "rows": R, "table_GB": R * table.row_stride * 4 / 1e9, "indices": "Zipf(1.05)" if zipf else "uniform",
            "frac_of_8TBps": steps * BATCH / dt * BYTES_STEP_FTRL / 1e9 / HBM_PEAK_GBPS}


def cpu_baseline(idx_pool, y_pool, sizes, seconds=12.0):
    """The oracle's flat FTRL step, timed on this host on a bounded sample of the same stream: the C port
    (oracle/fm_oracle.c; its OpenMP form on every hardware thread the process may use, and one thread beside it) when
    oracle/_build/liboracle.so is present, else the numpy port."""
    from oracle import fm_oracle as orc
    from oracle import c_oracle
    rng = np.random.default_rng(1)
    R = int(sum(sizes))
    h = {k: HYPER[k] for k in ("alpha", "beta", "l1", "l2")}
    V0 = (rng.normal(size=(R, K_EMB)) * 0.01).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1].astype(np.int64)
    use_c = c_oracle.available()
    x = None if use_c else np.ones(idx_pool.shape[1:], dtype=np.float32)
    rows_pool = [idx_pool[j].astype(np.int64) + offs[None, :] for j in range(idx_pool.shape[0])]

    def run(threads, seconds, max_steps):
        st = dict(zV=orc.ftrl_z_for_weight(V0, **h), nV=np.zeros((R, K_EMB), np.float32), zw=np.zeros(R, np.float32),
                  nw=np.zeros(R, np.float32), zb=np.float32(0), nb=np.float32(0))
        n, t0 = 0, time.perf_counter()
        while True:
            j = n % len(rows_pool)
            if use_c:
                c_oracle.fm_step(st, rows_pool[j], None, y_pool[j], "logits", "ftrl", HYPER, threads=threads)
            else:
                orc.flat_fm_step(st, rows_pool[j], x, y_pool[j], "logits", "ftrl", h)
            n += 1
            if time.perf_counter() - t0 > seconds or n >= max_steps:
                break
        return n, time.perf_counter() - t0
    B = idx_pool.shape[1]
    n1, dt1 = run(1, seconds / 2, 128)
    if not use_c:
        return dict(value=n1 * B / dt1, unit="samples/s", cores=1, kind="port",
                    sample=f"{n1} steps of B={B} of the same synthetic stream through oracle/fm_oracle.py flat_fm_step (numpy)")
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 32)                              # the update has 39-way parallelism; a one-GPU box's CPU share is 16
    run(cores, 1.0, 8)                                  # thread pool warm-up
    nm, dtm = run(cores, seconds / 2, 2048)
    dense = None
    try:                                                # what the reference's own formulation costs
        from oracle import dense_mode
        rate, thr = dense_mode.time_dense_steps(sizes, K_EMB, idx_pool[:4], y_pool[:4], n_steps=3, lr=HYPER["lr"], threads=cores)
        dense = {"value": rate, "unit": "samples/s", "cores": thr,
                 "sample": "3 steps of B=4096 through oracle/dense_mode.py: 78 nn.Embedding tables, dense gradients and a fresh "
                           "torch.optim.Adam over all 11 M parameters per step (the reference's formulation, its sign rule)"}
    except Exception as exc:                            # informational only
        dense = {"error": repr(exc)}
    return dict(reference_faithful_dense=dense, value=nm * B / dtm, unit="samples/s", cores=cores, kind="port",
                sample=f"{nm} steps of B={B} of the same synthetic stream through oracle/fm_oracle.c fmo_fm_step_mt (plain C, "
                       f"gcc -O2 -fopenmp, {cores} threads: forward over samples, update over the 39 fields)",
                one_thread={"value": n1 * B / dt1, "unit": "samples/s", "cores": 1,
                            "sample": f"{n1} steps through fmo_fm_step (1 thread)"})


def bench_deepfm(args, fmx, torch, dist, world, rank, dev, rehearsal):
    """BASELINE configs[3]: online DeepFM (bi-interaction + 3 x 256 relu MLP, SGD lr 1e-3) on the same synthetic Criteo
    stream, exact data parallelism (fmx.DeepFMTrainer).  The MLP section is fmx_mlp_section (fp32 MFMA GEMMs);
    FMX_MLP_NATIVE=0 runs it through PyTorch autograd instead (rocBLAS; FMX_MLP_GRAPH=1 replays it as a graph)."""
    import torch.nn as nn
    lr, hidden, n_layers = 1e-3, 256, 3
    torch.manual_seed(SEED)                                   # identical MLP replicas on every rank
    layers = [nn.Linear(K_EMB if j == 0 else hidden, hidden).to(dev) for j in range(n_layers)]
    owners = world > 1 and args.mp_mode == "owner"
    if owners:
        # N > 1: the scalable mode -- tables and update work sharded over field owners, the MLP replicated; per step one
        # all-to-all + one record all-gather on the critical path, ONE all-reduce of the MLP gradients beside the table update
        from fmx.owner import HipOwnerBackend
        obe = HipOwnerBackend(CRITEO_SIZES, K_EMB, fmx.Hyper(lr=lr), "sgd", "logits", rank, world, device=dev, max_local_batch=BATCH)
        table, eng = obe.table, obe.e
        tr = fmx.OwnerDeepFMTrainer(obe, layers, K_EMB, mlp_lr=lr)
    else:
        table = fmx.FlatTable(CRITEO_SIZES, K_EMB, layout="weights", device=dev)
        eng = fmx.FMEngine(table, max_batch=BATCH * world)
        tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, fmx.Hyper(lr=lr), "sgd"), layers, K_EMB, table.kp, mlp_lr=lr,
                               use_graph=os.environ.get("FMX_MLP_GRAPH", "1") == "1",
                               native_mlp=os.environ.get("FMX_MLP_NATIVE", "1") == "1")
    g = torch.Generator(device=dev).manual_seed(SEED + rank)
    table.rows[:, :K_EMB] = torch.randn((table.n_rows, K_EMB), generator=g, device=dev) * 0.01
    idx_np, y_np = synth_pool(N_POOL, BATCH, CRITEO_SIZES, SEED + 1000 * rank, zipf=args.zipf)
    idx_pool, y_pool = torch.from_numpy(idx_np).to(dev), torch.from_numpy(y_np).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # one GPU: the steps of the pool issued from ONE foreign call (fmx_deepfm_stream) -- through the Python trainer the step is bound
    # by its host side (84 us of calls per step for 67 us of kernels); that path is timed beside it as `through_trainer_step`
    native_loop = None
    import contextlib
    work = work_stream(torch, dev) if dev.type == "cuda" else None
    on_work = (lambda: torch.cuda.stream(work)) if work is not None else contextlib.nullcontext
    if work is not None:
        torch.cuda.synchronize()                              # the tables and the pool were written on torch's current stream
    if world == 1 and not owners and getattr(tr, "native", False) and os.environ.get("FMX_DEEPFM_STREAM", "1") == "1":
        loop_losses = torch.zeros(max(min(args.steps, 100), 10), device=dev)
        native_loop = tr.prepare_stream(idx_pool, y_pool, loss_out=loop_losses, stream=work)

    def run(n, first=0):
        out = None
        if native_loop is not None:
            native_loop(n)
            return loop_losses[n - 1]
        if owners:                                            # the index all-gather and the sort of the owned pieces run two steps ahead
            tokens = {d: tr.prefetch(idx_pool[(first + d) % N_POOL]) for d in range(min(2, n))}
            for s in range(n):
                j = (first + s) % N_POOL
                out = tr.step(idx_pool[j], y_pool[j], tokens.pop(s, None))
                if s + 2 < n:
                    tokens[s + 2] = tr.prefetch(idx_pool[(first + s + 2) % N_POOL])
            tr.finish()
            return out
        with on_work():                                        # (not torch's legacy default stream: every launch there is ordered against
            for s in range(n):                                 # every blocking stream of the process, a cross-stream wait each)
                out = tr.step(idx_pool[(first + s) % N_POOL], y_pool[(first + s) % N_POOL])
        return out
    steps, warm = min(args.steps, 100), min(args.warmup, 10)
    run(warm)
    barrier()
    t0 = time.perf_counter()
    last = run(steps, warm)
    barrier()
    dt = time.perf_counter() - t0
    eng.check_error_flag()
    loss = last.clone()
    through_trainer = None
    if native_loop is not None:                               # the same steps through DeepFMTrainer.step, for the record
        native_loop = None
        run(warm)
        barrier()
        t1 = time.perf_counter()
        run(steps, warm)
        barrier()
        through_trainer = steps * BATCH / (time.perf_counter() - t1)
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        ll = loss.cpu() if rehearsal else loss
        dist.all_reduce(ll)
        loss = ll
        dist.destroy_process_group()
    if rank != 0:
        return None
    mlp_params = sum(p.numel() for layer in layers for p in layer.parameters())
    flops = 6 * BATCH * (K_EMB * hidden + (n_layers - 1) * hidden * hidden)       # fwd + dgrad + wgrad of the three GEMMs
    return ({
        "metric": "samples/sec online-DeepFM (Criteo-39-field, k=16, 3x256 relu MLP) SGD", "value": steps * BATCH * world / dt,
        "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "online DeepFM fwd+bwd, fused SGD row update of the tables, "
                               + ("fp32-MFMA" if getattr(tr, "native", True) else "PyTorch") + " MLP 16-256-256-256 "
                               f"({mlp_params} parameters, {mlp_params * 4} B all-reduced per step when N > 1), synthetic "
                               f"Criteo-39 (R=1,006,628, k=16, B={BATCH} per GPU); BASELINE.json configs[3]",
                   "global_batch": BATCH * world,
                   "parallelism": "1 GPU" if world == 1 else
                   (f"field owners x{world}: tables and update work sharded, MLP replicated; per step an all-to-all of partial sums and ONE "
                    "all-gather of (S, dlogit, dL/dbi) records on the critical path, ONE all-reduce of the MLP gradients beside the table "
                    "update, the index all-gather ahead of time (exact)" if owners else
                    f"dp{world}: all-gather of (idx, S, dz, dL/dbi) + one all-reduce of the MLP gradients (exact, replicated tables)")},
        "mlp_section": {"flop_per_step": flops, "note": "3 x (forward + dgrad + wgrad) fp32-MFMA GEMMs; TFLOP/s over the WHOLE step "
                        "(tables included) is a lower bound of the section's rate", "TFLOPs_whole_step": flops / (dt / steps) / 1e12,
                        "frac_of_fp32_mfma_peak_whole_step": flops / (dt / steps) / 1e12 / 157.3},
        "final_loss": float(loss),
        **({"issued_by": "fmx_deepfm_stream: one foreign call for the timed steps", "through_trainer_step_samples_per_s": through_trainer}
           if through_trainer is not None else {})})


PUBLISHED_ONLINE = {"FMAdam": 39.1, "NFMAdam": 35.9, "NFMOnn": 27.9, "DeepFMAdam": 27.4, "DeepFMOnn": 18.4}


def bench_online(args, torch):
    """The reference's own published protocol (from its saved result pickle): run_experiment =
    predict + fit per sample, B = 1, Criteo vocabulary, k = 10, 5 x 10 MLP, 2,500 samples per batch -- through the drop-in
    classes, whose run_experiment runs the loop on the device.  value = the class named by --model; the others ride along."""
    import _experiment as ex
    N = 2500
    rng = np.random.default_rng(0)
    Xi = np.stack([rng.integers(0, s, size=N) for s in ex.feature_sizes], axis=1).tolist()
    Xv = [[1] * 39 for _ in range(N)]
    Y = (rng.uniform(size=N) < 0.7).astype(int).tolist()          # the 7:3 stream of main_experiment_2.py:33
    torch.manual_seed(0)
    rates = {}
    for model in ex.build_models():
        name = str(model).split("-")[0]
        model.run_experiment(Xi[:50], Xv[:50], Y[:50])            # warm-up
        torch.cuda.synchronize()
        best = 0.0
        for _ in range(max(1, min(args.steps, 5))):
            t0 = time.perf_counter()
            model.run_experiment(Xi, Xv, Y)
            torch.cuda.synchronize()
            best = max(best, N / (time.perf_counter() - t0))
        rates[name] = best
        del model
    v = rates[args.model]
    print(json.dumps({
        "metric": f"samples/sec online loop (predict + fit per sample, B=1) {args.model}", "value": v, "unit": "samples/s",
        "n_gpus": 1, "steps": N, "warmup": 50, "ms_per_step": 1e3 / v, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": v / PUBLISHED_ONLINE[args.model], "dtype": "f32", "data": "synthetic",
        "config": {"workload": "run_experiment of the reference's five classes on 2,500 synthetic Criteo-shaped samples "
                               "(39 fields, reference vocabulary, k=10, 5x10 MLP, n=1e-4), nested-list inputs as the reference "
                               "takes them, host conversion and the H2D copy inside the timed region",
                   "published": PUBLISHED_ONLINE, "published_hardware": "unstated (BASELINE.md section 1)"},
        "all_classes": rates, "vs_baseline_all": {k: rates[k] / PUBLISHED_ONLINE[k] for k in rates}}))


def bench_class_surface(torch, n_steps=80):
    """Mini-batch rate THROUGH THE DROP-IN CLASS (FMAdam.update_embedding, B = 4096, Criteo vocabulary, k = 16, the reference's
    rule): nested lists as the reference takes them (reference fm_adam.py:35-36 converts them per call, and so must we) against
    the array path fed by utils.data_preprocess.PinnedBatchStager (pinned int32 [B, 39], non-blocking copies, double-buffered)."""
    from models.models_online_deep.fm_adam import FMAdam
    from utils.data_preprocess import PinnedBatchStager
    rng = np.random.default_rng(5)
    N = BATCH * 8
    index = np.stack([rng.integers(0, s, size=N) for s in CRITEO_SIZES], axis=1).astype(np.int32)
    label = (rng.uniform(size=N) < 0.3).astype(np.int64)
    torch.manual_seed(0)
    m = FMAdam(CRITEO_SIZES, embedding_size=K_EMB, n=1e-4)
    m.strict_index_check = False
    out = {}
    # on a stream of its own: on torch's legacy default stream every launch is ordered against every other blocking stream of the
    # process (the copy stream of the stager, this bench's work stream): a cross-stream wait per launch (18 against 48 M samples/s)
    torch.cuda.synchronize()
    own = work_stream(torch, torch.device("cuda", torch.cuda.current_device()))
    with torch.cuda.stream(own):
        return _class_surface_body(torch, m, index, label, n_steps, out)


def _class_surface_body(torch, m, index, label, n_steps, out):
    from utils.data_preprocess import PinnedBatchStager
    Xi, Xv, Y = index[:BATCH].tolist(), [[1] * F for _ in range(BATCH)], label[:BATCH].tolist()
    m.update_embedding(Xi, Xv, Y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.update_embedding(Xi, Xv, Y)
    torch.cuda.synchronize()
    out["nested_lists_samples_per_s"] = 3 * BATCH / (time.perf_counter() - t0)
    def rate(stager):
        for _ in range(3):                                 # warm-up passes (the first pays one-time costs: ~1-2 ms per batch)
            for idx_d, xv_d, y_d in stager:
                m.update_embedding(idx_d, xv_d, y_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while n < n_steps:
            for idx_d, xv_d, y_d in stager:
                m.update_embedding(idx_d, xv_d, y_d)
                n += 1
        m.check_index_flag()
        torch.cuda.synchronize()
        return n * BATCH / (time.perf_counter() - t0)
    # staged (the default: a typed copy into pinned buffers per batch -- what a stream seen ONCE costs) and pinned in place (no host
    # pass; only for data passed over repeatedly, as here: the first copy out of freshly pinned pages costs ~0.8 ms per batch)
    st1 = PinnedBatchStager(index, label, BATCH)
    out["pinned_arrays_samples_per_s"] = rate(st1)
    st2 = PinnedBatchStager(index, label, BATCH, register_in_place=True, copy_stream=st1._copy)   # (one copy stream: see work_stream)
    out["pinned_in_place_repeated_passes_samples_per_s"] = rate(st2)
    st2.close()
    out["note"] = ("FMAdam.update_embedding at B = 4096 through the Python class; the typed copy into pinned memory (staged) or none "
                   "(pinned in place), the H2D copy and the step inside the timed region; one fmx_fm_step call (3 launches) per batch, "
                   "Python-bound")
    return out


def stream_read_probe(fmx, torch, dev):
    """HBM-read ceiling on this GPU, same run: 16-byte loads over a 4 GiB buffer (fmx_stream_read), 5 timed passes."""
    probe = torch.empty(1 << 30, dtype=torch.float32, device=dev)      # 4 GiB
    probe.normal_()
    sink = torch.zeros(1, device=dev)
    lib = fmx._lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        fmx._lib.check(lib.fmx_stream_read(probe.data_ptr(), probe.numel() * 4, sink.data_ptr(), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fmx._lib.check(lib.fmx_stream_read(probe.data_ptr(), probe.numel() * 4, sink.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
    gbps = 5 * probe.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    # informational gather ceilings: random 64-byte and 128-byte rows from a 64 MB window (Infinity-Cache resident)
    # and from the whole 4 GiB (HBM resident), 4 M rows per launch
    global GATHER_CEILINGS
    GATHER_CEILINGS = {}
    n_read = 4 << 20
    for name, nbytes in (("64MB", 64 << 20), ("4GB", probe.numel() * 4)):
        for rb in (64, 128):
            for _ in range(2):
                fmx._lib.check(lib.fmx_gather_read(probe.data_ptr(), nbytes, rb, n_read, 1, sink.data_ptr(), st))
            e0.record()
            for i in range(5):
                fmx._lib.check(lib.fmx_gather_read(probe.data_ptr(), nbytes, rb, n_read, 7 + i, sink.data_ptr(), st))
            e1.record()
            torch.cuda.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / 5
            GATHER_CEILINGS[f"rows_{rb}B_from_{name}"] = {"Grows_per_s": n_read / sec / 1e9, "GBps": n_read * rb / sec / 1e9}
    del probe
    torch.cuda.empty_cache()
    return gbps


GATHER_CEILINGS = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--zipf", action="store_true", help="Zipf(1.05) indices instead of uniform (secondary workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary.deepfm block (BASELINE configs[3])")
    ap.add_argument("--loop-only", action="store_true", help="only the timed online loop (no measuring pass, no spread pass): "
                                                             "what tools/profile_round.sh traces for the in-loop kernel durations")
    ap.add_argument("--row-stride", type=int, default=0)
    ap.add_argument("--owner-path", default="native", choices=["native", "torch"],
                    help="--gpus N, field owners: fmx_owner_step (one C call per step, own RCCL communicator) or torch.distributed collectives")
    ap.add_argument("--mp-mode", default="owner", choices=["owner", "replicated"],
                    help="N > 1: 'owner' = field-owner mode (fmx.owner: table and update work shard over the ranks), "
                         "'replicated' = every rank keeps the whole table and repeats the global update (fmx.DataParallelFM)")
    ap.add_argument("--model", default="FMAdam", choices=sorted(PUBLISHED_ONLINE), help="--workload online: the class `value` reports")
    ap.add_argument("--workload", default="fm", choices=["fm", "deepfm", "online"],
                    help="fm: the headline metric (BASELINE configs[1]+[2]); deepfm: configs[3], bi-interaction + 3x256 relu "
                         "MLP, SGD, data-parallel with one fused all-reduce of the dense gradients")
    ap.add_argument("--rule", default="ftrl", choices=["ftrl", "sgd", "signadam"],
                    help="update rule (the headline metric is ftrl; the others are for kernel comparisons)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import fmx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # FMX_BENCH_REHEARSAL=1: every rank uses GPU 0 and gloo (to rehearse the N > 1 code path on a one-GPU box; the
    # numbers of such a run mean nothing)
    rehearsal = os.environ.get("FMX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- resident state: FTRL table (z, n) with V ~ N(0, 0.01) folded into z, first-order weights 0 ----
    if args.workload == "online":
        return bench_online(args, torch)
    if args.workload == "deepfm":
        out = bench_deepfm(args, fmx, torch, dist, world, rank, dev, rehearsal)
        if out is not None:
            print(json.dumps(out))
        return
    RULE = args.rule
    hyper = fmx.Hyper(**HYPER)
    owner_mode = world > 1 and args.mp_mode == "owner"

    def init_rows(t, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        w0 = torch.randn((t.n_rows, K_EMB), generator=g, device=dev) * 0.01
        t.rows[:, :K_EMB] = w0
        if RULE == "ftrl":      # n = 0 and the z that reproduces V (first-order weights and bias start at 0)
            zo = t.z_offset
            t.rows[:, zo:zo + K_EMB] = fmx.table.ftrl_z_for_weight_torch(w0, t.ftrl)
    if owner_mode:
        # the field-owner mode (fmx.owner): every rank holds only the rows of its fields
        from fmx.owner import FieldOwnerFM, HipOwnerBackend
        obe = HipOwnerBackend(CRITEO_SIZES, K_EMB, hyper, RULE, "logits", rank, world, ftrl=HYPER, device=dev, max_local_batch=BATCH)
        init_rows(obe.table, SEED + rank)
        table, eng = obe.table, obe.e
    else:
        table = fmx.FlatTable(CRITEO_SIZES, K_EMB, layout="ftrl" if RULE == "ftrl" else "weights", device=dev,
                              row_stride=args.row_stride if args.row_stride else None, ftrl=HYPER)
        init_rows(table, SEED)
        eng = fmx.FMEngine(table, max_batch=BATCH * world)
    idx_np, y_np = synth_pool(N_POOL, BATCH, CRITEO_SIZES, SEED + 1000 * rank, zipf=args.zipf)
    idx_pool = torch.from_numpy(idx_np).to(dev)
    y_pool = torch.from_numpy(y_np).to(dev)
    loss_buf = torch.zeros(max(args.steps, args.warmup, 100), device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    kernel_ms = None
    short_run = None
    stream_gbps = None
    if os.environ.get("FMX_BENCH_PROBE_FIRST", "0") == "1" and rank == 0:
        stream_gbps = stream_read_probe(fmx, torch, dev)
    if world == 1:
        # ---- one GPU: the online loop of fmx_fm_stream over the resident pool.  Warm-up, then EXACTLY K timed steps ----
        # the loop runs on a stream of its own (the legacy default stream is slow to enqueue on) through a prepared call: the
        # structs, pointers and the stream handle are bound once, a call is one foreign call (FMEngine.prepare_stream)
        work = work_stream(torch, dev)
        barrier()
        run_loop = eng.prepare_stream(hyper, RULE, "logits", idx_pool, y_pool, loss_buf, stream=work)
        run_loop(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_loop(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        eng.check_error_flag()
        losses = loss_buf[:args.steps].cpu().numpy()
        assert np.isfinite(losses).all(), "non-finite loss in the timed region"
        # what a 20-step call costs (the driver's round-end run is --steps 20 --warmup 5: pipeline start included)
        if args.steps != 20 and not args.loop_only:
            ts = []
            for _ in range(5):
                barrier()
                tc = time.perf_counter()
                run_loop(20)
                barrier()
                ts.append(time.perf_counter() - tc)
            short_run = {"steps_per_call": 20, "calls": 5, "median_ms_per_step": float(np.median(ts)) / 20 * 1e3,
                         "median_samples_per_s": 20 * BATCH / float(np.median(ts))}
        # spread of the step time: 20 separately timed chunks of 100 steps (each chunk pays one pipeline start)
        chunk_us = []
        for _ in range(20 if args.steps >= 200 and not args.loop_only else 0):
            barrier()
            tc = time.perf_counter()
            run_loop(100)
            barrier()
            chunk_us.append((time.perf_counter() - tc) / 100 * 1e6)
        # the measuring pass (fmx.h, fmx_fm_stream with kernel_ms): groups of 8 steps -- one sort launch, the 8 forwards
        # back to back, the 8 updates back to back, every launch on a different batch of the pool -- each block between
        # two HIP events on the launch stream, an empty event pair's own cost subtracted: per-launch durations for the
        # roofline.  Not the pass timed above.
        barrier()
        n_meas = 0 if args.loop_only else min(8 * args.steps, 2400)
        if n_meas:
            if loss_buf.numel() < n_meas:
                loss_buf = torch.zeros(n_meas, device=dev)
            kernel_ms = eng.stream(hyper, RULE, "logits", idx_pool, y_pool, n_meas, loss_buf, timed=True)
        barrier()
    else:
        # ---- N GPUs: exact data parallelism (fmx.DataParallelFM): forward on the local slice, all-gather of the
        #      low-rank factors (idx, S, dlogit) over RCCL, identical row-reduced update of the replicas ----
        dp = None if owner_mode else fmx.DataParallelFM(fmx.HipBackend(eng, hyper, RULE, "logits"))
        fo, owner_path = None, None
        work = work_stream(torch, dev)                # not the legacy default stream (slow to enqueue on)
        if owner_mode:
            # the step as ONE C call with the library's own RCCL communicator (fmx_owner_step) where every rank has a GPU of its
            # own; agreed on by all ranks (a rank that cannot set it up takes everybody to the torch.distributed form of the
            # same step: identical results, more host time).  --owner-path torch forces the latter.
            ok = 0
            if not rehearsal and args.owner_path == "native":
                try:
                    from fmx.owner import NativeOwnerFM
                    fo = NativeOwnerFM(obe, stream=work)
                    ok = 1
                except Exception as exc:              # e.g. librccl.so.1 not loadable
                    print(f"[bench] rank {rank}: fmx_owner_step unavailable ({exc!r})", file=sys.stderr)
            flag = torch.tensor([ok], device="cpu" if rehearsal else dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                owner_path = "fmx_owner_step (one C call per step, the library's own RCCL communicator)"
            else:
                fo = FieldOwnerFM(obe)
                owner_path = "FieldOwnerFM (torch.distributed collectives between the kernels' calls)"

        def run_owner(n, first=0):
            # the index all-gather, the column pick and the sort of the owned fields run two steps ahead on the prefetch stream
            out, tokens = None, {}
            for d in range(min(2, n)):
                tokens[d] = fo.prefetch(idx_pool[(first + d) % N_POOL])
            for s in range(n):
                j = (first + s) % N_POOL
                out = fo.step(idx_pool[j], y_pool[j], tokens.pop(s, None))
                if s + 2 < n:
                    tokens[s + 2] = fo.prefetch(idx_pool[(first + s + 2) % N_POOL])
            return out

        def run(n, first=0):
            if owner_mode:
                return run_owner(n, first)
            # the index all-gather and the global sort of later steps run ahead on two prefetch streams: two steps ahead when
            # a step is one exact update (two global sorts in flight: one alone is as long as the step or longer), one
            # step (= all its sub-steps) ahead when the global batch is split
            depth = 2 if dp._sub_steps(BATCH) == 1 else 1
            out, tokens = None, {}
            for d in range(min(depth, n)):
                tokens[d] = dp.prefetch(idx_pool[(first + d) % N_POOL])
            for s in range(n):
                if s + depth < n:
                    tokens[s + depth] = dp.prefetch(idx_pool[(first + s + depth) % N_POOL])
                j = (first + s) % N_POOL
                out = dp.step(idx_pool[j], y_pool[j], tokens.pop(s, None))
            return out
        with torch.cuda.stream(work):
            run(args.warmup)
            barrier()
            t0 = time.perf_counter()
            last = run(args.steps, args.warmup)
            barrier()
            dt = time.perf_counter() - t0
        eng.check_error_flag()
        losses = last.cpu().numpy()
        assert np.isfinite(losses).all(), "non-finite loss in the timed region"
        tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    parallelism = (f"field owners x{world}: every rank holds and updates the rows of its fields only (table and update work "
                   "sharded); per step an all-gather of idx (ahead of time), an all-to-all of per-sample partial sums (144 B) and "
                   "an all-gather of (S, dlogit, loss) (80 B); bit-identical to one GPU on the same global batch"
                   if owner_mode else
                   f"dp{world}: replicated table, all-gather of (idx, S, dlogit), identical update on every replica (exact)")
    if world > 1:
        sub = 1 if (owner_mode and not hasattr(fo, "_sub_steps")) else (fo if owner_mode else dp)._sub_steps(BATCH)
        parallelism += f"; {sub} exact update(s) per step (global batch {BATCH * world})"
        if owner_path:
            parallelism += "; step issued through " + owner_path

    if stream_gbps is None:
        stream_gbps = stream_read_probe(fmx, torch, dev)

    samples = args.steps * BATCH * world
    value = samples / dt
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("k_fm_update_hbm_bytes_per_launch")
            traffic_src = ("REPLAYED, not measured in this run: profiles/traffic_latest.json (" + str(tj.get("source", "rocprofv3 --pmc "
                           "FETCH_SIZE / WRITE_SIZE passes of an earlier profiling run of this command")) + ")")
        except Exception:
            traffic = None
    step_gbps = value / world * BYTES_STEP_FTRL / 1e9
    out = {
        "metric": "samples/sec online-FM (Criteo-39-field, k=16) FTRL-proximal",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "online FM fwd+bwd + fused FTRL-proximal row update, synthetic Criteo-39 "
                               f"(R=1,006,628 rows, k=16, B={BATCH} per GPU, {'Zipf(1.05)' if args.zipf else 'uniform'} indices, "
                               "labels Bernoulli(0.3)); BASELINE.json configs[1]+[2]",
                   "global_batch": BATCH * world, "row_stride_bytes": table.row_stride * 4, "rule": RULE,
                   "hyper": HYPER, "pool_batches": N_POOL,
                   "parallelism": "1 GPU" if world == 1 else parallelism},
        "measured_stream_read_GBps": stream_gbps,
        "measured_gather_ceilings": GATHER_CEILINGS,
        "final_loss": float(losses[-1]),
    }
    # ---- roofline: the WHOLE STEP (samples/s x 10,772 algorithmic bytes per sample), timed in the
    #      loop above; per-kernel figures beside it ----
    roof = {"bound": "hbm", "scope": "whole step = sort + forward + update of one batch, as timed in `value`",
            "achieved": step_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": step_gbps / HBM_PEAK_GBPS,
            "frac_of_measured_stream_read": step_gbps / stream_gbps, "algorithmic_bytes_per_sample": BYTES_STEP_FTRL,
            "algorithmic_bytes_per_step": BYTES_STEP_FTRL * BATCH,
            "traffic": traffic, "traffic_scope": "k_fm_update, FABRIC-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE: the L2's memory-side requests, "
                             "Infinity-Cache hits included -- not HBM bytes proper; MI355X_MICROARCH.md, HBM)",
            "traffic_source": traffic_src}
    if kernel_ms is not None:
        sort_ms, fwd_ms, upd_ms, pair_ms = [v / n_meas for v in kernel_ms]
        if chunk_us:
            out["step_us_over_100_step_chunks"] = {q: float(np.percentile(chunk_us, p)) for q, p in (("p10", 10), ("p50", 50), ("p90", 90))}
        ach = BYTES_K_UPDATE * BATCH / (upd_ms * 1e-3) / 1e9
        ach_f = BYTES_K_FORWARD * BATCH / (fwd_ms * 1e-3) / 1e9
        roof["kernels"] = {
            "how": "HIP events on the launch stream around groups of 8 back-to-back launches of one kernel, each on a different "
                   "batch (empty event pair subtracted), in a measuring pass over the same pool AFTER the timed loop: back-to-back "
                   "launch durations, shorter than the same kernels inside the loop (dependent-launch gaps, the side-stream sort)",
            "k_fm_update": {"achieved": ach, "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": BYTES_K_UPDATE * BATCH,
                            "avg_launch_ms": upd_ms},
            "k_fm_forward": {"achieved": ach_f, "frac": ach_f / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": BYTES_K_FORWARD * BATCH,
                             "avg_launch_ms": fwd_ms},
            "k_sort_occ": {"avg_launch_ms": sort_ms, "batches_per_launch": 8, "stream": "side stream, beside the steps"},
            "empty_event_pair_ms": pair_ms,
            "sum_of_kernel_ms_per_step": fwd_ms + upd_ms, "loop_ms_per_step": dt / args.steps * 1e3}
    out["roofline"] = roof
    if short_run is not None:
        out["short_run"] = short_run
    if world > 1:
        dist.destroy_process_group()
    if world == 1 and not args.no_secondary and not args.loop_only:
        # BASELINE configs[3] in the same line, so that the round-end run observes it: online DeepFM (3 x 256 MLP, SGD)
        try:
            del eng, table
            torch.cuda.empty_cache()
            out["class_surface"] = bench_class_surface(torch)
        except Exception as exc:
            out["class_surface"] = {"error": repr(exc)}
        try:
            torch.cuda.empty_cache()
            ns = argparse.Namespace(steps=100, warmup=10, zipf=False, mp_mode="owner")
            first = bench_deepfm(ns, fmx, torch, dist, 1, 0, dev, False)          # (pays the one-time set-up: workspaces, first launches)
            sec = bench_deepfm(ns, fmx, torch, dist, 1, 0, dev, False)
            out["secondary"] = {"deepfm": {k: sec[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "mlp_section", "config", "issued_by",
                                                               "through_trainer_step_samples_per_s") if k in sec}}
            out["secondary"]["deepfm"]["passes"] = {"note": "two passes of 100 steps in this process; `value` is the SECOND (the first pays "
                                                    "one-time set-up); both are given", "first": first["value"], "second": sec["value"]}
            out["secondary"]["deepfm"]["roofline"] = mlp_section_probe(fmx, torch, dev)
        except Exception as exc:                            # the headline line must not be lost to the secondary workload
            out["secondary"] = {"deepfm": {"error": repr(exc)}}
        try:
            # the HBM-resident companion: the vocabulary list scaled 64 x (fields capped at 2^20 - 1 rows): 11.1 M rows,
            # 2.84 GB of FTRL rows -- 11 x the Infinity Cache -- and the Zipf(1.05) stream on the headline table
            torch.cuda.empty_cache()
            big = [min(s_ * 64, (1 << 20) - 1) for s_ in CRITEO_SIZES]
            out["secondary"]["fm_hbm_resident"] = fm_loop_probe(fmx, torch, dev, big, zipf=False)
            torch.cuda.empty_cache()
            out["secondary"]["fm_zipf"] = fm_loop_probe(fmx, torch, dev, CRITEO_SIZES, zipf=True)
        except Exception as exc:
            out["secondary"]["fm_companions_error"] = repr(exc)
    # the CPU baseline last: its OpenMP / torch thread pools keep the host cores busy for a while after they return
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(idx_np, y_np, CRITEO_SIZES)
        out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        out["cpu_baseline"]["cpu_model"] = cpu_model()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
