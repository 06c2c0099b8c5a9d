"""The field-owner multi-GPU mode through the HIP backend (fmx.owner, fmx.plan): the owners hold row-range PIECES of the index
columns at fixed positions of the forward tree, so G owners leave BIT-IDENTICAL rows, biases and losses to fmx_fm_step on ONE
table holding the same pieces at the same positions (OwnerPlan.table_whole) -- and that table's step meets the oracle at
north_star's tolerance.  Checked (a) with the G owners emulated in one process (the exchange done by slicing: exercises the
kernels for G = 1 ... 16, powers of two and not, Criteo- and Frappe-shaped), (b) with two processes sharing the test box's
one GPU, collectives over gloo with host staging (RCCL needs one GPU per rank), including a Frappe-shaped 10 M-row table
split over the two ranks (BASELINE configs[4]'s table)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import assert_ftrl_step_within_f64, assert_within_f64
from oracle import fm_oracle as orc

pytestmark = pytest.mark.gpu

CRITEO_SIZES = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887,
                632, 3, 41738, 5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86,
                44549]
HYP = dict(lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.001, l2=0.01)


def _weights(sizes, k, seed=5):
    rng = np.random.default_rng(seed)
    R = sum(sizes)
    return (rng.normal(size=(R, k)) * 0.3).astype(np.float32), (rng.normal(size=R) * 0.3).astype(np.float32)


def _columns(sizes, V, w):
    offs = np.concatenate([[0], np.cumsum(sizes)])
    return ([torch.from_numpy(w[offs[c]:offs[c + 1]].reshape(-1, 1)) for c in range(len(sizes))],
            [torch.from_numpy(V[offs[c]:offs[c + 1]]) for c in range(len(sizes))])


def _load(t, sizes, V, w):
    from fmx.plan import load_columns
    load_columns(t, *_columns(sizes, V, w))
    t.set_bias_weight(0.25)


def _batches(sizes, GB, steps, seed=9):
    rng = np.random.default_rng(seed)
    return [(np.stack([rng.integers(0, s, size=GB) for s in sizes], axis=1).astype(np.int32),
             (rng.uniform(size=GB) < 0.3).astype(np.float32)) for _ in range(steps)]


def _whole_run(fmx, plan, sizes, k, rule, batches):
    """The reference for every owner test: ONE table with every piece at its position, fmx_fm_step per global batch."""
    V, w = _weights(sizes, k)
    t = plan.table_whole(layout="ftrl" if rule == "ftrl" else "weights", ftrl=HYP)
    _load(t, sizes, V, w)
    eng = fmx.FMEngine(t, max_batch=batches[0][0].shape[0])
    hyp = fmx.Hyper(**HYP)
    losses = []
    for idx, y in batches:
        idx_d, _, y_d = eng.to_device(idx, None, y)
        eng.step(hyp, rule, "logits", idx_d, None, y_d)
        losses.append(float(eng.loss_out.item()))
    eng.check_error_flag()
    return t, losses


def _pieces(t):
    from fmx.plan import export_columns
    return {key: rows.numpy() for key, rows in export_columns(t, None).items()}


def _emulate(fmx, plan, sizes, k, B, rule, batches):
    """G owners in one process (each with its shard of the table), the all-gathers / all-to-all done by slicing."""
    from fmx.owner import HipOwnerBackend
    G, GB = plan.world, batches[0][0].shape[0]
    V, w = _weights(sizes, k)
    hyp = fmx.Hyper(**HYP)
    owners = []
    for g in range(G):
        be = HipOwnerBackend(sizes, k, hyp, rule, "logits", g, G, ftrl=HYP, max_local_batch=B, plan=plan)
        _load(be.table, sizes, V, w)
        owners.append(be)
    inv_b, REC = 1.0 / GB, 2 * owners[0].kp + 4
    all_losses = []
    for idx, y in batches:
        idx_all = torch.from_numpy(idx).cuda()
        y_all = torch.from_numpy(y).cuda()
        parts = [be.partial_forward(idx_all, B).clone().view(G, be.nlb, B, REC) for be in owners]      # destination-major per owner
        recs = []
        for r, be in enumerate(owners):                                                     # all-to-all: rank r's samples, block order
            mine = torch.cat([p[r] for p in parts]).contiguous()
            assert mine.shape[0] == plan.nb
            recs.append(be.finish(mine, y_all[r * B:(r + 1) * B].contiguous(), inv_b).clone())
        rec_g = torch.cat(recs).contiguous()                                                # all-gather
        all_losses.append([float(be.update(idx_all, rec_g, inv_b)[0]) for be in owners])
    torch.cuda.synchronize()
    for be in owners:
        be.check_error_flag()
    return owners, all_losses


FRAPPE_SMALL = [2000] * 10       # Frappe-shaped: 10 one-hot columns (BASELINE configs[4]), fewer columns than lane groups


@pytest.mark.parametrize("sizes,k,B,rule,G", [(CRITEO_SIZES, 16, 1024, "ftrl", 1), (CRITEO_SIZES, 16, 1024, "ftrl", 2),
                                              (CRITEO_SIZES, 16, 512, "ftrl", 8), (CRITEO_SIZES, 10, 256, "signadam", 4),
                                              (CRITEO_SIZES, 16, 128, "sgd", 16), (CRITEO_SIZES, 16, 256, "ftrl", 3),
                                              (CRITEO_SIZES, 16, 128, "ftrl", 6), (FRAPPE_SMALL, 16, 256, "ftrl", 4),
                                              (FRAPPE_SMALL, 16, 128, "ftrl", 8), ([7, 3, 50], 16, 64, "sgd", 4),
                                              ([7, 3, 50, 2, 9, 1000, 4, 4, 30] * 8, 4, 96, "ftrl", 4),
                                              ([5, 40, 3, 700, 2, 11], 32, 64, "sgd", 2)])
def test_emulated_owners_equal_the_one_table_step(sizes, k, B, rule, G):
    import fmx
    from fmx.plan import OwnerPlan
    GB, steps = B * G, 3
    plan = OwnerPlan(sizes, k, G, global_batch=GB)
    rows = plan.rows_per_owner()
    assert sum(rows) == sum(sizes) and min(rows) >= 1                                       # every row has one owner, every owner has rows
    batches = _batches(sizes, GB, steps)
    t_ref, ref_losses = _whole_run(fmx, plan, sizes, k, rule, batches)
    owners, losses = _emulate(fmx, plan, sizes, k, B, rule, batches)
    for step, ls in enumerate(losses):
        assert all(l == ref_losses[step] for l in ls), (ls, ref_losses[step])
    ref = _pieces(t_ref)
    seen = set()
    for be in owners:
        mine = _pieces(be.table)
        for key, r in mine.items():
            np.testing.assert_array_equal(r, ref[key], err_msg=f"owner {be.rank} piece {key}")
        seen |= set(mine)
        np.testing.assert_array_equal(be.table.bias.cpu().numpy(), t_ref.bias.cpu().numpy())
    assert seen == set(ref)


def test_one_owner_is_the_ordinary_table():
    """world == 1: the plan places column f at position f -- the ordinary (unmapped) table, the old bits."""
    import fmx
    from fmx.plan import OwnerPlan
    sizes, k, B = CRITEO_SIZES, 16, 1024
    plan = OwnerPlan(sizes, k, 1)
    t = plan.table_whole(layout="ftrl", ftrl=HYP)
    assert not t.mapped and t.feature_sizes == sizes


@pytest.mark.parametrize("sizes,G,B", [(CRITEO_SIZES, 8, 512), (FRAPPE_SMALL, 8, 512)])
def test_the_whole_table_step_meets_the_oracle(sizes, G, B):
    """The one-table equivalent of G owners (pieces at their positions: another order of the forward's additions than the
    ordinary table) against the float64 step at north_star's 1e-5 + the derived fp32 floor, touched rows and untouched ones;
    then the owners against it bit for bit.  So the owners' rows meet the oracle, not only each other."""
    import fmx
    from fmx.plan import OwnerPlan
    k, GB = 16, B * G
    plan = OwnerPlan(sizes, k, G, global_batch=GB)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    V, w = _weights(sizes, k)
    h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
    batches = _batches(sizes, GB, 1)
    idx, y = batches[0]
    t_ref, ref_losses = _whole_run(fmx, plan, sizes, k, "ftrl", batches)
    # the same start in the oracle's layout (column-major rows): n = 0, z reproducing the weights, bias 0.25
    st0 = dict(zV=orc.ftrl_z_for_weight(V, **h), nV=np.zeros_like(V), zw=orc.ftrl_z_for_weight(w, **h), nw=np.zeros_like(w),
               zb=np.float32(orc.ftrl_z_for_weight(np.float32(0.25), **h)), nb=np.float32(0.0))
    rows = idx.astype(np.int64) + offs[:-1][None, :]
    ref = orc.flat_fm_step_f64(st0, rows, np.ones(idx.shape, np.float32), y, "logits", "ftrl", h)
    assert_within_f64(ref_losses[0], ref["loss"], ref["floor"]["loss"], "loss")
    # gather the table's pieces back into the oracle's row order
    zo, kp = t_ref.z_offset, t_ref.kp
    got = dict(zV=np.zeros_like(V), nV=np.zeros_like(V), zw=np.zeros_like(w), nw=np.zeros_like(w))
    for (c, b, r), pr in _pieces(t_ref).items():
        lo = int(offs[c]) + b
        got["zV"][lo:lo + r], got["nV"][lo:lo + r] = pr[:, zo:zo + k], pr[:, zo + kp:zo + kp + k]
        got["zw"][lo:lo + r], got["nw"][lo:lo + r] = pr[:, kp + 1], pr[:, kp + 2]
    got["zb"], got["nb"] = t_ref.bias[0].item(), t_ref.bias[1].item()
    assert_ftrl_step_within_f64(got, ref, before=st0)
    owners, losses = _emulate(fmx, plan, sizes, k, B, "ftrl", batches)
    assert all(l == ref_losses[0] for l in losses[0])
    whole = _pieces(t_ref)
    for be in owners:
        for key, r in _pieces(be.table).items():
            np.testing.assert_array_equal(r, whole[key], err_msg=f"owner {be.rank} piece {key}")


def test_criteo_ownership_is_balanced_at_eight_ranks():
    from fmx.plan import OwnerPlan
    plan = OwnerPlan(CRITEO_SIZES, 16, 8, global_batch=8 * 4096)
    rows = plan.rows_per_owner()
    assert max(rows) <= 2 * min(rows), rows                                                # HBM footprint within 2x
    occ = [float(l[1]) for l in plan.rank_load]
    assert max(occ) <= 1.5 * min(occ), occ                                                 # expected update work within 1.5x
    plan10 = OwnerPlan([1_000_000] * 10, 16, 8, global_batch=8 * 1024)                     # configs[4]: 10 columns over 8 ranks
    assert min(plan10.rows_per_owner()) >= 1_000_000 and max(plan10.rows_per_owner()) <= 1_500_000


def test_a_stale_workspace_is_refused():
    """A workspace sized before the table's large pieces were split for a larger batch is too small afterwards at EVERY batch
    size: the C side checks the byte count and refuses (it used to write out of bounds)."""
    import fmx
    sizes, k = [300000, 50, 7], 16
    t = fmx.FlatTable(sizes, k, layout="weights")
    eng = fmx.FMEngine(t, max_batch=4096)
    old = eng.new_workspace(4096)
    t.sort_cap_override = 100000               # as a growth to a much larger batch would: the 300,000-row field becomes three pieces
    eng._alloc(4096)
    idx = torch.zeros((4096, 3), dtype=torch.int32, device="cuda")
    with pytest.raises(fmx._lib.FmxError, match="workspace of"):
        eng.sort(idx, workspace=old)
    eng.sort(idx)                              # the engine's own workspace was rebuilt with the table
    torch.cuda.synchronize()


def test_owner_class_single_rank_with_prefetch_equals_plain_steps():
    import fmx
    from fmx.owner import FieldOwnerFM, HipOwnerBackend
    from fmx.plan import OwnerPlan
    sizes, k, B, steps = CRITEO_SIZES, 16, 4096, 5
    batches = _batches(sizes, B, steps)
    plan = OwnerPlan(sizes, k, 1)
    t_ref, ref_losses = _whole_run(fmx, plan, sizes, k, "ftrl", batches)
    V, w = _weights(sizes, k)
    be = HipOwnerBackend(sizes, k, fmx.Hyper(**HYP), "ftrl", "logits", 0, 1, ftrl=HYP, max_local_batch=B)
    _load(be.table, sizes, V, w)
    fo = FieldOwnerFM(be)
    data = [(torch.from_numpy(i).cuda(), torch.from_numpy(y).cuda()) for i, y in batches]
    work = torch.cuda.Stream()
    losses = []
    with torch.cuda.stream(work):
        tokens = {0: fo.prefetch(data[0][0]), 1: fo.prefetch(data[1][0])}
        spare = fo.prefetch(data[2][0])                                                     # a token that is given back unused
        fo.cancel(spare)
        for i, (idx_d, y_d) in enumerate(data):
            losses.append(fo.step(idx_d, y_d, tokens.pop(i, None)).clone())
            if i + 2 < steps and i != 1:                                                    # step 3 runs with nothing prepared
                tokens[i + 2] = fo.prefetch(data[i + 2][0])
                assert tokens[i + 2] is not None
    torch.cuda.synchronize()
    fo.check_error_flag()
    assert not fo._pref
    assert [float(l[0]) for l in losses] == ref_losses
    np.testing.assert_array_equal(be.table.rows.cpu().numpy(), t_ref.rows.cpu().numpy())
    np.testing.assert_array_equal(be.table.bias.cpu().numpy(), t_ref.bias.cpu().numpy())


# ---- two processes sharing the GPU, gloo ----
FRAPPE_SIZES = [1_000_000] * 10                    # Frappe-shaped: 10 one-hot fields, 10 M rows in all (BASELINE configs[4])


def _seed_frappe(fmx, t, k):
    """A 2.5 GB table: seeded on the device, piece by piece (the same numbers for a row wherever it lives)."""
    zo = t.z_offset
    for f, (c, b, r) in enumerate(t.plan_fields):
        if not r:
            continue
        g = torch.Generator(device="cuda").manual_seed(100 + c)
        Vc = torch.randn((FRAPPE_SIZES[c], k), generator=g, device="cuda") * 0.3
        lo = int(t.offsets_host[f])
        t.rows[lo:lo + r, :k] = Vc[b:b + r]
        t.rows[lo:lo + r, zo:zo + k] = fmx.table.ftrl_z_for_weight_torch(Vc[b:b + r], t.ftrl)
        del Vc


def _touched(idx_batches, sizes):
    """Per column the sorted local indices any of the batches touches."""
    return [np.unique(np.concatenate([i[:, c] for i, _ in idx_batches])) for c in range(len(sizes))]


def _rows_of(t, touched):
    """{(column, local index): row} of the touched rows this table holds."""
    out, rows = {}, None
    for f, (c, b, r) in enumerate(t.plan_fields):
        if not r:
            continue
        sel = touched[c][(touched[c] >= b) & (touched[c] < b + r)]
        lo = int(t.offsets_host[f])
        got = t.rows[torch.from_numpy(lo + (sel - b)).cuda()].cpu().numpy()
        for i, li in enumerate(sel):
            out[(c, int(li))] = got[i]
    return out


def _owner_run(rank, world, sizes, k, B, steps, frappe, whole=False):
    import fmx
    from fmx.owner import FieldOwnerFM, HipOwnerBackend
    from fmx.plan import OwnerPlan
    hyp = fmx.Hyper(**HYP)
    batches = _batches(sizes, B * world, steps)
    plan = OwnerPlan(sizes, k, world, global_batch=B * world)
    if whole:                                       # the one-table equivalent of `world` owners, stepped by fmx_fm_step
        t = plan.table_whole(layout="ftrl", ftrl=HYP)
        eng = fmx.FMEngine(t, max_batch=B * world)
    else:
        be = HipOwnerBackend(sizes, k, hyp, "ftrl", "logits", rank, world, ftrl=HYP, max_local_batch=B, plan=plan)
        t = be.table
    if frappe:
        _seed_frappe(fmx, t, k)
    else:
        _load(t, sizes, *_weights(sizes, k))
    losses = []
    if whole:
        for idx, y in batches:
            idx_d, _, y_d = eng.to_device(idx, None, y)
            eng.step(hyp, "ftrl", "logits", idx_d, None, y_d)
            losses.append(float(eng.loss_out.item()))
        eng.check_error_flag()
    else:
        fo = FieldOwnerFM(be)
        sl = slice(rank * B, (rank + 1) * B)
        data = [(torch.from_numpy(i[sl].copy()).cuda(), torch.from_numpy(y[sl].copy()).cuda()) for i, y in batches]
        tok = fo.prefetch(data[0][0])
        for i, (idx_d, y_d) in enumerate(data):
            losses.append(float(fo.step(idx_d, y_d, tok)[0]))
            tok = fo.prefetch(data[i + 1][0]) if i + 1 < steps else None
        torch.cuda.synchronize()
        fo.check_error_flag()
    if frappe:                                      # the touched rows (a tiny part of 2.5 GB), each under its (column, index)
        return losses, _rows_of(t, _touched(batches, sizes)), t.bias.cpu().numpy()
    return losses, _pieces(t), t.bias.cpu().numpy()


def _worker(rank, world, port, q, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "fm-for-online-recommendation_amd"), os.path.join(root, "tests")]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank,) + _owner_run(rank, world, *args))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_owner_processes_equal_the_one_table_step(world):
    sizes, k, B, steps = CRITEO_SIZES, 16, 512, 4
    res = _spawn(world, (sizes, k, B, steps, False))
    ref_losses, ref, ref_bias = _owner_run(0, world, sizes, k, B, steps, False, whole=True)
    seen = set()
    for rank, losses, pieces, bias in res:
        assert losses == ref_losses
        for key, r in pieces.items():
            np.testing.assert_array_equal(r, ref[key], err_msg=f"rank {rank} piece {key}")
        seen |= set(pieces)
        np.testing.assert_array_equal(bias, ref_bias)
    assert seen == set(ref)


@pytest.mark.timeout(900)
def test_frappe_shaped_10m_row_table_split_over_two_owners():
    """BASELINE configs[4]'s table: 10 M rows (2.5 GB in the FTRL layout) -- here TRAINED (forward + fused FTRL update), split
    over two ranks (5 M rows each); against ONE process holding all of it at the same positions: every touched ROW compared bit
    for bit (not a digest), and those rows against the oracle's float64 step at 1e-5 + the fp32 floor."""
    sizes, k, B, steps = FRAPPE_SIZES, 16, 1024, 3
    res = _spawn(2, (sizes, k, B, steps, True))
    one_losses, one_rows, one_bias = _owner_run(0, 2, sizes, k, B, steps, True, whole=True)
    got = {}
    for rank, losses, rows, bias in res:
        assert losses == one_losses
        np.testing.assert_array_equal(bias, one_bias)
        assert 0 < len(rows) < len(one_rows)
        got.update(rows)
    assert set(got) == set(one_rows)
    for key, r in one_rows.items():
        np.testing.assert_array_equal(got[key], r, err_msg=f"row {key}")
    # ---- the touched rows of the FIRST step against the oracle (float64 step over just those rows' state) ----
    one1_losses, rows1, bias1 = _owner_run(0, 2, sizes, k, B, 1, True, whole=True)
    idx, y = _batches(sizes, B * 2, 1)[0]
    keys = sorted(rows1)                                                                    # (column, index) of every touched row
    pos = {key: i for i, key in enumerate(keys)}
    h = dict(alpha=HYP["alpha"], beta=HYP["beta"], l1=HYP["l1"], l2=HYP["l2"])
    V0 = np.zeros((len(keys), k), np.float32)
    for c in range(len(sizes)):                                                             # the seeded start of exactly those rows
        g = torch.Generator(device="cuda").manual_seed(100 + c)
        Vc = (torch.randn((sizes[c], k), generator=g, device="cuda") * 0.3).cpu().numpy()
        for (cc, li), i in pos.items():
            if cc == c:
                V0[i] = Vc[li]
    st0 = dict(zV=orc.ftrl_z_for_weight(V0, **h), nV=np.zeros_like(V0), zw=np.zeros(len(keys), np.float32), nw=np.zeros(len(keys), np.float32),
               zb=np.float32(0.0), nb=np.float32(0.0))
    rows = np.array([[pos[(c, int(idx[b, c]))] for c in range(len(sizes))] for b in range(idx.shape[0])], dtype=np.int64)
    ref = orc.flat_fm_step_f64(st0, rows, np.ones(idx.shape, np.float32), y, "logits", "ftrl", h)
    assert_within_f64(one1_losses[0], ref["loss"], ref["floor"]["loss"], "loss")
    kp, zo = 16, 32
    hip = dict(zV=np.stack([rows1[key][zo:zo + k] for key in keys]), nV=np.stack([rows1[key][zo + kp:zo + kp + k] for key in keys]),
               zw=np.array([rows1[key][kp + 1] for key in keys]), nw=np.array([rows1[key][kp + 2] for key in keys]),
               zb=float(bias1[0]), nb=float(bias1[1]))
    assert_ftrl_step_within_f64(hip, ref, before=st0)


def test_rccl_collectives_of_a_step_with_one_rank():
    """The RCCL calls themselves (gloo stands in for them in the two-process tests above, which share one GPU): one rank,
    backend nccl, FMX_FORCE_COLLECTIVES=1 -- 20 steps through FieldOwnerFM (prefetch tokens) and DataParallelFM end
    bit-identical to fmx_fm_stream (tools/nccl_world1_check.py, run as the driver launches bench.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(root, "tools", "nccl_world1_check.py")], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "field owners, nccl, 1 rank, forced collectives: 20 steps bit-identical" in r.stdout
    assert "replicated mode, nccl, 1 rank, forced collectives: bit-identical" in r.stdout
    assert "fmx_owner_step, own RCCL communicator, 1 rank, forced collectives: 20 steps bit-identical" in r.stdout


def test_native_owner_step_single_rank_equals_plain_steps():
    """fmx_owner_prefetch / fmx_owner_step (one C call per step; with one rank nothing is exchanged and RCCL is not loaded)
    against fmx_fm_step on the same batches: the same bits, with and without prepared slots."""
    import fmx
    from fmx.owner import HipOwnerBackend, NativeOwnerFM
    from fmx.plan import OwnerPlan
    sizes, k, B, steps = CRITEO_SIZES, 16, 2048, 6
    batches = _batches(sizes, B, steps)
    t_ref, ref_losses = _whole_run(fmx, OwnerPlan(sizes, k, 1), sizes, k, "ftrl", batches)
    be = HipOwnerBackend(sizes, k, fmx.Hyper(**HYP), "ftrl", "logits", 0, 1, ftrl=HYP, max_local_batch=B)
    _load(be.table, sizes, *_weights(sizes, k))
    data = [(torch.from_numpy(i).cuda(), torch.from_numpy(y).cuda()) for i, y in batches]
    work = torch.cuda.Stream()
    torch.cuda.synchronize()
    nat = NativeOwnerFM(be, stream=work)                                                    # the stream its calls go to
    losses = []
    with torch.cuda.stream(work):
        tokens = {0: nat.prefetch(data[0][0]), 1: nat.prefetch(data[1][0])}
        for i, (idx_d, y_d) in enumerate(data):
            losses.append(nat.step(idx_d, y_d, tokens.pop(i, None)).clone())
            if i + 2 < steps and i != 1:                                                    # step 3 runs with nothing prepared
                tokens[i + 2] = nat.prefetch(data[i + 2][0])
    torch.cuda.synchronize()
    nat.check_error_flag()
    assert [float(l[0]) for l in losses] == ref_losses
    np.testing.assert_array_equal(be.table.rows.cpu().numpy(), t_ref.rows.cpu().numpy())
    np.testing.assert_array_equal(be.table.bias.cpu().numpy(), t_ref.bias.cpu().numpy())


def test_owner_entry_points_reject_bad_arguments():
    import ctypes as C
    import fmx
    from fmx.owner import HipOwnerBackend, NativeOwnerFM
    lib = fmx._lib.load()
    be = HipOwnerBackend([50, 7, 300], 16, fmx.Hyper(**HYP), "sgd", "logits", 0, 1, max_local_batch=256)
    nat = NativeOwnerFM(be)
    idx = torch.zeros((256, 3), dtype=torch.int32, device="cuda")
    y = torch.zeros(256, device="cuda")
    ws = nat._slot(0, 256)[2]
    idx_all = torch.zeros((256, 3), dtype=torch.int32, device="cuda")
    bufs = nat._step_bufs(256)
    err, st = be.e.error.data_ptr(), 0
    t, h = be.table.c_struct(), be.hyper.ref()
    E = fmx._lib
    assert lib.fmx_owner_prefetch(None, t, idx.data_ptr(), 256, 0, idx_all.data_ptr(), ws.data_ptr(), ws.numel() * 4, err, st) == E.ERR_ARG
    assert lib.fmx_owner_prefetch(nat.comm, t, idx.data_ptr(), 256, 9, idx_all.data_ptr(), ws.data_ptr(), ws.numel() * 4, err, st) == E.ERR_ARG
    assert lib.fmx_owner_prefetch(nat.comm, t, None, 256, 0, idx_all.data_ptr(), ws.data_ptr(), ws.numel() * 4, err, st) == E.ERR_ARG
    assert lib.fmx_owner_prefetch(nat.comm, t, idx.data_ptr(), 256, 0, idx_all.data_ptr(), ws.data_ptr(), 64, err, st) == E.ERR_SHAPE
    assert lib.fmx_owner_prefetch(nat.comm, t, idx.data_ptr(), 40000, 0, idx_all.data_ptr(), ws.data_ptr(), ws.numel() * 4, err, st) == E.ERR_UNSUPPORTED
    assert b"exceed one exact step" in lib.fmx_last_error_string()
    step = lambda *a: lib.fmx_owner_step(*a)
    assert step(nat.comm, t, h, E.RULE_SGD, E.LOSS_BCE_LOGITS, idx_all.data_ptr(), y.data_ptr(), 256, 0, ws.data_ptr(), ws.numel() * 4, None,
                be.e.loss_out.data_ptr(), err, st) == E.ERR_ARG
    assert step(nat.comm, t, h, E.RULE_FTRL, E.LOSS_BCE_LOGITS, idx_all.data_ptr(), y.data_ptr(), 256, 0, ws.data_ptr(), ws.numel() * 4,
                C.byref(bufs), be.e.loss_out.data_ptr(), err, st) == E.ERR_ARG                 # the rule does not fit the table's layout
    assert step(nat.comm, t, h, E.RULE_SGD, E.LOSS_NONE, idx_all.data_ptr(), y.data_ptr(), 256, 0, ws.data_ptr(), ws.numel() * 4,
                C.byref(bufs), be.e.loss_out.data_ptr(), err, st) == E.ERR_ARG
    bad = fmx._lib.OwnerBufs(bufs.parts_send + 4, bufs.parts_recv, bufs.rec_local, bufs.rec_all)
    assert step(nat.comm, t, h, E.RULE_SGD, E.LOSS_BCE_LOGITS, idx_all.data_ptr(), y.data_ptr(), 256, 0, ws.data_ptr(), ws.numel() * 4,
                C.byref(bad), be.e.loss_out.data_ptr(), err, st) == E.ERR_ALIGN
    ids = (C.c_char * 256)()
    out = C.c_void_p()
    assert lib.fmx_comm_create(ids, 3, 2, None, 0, C.byref(out)) == E.ERR_ARG                   # rank outside the world
    assert lib.fmx_comm_create(None, 0, 2, None, 0, C.byref(out)) == E.ERR_ARG                  # two ranks need the ids
    assert lib.fmx_comm_create(None, 0, 1, None, 0, None) == E.ERR_ARG
