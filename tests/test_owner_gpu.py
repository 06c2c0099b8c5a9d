"""The field-owner multi-GPU mode through the HIP backend (fmx.owner): every additions tree of the one-GPU kernels is
reproduced, so G owners leave BIT-IDENTICAL rows, biases and losses to fmx_fm_step on one table holding every field.
Checked (a) with the G owners emulated in one process (the exchange done by slicing: exercises the kernels for every
G = 1 ... 16), (b) with two processes sharing the test box's one GPU, collectives over gloo with host staging (RCCL needs
one GPU per rank), including a Frappe-shaped 10 M-row table split over the two ranks (BASELINE configs[4]'s table)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CRITEO_SIZES = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887,
                632, 3, 41738, 5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86,
                44549]
HYP = dict(lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.001, l2=0.01)


def _weights(sizes, k, seed=5):
    rng = np.random.default_rng(seed)
    R = sum(sizes)
    return (rng.normal(size=(R, k)) * 0.3).astype(np.float32), (rng.normal(size=R) * 0.3).astype(np.float32)


def _load(t, sizes, fields, V, w):
    offs = np.concatenate([[0], np.cumsum(sizes)])
    t.load_reference([torch.from_numpy(w[offs[f]:offs[f + 1]].reshape(-1, 1)) for f in fields],
                     [torch.from_numpy(V[offs[f]:offs[f + 1]]) for f in fields])
    t.set_bias_weight(0.25)


def _batches(sizes, GB, steps, seed=9):
    rng = np.random.default_rng(seed)
    return [(np.stack([rng.integers(0, s, size=GB) for s in sizes], axis=1).astype(np.int32),
             (rng.uniform(size=GB) < 0.3).astype(np.float32)) for _ in range(steps)]


def _plain_run(fmx, sizes, k, rule, batches):
    """The reference for every owner test: one table with every field, fmx_fm_step per global batch."""
    V, w = _weights(sizes, k)
    t = fmx.FlatTable(sizes, k, layout="ftrl" if rule == "ftrl" else "weights", ftrl=HYP)
    _load(t, sizes, range(len(sizes)), V, w)
    eng = fmx.FMEngine(t, max_batch=batches[0][0].shape[0])
    hyp = fmx.Hyper(**HYP)
    losses = []
    for idx, y in batches:
        idx_d, _, y_d = eng.to_device(idx, None, y)
        eng.step(hyp, rule, "logits", idx_d, None, y_d)
        losses.append(float(eng.loss_out.item()))
    eng.check_error_flag()
    return t, losses


def _field_rows(t, sizes_local):
    offs = np.concatenate([[0], np.cumsum(sizes_local)])
    rows = t.rows.cpu().numpy()
    return [rows[offs[i]:offs[i + 1]] for i in range(len(sizes_local))]


@pytest.mark.parametrize("sizes,k,B,rule,G", [(CRITEO_SIZES, 16, 1024, "ftrl", 1), (CRITEO_SIZES, 16, 1024, "ftrl", 2),
                                              (CRITEO_SIZES, 16, 512, "ftrl", 8), (CRITEO_SIZES, 10, 256, "signadam", 4),
                                              (CRITEO_SIZES, 16, 128, "sgd", 16), ([7, 3, 50, 2, 9, 1000, 4, 4, 30] * 8, 4, 96, "ftrl", 4),
                                              ([5, 40, 3, 700, 2, 11], 32, 64, "sgd", 2)])
def test_emulated_owners_equal_the_one_table_step(sizes, k, B, rule, G):
    """G owners in one process (each with its shard of the table), the all-gathers / all-to-all done by slicing."""
    import fmx
    from fmx.owner import HipOwnerBackend, owner_fields
    GB, steps = B * G, 3
    batches = _batches(sizes, GB, steps)
    t_ref, ref_losses = _plain_run(fmx, sizes, k, rule, batches)
    V, w = _weights(sizes, k)
    hyp = fmx.Hyper(**HYP)
    owners = []
    for g in range(G):
        be = HipOwnerBackend(sizes, k, hyp, rule, "logits", g, G, ftrl=HYP, max_local_batch=B)
        _load(be.table, sizes, be.fields, V, w)
        owners.append(be)
    assert sorted(sum((be.fields for be in owners), [])) == list(range(len(sizes)))
    inv_b = 1.0 / GB
    for step, (idx, y) in enumerate(batches):
        idx_all = torch.from_numpy(idx).cuda()
        y_all = torch.from_numpy(y).cuda()
        own = [be.select(idx_all) for be in owners]
        parts = [be.partial_forward(o).clone() for be, o in zip(owners, own)]               # [GB, 2 kp + 4] per owner
        recs = []
        for r, be in enumerate(owners):                                                     # all-to-all: rank r's samples
            mine = torch.stack([p[r * B:(r + 1) * B] for p in parts]).contiguous()
            recs.append(be.finish(mine, y_all[r * B:(r + 1) * B].contiguous(), inv_b).clone())
        rec_g = torch.cat(recs).contiguous()                                                # all-gather
        losses = [float(be.update(o, rec_g, inv_b)[0]) for be, o in zip(owners, own)]
        assert all(l == ref_losses[step] for l in losses), (losses, ref_losses[step])
    torch.cuda.synchronize()
    ref_fields = _field_rows(t_ref, sizes)
    for be in owners:
        be.e.check_error_flag()
        for rows, f in zip(_field_rows(be.table, [sizes[f] for f in be.fields]), be.fields):
            np.testing.assert_array_equal(rows, ref_fields[f], err_msg=f"owner {be.rank} field {f}")
        np.testing.assert_array_equal(be.table.bias.cpu().numpy(), t_ref.bias.cpu().numpy())


def test_an_owner_without_fields_is_refused():
    import fmx
    from fmx.owner import HipOwnerBackend
    with pytest.raises(ValueError, match="would own no field"):
        HipOwnerBackend([7, 3, 50], 4, fmx.Hyper(**HYP), "sgd", "logits", 1, 4)      # 3 fields in 64 lane groups: rank 1 of 4 gets none


def test_owner_class_single_rank_with_prefetch_equals_plain_steps():
    import fmx
    from fmx.owner import FieldOwnerFM, HipOwnerBackend
    sizes, k, B, steps = CRITEO_SIZES, 16, 4096, 5
    batches = _batches(sizes, B, steps)
    t_ref, ref_losses = _plain_run(fmx, sizes, k, "ftrl", batches)
    V, w = _weights(sizes, k)
    be = HipOwnerBackend(sizes, k, fmx.Hyper(**HYP), "ftrl", "logits", 0, 1, ftrl=HYP, max_local_batch=B)
    _load(be.table, sizes, be.fields, V, w)
    fo = FieldOwnerFM(be)
    data = [(torch.from_numpy(i).cuda(), torch.from_numpy(y).cuda()) for i, y in batches]
    work = torch.cuda.Stream()
    losses = []
    with torch.cuda.stream(work):
        tokens = {0: fo.prefetch(data[0][0]), 1: fo.prefetch(data[1][0])}
        for i, (idx_d, y_d) in enumerate(data):
            losses.append(fo.step(idx_d, y_d, tokens.pop(i, None)).clone())
            if i + 2 < steps and i != 1:                                                    # step 3 runs with nothing prepared
                tokens[i + 2] = fo.prefetch(data[i + 2][0])
    torch.cuda.synchronize()
    be.e.check_error_flag()
    assert [float(l[0]) for l in losses] == ref_losses
    np.testing.assert_array_equal(be.table.rows.cpu().numpy(), t_ref.rows.cpu().numpy())
    np.testing.assert_array_equal(be.table.bias.cpu().numpy(), t_ref.bias.cpu().numpy())


# ---- two processes sharing the GPU, gloo ----
FRAPPE_SIZES = [1_000_000] * 10                    # Frappe-shaped: 10 one-hot fields, 10 M rows in all (BASELINE configs[4])


def _owner_run(rank, world, sizes, k, B, steps, frappe):
    import fmx
    from fmx.owner import FieldOwnerFM, HipOwnerBackend
    hyp = fmx.Hyper(**HYP)
    be = HipOwnerBackend(sizes, k, hyp, "ftrl", "logits", rank, world, ftrl=HYP, max_local_batch=B)
    if frappe:                                      # a 2.5 GB table: seeded on the device, field by field (the same numbers on every rank)
        offs = np.concatenate([[0], np.cumsum([sizes[f] for f in be.fields])])
        for l, f in enumerate(be.fields):
            g = torch.Generator(device="cuda").manual_seed(100 + f)
            Vf = torch.randn((sizes[f], k), generator=g, device="cuda") * 0.3
            be.table.rows[offs[l]:offs[l + 1], :k] = Vf
            zo = be.table.z_offset
            be.table.rows[offs[l]:offs[l + 1], zo:zo + k] = fmx.table.ftrl_z_for_weight_torch(Vf, be.table.ftrl)
    else:
        V, w = _weights(sizes, k)
        _load(be.table, sizes, be.fields, V, w)
    fo = FieldOwnerFM(be)
    sl = slice(rank * B, (rank + 1) * B)
    data = [(torch.from_numpy(i[sl].copy()).cuda(), torch.from_numpy(y[sl].copy()).cuda()) for i, y in _batches(sizes, B * world, steps)]
    losses, tok = [], fo.prefetch(data[0][0])
    for i, (idx_d, y_d) in enumerate(data):
        losses.append(float(fo.step(idx_d, y_d, tok)[0]))
        tok = fo.prefetch(data[i + 1][0]) if i + 1 < steps else None
    torch.cuda.synchronize()
    be.e.check_error_flag()
    if frappe:                                      # digest: the touched rows are a tiny part of 2.5 GB
        r = be.table.rows
        digest = [float(r[:, :k].double().sum()), float(r[:, be.table.z_offset:].double().abs().sum()), float((r[:, k + 1] != 0).sum())]
        return losses, be.fields, digest, be.table.bias.cpu().numpy()
    return losses, be.fields, _field_rows(be.table, [sizes[f] for f in be.fields]), be.table.bias.cpu().numpy()


def _worker(rank, world, port, q, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "fm-for-online-recommendation_amd")]
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
def test_two_owner_processes_equal_the_one_table_step():
    import fmx
    sizes, k, B, steps = CRITEO_SIZES, 16, 512, 4
    res = _spawn(2, (sizes, k, B, steps, False))
    t_ref, ref_losses = _plain_run(fmx, sizes, k, "ftrl", _batches(sizes, B * 2, steps))
    ref_fields = _field_rows(t_ref, sizes)
    for rank, losses, fields, rows, bias in res:
        assert losses == ref_losses
        for r, f in zip(rows, fields):
            np.testing.assert_array_equal(r, ref_fields[f], err_msg=f"rank {rank} field {f}")
        np.testing.assert_array_equal(bias, t_ref.bias.cpu().numpy())
    assert sorted(res[0][2] + res[1][2]) == list(range(len(sizes)))


@pytest.mark.timeout(900)
def test_frappe_shaped_10m_row_table_split_over_two_owners():
    """BASELINE configs[4]'s table: 10 M rows (2.5 GB in the FTRL layout) -- here TRAINED (forward + fused FTRL update), split
    over two ranks by lane group (8 and 2 of the 10 fields: ownership follows the forward wavefront's 16 lane groups, so
    fewer fields than groups split unevenly); against one process holding all of it (the same seeded rows)."""
    sizes, k, B, steps = FRAPPE_SIZES, 16, 1024, 3
    res = _spawn(2, (sizes, k, B, steps, True))
    one = _owner_run(0, 1, sizes, k, B * 2, steps, True)                          # world 1: every field on one rank, global batches
    for rank, losses, fields, digest, bias in res:
        assert losses == one[0]
        np.testing.assert_array_equal(bias, one[3])
        assert len(fields) == (8, 2)[rank]      # lane groups 0-7 | 8-15 of the forward wavefront: fields 0-7 | 8-9 (10 fields, 16 groups)
    # the digests of the two shards add up to the one table's (sums over disjoint row sets; float64 accumulation)
    for j in range(3):
        tot = res[0][3][j] + res[1][3][j]
        assert abs(tot - one[2][j]) <= 1e-9 * max(1.0, abs(one[2][j])), (j, tot, one[2][j])


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
