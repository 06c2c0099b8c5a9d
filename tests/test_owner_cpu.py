"""The field-owner multi-GPU mode (fmx.owner.FieldOwnerFM) on CPU: a world_size-2 gloo run with an oracle-backed compute
backend (test infrastructure: the product has only the HIP backend) must leave the SAME BITS as one process stepping the
same global batches -- every row updated by its owner only, the replicated bias identically on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fm_oracle as orc

SIZES = [3, 9, 200, 1000, 17, 5, 64, 2, 31, 400, 7, 7, 90, 11, 1500, 6, 25, 300, 4, 50]      # 20 fields: two passes of 16 lane groups
K, B_LOCAL, STEPS = 16, 40, 3
SLOTS = 16
HYP = dict(alpha=0.05, beta=1.0, l1=0.001, l2=0.01)
f32 = np.float32


def tree(parts):
    """Adjacent pairs, then pairs of pairs, ...: the butterfly's order (len(parts) a power of two)."""
    parts = list(parts)
    while len(parts) > 1:
        parts = [(parts[i] + parts[i + 1]).astype(f32) for i in range(0, len(parts), 2)]
    return parts[0]


class OracleOwnerBackend:
    """numpy stand-in for fmx.owner.HipOwnerBackend with the same summation tree, so that G ranks and one rank add the same
    numbers in the same order: the rank's pieces (fmx.plan.OwnerPlan) sit at fixed positions; per lane group the passes in
    order, a pairwise tree over the lane groups of a block, then over the blocks.  The state holds EVERY row (column-major,
    plus one scratch row at the end that absent occurrences are pointed at with x = 0); a rank only touches its pieces' rows."""

    def __init__(self, state, offs, rank, world, plan):
        self.st, self.offs, self.rank, self.world, self.plan = state, offs, rank, world, plan
        self.fields = plan.owner_fields(rank)                       # local fields in position order: (column, first index, rows)
        self.nb, self.nlb, self.sl, self.np = plan.nb, plan.block_count[rank], plan.sl, plan.np
        self.block_count = list(plan.block_count)
        self.scratch = int(offs[-1])

    def _occ(self, idx_all):
        """rows [GB, n_local_fields] (the scratch row where the sample's index lies outside the piece) and the 0/1 mask."""
        idx = idx_all.numpy().astype(np.int64)
        rows = np.full((idx.shape[0], len(self.fields)), self.scratch, dtype=np.int64)
        ok = np.zeros(rows.shape, dtype=bool)
        for l, (c, b, r) in enumerate(self.fields):
            ok[:, l] = (idx[:, c] >= b) & (idx[:, c] < b + r)
            rows[ok[:, l], l] = self.offs[c] + idx[ok[:, l], c]
        return rows, ok

    def partial_forward(self, idx_all, B_local):
        st = self.st
        V = orc.ftrl_weight(st["zV"], st["nV"], **HYP)
        w = orc.ftrl_weight(st["zw"], st["nw"], **HYP)
        rows, ok = self._occ(idx_all)
        GB, G = rows.shape[0], rows.shape[0] // B_local
        rec = np.zeros((G, self.nlb, B_local, 2 * K + 4), f32)
        for lb in range(self.nlb):
            zero = np.zeros((GB, K), f32)
            S_slot = [zero.copy() for _ in range(self.sl)]
            SS_slot = [zero.copy() for _ in range(self.sl)]
            fo_slot = [np.zeros(GB, f32) for _ in range(self.sl)]
            for p in range(self.np):                                  # passes in order per lane group; absent rows add nothing
                for s in range(self.sl):
                    l = (lb * self.np + p) * self.sl + s
                    m = ok[:, l]
                    e = V[rows[:, l]]
                    S_slot[s] = np.where(m[:, None], (S_slot[s] + e).astype(f32), S_slot[s])
                    SS_slot[s] = np.where(m[:, None], (SS_slot[s] + e * e).astype(f32), SS_slot[s])
                    fo_slot[s] = np.where(m, (fo_slot[s] + w[rows[:, l]]).astype(f32), fo_slot[s])
            S, SS, fo = tree(S_slot), tree(SS_slot), tree(fo_slot)
            rec[:, lb, :, :K] = S.reshape(G, B_local, K)
            rec[:, lb, :, K:2 * K] = SS.reshape(G, B_local, K)
            rec[:, lb, :, 2 * K] = fo.reshape(G, B_local)
        return torch.from_numpy(rec.reshape(-1, 2 * K + 4))

    def finish(self, mine, y_local, inv_b):
        m = mine.numpy()
        S, SS, fo = tree([m[r, :, :K] for r in range(m.shape[0])]), tree([m[r, :, K:2 * K] for r in range(m.shape[0])]), \
            tree([m[r, :, 2 * K] for r in range(m.shape[0])])
        bi = ((S * S - SS) * f32(0.5)).astype(f32)
        b = orc.ftrl_weight(self.st["zb"], self.st["nb"], **HYP)
        logit = (fo + bi.sum(axis=1, dtype=f32) + b).astype(f32)
        y = y_local.numpy()
        rec = np.zeros((len(y), K + 4), f32)
        rec[:, :K] = S
        rec[:, K] = orc.dloss_dlogit(logit, y, "logits", inv_b)
        rec[:, K + 1] = orc.loss_value(logit, y, "logits")
        return torch.from_numpy(rec)

    def update(self, idx_all, rec_g, inv_b):
        st = self.st
        V = orc.ftrl_weight(st["zV"], st["nV"], **HYP)
        rows, ok = self._occ(idx_all)
        rec = rec_g.numpy()
        S, dz, loss_g = np.ascontiguousarray(rec[:, :K]), np.ascontiguousarray(rec[:, K]), rec[:, K + 1]
        x = ok.astype(f32)                                            # absent occurrences: x = 0 on the scratch row
        u, dV, dw = orc.flat_row_gradients(V, rows, x, S, dz, np.repeat(dz[:, None], K, axis=1))
        keep = u != self.scratch
        u, dV, dw = u[keep], dV[keep], dw[keep]
        st["zV"][u], st["nV"][u] = orc.ftrl_step(st["zV"][u], st["nV"][u], dV, **HYP)
        st["zw"][u], st["nw"][u] = orc.ftrl_step(st["zw"][u], st["nw"][u], dw, **HYP)
        st["zb"], st["nb"] = orc.ftrl_step(st["zb"], st["nb"], dz.sum(dtype=f32), **HYP)      # replicated, identical everywhere
        return torch.tensor([float(loss_g.sum(dtype=f32) * f32(inv_b))])


def make_state():
    rng = np.random.default_rng(3)
    offs = np.concatenate([[0], np.cumsum(SIZES)]).astype(np.int64)
    R = int(offs[-1])
    V = (rng.normal(size=(R + 1, K)) * 0.3).astype(f32)            # (+ the scratch row)
    w = (rng.normal(size=R + 1) * 0.3).astype(f32)
    st = dict(zV=orc.ftrl_z_for_weight(V, **HYP), nV=np.zeros_like(V), zw=orc.ftrl_z_for_weight(w, **HYP),
              nw=np.zeros_like(w), zb=f32(0.1), nb=f32(0.0))
    return st, offs


def make_batches(world):
    rng = np.random.default_rng(9)
    GB = B_LOCAL * world
    return [(np.stack([rng.integers(0, s, size=GB) for s in SIZES], axis=1).astype(np.int32),
             (rng.uniform(size=GB) < 0.3).astype(f32)) for _ in range(STEPS)]


WORLDS = (2, 3)


def make_plan(world):
    from fmx.plan import OwnerPlan
    return OwnerPlan(SIZES, K, world, global_batch=B_LOCAL * world)


class WholePlan:
    """The same pieces at the same positions held by ONE rank: what `world` owners must be bit-identical to."""

    def __init__(self, plan):
        self.nb, self.sl, self.np, self.block_count = plan.nb, plan.sl, plan.np, [plan.nb]
        self._fields = sum((plan._fields_of_block(b) for b in range(plan.nb)), [])

    def owner_fields(self, g):
        return self._fields


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fmx.owner import FieldOwnerFM
    st, offs = make_state()
    be = OracleOwnerBackend(st, offs, rank, world, make_plan(world))
    fo = FieldOwnerFM(be)
    sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
    losses = [float(fo.step(torch.from_numpy(idx[sl].copy()), torch.from_numpy(y[sl].copy()))[0]) for idx, y in make_batches(world)]
    q.put((rank, losses, [f for f in be.fields if f[2]], {k: np.asarray(v).copy() for k, v in st.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_plans_partition_the_rows():
    """Every row has exactly one owner, every owner has rows, the positions of the pieces are distinct -- for any number of
    columns and any world size up to the number of lane groups (powers of two or not)."""
    from fmx.plan import OwnerPlan
    cases = ((SIZES, 16), ([1000] * 10, 16), ([7, 3, 50], 16), ([5] * 70, 4), ([40000, 3, 9], 64), (list(range(1, 40)), 10))
    for sizes, k in cases:
        slots = 64 // (max(4, 1 << (k - 1).bit_length()) // 4)
        for world in range(1, min(slots, 9) + 1):
            if sum(sizes) < world:
                continue
            plan = OwnerPlan(sizes, k, world, global_batch=256 * world)
            covered = [np.zeros(s, dtype=np.int32) for s in sizes]
            for g in range(world):
                fields = plan.owner_fields(g)
                assert len(fields) == plan.block_count[g] * plan.np * plan.sl
                assert sum(r for _, _, r in fields) >= 1, (sizes, world, g)
                for c, b, r in fields:
                    covered[c][b:b + r] += 1
            assert all((c == 1).all() for c in covered), (sizes, world)
            whole = plan.whole_fields()
            assert sorted(f for f in whole if f[2]) == sorted(plan.pieces)
            if world == 1:
                assert [f for f in whole[:len(sizes)]] == [(c, 0, r) for c, r in enumerate(sizes)]     # the ordinary table
    with pytest.raises(ValueError):
        OwnerPlan([1, 1], 16, 4)                                                 # two rows cannot be dealt over four owners


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_owners_equal_one_owner_bit_for_bit(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from fmx.owner import FieldOwnerFM
    st, offs = make_state()
    one = FieldOwnerFM(OracleOwnerBackend(st, offs, 0, 1, WholePlan(make_plan(world))))
    ref_losses = [float(one.step(torch.from_numpy(idx), torch.from_numpy(y))[0]) for idx, y in make_batches(world)]
    st0, _ = make_state()
    owner_of = {}
    for rank, losses, fields, state in res:
        assert losses == ref_losses                                              # the same bits, not "close"
        assert state["zb"] == st["zb"] and state["nb"] == st["nb"]               # replicated bias: identical on every rank
        mine = np.zeros(int(offs[-1]) + 1, dtype=bool)
        for c, b, r in fields:
            mine[int(offs[c]) + b:int(offs[c]) + b + r] = True
            assert (c, b, r) not in owner_of
            owner_of[(c, b, r)] = rank
        for k in ("zV", "nV", "zw", "nw"):
            np.testing.assert_array_equal(state[k][mine], st[k][mine], err_msg=f"rank {rank} {k}: its own rows")   # the owner holds the trained rows ...
            np.testing.assert_array_equal(state[k][~mine], st0[k][~mine], err_msg=f"rank {rank} {k}: foreign rows")  # ... nobody else touches them
    assert sum(r for _, _, r in owner_of) == sum(SIZES)
    assert any((st[k] != st0[k]).any() for k in ("zV", "zw"))                     # the steps did train


# ---- DeepFM on field owners (fmx.deep.OwnerDeepFMTrainer) on CPU: the exchange logic of the scalable DeepFM step ----
class OracleDeepOwnerBackend(OracleOwnerBackend):
    """The owner backend for a weights-layout table under SGD plus what OwnerDeepFMTrainer needs: the finish without a loss, the
    MLP section (PyTorch autograd here; fmx_mlp_section in the product) and the update from (S | dlogit | dL/dbi) records."""
    kp = K
    LR = 0.01

    def _weights(self):
        return self.st["V"], self.st["w"]

    def partial_forward(self, idx_all, B_local):
        V, w = self._weights()
        rows, ok = self._occ(idx_all)
        GB, G = rows.shape[0], rows.shape[0] // B_local
        rec = np.zeros((G, self.nlb, B_local, 2 * K + 4), f32)
        for lb in range(self.nlb):
            S_slot = [np.zeros((GB, K), f32) for _ in range(self.sl)]
            SS_slot = [np.zeros((GB, K), f32) for _ in range(self.sl)]
            fo_slot = [np.zeros(GB, f32) for _ in range(self.sl)]
            for p in range(self.np):
                for s in range(self.sl):
                    l = (lb * self.np + p) * self.sl + s
                    m = ok[:, l]
                    e = V[rows[:, l]]
                    S_slot[s] = np.where(m[:, None], (S_slot[s] + e).astype(f32), S_slot[s])
                    SS_slot[s] = np.where(m[:, None], (SS_slot[s] + e * e).astype(f32), SS_slot[s])
                    fo_slot[s] = np.where(m, (fo_slot[s] + w[rows[:, l]]).astype(f32), fo_slot[s])
            rec[:, lb, :, :K] = tree(S_slot).reshape(G, B_local, K)
            rec[:, lb, :, K:2 * K] = tree(SS_slot).reshape(G, B_local, K)
            rec[:, lb, :, 2 * K] = tree(fo_slot).reshape(G, B_local)
        return torch.from_numpy(rec.reshape(-1, 2 * K + 4))

    def finish_bi(self, mine):
        m = mine.numpy()
        S, SS, fo = tree([m[r, :, :K] for r in range(m.shape[0])]), tree([m[r, :, K:2 * K] for r in range(m.shape[0])]), \
            tree([m[r, :, 2 * K] for r in range(m.shape[0])])
        bi = ((S * S - SS) * f32(0.5)).astype(f32)
        rec = np.zeros((S.shape[0], 2 * K + 4), f32)
        rec[:, :K] = S
        logit = (fo + bi.sum(axis=1, dtype=f32) + self.st["bias"]).astype(f32)
        return torch.from_numpy(rec), torch.from_numpy(bi), torch.from_numpy(fo.astype(f32)), torch.from_numpy(logit)

    def bias_weight(self):
        return torch.tensor(float(self.st["bias"]))

    def mlp_section(self, flat, gflat, k, hidden, n_layers, loss, bi, base, y, B, inv_b, lr_apply):
        import torch.nn.functional as F
        ws, off = [], 0
        for l in range(n_layers):
            n_in = k if l == 0 else hidden
            W = flat[off:off + hidden * n_in].view(hidden, n_in).detach().clone().requires_grad_(True)
            off += hidden * n_in
            b = flat[off:off + hidden].detach().clone().requires_grad_(True)
            off += hidden
            ws += [W, b]
        bi_l = bi[:, :k].detach().clone().requires_grad_(True)
        base_l = base.detach().clone().requires_grad_(True)
        x = bi_l
        for l in range(n_layers):
            x = F.relu(F.linear(x, ws[2 * l], ws[2 * l + 1]))
        lo = F.binary_cross_entropy_with_logits(base_l + x.sum(1), y, reduction="sum") * inv_b
        lo.backward()
        gflat.copy_(torch.cat([p.grad.reshape(-1) for p in ws]))
        if lr_apply:
            flat.sub_(gflat, alpha=lr_apply)
        return lo.detach().reshape(1), base_l.grad.contiguous(), bi_l.grad.contiguous()

    def update_deep(self, idx_all, rec_all, fm_term, inv_b):
        V, w = self._weights()
        rows, ok = self._occ(idx_all)
        rec = rec_all.numpy()
        S, dz, gbi = np.ascontiguousarray(rec[:, :K]), np.ascontiguousarray(rec[:, K]), np.ascontiguousarray(rec[:, K + 4:])
        G = (gbi + dz[:, None]).astype(f32) if fm_term else gbi
        u, dV, dw = orc.flat_row_gradients(V, rows, ok.astype(f32), S, dz, G)
        keep = u != self.scratch
        u, dV, dw = u[keep], dV[keep], dw[keep]
        V[u] = (V[u] - f32(self.LR) * dV).astype(f32)
        w[u] = (w[u] - f32(self.LR) * dw).astype(f32)
        self.st["bias"] = f32(self.st["bias"] - f32(self.LR) * dz.sum(dtype=f32))


def _deep_state():
    rng = np.random.default_rng(4)
    offs = np.concatenate([[0], np.cumsum(SIZES)]).astype(np.int64)
    R = int(offs[-1])
    return dict(V=(rng.normal(size=(R + 1, K)) * 0.3).astype(f32), w=(rng.normal(size=R + 1) * 0.3).astype(f32), bias=f32(0.1)), offs


def _deep_trainer(be):
    import torch.nn as nn
    from fmx.deep import OwnerDeepFMTrainer
    torch.manual_seed(11)
    layers = [nn.Linear(K if j == 0 else 32, 32) for j in range(2)]
    return OwnerDeepFMTrainer(be, layers, K, mlp_lr=0.01)


def _deep_owner_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st, offs = _deep_state()
    be = OracleDeepOwnerBackend(st, offs, rank, world, make_plan(world))
    tr = _deep_trainer(be)
    sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
    losses = [float(tr.step(torch.from_numpy(idx[sl].copy()), torch.from_numpy(y[sl].copy()))) for idx, y in make_batches(world)]
    q.put((rank, losses, [f for f in be.fields if f[2]], {k: np.asarray(v).copy() for k, v in st.items()}, tr.flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_deepfm_on_field_owners_two_ranks_equal_one_rank():
    """The scalable DeepFM step (fmx.deep.OwnerDeepFMTrainer) on CPU, world size 2 over gloo, oracle-backed compute: per step ONE
    all-to-all and ONE record all-gather on the critical path plus ONE all-reduce of the flattened MLP gradients; against one
    rank holding every block of the same plan.  First-step table rows identical; then 1e-5 (the all-reduce's summation order)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_deep_owner_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st, offs = _deep_state()
    one = _deep_trainer(OracleDeepOwnerBackend(st, offs, 0, 1, WholePlan(make_plan(world))))
    ref_losses = [float(one.step(torch.from_numpy(idx), torch.from_numpy(y))) for idx, y in make_batches(world)]
    st0, _ = _deep_state()
    np.testing.assert_array_equal(res[0][4], res[1][4])                          # the replicated MLP: the same on both ranks
    np.testing.assert_allclose(res[0][4], one.flat.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([a + b for a, b in zip(res[0][1], res[1][1])], ref_losses, rtol=1e-5)
    for rank, losses, fields, state, _ in res:
        mine = np.zeros(int(offs[-1]) + 1, dtype=bool)
        for c, b, r in fields:
            mine[int(offs[c]) + b:int(offs[c]) + b + r] = True
        for k in ("V", "w"):
            np.testing.assert_allclose(state[k][mine], st[k][mine], rtol=1e-5, atol=1e-6, err_msg=f"rank {rank} {k}")
            np.testing.assert_array_equal(state[k][~mine], st0[k][~mine], err_msg=f"rank {rank} {k}: foreign rows")
        np.testing.assert_allclose(state["bias"], st["bias"], rtol=1e-5)
    assert (st["V"] != st0["V"]).any()
