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
    numbers in the same order: per lane group the fields slot, 16 + slot, ... in order, then the pairwise tree."""

    def __init__(self, state, offs, rank, world):
        from fmx.owner import owner_fields
        self.st, self.offs, self.rank, self.world = state, offs, rank, world
        self.fields = owner_fields(len(SIZES), K, world, rank)
        self.sl = SLOTS // world
        self.cols = torch.tensor(self.fields, dtype=torch.long)

    def select(self, idx_all):
        return idx_all.index_select(1, self.cols).contiguous()

    def _rows(self, idx_own):
        return idx_own.numpy().astype(np.int64) + self.offs[self.fields][None, :]

    def partial_forward(self, idx_own):
        st = self.st
        V = orc.ftrl_weight(st["zV"], st["nV"], **HYP)
        w = orc.ftrl_weight(st["zw"], st["nw"], **HYP)
        rows = self._rows(idx_own)
        GB = rows.shape[0]
        zero = np.zeros((GB, K), f32)
        S_slot, SS_slot, fo_slot = [zero.copy() for _ in range(self.sl)], [zero.copy() for _ in range(self.sl)], [np.zeros(GB, f32) for _ in range(self.sl)]
        for l, f in enumerate(self.fields):           # local field l = pass * SL + lane group: passes in order per lane group
            s = l % self.sl
            e = V[rows[:, l]]
            S_slot[s] = (S_slot[s] + e).astype(f32)
            SS_slot[s] = (SS_slot[s] + e * e).astype(f32)
            fo_slot[s] = (fo_slot[s] + w[rows[:, l]]).astype(f32)
        rec = np.zeros((GB, 2 * K + 4), f32)
        rec[:, :K], rec[:, K:2 * K], rec[:, 2 * K] = tree(S_slot), tree(SS_slot), tree(fo_slot)
        return torch.from_numpy(rec)

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

    def update(self, idx_own, rec_g, inv_b):
        st = self.st
        V = orc.ftrl_weight(st["zV"], st["nV"], **HYP)
        rows = self._rows(idx_own)
        rec = rec_g.numpy()
        S, dz, loss_g = np.ascontiguousarray(rec[:, :K]), np.ascontiguousarray(rec[:, K]), rec[:, K + 1]
        x = np.ones(rows.shape, dtype=f32)
        u, dV, dw = orc.flat_row_gradients(V, rows, x, S, dz, np.repeat(dz[:, None], K, axis=1))
        st["zV"][u], st["nV"][u] = orc.ftrl_step(st["zV"][u], st["nV"][u], dV, **HYP)
        st["zw"][u], st["nw"][u] = orc.ftrl_step(st["zw"][u], st["nw"][u], dw, **HYP)
        st["zb"], st["nb"] = orc.ftrl_step(st["zb"], st["nb"], dz.sum(dtype=f32), **HYP)      # replicated, identical everywhere
        return torch.tensor([float(loss_g.sum(dtype=f32) * f32(inv_b))])


def make_state():
    rng = np.random.default_rng(3)
    offs = np.concatenate([[0], np.cumsum(SIZES)]).astype(np.int64)
    R = int(offs[-1])
    V = (rng.normal(size=(R, K)) * 0.3).astype(f32)
    w = (rng.normal(size=R) * 0.3).astype(f32)
    st = dict(zV=orc.ftrl_z_for_weight(V, **HYP), nV=np.zeros_like(V), zw=orc.ftrl_z_for_weight(w, **HYP),
              nw=np.zeros_like(w), zb=f32(0.1), nb=f32(0.0))
    return st, offs[:-1]


def make_batches(world):
    rng = np.random.default_rng(9)
    GB = B_LOCAL * world
    return [(np.stack([rng.integers(0, s, size=GB) for s in SIZES], axis=1).astype(np.int32),
             (rng.uniform(size=GB) < 0.3).astype(f32)) for _ in range(STEPS)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fmx.owner import FieldOwnerFM
    st, offs = make_state()
    be = OracleOwnerBackend(st, offs, rank, world)
    fo = FieldOwnerFM(be)
    sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
    losses = [float(fo.step(torch.from_numpy(idx[sl].copy()), torch.from_numpy(y[sl].copy()))[0]) for idx, y in make_batches(world)]
    q.put((rank, losses, be.fields, {k: np.asarray(v).copy() for k, v in st.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_fields_partition_the_fields():
    from fmx.owner import owner_fields
    for F, k in ((39, 16), (39, 10), (10, 16), (70, 4), (5, 64), (20, 16)):
        slots = 64 // (max(4, 1 << (k - 1).bit_length()) // 4) if k > 4 else 64
        for world in (1, 2, 4, 8, 16):
            if slots % world:
                with pytest.raises(ValueError):
                    owner_fields(F, k, world, 0)
                continue
            owned = [owner_fields(F, k, world, g) for g in range(world)]
            assert sorted(sum(owned, [])) == list(range(F))                     # every field has exactly one owner
            sl = slots // world
            for g, fs in enumerate(owned):                                      # local field l is in lane group l % SL of rank g
                assert all((f % slots) == g * sl + (l % sl) and f // slots == l // sl for l, f in enumerate(fs))
    with pytest.raises(ValueError):
        owner_fields(39, 16, 3, 0)


@pytest.mark.timeout(300)
def test_two_owners_equal_one_owner_bit_for_bit():
    world = 2
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
    one = FieldOwnerFM(OracleOwnerBackend(st, offs, 0, 1))
    ref_losses = [float(one.step(torch.from_numpy(idx), torch.from_numpy(y))[0]) for idx, y in make_batches(world)]
    st0, _ = make_state()
    offs_full = np.concatenate([offs, [sum(SIZES)]])
    seen = set()
    for rank, losses, fields, state in res:
        assert losses == ref_losses                                              # the same bits, not "close"
        assert state["zb"] == st["zb"] and state["nb"] == st["nb"]               # replicated bias: identical on every rank
        for f in range(len(SIZES)):
            rows = slice(int(offs_full[f]), int(offs_full[f + 1]))
            for k in ("zV", "nV", "zw", "nw"):
                if f in fields:                                                  # the owner holds the trained rows ...
                    np.testing.assert_array_equal(state[k][rows], st[k][rows], err_msg=f"rank {rank} field {f} {k}")
                else:                                                            # ... and nobody else ever touches them
                    np.testing.assert_array_equal(state[k][rows], st0[k][rows], err_msg=f"rank {rank} foreign field {f} {k}")
        seen |= set(fields)
    assert seen == set(range(len(SIZES)))
    assert any((st[k] != st0[k]).any() for k in ("zV", "zw"))                     # the steps did train
