"""The N > 1 host logic on CPU: world_size-2 gloo run of fmx.DataParallelFM with an oracle-backed compute backend
(test infrastructure) must equal the single-process step on the same global batch, and both ranks must end with
bit-identical replicas."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fm_oracle as orc

SIZES = [3, 9, 200, 1000, 17]
K, B_LOCAL, STEPS = 8, 48, 3
HYP = dict(alpha=0.05, beta=1.0, l1=0.001, l2=0.01)


class OracleBackend:
    """Test stand-in for fmx.HipBackend: the numpy oracle behind the same two calls (FTRL rule, BCEwl(z))."""

    def __init__(self, state, offs):
        self.st, self.offs = state, offs

    def forward(self, idx, y, inv_b):
        st = self.st
        V = orc.ftrl_weight(st["zV"], st["nV"], **HYP)
        w = orc.ftrl_weight(st["zw"], st["nw"], **HYP)
        b = orc.ftrl_weight(st["zb"], st["nb"], **HYP)
        rows = idx.numpy().astype(np.int64) + self.offs[None, :]
        fw = orc.flat_forward(V, w, b, rows, np.ones(rows.shape, dtype=np.float32))
        yv = y.numpy()
        rec = np.zeros((len(yv), K + 4), dtype=np.float32)      # the product's record: S | dz | loss | pad
        rec[:, :K] = fw["S"]
        rec[:, K] = orc.dloss_dlogit(fw["logit"], yv, "logits", inv_b)
        rec[:, K + 1] = orc.loss_value(fw["logit"], yv, "logits")
        return torch.from_numpy(rec)

    def update(self, idx_g, rec_g, inv_b):
        st = self.st
        V = orc.ftrl_weight(st["zV"], st["nV"], **HYP)
        rows = idx_g.numpy().astype(np.int64) + self.offs[None, :]
        rec = rec_g.numpy()
        S, dz, loss_g = np.ascontiguousarray(rec[:, :K]), np.ascontiguousarray(rec[:, K]), rec[:, K + 1]
        x = np.ones(rows.shape, dtype=np.float32)
        u, dV, dw = orc.flat_row_gradients(V, rows, x, S, dz, np.repeat(dz[:, None], V.shape[1], axis=1))
        st["zV"][u], st["nV"][u] = orc.ftrl_step(st["zV"][u], st["nV"][u], dV, **HYP)
        st["zw"][u], st["nw"][u] = orc.ftrl_step(st["zw"][u], st["nw"][u], dw, **HYP)
        st["zb"], st["nb"] = orc.ftrl_step(st["zb"], st["nb"], dz.sum(dtype=np.float32), **HYP)
        return torch.tensor([float(loss_g.sum(dtype=np.float32) * np.float32(inv_b))])


def make_state():
    rng = np.random.default_rng(3)
    offs = np.concatenate([[0], np.cumsum(SIZES)]).astype(np.int64)
    R = int(offs[-1])
    V = (rng.normal(size=(R, K)) * 0.3).astype(np.float32)
    w = (rng.normal(size=R) * 0.3).astype(np.float32)
    st = dict(zV=orc.ftrl_z_for_weight(V, **HYP), nV=np.zeros_like(V), zw=orc.ftrl_z_for_weight(w, **HYP),
              nw=np.zeros_like(w), zb=np.float32(0.0), nb=np.float32(0.0))
    return st, offs[:-1]


def make_batches(world):
    rng = np.random.default_rng(9)
    GB = B_LOCAL * world
    out = []
    for _ in range(STEPS):
        idx = np.stack([rng.integers(0, s, size=GB) for s in SIZES], axis=1).astype(np.int32)
        y = (rng.uniform(size=GB) < 0.3).astype(np.float32)
        out.append((idx, y))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fmx
    st, offs = make_state()
    dp = fmx.DataParallelFM(OracleBackend(st, offs))
    losses = []
    sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
    batches = [(torch.from_numpy(idx[sl].copy()), torch.from_numpy(y[sl].copy())) for idx, y in make_batches(world)]
    tok = None
    for i, (idx_t, y_t) in enumerate(batches):
        losses.append(float(dp.step(idx_t, y_t, tok)[0]))
        tok = dp.prefetch(batches[i + 1][0]) if (i + 1 < len(batches) and i % 2 == 0) else None   # (CPU tensors: no token, the in-line path)
    q.put((rank, losses, {k: np.asarray(v).copy() for k, v in st.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process on the same global batches
    import fmx
    st, offs = make_state()
    dp1 = fmx.DataParallelFM(OracleBackend(st, offs))
    ref_losses = [float(dp1.step(torch.from_numpy(idx), torch.from_numpy(y))[0]) for idx, y in make_batches(world)]
    for rank, losses, state in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-6)
        for k in st:
            np.testing.assert_array_equal(state[k], np.asarray(st[k]), err_msg=f"rank {rank} {k}")   # bit-identical
    for k in st:
        np.testing.assert_array_equal(res[0][2][k], res[1][2][k])


# ---------------------------------------------------------------------------------------------------------
# DeepFM: exact table data parallelism + one fused all-reduce of the dense (MLP) gradients
# ---------------------------------------------------------------------------------------------------------
class OracleDeepBackend:
    """Test stand-in for fmx.HipDeepBackend (weights layout, SGD rule on the tables)."""

    def __init__(self, V, w, bias, offs, lr):
        self.V, self.w, self.b, self.offs, self.lr = V, w, np.float32(bias), offs, lr

    def forward(self, idx):
        rows = idx.numpy().astype(np.int64) + self.offs[None, :]
        fw = orc.flat_forward(self.V, self.w, self.b, rows, np.ones(rows.shape, dtype=np.float32))
        return (torch.from_numpy(fw["S"]), torch.from_numpy(fw["bi"]), torch.from_numpy(fw["sfirst"]),
                torch.from_numpy(fw["logit"]))

    def bias(self):
        return torch.tensor(float(self.b))

    def update(self, idx_g, S_g, dz_g, gbi_g, fm_term, inv_b):
        rows = idx_g.numpy().astype(np.int64) + self.offs[None, :]
        dz, gbi = dz_g.numpy(), gbi_g.numpy()[:, :self.V.shape[1]]
        G = (gbi + dz[:, None]).astype(np.float32) if fm_term else gbi.astype(np.float32)
        u, dV, dw = orc.flat_row_gradients(self.V, rows, np.ones(rows.shape, dtype=np.float32), S_g.numpy(), dz, G)
        self.V[u] = orc.sgd_step(self.V[u], dV, self.lr)
        self.w[u] = orc.sgd_step(self.w[u], dw, self.lr)
        self.b = orc.sgd_step(self.b, dz.sum(dtype=np.float32), self.lr)


def _make_deep(world):
    import fmx
    import torch.nn as nn
    rng = np.random.default_rng(4)
    offs = np.concatenate([[0], np.cumsum(SIZES)]).astype(np.int64)
    R = int(offs[-1])
    V = (rng.normal(size=(R, K)) * 0.3).astype(np.float32)
    w = (rng.normal(size=R) * 0.3).astype(np.float32)
    be = OracleDeepBackend(V, w, 0.2, offs[:-1], lr=0.05)
    torch.manual_seed(7)
    layers = [nn.Linear(K, 12), nn.Linear(12, 12)]
    return be, layers, fmx.DeepFMTrainer(be, layers, K, K, mlp_lr=0.05, fm_term=True, loss="logits")


def _deep_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    be, layers, tr = _make_deep(world)
    losses = []
    for idx, y in make_batches(world):
        sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
        l = tr.step(torch.from_numpy(idx[sl]), torch.from_numpy(y[sl])).clone()
        dist.all_reduce(l)
        losses.append(float(l))
    q.put((rank, losses, be.V.copy(), be.w.copy(), float(be.b), [p.detach().numpy().copy() for l_ in layers for p in l_.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_deepfm_two_ranks_equal_one_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.set_num_threads(1)
    be, layers, tr = _make_deep(1)
    ref_losses = [float(tr.step(torch.from_numpy(idx), torch.from_numpy(y))) for idx, y in make_batches(world)]
    ref_params = [p.detach().numpy() for l_ in layers for p in l_.parameters()]
    for rank, losses, V, w, b, params in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-6)
        np.testing.assert_allclose(V, be.V, rtol=1e-5, atol=1e-7)       # dz / gbi come from per-rank GEMMs: not bitwise
        np.testing.assert_allclose(w, be.w, rtol=1e-5, atol=1e-7)
        assert abs(b - float(be.b)) <= 1e-6
        for a, r in zip(params, ref_params):
            np.testing.assert_allclose(a, r, rtol=1e-5, atol=1e-7)
    # the two replicas of the 2-rank run are identical to each other, bit for bit
    np.testing.assert_array_equal(res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][3], res[1][3])
    for a, b_ in zip(res[0][5], res[1][5]):
        np.testing.assert_array_equal(a, b_)
