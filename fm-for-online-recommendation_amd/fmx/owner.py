"""Field-owner parallel online FM over the GPUs of one node: the mode in which the update WORK and the TABLE shard.

The replicated mode (fmx.distributed.DataParallelFM) keeps a full table on every rank and repeats the row-reduced update
of the GLOBAL batch on every rank: exact, but no faster than one GPU.  Here rank g OWNS a subset of the fields -- the
fields whose lane groups [g SL, (g + 1) SL) it would occupy in k_fm_forward's wavefront (include/fmx.h, "the forward pass
split over field owners") -- and holds only those fields' rows (a 10 M-row table spreads over the ranks' HBM).  One step on
a global batch of G x B samples, rank-major:

    all-gather   idx [B, F]            -> every rank picks the columns of its fields: idx_own [G B, F_own]     (weights-free:
    sort         idx_own               -> occurrence lists of the owned fields over the global batch            runs ahead, prefetch())
    partial fwd  fmx_fm_forward_partial -> per sample (S_part, ss_part, first_part) over the owned fields      [G B, 2 kp + 4]
    all-to-all   the records of rank r's samples go to rank r                                                    [G, B, 2 kp + 4]
    finish       fmx_fm_forward_finish -> S, logit, loss, dlogit of the B local samples (owner tree order)     [B, kp + 4]
    all-gather   (S, dlogit, loss)                                                                               [G B, kp + 4]
    update       k_fm_update on the owned fields' lists: every row is updated by its owner only; the bias (replicated) by
                 everybody, identically

Per rank the gather, the sort and the update touch G B x F / G = B x F occurrences: constant work per GPU as G grows.  No
embedding row crosses xGMI; per step two small collectives sit on the critical path (144 B and 80 B per sample at k = 16).
The additions are those of the one-GPU kernels in the same order, so G ranks give bit-identical tables, losses and biases
to one GPU stepping the same global batch (tests/test_owner_*.py).

The compute behind a step is a small backend interface so that the sharding / exchange logic runs on CPU with gloo in the
tests (an oracle-backed backend injected there; the product has only the HIP one).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib
from .engine import FMEngine, _ptr
from .table import FlatTable, padded_k


def owner_fields(n_fields, k, world, rank):
    """The global fields rank `rank` of `world` owns, in the order of its local table.  Lane group `slot` of the forward
    wavefront (SLOTS = 64 / (kp / 4) groups) adds the fields slot, SLOTS + slot, ...; rank g takes the groups
    [g SL, (g + 1) SL), SL = SLOTS / world.  Local field l = p * SL + sl is global field p * SLOTS + g * SL + sl; fields
    that do not exist (>= n_fields) can only be a suffix of the last pass, so the local numbering has no holes."""
    slots = 64 // (padded_k(k) // 4)
    if world < 1 or world & (world - 1) or slots % world:
        raise ValueError(f"field-owner mode needs a power-of-two number of ranks dividing {slots} (k = {k}); got {world}")
    sl = slots // world
    passes = (n_fields + slots - 1) // slots
    out = [p * slots + rank * sl + s for p in range(passes) for s in range(sl) if p * slots + rank * sl + s < n_fields]
    assert out == sorted(out)
    return out


class HipOwnerBackend:
    """The product backend: this rank's shard of the table (fmx.FlatTable over the owned fields) and an FMEngine on it."""

    N_SLOTS = 3   # batches whose indices may be gathered + sorted ahead of their update
    takes_stream = True

    def __init__(self, feature_sizes, k, hyper, rule, loss, rank, world, layout=None, ftrl=None, device=None, max_local_batch=4096):
        self.rank, self.world, self.rule, self.loss, self.hyper = rank, world, rule, loss, hyper
        self.feature_sizes = [int(s) for s in feature_sizes]
        self.fields = owner_fields(len(self.feature_sizes), k, world, rank)
        if not self.fields:
            raise ValueError(f"rank {rank} of {world} would own no field: {len(self.feature_sizes)} fields occupy fewer lane groups "
                             f"than there are ranks at k = {k}; use fewer ranks")
        layout = layout or ("ftrl" if rule == "ftrl" else "weights")
        self.table = FlatTable([self.feature_sizes[f] for f in self.fields], k, layout=layout, device=device, ftrl=ftrl)
        self.e = FMEngine(self.table, max_batch=max_local_batch * world)
        self.device = self.table.device
        self.cols = torch.tensor(self.fields, dtype=torch.long, device=self.device)
        self.kp, self.rec_in, self.rec_out = self.table.kp, 2 * self.table.kp + 4, self.table.kp + 4
        self.max_global_batch = _max_step_batch(max(self.feature_sizes))       # the same cap on every rank
        self._bufs, self._ws = {}, {}

    def _buf(self, name, shape, dtype=torch.float32):
        t = self._bufs.get((name, shape))
        if t is None:
            t = self._bufs[(name, shape)] = torch.zeros(shape, dtype=dtype, device=self.device)
        return t

    def select(self, idx_all):
        """[G B, F] gathered indices -> this rank's columns, contiguous int32 [G B, F_own]."""
        return idx_all.index_select(1, self.cols).contiguous()

    def partial_forward(self, idx_own, stream=None):
        GB = idx_own.shape[0]
        parts = self._buf("parts", (GB, self.rec_in))
        _lib.check(self.e.lib.fmx_fm_forward_partial(self.table.c_struct(), idx_own.data_ptr(), None, GB, self.world,
                                                    parts.data_ptr(), self.e.error.data_ptr(), self.e._stream(stream)))
        return parts

    def finish(self, parts_mine, y_local, inv_b, stream=None):
        """parts_mine [G, B, 2 kp + 4] -> records [B, kp + 4] = (S, dlogit, loss, pad) of the local samples."""
        G, B = parts_mine.shape[0], parts_mine.shape[1]
        rec = self._buf("rec", (B, self.rec_out))
        out = self._bufs.get(("fwd_out", B))
        if out is None:                              # the output struct only changes with the buffer it points at
            out = self._bufs[("fwd_out", B)] = _lib.FwdOut()
            base = rec.data_ptr()
            out.S, out.dz, out.loss, out.sample_ld = base, base + 4 * self.kp, base + 4 * (self.kp + 1), self.rec_out
            out.error = self.e.error.data_ptr()
        t = self.table
        _lib.check(self.e.lib.fmx_fm_forward_finish(self.hyper.ref(), t.bias.data_ptr(),
                                                   _lib.LAYOUT_WEIGHTS if t.layout == "weights" else _lib.LAYOUT_FTRL, t.kp,
                                                   parts_mine.data_ptr(), B * self.rec_in, G, y_local.data_ptr(), B,
                                                   _lib.LOSSES[self.loss], inv_b, C.byref(out), self.e._stream(stream)))
        return rec

    def _slot_ws(self, slot, GB):
        ws = self._ws.get((slot, GB))
        if ws is None:
            ws = self._ws[(slot, GB)] = self.e.new_workspace(GB)
        return ws

    def start_sort(self, idx_own, slot, stream=None):
        GB = idx_own.shape[0]
        self.e._ensure(GB)
        had = (slot, GB) in self._ws
        ws = self._slot_ws(slot, GB)
        if not had and stream is not None and not isinstance(stream, int):
            stream.wait_stream(torch.cuda.current_stream(self.device))   # the zero fill of a new workspace comes first
        self.e.sort(idx_own, workspace=ws, stream=stream)

    def update(self, idx_own, rec_g, inv_b, slot=None, stream=None):
        GB = idx_own.shape[0]
        self.e._ensure(GB)
        if slot is None:
            self.e.sort(idx_own, stream=stream)
            ws = None
        else:
            ws = self._slot_ws(slot, GB)
        self.e.update(self.hyper, self.rule, GB, None, None, inv_b=inv_b, with_loss=True, records=rec_g, workspace=ws, stream=stream)
        return self.e.loss_out


def _max_step_batch(max_field_rows):
    from .distributed import max_step_batch
    return max_step_batch(max_field_rows)


class FieldOwnerFM:
    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"
        # FMX_FORCE_COLLECTIVES=1: issue the collectives even with one rank (tools/nccl_world1_check.py: the RCCL calls of a
        # step on a one-GPU box)
        self._force = dist.is_initialized() and os.environ.get("FMX_FORCE_COLLECTIVES") == "1"
        self._bufs = {}
        self._pref, self._next_slot, self._pf = {}, 0, None

    # ---- collectives (gloo with CUDA tensors -- several ranks rehearsing on one GPU -- is staged through the host) ----
    def _out(self, name, shape, like):
        t = self._bufs.get((name, shape))
        if t is None or t.dtype != like.dtype or t.device != like.device:
            t = self._bufs[(name, shape)] = torch.empty(shape, dtype=like.dtype, device=like.device)
        return t

    def _all_gather(self, name, local):
        if self.world == 1 and not self._force:
            return local
        out = self._out(name, (self.world * local.shape[0],) + tuple(local.shape[1:]), local)
        if self._gloo and local.is_cuda:
            host = torch.empty(out.shape, dtype=local.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def _all_to_all(self, name, parts, B):
        """parts [G B, R] (rank-major samples) -> [G, B, R]: block r = rank r's records of THIS rank's samples."""
        if self.world == 1 and not self._force:
            return parts.view(1, B, parts.shape[1])
        out = self._out(name, (self.world * B, parts.shape[1]), parts)
        if self._gloo and parts.is_cuda:
            host = torch.empty(out.shape, dtype=parts.dtype)
            dist.all_to_all_single(host, parts.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_to_all_single(out, parts.contiguous(), group=self.group)
        return out.view(self.world, B, parts.shape[1])

    # ---- the weights-free part of a later step, ahead of time ----
    def prefetch(self, idx_next):
        """Gather a LATER step's indices, pick the owned columns and sort them now, on a prefetch stream, into a slot of their
        own.  Returns a token to pass to step(); None when no slot is free (step() then does this work in line)."""
        be = self.backend
        n_slots = getattr(be, "N_SLOTS", 0)
        if not (idx_next.is_cuda and hasattr(be, "start_sort")) or len(self._pref) >= n_slots or self._sub_steps(idx_next.shape[0]) > 1:
            return None
        dev = idx_next.device
        cur = torch.cuda.current_stream(dev)
        if self._pf is None:
            self._pf = (torch.cuda.Stream(device=dev), [torch.cuda.Event() for _ in range(n_slots)],
                        [torch.cuda.Event() for _ in range(n_slots)])
        pf, ready, free = self._pf
        slot = self._next_slot
        self._next_slot = (slot + 1) % n_slots
        pf.wait_stream(cur)                     # whatever wrote idx_next (an H2D copy, a refill) is ordered before the gather
        pf.wait_event(free[slot])               # the update that last used this slot has run
        with torch.cuda.stream(pf):
            idx_own = be.select(self._all_gather(f"idx_slot{slot}", idx_next))
            be.start_sort(idx_own, slot, stream=pf)
        ready[slot].record(pf)
        token = object()
        self._pref[id(token)] = (token, idx_own, slot)
        return token

    def _sub_steps(self, B):
        cap = getattr(self.backend, "max_global_batch", None)
        n_sub = 1
        while cap is not None and (B // n_sub) * self.world > cap and (B // n_sub) % 2 == 0:
            n_sub *= 2
        return n_sub

    def step(self, idx_local, y_local, token=None):
        """One exact step on the global batch (the rank-major concatenation of the ranks' idx_local [B, F] / y_local [B]).
        Returns the global mean-loss tensor [1].  When G x B exceeds what one exact step can sort (backend.max_global_batch)
        the batch is processed as consecutive exact steps over equal slices of every rank's samples."""
        B = idx_local.shape[0]
        n_sub = self._sub_steps(B)
        if n_sub > 1:
            out, Bs = None, B // n_sub
            for j in range(n_sub):
                out = self._step(idx_local[j * Bs:(j + 1) * Bs], y_local[j * Bs:(j + 1) * Bs], None)
            return out
        return self._step(idx_local, y_local, token)

    def _step(self, idx_local, y_local, token):
        be, B = self.backend, idx_local.shape[0]
        inv_b = 1.0 / (B * self.world)
        pref = self._pref.pop(id(token), None) if token is not None else None
        if pref is not None:
            _, idx_own, slot = pref
            torch.cuda.current_stream(idx_local.device).wait_event(self._pf[1][slot])
        else:
            idx_own, slot = be.select(self._all_gather("idx", idx_local)), None
        # the launch stream is looked up once per step (torch.cuda.current_stream costs several microseconds per call)
        kw = {"stream": torch.cuda.current_stream(idx_local.device)} if idx_local.is_cuda and getattr(be, "takes_stream", False) else {}
        parts = be.partial_forward(idx_own, **kw)
        mine = self._all_to_all("parts", parts, B)
        rec = be.finish(mine, y_local, inv_b, **kw)
        rec_g = self._all_gather("rec", rec)
        out = be.update(idx_own, rec_g, inv_b, slot, **kw) if slot is not None else be.update(idx_own, rec_g, inv_b, **kw)
        if slot is not None:
            self._pf[2][slot].record(kw.get("stream") or torch.cuda.current_stream(idx_local.device))
        return out
