"""Field-owner parallel online FM over the GPUs of one node: the mode in which the update WORK and the TABLE shard.

The replicated mode (fmx.distributed.DataParallelFM) keeps a full table on every rank and repeats the row-reduced update
of the GLOBAL batch on every rank: exact, but no faster than one GPU.  Here every rank OWNS row ranges of the index columns
(pieces, fmx.plan.OwnerPlan: cut and dealt so that rows and expected occurrences balance, for any number of columns and any
world size) and holds only those rows (a 10 M-row table spreads over the ranks' HBM).  The pieces sit at fixed positions of
the forward pass's additions tree; a rank owns the blocks of lane groups its pieces sit in (include/fmx.h, "the forward pass
split over field owners").  One step on a global batch of G x B samples, rank-major:

    all-gather   idx [B, F]             -> idx_all [G B, F]; the kernels pick the owned columns themselves       (weights-free:
    sort         idx_all                -> occurrence lists of the owned pieces over the global batch              runs ahead, prefetch())
    partial fwd  fmx_fm_forward_partial -> per sample and owned block (S_part, ss_part, first_part)              [G][blocks][B, 2 kp + 4]
    all-to-all   the records of rank r's samples go to rank r                                                     [NB, B, 2 kp + 4]
    finish       fmx_fm_forward_finish  -> S, logit, loss, dlogit of the B local samples (tree order)            [B, kp + 4]
    all-gather   (S, dlogit, loss)                                                                                [G B, kp + 4]
    update       k_fm_update on the owned pieces' lists: every row is updated by its owner only; the bias (replicated) by
                 everybody, identically

Per rank the gather, the sort and the update touch about G B x F / G = B x F occurrences: constant work per GPU as G grows.
No embedding row crosses xGMI; per step two small collectives sit on the critical path (144 B per sample and block, 80 B per
sample at k = 16).  The additions are those of ONE table holding the same pieces at the same positions
(OwnerPlan.table_whole()) in the same order, so G ranks give bit-identical rows, losses and biases to one GPU stepping the
same global batch on that table (tests/test_owner_*.py); with one rank the plan IS the ordinary table.

The compute behind a step is a small backend interface so that the sharding / exchange logic runs on CPU with gloo in the
tests (an oracle-backed backend injected there; the product has only the HIP one).  fmx_owner_step (include/fmx.h) is the
same step as ONE C call with the library's own RCCL communicator: NativeOwnerFM below.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib
from .engine import FMEngine, _ptr
from .plan import OwnerPlan


class HipOwnerBackend:
    """The product backend: this rank's shard of the table (its pieces, fmx.plan.OwnerPlan.table_for_owner) and an FMEngine on it."""

    N_SLOTS = 3   # batches whose indices may be gathered + sorted ahead of their update
    takes_stream = True

    def __init__(self, feature_sizes, k, hyper, rule, loss, rank, world, layout=None, ftrl=None, device=None, max_local_batch=4096,
                 plan=None):
        self.rank, self.world, self.rule, self.loss, self.hyper = rank, world, rule, loss, hyper
        self.feature_sizes = [int(s) for s in feature_sizes]
        self.plan = plan if plan is not None else OwnerPlan(self.feature_sizes, k, world, global_batch=max_local_batch * world)
        assert self.plan.world == world and self.plan.sizes == self.feature_sizes
        layout = layout or ("ftrl" if rule == "ftrl" else "weights")
        self.table = self.plan.table_for_owner(rank, layout=layout, device=device, ftrl=ftrl)
        self.fields = self.table.plan_fields                                    # [(column, first index, rows)] per local field
        self.e = FMEngine(self.table, max_batch=max_local_batch * world)
        self.device = self.table.device
        self.nb, self.nlb, self.block_count = self.plan.nb, self.plan.block_count[rank], list(self.plan.block_count)
        self.kp, self.rec_in, self.rec_out = self.table.kp, 2 * self.table.kp + 4, self.table.kp + 4
        self.max_global_batch = _max_step_batch(max(r for _, _, r in self.fields))   # (every rank sorts pieces of at most this size)
        self._bufs, self._ws = {}, {}

    def _buf(self, name, shape, dtype=torch.float32):
        t = self._bufs.get((name, shape))
        if t is None:
            t = self._bufs[(name, shape)] = torch.zeros(shape, dtype=dtype, device=self.device)
        return t

    def partial_forward(self, idx_all, B_local, stream=None):
        """idx_all [G B, F] -> records [G * blocks * B, 2 kp + 4]: destination-major, then block, then sample."""
        GB = idx_all.shape[0]
        parts = self._buf("parts", (GB * self.nlb, self.rec_in))
        _lib.check(self.e.lib.fmx_fm_forward_partial(self.table.c_struct(), idx_all.data_ptr(), None, GB, self.nb, self.nlb, B_local,
                                                    parts.data_ptr(), self.e.error.data_ptr(), self.e._stream(stream)))
        return parts

    def finish(self, parts_mine, y_local, inv_b, stream=None):
        """parts_mine [NB, B, 2 kp + 4] (block-major) -> records [B, kp + 4] = (S, dlogit, loss, pad) of the local samples."""
        NB, B = parts_mine.shape[0], parts_mine.shape[1]
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
                                                   parts_mine.data_ptr(), B * self.rec_in, NB, y_local.data_ptr(), B,
                                                   _lib.LOSSES[self.loss], inv_b, C.byref(out), self.e._stream(stream)))
        return rec

    def _slot_ws(self, slot, GB):
        """The per-slot workspace: sized for the table's CURRENT sort fields (a table whose large pieces were split for a larger
        batch needs a larger workspace at every batch size; the C side checks the byte count it is given)."""
        key = (slot, GB, self.table._sort_split[1].numel() if self.table._sort_split is not None else 0)
        ws = self._ws.get(key)
        if ws is None:
            for old in [k for k in self._ws if k[:2] == (slot, GB)]:
                del self._ws[old]
            ws = self._ws[key] = self.e.new_workspace(GB)
            return ws, True
        return ws, False

    def start_sort(self, idx_all, slot, stream=None):
        GB = idx_all.shape[0]
        self.e._ensure(GB)
        ws, fresh = self._slot_ws(slot, GB)
        if fresh and stream is not None and not isinstance(stream, int):
            stream.wait_stream(torch.cuda.current_stream(self.device))   # the zero fill of a new workspace comes first
        self.e.sort(idx_all, workspace=ws, stream=stream)

    def update(self, idx_all, rec_g, inv_b, slot=None, stream=None):
        GB = idx_all.shape[0]
        self.e._ensure(GB)
        if slot is None:
            self.e.sort(idx_all, stream=stream)
            ws = None
        else:
            ws, fresh = self._slot_ws(slot, GB)
            assert not fresh, "the slot's workspace was rebuilt between its sort and its update"
        self.e.update(self.hyper, self.rule, GB, None, None, inv_b=inv_b, with_loss=True, records=rec_g, workspace=ws, stream=stream)
        return self.e.loss_out

    # ---- DeepFM / NFM on field owners (fmx.deep.OwnerDeepFMTrainer): the finish without a loss, records carrying dL/dbi ----
    def deep_record_ld(self):
        """floats per sample of the record the ranks all-gather: S [kp] | dlogit, 0, 0, 0 | dL/dbi [kp]."""
        return 2 * self.kp + 4

    def finish_bi(self, parts_mine, stream=None):
        """parts_mine [NB, B, 2 kp + 4] -> (rec [B, 2 kp + 4] with S in its first kp floats, bi [B, kp], sfirst [B], logit [B]) of
        the local samples; no loss here: it comes out of the MLP section."""
        NB, B = parts_mine.shape[0], parts_mine.shape[1]
        ld = self.deep_record_ld()
        rec = self._buf("deep_rec", (B, ld))
        bi, sfirst, logit = self._buf("deep_bi", (B, self.kp)), self._buf("deep_sfirst", (B,)), self._buf("deep_logit", (B,))
        out = self._bufs.get(("deep_out", B))
        if out is None:
            out = self._bufs[("deep_out", B)] = _lib.FwdOut()
            out.S, out.sample_ld = rec.data_ptr(), ld
            out.bi, out.sfirst, out.logit, out.error = bi.data_ptr(), sfirst.data_ptr(), logit.data_ptr(), self.e.error.data_ptr()
        t = self.table
        _lib.check(self.e.lib.fmx_fm_forward_finish(self.hyper.ref(), t.bias.data_ptr(),
                                                   _lib.LAYOUT_WEIGHTS if t.layout == "weights" else _lib.LAYOUT_FTRL, t.kp,
                                                   parts_mine.data_ptr(), B * self.rec_in, NB, None, B, _lib.LOSS_NONE, 1.0,
                                                   C.byref(out), self.e._stream(stream)))
        return rec, bi, sfirst, logit

    def update_deep(self, idx_all, rec_all, fm_term, inv_b, slot=None, stream=None):
        """The owned rows from the gathered records [G B, 2 kp + 4] = (S | dlogit, - | dL/dbi): G = (fm_term ? dlogit : 0) + dL/dbi."""
        GB, kp, ld = idx_all.shape[0], self.kp, rec_all.shape[1]
        self.e._ensure(GB)
        if slot is None:
            self.e.sort(idx_all, stream=stream)
            ws = self.e.workspace
        else:
            ws, fresh = self._slot_ws(slot, GB)
            assert not fresh
        base = rec_all.data_ptr()
        dz = base + 4 * kp
        _lib.check(self.e.lib.fmx_fm_update(self.table.c_struct(), self.hyper.ref(), _lib.RULES[self.rule], ws.data_ptr(),
                                           ws.numel() * ws.element_size(), None, base, dz, dz if fm_term else None, base + 4 * (kp + 4),
                                           GB, ld, None, inv_b, None, self.e._stream(stream)))

    def bias_weight(self):
        return self.table.bias_weight()

    def mlp_section(self, flat, gflat, k, hidden, n_layers, loss, bi, base, y, B, inv_b, lr_apply):
        """The replicated MLP on the local samples' bi (fmx_mlp_section: fp32 MFMA) -> (loss [1], dlogit [B], dL/dbi [B, kp])."""
        return self.e.mlp_section(flat, gflat, k, hidden, n_layers, loss, bi, base, y, B, inv_b, lr_apply)

    def check_error_flag(self):
        self.e.check_error_flag()


class NativeOwnerFM:
    """The field-owner step as ONE C call per step (fmx_owner_prefetch / fmx_owner_step, include/fmx.h) with the library's own
    RCCL communicator: the kernels of a step, its two exchanges and the events between the step's stream and the prefetch
    stream are issued from C.  Same interface as FieldOwnerFM (prefetch -> token, step, cancel, check_error_flag), same bits.
    Needs one GPU per rank (RCCL); torch.distributed is used once, to hand the communicator's ids to the ranks."""

    def __init__(self, backend, group=None, force_collectives=False, stream=None):
        """stream: the HIP stream (torch stream or int handle) the steps are issued on when a call names none; default: the
        stream current at construction."""
        be = self.backend = backend
        self.lib = be.e.lib
        self._st = be.e._stream(stream)
        self._prefetch_fn, self._step_fn = self.lib.fmx_owner_prefetch, self.lib.fmx_owner_step
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert (self.world, self.rank) == (be.world, be.rank)
        dev = be.device
        ids = torch.zeros(2 * _lib.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        exchange = self.world > 1 or force_collectives
        if exchange:
            if self.rank == 0:
                buf = (C.c_char * (2 * _lib.COMM_ID_BYTES))()
                _lib.check(self.lib.fmx_comm_unique_id(buf))
                ids.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
            if self.world > 1:
                if dist.get_backend(group) == "gloo":
                    host = ids.cpu()
                    dist.broadcast(host, src=0, group=group)
                    ids.copy_(host)
                else:
                    dist.broadcast(ids, src=0, group=group)
        raw = bytes(ids.cpu().numpy().tobytes())
        counts = (C.c_int32 * self.world)(*be.block_count)
        handle = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(self.lib.fmx_comm_create(raw if exchange else None, self.rank, self.world, counts, 1 if force_collectives else 0,
                                                C.byref(handle)))
        self.comm = handle
        self._exchange = exchange
        self._slots = {}          # slot -> (idx_all, workspace) sized for one global batch
        self._pref, self._next_slot = {}, 0
        self._bufs = None

    def close(self):
        """Release the communicator (its RCCL communicators, its stream and events); the caller has synchronised."""
        comm, self.comm = getattr(self, "comm", None), None
        if comm:
            self.lib.fmx_comm_destroy(comm)

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the library or the runtime may already be gone
            pass

    def _slot(self, slot, B):
        """The slot's buffers for a global batch of world x B samples, and its pre-bound call arguments (a call is then one
        foreign call plus a pointer or two: the structs, buffers and sizes of a slot only change when the engine grows)."""
        be, GB = self.backend, B * self.world
        be.e._ensure(GB)
        key = (GB, be.table._sort_split[1].numel() if be.table._sort_split is not None else 0)
        cur = self._slots.get(slot)
        if cur is None or cur[0] != key:
            idx_all = torch.empty((GB, be.table.n_cols), dtype=torch.int32, device=be.device) if self._exchange else None
            ws = be.e.new_workspace(GB)
            torch.cuda.current_stream(be.device).synchronize()      # (a new zero-filled workspace: rare, and ordered before the prefetch stream uses it)
            bufs = self._step_bufs(B)
            t, e = be.table.c_struct(), be.e
            pre = (_ptr(idx_all), ws.data_ptr(), ws.numel() * 4, e.error.data_ptr())
            post = (B, slot, ws.data_ptr(), ws.numel() * 4, C.byref(bufs), e.loss_out.data_ptr(), e.error.data_ptr())
            head = (self.comm, t, be.hyper.ref(), _lib.RULES[be.rule], _lib.LOSSES[be.loss])
            cur = self._slots[slot] = (key, idx_all, ws, pre, post, head, t)
        return cur

    def _step_bufs(self, B):
        be = self.backend
        if self._bufs is None or self._bufs[0] != B:
            f32 = dict(dtype=torch.float32, device=be.device)
            send = torch.zeros((self.world * be.nlb * B, be.rec_in), **f32)
            rec = torch.zeros((B, be.rec_out), **f32)
            recv = torch.zeros((be.nb * B, be.rec_in), **f32) if self._exchange else send
            rec_all = torch.zeros((self.world * B, be.rec_out), **f32) if self._exchange else rec
            c = _lib.OwnerBufs(send.data_ptr(), recv.data_ptr(), rec.data_ptr(), rec_all.data_ptr())
            self._bufs = (B, c, (send, recv, rec, rec_all))
        return self._bufs[1]

    def prefetch(self, idx_next, stream=None):
        """The weights-free part of a LATER step, now, on the communicator's own stream (behind `stream` as it is at this call):
        all-gather of the ranks' idx_next [B, F], occurrence sort of the owned pieces.  -> token for step(), None when every
        slot is taken.  With one rank nothing is copied: idx_next must stay unchanged until its step has run."""
        if len(self._pref) >= _lib.COMM_SLOTS:
            return None
        be, B = self.backend, idx_next.shape[0]
        busy = {s for _, s, _, _ in self._pref.values()}
        slot = self._next_slot
        while slot in busy:
            slot = (slot + 1) % _lib.COMM_SLOTS
        self._next_slot = (slot + 1) % _lib.COMM_SLOTS
        cur = self._slot(slot, B)
        _lib.check(self._prefetch_fn(self.comm, cur[6], idx_next.data_ptr(), B, slot, *cur[3], self._st if stream is None else be.e._stream(stream)))
        token = object()
        self._pref[id(token)] = (token, slot, B, idx_next)
        return token

    def cancel(self, token):
        self._pref.pop(id(token), None)

    def check_error_flag(self):
        self.backend.check_error_flag()

    def step(self, idx_local, y_local, token=None, stream=None):
        be, B = self.backend, idx_local.shape[0]
        pref = self._pref.pop(id(token), None) if token is not None else None
        if pref is None:                             # nothing prepared: the gather and the sort now, then the step
            token = self.prefetch(idx_local, stream=stream)
            pref = self._pref.pop(id(token))
        _, slot, Bp, idx_src = pref
        assert Bp == B
        cur = self._slot(slot, B)
        idx_all = cur[1].data_ptr() if self._exchange else idx_src.data_ptr()
        _lib.check(self._step_fn(*cur[5], idx_all, y_local.data_ptr(), *cur[4], self._st if stream is None else be.e._stream(stream)))
        return be.e.loss_out


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
        self.block_count = list(getattr(backend, "block_count", [1] * self.world))   # tree blocks per rank (OwnerPlan)
        assert len(self.block_count) == self.world

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
        """parts [G * blocks_mine * B, R] (destination-major) -> [NB, B, R]: every block's records of THIS rank's samples, in
        block order (rank r's blocks are consecutive: the messages arrive in rank order)."""
        R = parts.shape[1]
        nb = sum(self.block_count)
        if self.world == 1 and not self._force:
            return parts.view(nb, B, R)
        out = self._out(name, (nb * B, R), parts)
        mine = self.block_count[self.rank] * B
        in_split = [mine] * self.world
        out_split = [c * B for c in self.block_count]
        if self._gloo and parts.is_cuda:
            host = torch.empty(out.shape, dtype=parts.dtype)
            dist.all_to_all_single(host, parts.contiguous().cpu(), out_split, in_split, group=self.group)
            out.copy_(host)
        else:
            dist.all_to_all_single(out, parts.contiguous(), out_split, in_split, group=self.group)
        return out.view(nb, B, R)

    # ---- the weights-free part of a later step, ahead of time ----
    def prefetch(self, idx_next):
        """Gather a LATER step's indices and sort the owned pieces' occurrences now, on a prefetch stream, into a slot of their
        own.  Returns a token to pass to step() (or to cancel()); None when no slot is free (step() then does this work in line)."""
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
        busy = {s for _, _, s in self._pref.values()}
        slot = next(s for s in ((self._next_slot + i) % n_slots for i in range(n_slots)) if s not in busy)
        self._next_slot = (slot + 1) % n_slots
        pf.wait_stream(cur)                     # whatever wrote idx_next (an H2D copy, a refill) is ordered before the gather
        pf.wait_event(free[slot])               # the update that last used this slot has run
        with torch.cuda.stream(pf):
            idx_all = self._all_gather(f"idx_slot{slot}", idx_next)
            be.start_sort(idx_all, slot, stream=pf)
        ready[slot].record(pf)
        token = object()
        self._pref[id(token)] = (token, idx_all, slot)
        return token

    def cancel(self, token):
        """Give a prefetch token back unused (its slot is free again once the prefetch stream has passed it)."""
        pref = self._pref.pop(id(token), None) if token is not None else None
        if pref is not None:
            self._pf[2][pref[2]].record(self._pf[0])

    def check_error_flag(self):
        """The device error word of the backend (IndexError / HandOffTimeout as FMEngine.check_error_flag).  Synchronises."""
        if hasattr(self.backend, "check_error_flag"):
            self.backend.check_error_flag()

    def _sub_steps(self, B):
        cap = getattr(self.backend, "max_global_batch", None)
        n_sub = 1
        while cap is not None and (B // n_sub) * self.world > cap and (B // n_sub) % 2 == 0:
            n_sub *= 2
        return n_sub

    def step(self, idx_local, y_local, token=None):
        """One exact step on the global batch (the rank-major concatenation of the ranks' idx_local [B, F] / y_local [B]).
        Returns the global mean-loss tensor [1].  When G x B exceeds what one exact step can sort (backend.max_global_batch)
        the batch is processed as consecutive exact steps over equal slices of every rank's samples (a token is given back)."""
        B = idx_local.shape[0]
        n_sub = self._sub_steps(B)
        if n_sub > 1:
            self.cancel(token)
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
            _, idx_all, slot = pref
            torch.cuda.current_stream(idx_local.device).wait_event(self._pf[1][slot])
        else:
            idx_all, slot = self._all_gather("idx", idx_local), None
        # the launch stream is looked up once per step (torch.cuda.current_stream costs several microseconds per call)
        kw = {"stream": torch.cuda.current_stream(idx_local.device)} if idx_local.is_cuda and getattr(be, "takes_stream", False) else {}
        parts = be.partial_forward(idx_all, B, **kw)
        mine = self._all_to_all("parts", parts, B)
        rec = be.finish(mine, y_local, inv_b, **kw)
        rec_g = self._all_gather("rec", rec)
        out = be.update(idx_all, rec_g, inv_b, slot, **kw) if slot is not None else be.update(idx_all, rec_g, inv_b, **kw)
        if slot is not None:
            self._pf[2][slot].record(kw.get("stream") or torch.cuda.current_stream(idx_local.device))
        return out
