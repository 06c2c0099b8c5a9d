"""Data-parallel online FM over the GPUs of one node (one process per GPU, torch.distributed: "nccl" is RCCL on ROCm).

Exact by construction: every rank keeps a full replica of the table, runs the forward pass on its own slice of the
global mini-batch, and the ranks all-gather what the backward needs.  For FM that is NOT the row gradients
(39 x 68 B per sample) but their low-rank factors: the sample's indices, its sum vector S_b and its dlogit --
156 B of indices + one 80-byte record (S, dlogit, loss) per sample, two collectives per step -- because every row gradient of a sample is  x (S_b - x V_row) dz_b.  Each rank then
runs the same deterministic sort + row-reduced update over the GLOBAL batch, so the replicas stay bit-identical,
no embedding row ever crosses xGMI, and the G-GPU step equals the 1-GPU step on the same global batch
(SURVEY.md section 8(e)).  The bias gradient is the sum of the gathered dlogits, so pure FM needs no all-reduce; a
dense network on top (DeepFM / NFM) adds one fused all-reduce of its gradients.

The compute behind a step is a small backend interface so that the sharding / gather logic can be exercised on CPU
with the gloo backend in tests (tests inject an oracle-backed backend; the product has only the HIP one).
"""
import os

import torch
import torch.distributed as dist


def max_step_batch(max_field_rows):
    """Largest batch one exact step can take: 32,768 (a field's occurrence list is merged inside one workgroup's LDS;
    include/fmx.h).  The 32-bit (index, sample) composite no longer bounds it: large fields are cut into sort pieces."""
    return 32768       # fields too large for a 32-bit (index, sample) composite are cut into sort pieces (FlatTable.ensure_sort_split)


class HipBackend:
    """The product backend: fmx.FMEngine on this rank's GPU."""

    def __init__(self, engine, hyper, rule, loss):
        self.e, self.hyper, self.rule, self.loss = engine, hyper, rule, loss
        self.max_global_batch = max_step_batch(max(engine.table.feature_sizes))

    def forward(self, idx, y, inv_b, stream=None):
        """-> records [B, kp + 4]: per sample (S[kp], dz, loss, pad) -- one buffer, so ONE all-gather carries everything
        the global update needs from this rank."""
        B = idx.shape[0]
        if getattr(self, "_rec", None) is None or self._rec.shape[0] < B:
            self._rec = torch.zeros((B, self.e.table.kp + 4), dtype=torch.float32, device=self.e.device)
            self._rec_views = {}
        rec = self._rec_views.get(B)
        if rec is None:
            rec = self._rec_views[B] = self._rec[:B]
        self.e.forward(self.hyper, idx, None, y, loss=self.loss, inv_b=inv_b, want_first=False, want_bi=False, records=self._rec,
                       stream=stream)
        return rec

    N_SLOTS = 4   # batches that may be gathered + sorted ahead of their update, each with a workspace of its own
    takes_stream = True   # forward / update accept the caller's stream (saves a current_stream() lookup per launch)

    def _slot_ws(self, slot, GB):
        if not hasattr(self, "_ws"):
            self._ws, self._ws_need = {}, {}
        need = self._ws_need.get(GB)
        if need is None:
            need = self._ws_need[GB] = int(self.e.lib.fmx_workspace_bytes(self.e.table.c_struct(), GB)) // 4
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = self._ws[slot] = self.e.new_workspace(GB)
        return ws

    def start_sort(self, idx_g, slot=0, stream=None):
        """The global occurrence sort only needs the gathered indices; it is launched on `stream` (the caller's prefetch
        stream; default: the current stream) into the slot's own workspace."""
        e = self.e
        GB = idx_g.shape[0]
        e._ensure(GB)
        had = getattr(self, "_ws", {}).get(slot)
        ws = self._slot_ws(slot, GB)
        if ws is not had and stream is not None and not isinstance(stream, int):
            # a workspace allocated just now was zero-filled on the CURRENT stream: the sort must not overtake the fill
            stream.wait_stream(torch.cuda.current_stream(e.device))
        e.sort(idx_g, workspace=ws, stream=stream)

    def update(self, idx_g, rec_g, inv_b, slot=None, stream=None):
        """Row-reduced update over the global batch (sorted into `slot` by start_sort, or here); returns the mean-loss
        tensor [1]."""
        e = self.e
        GB = idx_g.shape[0]
        e._ensure(GB)
        if slot is None:                                      # nothing prepared: sort here, in the engine's own workspace
            e.sort(idx_g, stream=stream)
            ws = None
        else:
            ws = self._slot_ws(slot, GB)
        e.update(self.hyper, self.rule, GB, None, None, inv_b=inv_b, with_loss=True, records=rec_g, workspace=ws, stream=stream)
        return e.loss_out


class DataParallelFM:
    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"     # looked up once, not per collective
        self._bufs = {}

    def _gathered(self, name, local):
        """all_gather `local` ([B, ...] contiguous) into a rank-major [G*B, ...] buffer."""
        if self.world == 1 and os.environ.get("FMX_FORCE_COLLECTIVES") != "1":  # (=1: tools/nccl_world1_check.py)
            return local
        key = (name, local.shape[0])
        out = self._bufs.get(key)
        if out is None or out.dtype != local.dtype or out.device != local.device or out.shape[1:] != local.shape[1:]:
            shape = (self.world * local.shape[0],) + tuple(local.shape[1:])
            out = self._bufs[key] = torch.empty(shape, dtype=local.dtype, device=local.device)
        shape = out.shape
        if self._gloo and local.is_cuda:
            # rehearsal only (several ranks sharing one GPU, where RCCL cannot run): stage through the host
            host = torch.empty(shape, dtype=local.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def step(self, idx_local, y_local, token=None):
        """One exact data-parallel step.  idx_local [B,F] int32 and y_local [B] fp32 are this rank's slice; the global
        batch is the rank-major concatenation.  token: what prefetch() returned for this batch (the gathered indices and the
        global sort of every sub-step are then already under way), or None.  Returns the global mean-loss tensor [1] (of the
        last sub-step, see below).

        When G * B exceeds what one exact step can take (backend.max_global_batch), the batch is processed as consecutive exact
        steps over equal slices of every rank's samples -- still exact online learning, with a smaller global batch per update."""
        B = idx_local.shape[0]
        n_sub = self._sub_steps(B)
        pref = self._pref.pop(id(token), None) if (token is not None and hasattr(self, "_pref")) else None
        if pref is not None and (pref[2] != B or len(pref[1]) != n_sub):
            self._release(pref[1])
            pref = None
        if n_sub > 1:
            out = None
            Bs = B // n_sub
            for j in range(n_sub):
                out = self._step(idx_local[j * Bs:(j + 1) * Bs], y_local[j * Bs:(j + 1) * Bs], pref[1][j] if pref else None)
            return out
        return self._step(idx_local, y_local, pref[1][0] if pref else None)

    def _sub_steps(self, B):
        cap = getattr(self.backend, "max_global_batch", None)
        n_sub = 1
        while cap is not None and (B // n_sub) * self.world > cap and (B // n_sub) % 2 == 0:
            n_sub *= 2
        return n_sub

    def _release(self, parts):
        """Give prefetched slots back unused (their buffers are free once the prefetch stream has passed them)."""
        for _, slot, _ in parts:
            st = self._pf_streams[slot]
            st[2].record(st[0])

    def cancel(self, token):
        pref = self._pref.pop(id(token), None) if (token is not None and hasattr(self, "_pref")) else None
        if pref is not None:
            self._release(pref[1])

    def prefetch(self, idx_next):
        """Gather and sort a LATER step's indices now (they do not depend on the weights): the index all-gather and the
        global occurrence sort of every sub-step run on a prefetch stream of their own, into a slot (gathered-index
        buffer + workspace) of their own, beside the steps in front of them.  Returns a TOKEN to hand to step() with that
        batch (or to cancel()); None when the backend cannot or no slot is free (step() then gathers and sorts by itself).
        Up to backend.N_SLOTS sub-steps may be in flight.  (Rounds 1-2 matched prepared work by the batch tensor's address: a
        recycled address with new contents would have consumed a stale sorted list.)"""
        if not (idx_next.is_cuda and hasattr(self.backend, "start_sort")):
            return None
        if not hasattr(self, "_pref"):
            self._pref, self._next_slot, self._pf_streams = {}, 0, {}
        B = idx_next.shape[0]
        n_sub = self._sub_steps(B)
        Bs = B // n_sub
        n_slots = getattr(self.backend, "N_SLOTS", 1)
        busy = {slot for _, parts, _ in self._pref.values() for _, slot, _ in parts}
        if len(busy) + n_sub > n_slots:
            return None                                       # no free slot: the step will gather and sort by itself
        dev = idx_next.device
        cur = torch.cuda.current_stream(dev)
        parts = []
        for j in range(n_sub):
            part = idx_next[j * Bs:(j + 1) * Bs] if n_sub > 1 else idx_next
            slot = self._next_slot
            while slot in busy:
                slot = (slot + 1) % n_slots
            busy.add(slot)
            self._next_slot = (slot + 1) % n_slots
            st = self._pf_streams.get(slot)
            if st is None:
                # per slot: a "gathered + sorted" event and a "slot free again" event, both reused; the slots share TWO
                # prefetch streams (a process gets few hardware queues: with a stream per slot some prefetch stream ends up
                # in the queue of the main stream and its sort waits behind the step it was meant to overlap)
                shared = self._pf_streams.get(("stream", slot % 2))
                if shared is None:
                    shared = self._pf_streams[("stream", slot % 2)] = torch.cuda.Stream(device=dev)
                st = self._pf_streams[slot] = (shared, torch.cuda.Event(), torch.cuda.Event())
            else:
                st[0].wait_event(st[2])                       # the update that last used this slot has run
            st[0].wait_stream(cur)                            # whatever wrote idx_next (an H2D copy, a refill) comes first
            pf, ready = st[0], st[1]
            if self.world > 1 or os.environ.get("FMX_FORCE_COLLECTIVES") == "1":   # the collective goes to whatever stream is current
                with torch.cuda.stream(pf):
                    idx_g = self._gathered(f"idx_slot{slot}", part)
                    self.backend.start_sort(idx_g, slot, stream=pf)
            else:
                idx_g = part
                self.backend.start_sort(idx_g, slot, stream=pf)
            ready.record(pf)
            parts.append((idx_g, slot, ready))
        token = object()
        self._pref[id(token)] = (token, parts, B)
        return token

    def _step(self, idx_local, y_local, pref=None):
        B = idx_local.shape[0]
        inv_b = 1.0 / (B * self.world)
        cur = torch.cuda.current_stream(idx_local.device) if idx_local.is_cuda else None   # looked up once per step
        kw = {"stream": cur} if cur is not None and getattr(self.backend, "takes_stream", False) else {}
        rec = self.backend.forward(idx_local, y_local, inv_b, **kw)
        rec_g = self._gathered("rec", rec)
        if pref is None:                                      # nothing prepared: gather and sort in line
            idx_g = self._gathered("idx", idx_local)
            return self.backend.update(idx_g, rec_g, inv_b, **kw)
        idx_g, slot, ready = pref
        cur.wait_event(ready)
        out = self.backend.update(idx_g, rec_g, inv_b, slot, **kw)
        self._pf_streams[slot][2].record(cur)                 # the slot's buffers may be overwritten after this update
        return out
