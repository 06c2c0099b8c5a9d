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
import torch
import torch.distributed as dist


def max_step_batch(max_field_rows):
    """Largest batch one exact step can take: the occurrence sort packs (local index, sample) into 32 bits and sorts a
    field inside one workgroup's LDS (include/fmx.h), so  (max_field_rows - 1) < (0xFFFFFFFF >> log2(batch))  and
    batch <= 32768."""
    bbits = 15
    while bbits > 6 and (max_field_rows - 1) >= (0xFFFFFFFF >> bbits):
        bbits -= 1
    return 1 << bbits


class HipBackend:
    """The product backend: fmx.FMEngine on this rank's GPU."""

    def __init__(self, engine, hyper, rule, loss):
        self.e, self.hyper, self.rule, self.loss = engine, hyper, rule, loss
        self.max_global_batch = max_step_batch(max(engine.table.feature_sizes))

    def forward(self, idx, y, inv_b):
        """-> records [B, kp + 4]: per sample (S[kp], dz, loss, pad) -- one buffer, so ONE all-gather carries everything
        the global update needs from this rank."""
        B = idx.shape[0]
        if getattr(self, "_rec", None) is None or self._rec.shape[0] < B:
            self._rec = torch.zeros((B, self.e.table.kp + 4), dtype=torch.float32, device=self.e.device)
        rec = self._rec[:B]
        self.e.forward(self.hyper, idx, None, y, loss=self.loss, inv_b=inv_b, want_first=False, want_bi=False, records=rec)
        return rec

    def start_sort(self, idx_g):
        """The global occurrence sort only needs the gathered indices: it runs on a side stream while this rank's
        forward pass and the S / dlogit gathers proceed."""
        e = self.e
        e._ensure(idx_g.shape[0])
        cur = torch.cuda.current_stream(e.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=e.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            e.sort(idx_g)
        self._sorted_for = idx_g.data_ptr()

    def update(self, idx_g, rec_g, inv_b):
        """Row-reduced update over the global batch (sorted by start_sort, or here); returns the mean-loss tensor [1]."""
        e = self.e
        GB = idx_g.shape[0]
        e._ensure(GB)
        if getattr(self, "_sorted_for", None) == idx_g.data_ptr() and getattr(self, "_side", None) is not None:
            torch.cuda.current_stream(e.device).wait_stream(self._side)
        else:
            e.sort(idx_g)
        self._sorted_for = None
        e.update(self.hyper, self.rule, GB, None, None, inv_b=inv_b, with_loss=True, records=rec_g)
        return e.loss_out


class DataParallelFM:
    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._bufs = {}

    def _gathered(self, name, local):
        """all_gather `local` ([B, ...] contiguous) into a rank-major [G*B, ...] buffer."""
        if self.world == 1:
            return local
        shape = (self.world * local.shape[0],) + tuple(local.shape[1:])
        key = (name, shape, local.dtype, local.device)
        out = self._bufs.get(key)
        if out is None:
            out = self._bufs[key] = torch.empty(shape, dtype=local.dtype, device=local.device)
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks sharing one GPU, where RCCL cannot run): stage through the host
            host = torch.empty(shape, dtype=local.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def step(self, idx_local, y_local):
        """One exact data-parallel step.  idx_local [B,F] int32 and y_local [B] fp32 are this rank's slice; the global
        batch is the rank-major concatenation.  Returns the global mean-loss tensor [1] (of the last sub-step, see below).

        When G * B exceeds what one exact step can take (backend.max_global_batch: 16,384 for the Criteo vocabulary,
        whose largest field needs 18 index bits), the batch is processed as consecutive exact steps over equal slices of
        every rank's samples -- still exact online learning, with a smaller global batch per update."""
        B = idx_local.shape[0]
        cap = getattr(self.backend, "max_global_batch", None)
        n_sub = 1
        while cap is not None and (B // n_sub) * self.world > cap and (B // n_sub) % 2 == 0:
            n_sub *= 2
        if n_sub > 1:
            out = None
            Bs = B // n_sub
            for j in range(n_sub):
                out = self._step(idx_local[j * Bs:(j + 1) * Bs], y_local[j * Bs:(j + 1) * Bs])
            return out
        return self._step(idx_local, y_local)

    def prefetch(self, idx_next):
        """Gather (and start sorting) the NEXT step's indices now: they do not depend on the weights, and it takes the
        index all-gather off the next step's critical path.  Two alternating buffers, so a prefetch issued while a step
        still reads its own gathered indices does not overwrite them."""
        self._pf_toggle = 1 - getattr(self, "_pf_toggle", 0)
        idx_g = self._gathered(f"idx_pf{self._pf_toggle}", idx_next)
        self._pref = (idx_next.data_ptr(), idx_g)
        cap = getattr(self.backend, "max_global_batch", None)
        if hasattr(self.backend, "start_sort") and (cap is None or idx_g.shape[0] <= cap):
            self.backend.start_sort(idx_g)

    def _step(self, idx_local, y_local):
        B = idx_local.shape[0]
        inv_b = 1.0 / (B * self.world)
        pref, self._pref = getattr(self, "_pref", None), None
        if pref is not None and pref[0] == idx_local.data_ptr() and pref[1].shape[0] == B * self.world:
            idx_g = pref[1]                                   # gathered (and being sorted) since the previous step
        else:
            idx_g = self._gathered("idx", idx_local)          # independent of the weights: issued first
            if hasattr(self.backend, "start_sort"):
                self.backend.start_sort(idx_g)
        rec = self.backend.forward(idx_local, y_local, inv_b)
        rec_g = self._gathered("rec", rec)
        return self.backend.update(idx_g, rec_g, inv_b)
