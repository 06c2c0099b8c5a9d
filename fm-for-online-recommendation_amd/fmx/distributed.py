"""Data-parallel online FM over the GPUs of one node (one process per GPU, torch.distributed: "nccl" is RCCL on ROCm).

Exact by construction: every rank keeps a full replica of the table, runs the forward pass on its own slice of the
global mini-batch, and the ranks all-gather what the backward needs.  For FM that is NOT the row gradients
(39 x 68 B per sample) but their low-rank factors: the sample's indices, its sum vector S_b and its dlogit --
(156 + 64 + 8) B per sample -- because every row gradient of a sample is  x (S_b - x V_row) dz_b.  Each rank then
runs the same deterministic sort + row-reduced update over the GLOBAL batch, so the replicas stay bit-identical,
no embedding row ever crosses xGMI, and the G-GPU step equals the 1-GPU step on the same global batch
(SURVEY.md section 8(e)).  The bias gradient is the sum of the gathered dlogits, so pure FM needs no all-reduce; a
dense network on top (DeepFM / NFM) adds one fused all-reduce of its gradients.

The compute behind a step is a small backend interface so that the sharding / gather logic can be exercised on CPU
with the gloo backend in tests (tests inject an oracle-backed backend; the product has only the HIP one).
"""
import torch
import torch.distributed as dist


class HipBackend:
    """The product backend: fmx.FMEngine on this rank's GPU."""

    def __init__(self, engine, hyper, rule, loss):
        self.e, self.hyper, self.rule, self.loss = engine, hyper, rule, loss

    def forward(self, idx, y, inv_b):
        """-> (S [B,kp], dz [B], loss_b [B]) views valid until the next forward."""
        B = self.e.forward(self.hyper, idx, None, y, loss=self.loss, inv_b=inv_b, want_first=False, want_bi=False)
        return self.e.S[:B], self.e.dz[:B], self.e.loss_b[:B]

    def update(self, idx_g, S_g, dz_g, loss_g, inv_b):
        """Sort + row-reduced update over the global batch; returns the mean-loss tensor [1] (no sync)."""
        e = self.e
        GB = idx_g.shape[0]
        e._ensure(GB)
        e.sort(idx_g)
        e.update(self.hyper, self.rule, GB, None, dz_g, dz_g, None, inv_b=inv_b, with_loss=True, S=S_g, loss_b=loss_g)
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
        batch is the rank-major concatenation.  Returns the global mean-loss tensor [1]."""
        B = idx_local.shape[0]
        inv_b = 1.0 / (B * self.world)
        idx_g = self._gathered("idx", idx_local)              # independent of the weights: issued first
        S, dz, loss_b = self.backend.forward(idx_local, y_local, inv_b)
        S_g = self._gathered("S", S)
        dz_g = self._gathered("dz", dz)
        loss_g = self._gathered("loss", loss_b)
        return self.backend.update(idx_g, S_g, dz_g, loss_g, inv_b)
