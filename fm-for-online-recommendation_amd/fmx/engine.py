"""FMEngine: workspace + launch sequencing for one FlatTable (sort -> forward -> update on the current stream)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .table import FlatTable


class Hyper:
    """Update-rule hyper-parameters (include/fmx.h fmx_hyper_t)."""

    def __init__(self, lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.0, l2=0.0):
        self.c = _lib.Hyper(lr, eps, alpha, beta, l1, l2)

    def ref(self):
        return C.byref(self.c)


def normalize_inputs(Xi, Xv, n_fields, feature_sizes=None):
    """The reference's input convention (reference fm_adam.py:35-36): nested lists (or arrays), a single sample may
    be 1-D.  Returns (idx int32 [B,F] numpy, xv float32 [B,F] numpy or None when every value is exactly 1).
    Out-of-range indices raise IndexError like nn.Embedding does."""
    idx = np.asarray(Xi, dtype=np.int64).reshape(-1, n_fields)
    xv = np.asarray(Xv, dtype=np.float32).reshape(-1, n_fields)
    if idx.shape != xv.shape:
        raise ValueError(f"Xi {idx.shape} and Xv {xv.shape} disagree")
    if feature_sizes is not None:
        sizes = np.asarray(feature_sizes, dtype=np.int64)[None, :]
        if (idx < 0).any() or (idx >= sizes).any():
            raise IndexError("index out of range in self")
    if np.all(xv == 1.0):
        xv = None
    return idx.astype(np.int32), xv


def _ptr(t):
    return None if t is None else t.data_ptr()


class HandOffTimeout(RuntimeError):
    """Device error word 2 (k_fm_update: a crossing run's row update was skipped): an in-launch hand-off ran into its spin
    bound.  Never observed; the table must be considered corrupt."""

    def __init__(self, code):
        super().__init__(f"fmx: in-launch hand-off timed out (device error word {code}); the table is not the exact result")
        self.code = code


class FMEngine:
    def __init__(self, table: FlatTable, max_batch=4096):
        if not torch.cuda.is_available():
            raise RuntimeError("fmx needs a ROCm GPU: the hot path has no CPU implementation")
        self.lib = _lib.load()
        self.table = table
        self.device = table.device
        self.max_batch = 0
        self._alloc(max_batch)

    def _alloc(self, B):
        B = int(B)
        t, dev = self.table, self.device
        t.ensure_sort_split(B)                       # large fields are cut into sort pieces when (index, sample) would not fit 32 bits
        Bp = self.lib.fmx_sorted_width(B)
        f32 = dict(dtype=torch.float32, device=dev)
        nbytes = int(self.lib.fmx_workspace_bytes(t.c_struct(), B))
        if nbytes < 0:
            _lib.check(nbytes)
        self.workspace = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
        nsf = t.n_fields if t._sort_split is None else t._sort_split[1].numel()
        self.sorted = self.workspace[:nsf * Bp].view(nsf, Bp)     # the occurrence lists (uint32 bits), one per sort field
        self.S = torch.empty((B, t.kp), **f32)
        self.bi = torch.empty((B, t.kp), **f32)
        self.first = torch.empty((B, t.n_fields), **f32)
        self.sfirst = torch.empty(B, **f32)
        self.sbi = torch.empty(B, **f32)
        self.logit = torch.empty(B, **f32)
        self.loss_b = torch.empty(B, **f32)
        self.dz = torch.empty(B, **f32)
        self.loss_out = torch.zeros(1, **f32)
        self.error = torch.zeros(1, dtype=torch.int32, device=dev)
        self.max_batch = B

    def _ws_bytes(self):
        return self.workspace.numel() * self.workspace.element_size()

    def _ensure(self, B):
        if B > self.max_batch:
            self._alloc(B)

    def _stream(self, stream=None):
        """The HIP stream handle a launch goes to: `stream` (an int handle or a torch stream) when the caller already has
        it -- torch.cuda.current_stream() costs several microseconds per call -- else torch's current stream."""
        if stream is None:
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the handle itself: ~0.3 us against ~3 for the Stream object
            if raw is not None and self.device.index is not None:
                return raw(self.device.index)
            return torch.cuda.current_stream(self.device).cuda_stream
        return stream if isinstance(stream, int) else stream.cuda_stream

    def _fwd_out(self, want_first=True, want_bi=True):
        o = _lib.FwdOut()
        o.S, o.sfirst, o.sbi, o.logit = self.S.data_ptr(), self.sfirst.data_ptr(), self.sbi.data_ptr(), self.logit.data_ptr()
        o.bi = self.bi.data_ptr() if want_bi else None
        o.first = self.first.data_ptr() if want_first else None
        o.loss, o.dz, o.error = self.loss_b.data_ptr(), self.dz.data_ptr(), self.error.data_ptr()
        return o

    # ---- device inputs ----
    def to_device(self, idx, xv=None, y=None):
        dev = self.device
        idx_d = torch.as_tensor(idx, dtype=torch.int32).to(dev).contiguous()
        xv_d = None if xv is None else torch.as_tensor(xv, dtype=torch.float32).to(dev).contiguous()
        y_d = None if y is None else torch.as_tensor(np.asarray(y, dtype=np.float32)).reshape(-1).to(dev).contiguous()
        return idx_d, xv_d, y_d

    # ---- kernels ----
    def forward(self, hyper, idx_d, xv_d=None, y_d=None, loss=None, inv_b=None, want_first=True, want_bi=True, records=None,
                stream=None):
        """records: optional [B, ld] fp32 tensor (ld >= kp + 2, multiple of 4): S, dz and loss are written as fields of one
        per-sample record (S = rec[:kp], dz = rec[kp], loss = rec[kp + 1]) instead of the engine's dense buffers."""
        B = idx_d.shape[0]
        self._ensure(B)
        key = (want_first, want_bi, None if records is None else (records.data_ptr(), records.shape[1]), self.S.data_ptr())
        cached = getattr(self, "_fwd_cache", None)
        if cached is None or cached[0] != key:     # the output struct only changes with the buffers it points at
            out = self._fwd_out(want_first, want_bi)
            if records is not None:
                kp, ld = self.table.kp, records.shape[1]
                assert ld >= kp + 2 and ld % 4 == 0 and records.is_contiguous()
                base = records.data_ptr()
                out.S, out.dz, out.loss, out.sample_ld = base, base + 4 * kp, base + 4 * (kp + 1), ld
            self._fwd_cache = cached = (key, out)
        if records is not None:
            assert records.shape[0] >= B
        inv_b = 1.0 / B if inv_b is None else inv_b
        _lib.check(self.lib.fmx_fm_forward(self.table.c_struct(), hyper.ref(), idx_d.data_ptr(), _ptr(xv_d), _ptr(y_d), B,
                                           _lib.LOSSES[loss], inv_b, C.byref(cached[1]), self._stream(stream)))
        return B

    def new_workspace(self, B):
        """A further (zero-filled) per-step workspace for batches up to B: a caller that sorts several batches ahead keeps
        one per batch in flight and passes it to sort() / update()."""
        nbytes = int(self.lib.fmx_workspace_bytes(self.table.c_struct(), B))
        return torch.zeros(nbytes // 4, dtype=torch.int32, device=self.device)

    def sort(self, idx_d, workspace=None, stream=None):
        B = idx_d.shape[0]
        self._ensure(B)
        ws = self.workspace if workspace is None else workspace
        _lib.check(self.lib.fmx_sort_occurrences(self.table.c_struct(), idx_d.data_ptr(), B, ws.data_ptr(), ws.numel() * ws.element_size(),
                                                 self.error.data_ptr(), self._stream(stream)))

    def update(self, hyper, rule, B, xv_d, dz_first, dz_bi=None, gbi=None, inv_b=None, with_loss=True, S=None, loss_b=None,
               records=None, fm_term=True, workspace=None, stream=None):
        """Row-reduced backward + fused update for the batch whose occurrences self.sort() just listed.
        S / loss_b default to the buffers the last forward() filled (a data-parallel caller passes gathered ones, or
        `records` [B, ld] as written by forward(records=...): then dz_first = dz_bi = the records' dz field)."""
        inv_b = 1.0 / B if inv_b is None else inv_b
        ld = 0
        if records is not None:
            kp, ld = self.table.kp, records.shape[1]
            base = records.data_ptr()
            S_p, dzf_p, loss_p = base, base + 4 * kp, base + 4 * (kp + 1)
            dzb_p = dzf_p if fm_term else None
        else:
            S = self.S if S is None else S
            loss_b = self.loss_b if loss_b is None else loss_b
            S_p, dzf_p, dzb_p, loss_p = S.data_ptr(), dz_first.data_ptr(), _ptr(dz_bi), loss_b.data_ptr()
        ws = self.workspace if workspace is None else workspace
        _lib.check(self.lib.fmx_fm_update(self.table.c_struct(), hyper.ref(), _lib.RULES[rule], ws.data_ptr(), ws.numel() * ws.element_size(),
                                          _ptr(xv_d), S_p, dzf_p, dzb_p, _ptr(gbi), B, ld,
                                          loss_p if with_loss else None, inv_b,
                                          self.loss_out.data_ptr() if with_loss else None, self._stream(stream)))

    def step(self, hyper, rule, loss, idx_d, xv_d, y_d, inv_b=None):
        """One pure-FM mini-batch step; the mean loss lands in self.loss_out[0] (no sync here)."""
        B = idx_d.shape[0]
        self._ensure(B)
        cached = getattr(self, "_step_out", None)
        if cached is None or cached[0] != self.S.data_ptr():       # the output struct only changes with the buffers it points at
            self._step_out = cached = (self.S.data_ptr(), self._fwd_out(want_first=False, want_bi=False))
        out = cached[1]
        inv_b = 1.0 / B if inv_b is None else inv_b
        _lib.check(self.lib.fmx_fm_step(self.table.c_struct(), hyper.ref(), _lib.RULES[rule], _lib.LOSSES[loss],
                                        idx_d.data_ptr(), _ptr(xv_d), y_d.data_ptr(), B, inv_b, self.workspace.data_ptr(), self._ws_bytes(),
                                        C.byref(out), self.loss_out.data_ptr(), self._stream()))

    def stream(self, hyper, rule, loss, idx_pool, y_pool, n_steps, loss_out=None, timed=False):
        """The online loop over a resident pool of batches (fmx_fm_stream).  Returns per-launch ms [sort, forward, update, empty event pair] when timed (the measuring mode repeats launches: see fmx.h)."""
        n_pool, B, F = idx_pool.shape
        assert F == self.table.n_fields and y_pool.shape == (n_pool, B)
        self._ensure(B)
        out = self._fwd_out(want_first=False, want_bi=False)
        ms = (C.c_float * 4)() if timed else None
        _lib.check(self.lib.fmx_fm_stream(self.table.c_struct(), hyper.ref(), _lib.RULES[rule], _lib.LOSSES[loss],
                                          idx_pool.data_ptr(), y_pool.data_ptr(), n_pool, B, 1.0 / B, n_steps,
                                          self.workspace.data_ptr(), self._ws_bytes(), C.byref(out), _ptr(loss_out), ms, self._stream()))
        return None if ms is None else [float(v) for v in ms]

    def prepare_stream(self, hyper, rule, loss, idx_pool, y_pool, loss_out=None, stream=None):
        """-> run(n_steps): the online loop of stream() with every argument but the step count bound once (the structs, the
        device pointers and the HIP stream handle: `stream`, a torch stream or an int handle, else the stream current NOW).
        A call is then one foreign call -- what stream() spends in Python before it (building the output struct, looking up
        the current stream: tens of microseconds) is as long as two steps of the loop.  The caller keeps idx_pool, y_pool and
        loss_out alive and does not grow the engine between prepare and run."""
        n_pool, B, F = idx_pool.shape
        assert F == self.table.n_fields and y_pool.shape == (n_pool, B)
        assert loss_out is None or loss_out.is_contiguous()
        self._ensure(B)
        out = self._fwd_out(want_first=False, want_bi=False)
        fn, check = self.lib.fmx_fm_stream, _lib.check
        fixed = (self.table.c_struct(), hyper.ref(), _lib.RULES[rule], _lib.LOSSES[loss], idx_pool.data_ptr(), y_pool.data_ptr(),
                 n_pool, B, 1.0 / B)
        tail = (self.workspace.data_ptr(), self._ws_bytes(), C.byref(out), _ptr(loss_out), None, self._stream(stream))
        cap = None if loss_out is None else loss_out.numel()
        keep = (out, hyper, idx_pool, y_pool, loss_out, self.workspace)   # referenced by the closure: stay alive with it

        def run(n_steps):
            if cap is not None and n_steps > cap:
                raise ValueError(f"loss_out holds {cap} steps, {n_steps} asked for")
            check(fn(*fixed, n_steps, *tail))
        run.keep = keep
        return run

    def prepare_deepfm_stream(self, hyper, rule, loss, params, grads, k, hidden, n_layers, lr_mlp, idx_pool, y_pool, loss_out=None,
                              stream=None, fm_term=True):
        """-> run(n_steps): the mini-batch DeepFM loop over a resident pool (fmx_deepfm_stream), every argument but the step count
        bound once; per step forward, MLP section (SGD of the MLP applied in it: lr_mlp), table update, all issued from one call
        (fm_term=False: NFM -- the network's base is first-order + bias; weights-layout tables).
        `params` / `grads`: the flat MLP buffers (W_l then b_l per layer).  The caller keeps the tensors alive and does not grow
        the engine between prepare and run."""
        n_pool, B, F = idx_pool.shape
        assert F == self.table.n_fields and y_pool.shape == (n_pool, B)
        assert loss_out is None or loss_out.is_contiguous()
        self._ensure(B)
        out = self._fwd_out(want_first=False, want_bi=True)
        m = self._mlp_struct(params, k, hidden, n_layers)
        self._mlp_big_buffers(m, k, hidden, n_layers, B)
        fn, check = self.lib.fmx_deepfm_stream, _lib.check
        fixed = (self.table.c_struct(), hyper.ref(), _lib.RULES[rule], C.byref(m), _lib.LOSSES[loss], int(bool(fm_term)), idx_pool.data_ptr(), y_pool.data_ptr(),
                 n_pool, B, 1.0 / B)
        tail = (self.workspace.data_ptr(), self._ws_bytes(), self._mlp_ws.data_ptr(), C.byref(out), self._mlp_dz.data_ptr(),
                self._mlp_gbi.data_ptr(), grads.data_ptr(), lr_mlp, _ptr(loss_out), self._stream(stream))
        cap = None if loss_out is None else loss_out.numel()
        keep = (out, m, hyper, params, grads, idx_pool, y_pool, loss_out, self.workspace, self._mlp_ws, self._mlp_dz, self._mlp_gbi)

        def run(n_steps):
            if cap is not None and n_steps > cap:
                raise ValueError(f"loss_out holds {cap} steps, {n_steps} asked for")
            check(fn(*fixed, n_steps, *tail))
        run.keep = keep
        return run

    # ---- the small fused MLP (online steps of DeepFM / NFM / ONN); shapes beyond its limits raise FmxError(UNSUPPORTED) ----
    MLP_MAX_B, MLP_MAX_W, MLP_MAX_L = 16, 64, 8

    @staticmethod
    def mlp_fits(B, k, hidden, n_layers, mode):
        ok = B <= 16 and hidden <= 64 and 1 <= n_layers <= 8
        if mode == "fit":
            return ok and k <= 63
        if mode == "hedge":
            return ok and k + n_layers <= 64
        return ok and k <= 64

    def _mlp_struct(self, params, k, hidden, n_layers):
        return _lib.Mlp(params.data_ptr(), n_layers, k, hidden, 0)

    def mlp_forward(self, params, k, hidden, n_layers, base, B, want_layers):
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        layers = torch.empty((n_layers, B), dtype=torch.float32, device=self.device) if want_layers else None
        m = self._mlp_struct(params, k, hidden, n_layers)
        _lib.check(self.lib.fmx_mlp_forward(C.byref(m), self.bi.data_ptr(), self.table.kp, base.data_ptr(), B, out.data_ptr(),
                                            _ptr(layers), self._stream()))
        return out, layers

    def mlp_fit(self, params, k, hidden, n_layers, hyper, rule, loss, base, y_d, B, inv_b=None):
        """-> (dz [B], gbi [B, kp]) for self.update(); the hidden layers in `params` are updated in place."""
        dz = torch.empty(B, dtype=torch.float32, device=self.device)
        gbi = torch.empty((B, self.table.kp), dtype=torch.float32, device=self.device)
        m = self._mlp_struct(params, k, hidden, n_layers)
        _lib.check(self.lib.fmx_mlp_fit(C.byref(m), hyper.ref(), _lib.RULES[rule], _lib.LOSSES[loss], self.bi.data_ptr(),
                                        self.table.kp, base.data_ptr(), y_d.data_ptr(), B, 1.0 / B if inv_b is None else inv_b,
                                        dz.data_ptr(), gbi.data_ptr(), self.loss_out.data_ptr(), self._stream()))
        return dz, gbi

    def mlp_hedge_fit(self, params, k, hidden, n_layers, lr, hedge_b, hedge_s, alpha, base, y_d, B):
        m = self._mlp_struct(params, k, hidden, n_layers)
        _lib.check(self.lib.fmx_mlp_hedge_fit(C.byref(m), lr, hedge_b, hedge_s, alpha.data_ptr(), self.bi.data_ptr(),
                                              self.table.kp, base.data_ptr(), y_d.data_ptr(), B, None, self._stream()))

    def online_run(self, hyper, rule, loss, idx_d, xv_d, y_d, want_loss=False):
        """The reference's online protocol for pure FM on N device-resident samples (fmx_fm_online_run): per sample
        predict, then fit on that sample.  -> (pred uint8 [N], loss [N] or None)."""
        N = idx_d.shape[0]
        pred = torch.empty(N, dtype=torch.uint8, device=self.device)
        loss_b = torch.empty(N, dtype=torch.float32, device=self.device) if want_loss else None
        _lib.check(self.lib.fmx_fm_online_run(self.table.c_struct(), hyper.ref(), _lib.RULES[rule], _lib.LOSSES[loss],
                                              idx_d.data_ptr(), _ptr(xv_d), y_d.data_ptr(), N, pred.data_ptr(), _ptr(loss_b),
                                              self.error.data_ptr(), self._stream()))
        return pred, loss_b

    def online_run_mlp(self, hyper, rule, loss, params, k, hidden, n_layers, hedge, fm_term, hedge_b, hedge_s, alpha,
                       idx_d, xv_d, y_d):
        """The online protocol for the classes with an MLP (fmx_online_run_mlp) -> forward() value per sample [N]."""
        N = idx_d.shape[0]
        self._ensure(1)
        out = self._fwd_out(want_first=False, want_bi=True)
        m = self._mlp_struct(params, k, hidden, n_layers)
        pred = torch.empty(N, dtype=torch.float32, device=self.device)
        if getattr(self, "_online_scratch", None) is None:
            self._online_scratch = torch.zeros(self.table.kp + 8, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fmx_online_run_mlp(self.table.c_struct(), hyper.ref(), _lib.RULES[rule], _lib.LOSSES[loss], C.byref(m),
                                               1 if hedge else 0, 1 if fm_term else 0, hedge_b, hedge_s, _ptr(alpha),
                                               idx_d.data_ptr(), _ptr(xv_d), y_d.data_ptr(), N, self.workspace.data_ptr(), self._ws_bytes(),
                                               C.byref(out), self._online_scratch.data_ptr(), pred.data_ptr(), self._stream()))
        return pred

    @staticmethod
    def online_run_fits(n_fields, kp):
        return n_fields <= 4 * (64 // (kp // 4))

    def _mlp_big_buffers(self, m, k, hidden, n_layers, B):
        key = (k, hidden, n_layers, B)
        if getattr(self, "_mlp_ws_key", None) != key:
            nbytes = int(self.lib.fmx_mlp_section_workspace_bytes(C.byref(m), B))
            if nbytes < 0:
                raise _lib.FmxError(nbytes, self.lib.fmx_last_error_string().decode())
            self._mlp_ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            self._mlp_dz = torch.empty(B, dtype=torch.float32, device=self.device)
            self._mlp_gbi = torch.empty((B, self.table.kp), dtype=torch.float32, device=self.device)
            self._mlp_loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._mlp_ws_key = key

    def mlp_forward_batch(self, params, k, hidden, n_layers, bi, base, B, want_layers):
        """forward() of the MLP at mini-batch sizes (fmx_mlp_forward_batch) -> (logit [B], layers [L, B] or None)."""
        m = self._mlp_struct(params, k, hidden, n_layers)
        self._mlp_big_buffers(m, k, hidden, n_layers, B)
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        layers = torch.empty((n_layers, B), dtype=torch.float32, device=self.device) if want_layers else None
        _lib.check(self.lib.fmx_mlp_forward_batch(C.byref(m), bi.data_ptr(), bi.stride(0), base.data_ptr(), B,
                                                  self._mlp_ws.data_ptr(), out.data_ptr(), _ptr(layers), self._stream()))
        return out, layers

    def mlp_hedge_section(self, params, grads, k, hidden, n_layers, lr, hedge_b, hedge_s, alpha, bi, base, y_d, B):
        """Hedge backprop at mini-batch sizes (fmx_mlp_hedge_section): hidden layers and alpha updated in place."""
        m = self._mlp_struct(params, k, hidden, n_layers)
        self._mlp_big_buffers(m, k, hidden, n_layers, B)
        _lib.check(self.lib.fmx_mlp_hedge_section(C.byref(m), lr, hedge_b, hedge_s, alpha.data_ptr(), bi.data_ptr(),
                                                  bi.stride(0), base.data_ptr(), y_d.data_ptr(), B, self._mlp_ws.data_ptr(),
                                                  grads.data_ptr(), None, self._stream()))

    def mlp_section(self, params, grads, k, hidden, n_layers, loss, bi, base, y_d, B, inv_b, lr_apply=0.0):
        """The MLP on `bi` at mini-batch sizes (fmx_mlp_section: fp32 MFMA GEMMs): forward, loss, backward.
        -> (loss [1], dz [B], gbi [B, kp]); `grads` (flat, the layout of `params`) is filled; lr_apply != 0 also applies SGD."""
        m = self._mlp_struct(params, k, hidden, n_layers)
        self._mlp_big_buffers(m, k, hidden, n_layers, B)
        _lib.check(self.lib.fmx_mlp_section(C.byref(m), _lib.LOSSES[loss], bi.data_ptr(), bi.stride(0), base.data_ptr(),
                                            y_d.data_ptr(), B, inv_b, self._mlp_ws.data_ptr(), None, self._mlp_dz.data_ptr(),
                                            self._mlp_gbi.data_ptr(), self.table.kp, grads.data_ptr(), lr_apply,
                                            self._mlp_loss.data_ptr(), self._stream()))
        return self._mlp_loss, self._mlp_dz, self._mlp_gbi

    def check_error_flag(self):
        """The device-side error word (include/fmx.h, Conventions): 1 -> IndexError like nn.Embedding; 2 -> HandOffTimeout
        (an in-launch hand-off ran into its spin bound: the table is no longer the exact result).  Synchronises."""
        code = int(self.error.item())
        if code != 0:
            self.error.zero_()
            if code == 1:
                raise IndexError("index out of range in self (flagged by the fmx kernels)")
            raise HandOffTimeout(code)
