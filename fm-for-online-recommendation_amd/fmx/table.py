"""FlatTable: every field's first-order weight and second-order embedding row in ONE HBM buffer.

The reference keeps 2 x 39 nn.Embedding modules (reference fm_adam.py:29-32).  Here field f owns rows
[offsets[f], offsets[f+1]) of a single [R, row_stride] fp32 buffer, so a sample's 39 active features are 39 row
gathers of one 64-byte-aligned record each (layouts in include/fmx.h):

    weights layout   [ V[0..kp) | w | pad ]                     (rules 'signadam', 'sgd')
    ftrl layout      [ zV[0..kp) | nV[0..kp) | zw | nw | pad ]  (rule 'ftrl'; the weights are derived, never stored)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def padded_k(k):
    for kp in (4, 8, 16, 32, 64):
        if k <= kp:
            return kp
    raise ValueError(f"embedding_size {k} > 64 is not supported by the gfx950 kernels")


def _round_up(x, m):
    return (x + m - 1) // m * m


def field_offsets(feature_sizes):
    sizes = [int(s) for s in feature_sizes]
    if any(s < 1 for s in sizes):
        raise ValueError("every field needs at least one row")
    return np.concatenate([[0], np.cumsum(sizes, dtype=np.int64)]).astype(np.int64)


class FlatTable:
    def __init__(self, feature_sizes, k, layout="weights", device=None, row_stride=None):
        if device is None:
            device = torch.device("cuda")
        self.device = torch.device(device)
        self.feature_sizes = [int(s) for s in feature_sizes]
        self.n_fields = len(self.feature_sizes)
        self.k = int(k)
        self.kp = padded_k(self.k)
        self.layout = layout
        if layout not in ("weights", "ftrl"):
            raise ValueError(layout)
        need = (self.kp if layout == "weights" else 2 * self.kp) + 4
        self.row_stride = _round_up(need, 16) if row_stride is None else int(row_stride)
        if self.row_stride % 4 or self.row_stride < need:
            raise ValueError(f"row_stride must be a multiple of 4 and >= {need}")
        offs = field_offsets(self.feature_sizes)
        self.offsets_host = offs
        self.n_rows = int(offs[-1])
        self.rows = torch.zeros((self.n_rows, self.row_stride), dtype=torch.float32, device=self.device)
        self.offsets = torch.from_numpy(offs).to(self.device)
        self.bias = torch.zeros(1 if layout == "weights" else 2, dtype=torch.float32, device=self.device)
        self._cstruct = None

    # ---- C view ----
    def c_struct(self):
        if self._cstruct is None:
            t = _lib.Table()
            t.rows = self.rows.data_ptr()
            t.field_offsets = self.offsets.data_ptr()
            t.bias = self.bias.data_ptr()
            t.n_rows = self.n_rows
            t.n_fields = self.n_fields
            t.k, t.kp, t.row_stride = self.k, self.kp, self.row_stride
            t.layout = _lib.LAYOUT_WEIGHTS if self.layout == "weights" else _lib.LAYOUT_FTRL
            t.max_field_rows = max(self.feature_sizes)
            self._cstruct = t
        return C.byref(self._cstruct)

    # ---- strided views into the flat buffer (weights layout) ----
    @property
    def V(self):
        return self.rows[:, :self.k]

    @property
    def w(self):
        return self.rows[:, self.kp if self.layout == "weights" else 2 * self.kp]

    def field_V(self, f):
        lo, hi = int(self.offsets_host[f]), int(self.offsets_host[f + 1])
        return self.rows[lo:hi, :self.k]

    def field_w(self, f):
        lo, hi = int(self.offsets_host[f]), int(self.offsets_host[f + 1])
        return self.rows[lo:hi, self.kp:self.kp + 1]

    # ---- reference-shaped import / export (weights layout) ----
    def load_reference(self, first_list, second_list):
        """first_list[f]: [size_f, 1]; second_list[f]: [size_f, k] (reference nn.Embedding weights)."""
        assert self.layout == "weights"
        V = torch.cat([torch.as_tensor(t, dtype=torch.float32).reshape(s, self.k)
                       for t, s in zip(second_list, self.feature_sizes)])
        w = torch.cat([torch.as_tensor(t, dtype=torch.float32).reshape(s)
                       for t, s in zip(first_list, self.feature_sizes)])
        self.rows.zero_()
        self.rows[:, :self.k] = V.to(self.device)
        self.rows[:, self.kp] = w.to(self.device)

    def export_reference(self):
        assert self.layout == "weights"
        rows = self.rows.detach().cpu()
        first, second = [], []
        for f in range(self.n_fields):
            lo, hi = int(self.offsets_host[f]), int(self.offsets_host[f + 1])
            first.append(rows[lo:hi, self.kp:self.kp + 1].clone())
            second.append(rows[lo:hi, :self.k].clone())
        return first, second

    # ---- ftrl layout helpers ----
    def load_ftrl_state(self, zV, nV, zw, nw):
        assert self.layout == "ftrl"
        kp, k = self.kp, self.k
        self.rows.zero_()
        self.rows[:, :k] = torch.as_tensor(zV, dtype=torch.float32).to(self.device)
        self.rows[:, kp:kp + k] = torch.as_tensor(nV, dtype=torch.float32).to(self.device)
        self.rows[:, 2 * kp] = torch.as_tensor(zw, dtype=torch.float32).to(self.device)
        self.rows[:, 2 * kp + 1] = torch.as_tensor(nw, dtype=torch.float32).to(self.device)

    def export_ftrl_state(self):
        assert self.layout == "ftrl"
        kp, k = self.kp, self.k
        r = self.rows.detach().cpu()
        return (r[:, :k].clone(), r[:, kp:kp + k].clone(), r[:, 2 * kp].clone(), r[:, 2 * kp + 1].clone())
