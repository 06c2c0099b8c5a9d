"""FlatTable: every field's first-order weight and second-order embedding row in ONE HBM buffer.

The reference keeps 2 x 39 nn.Embedding modules (reference fm_adam.py:29-32).  Here field f owns rows
[offsets[f], offsets[f+1]) of a single [R, row_stride] fp32 buffer, so a sample's 39 active features are 39 row
gathers of one 64-byte request each, inside one 128-byte line (layouts in include/fmx.h):

    weights layout   [ V[0..kp) | w | pad ]                                         (rules 'signadam', 'sgd')
    ftrl layout      [ V[0..kp) | w, zw, nw, 0 | pad | zV[0..kp) | nV[0..kp) ]      (rule 'ftrl'; state is (z, n), V and
                                                                                     w are the weights derived from it)
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


def field_offsets(feature_sizes, allow_empty=False):
    sizes = [int(s) for s in feature_sizes]
    if any(s < (0 if allow_empty else 1) for s in sizes) or sum(sizes) < 1:
        raise ValueError("every field needs at least one row")
    return np.concatenate([[0], np.cumsum(sizes, dtype=np.int64)]).astype(np.int64)


def ftrl_weight_torch(z, n, h):
    """w(z, n) of FTRL-proximal in torch (host-side plumbing for import / export; the kernels have their own)."""
    w = -(z - torch.sign(z) * h["l1"]) / ((h["beta"] + torch.sqrt(n)) / h["alpha"] + h["l2"])
    return torch.where(z.abs() <= h["l1"], torch.zeros_like(w), w)


def ftrl_z_for_weight_torch(w, h):
    """z with ftrl_weight(z, n=0) == w: FTRL starts from given weights (FM needs V != 0 to learn at all)."""
    return -w * (h["beta"] / h["alpha"] + h["l2"]) - torch.sign(w) * h["l1"]


class FlatTable:
    def __init__(self, feature_sizes, k, layout="weights", device=None, row_stride=None, ftrl=None, field_cols=None,
                 field_base=None, n_cols=None):
        """field_cols / field_base / n_cols (include/fmx.h, "fields as pieces of index columns"; the multi-GPU owner mode): field f
        holds the indices [field_base[f], field_base[f] + feature_sizes[f]) of column field_cols[f] of an idx with n_cols
        columns; such a table may have empty fields (holes of the forward tree)."""
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
        self.ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=0.0)
        if ftrl:
            self.ftrl.update({k_: float(v) for k_, v in ftrl.items() if k_ in self.ftrl})
        kp = self.kp
        if layout == "weights":
            self.z_offset = 0
            need = kp + 4
            default = _round_up(need, 32) if kp >= 16 else 2 * kp      # k = 16: one 128-byte line per row
        else:
            # the (z, n) half starts on its own 128-byte line for kp >= 16; smaller rows share one line
            self.z_offset = _round_up(kp + 4, 32) if kp >= 16 else 2 * kp
            need = self.z_offset + 2 * kp
            default = _round_up(need, 32) if kp >= 16 else 4 * kp
        self.row_stride = default if row_stride is None else int(row_stride)
        if self.row_stride % 4 or self.row_stride < need:
            raise ValueError(f"row_stride must be a multiple of 4 and >= {need}")
        self.mapped = field_cols is not None
        offs = field_offsets(self.feature_sizes, allow_empty=self.mapped)
        if self.mapped:
            fb = [0] * self.n_fields if field_base is None else [int(v) for v in field_base]
            self.field_cols_host, self.field_base_host = [int(c) for c in field_cols], fb
            assert len(self.field_cols_host) == self.n_fields == len(fb)
            self.n_cols = int(n_cols) if n_cols is not None else max(self.field_cols_host) + 1
            assert 0 <= min(self.field_cols_host) and max(self.field_cols_host) < self.n_cols
        else:
            assert field_base is None and n_cols is None
            self.n_cols = self.n_fields
        self.offsets_host = offs
        self.n_rows = int(offs[-1])
        self.rows = torch.zeros((self.n_rows, self.row_stride), dtype=torch.float32, device=self.device)
        self.offsets = torch.from_numpy(offs).to(self.device)
        if self.mapped:
            self.field_cols = torch.tensor(self.field_cols_host, dtype=torch.int32, device=self.device)
            self.field_base = torch.tensor(self.field_base_host, dtype=torch.int32, device=self.device)
        self.bias = torch.zeros(1 if layout == "weights" else 2, dtype=torch.float32, device=self.device)
        self._cstruct = None
        self._sort_split = None
        self.sort_cap_override = None

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
            t.z_offset = self.z_offset
            t.max_field_rows = max(self.feature_sizes)
            if self._sort_split is not None:
                so, sc, mx = self._sort_split
                t.sort_offsets, t.sort_cols, t.n_sort_fields, t.max_sort_field_rows = so.data_ptr(), sc.data_ptr(), sc.numel(), mx
            if self.mapped:
                t.field_cols, t.field_base, t.n_cols = self.field_cols.data_ptr(), self.field_base.data_ptr(), self.n_cols
            self._cstruct = t
        return C.byref(self._cstruct)

    # ---- sort fields: pieces of the large fields, so that (index, sample) fits 32 bits at large batches ----
    def ensure_sort_split(self, B):
        """An exact step over B samples packs (index within the sort field, sample) into 32 bits (include/fmx.h, "SORT
        FIELDS"): fields with more rows than that leaves room for are cut into equal consecutive pieces.  Called by FMEngine
        whenever its largest batch grows; a no-op for the Criteo vocabulary up to 16,384 samples."""
        bbits = max(6, int(np.ceil(np.log2(max(int(B), 1)))))
        cap = 0xFFFFFFFF >> bbits                              # rows a sort field may have
        if self.sort_cap_override:                             # tests: force a finer split (identical results, more sort fields)
            cap = min(cap, int(self.sort_cap_override))
        if max(self.feature_sizes) <= cap:
            if self._sort_split is not None:
                self._sort_split, self._cstruct = None, None
            return
        offs, cols, mx = [0], [], 0
        for f, size in enumerate(self.feature_sizes):
            n = max(1, (size + cap - 1) // cap)                  # (an empty field of a mapped table stays one empty sort field)
            base, lo = size // n, int(self.offsets_host[f])
            for j in range(n):
                rows = base + (1 if j < size % n else 0)
                lo += rows
                offs.append(lo)
                cols.append(f)
                mx = max(mx, rows)
        so = torch.tensor(offs, dtype=torch.int64, device=self.device)
        sc = torch.tensor(cols, dtype=torch.int32, device=self.device)
        if self._sort_split is None or self._sort_split[1].numel() != sc.numel() or self._sort_split[2] != mx:
            self._sort_split, self._cstruct = (so, sc, mx), None

    # ---- strided views into the flat buffer (both layouts keep [ V | w ] at the head of the row) ----
    @property
    def V(self):
        return self.rows[:, :self.k]

    @property
    def w(self):
        return self.rows[:, self.kp]

    def field_V(self, f):
        lo, hi = int(self.offsets_host[f]), int(self.offsets_host[f + 1])
        return self.rows[lo:hi, :self.k]

    def field_w(self, f):
        lo, hi = int(self.offsets_host[f]), int(self.offsets_host[f + 1])
        return self.rows[lo:hi, self.kp:self.kp + 1]

    # ---- reference-shaped import / export ----
    def load_reference(self, first_list, second_list):
        """first_list[f]: [size_f, 1]; second_list[f]: [size_f, k] (reference nn.Embedding weights).
        FTRL layout: n = 0 and the z that reproduces these weights."""
        V = torch.cat([torch.as_tensor(t, dtype=torch.float32).reshape(s, self.k)
                       for t, s in zip(second_list, self.feature_sizes)]).to(self.device)
        w = torch.cat([torch.as_tensor(t, dtype=torch.float32).reshape(s)
                       for t, s in zip(first_list, self.feature_sizes)]).to(self.device)
        if self.layout == "weights":
            self.rows.zero_()
            self.rows[:, :self.k] = V
            self.rows[:, self.kp] = w
        else:
            self.load_ftrl_state(ftrl_z_for_weight_torch(V, self.ftrl), torch.zeros_like(V),
                                 ftrl_z_for_weight_torch(w, self.ftrl), torch.zeros_like(w))

    def export_reference(self):
        """Per-field (first [size,1], second [size,k]) weights on the CPU (for FTRL: the derived weights)."""
        rows = self.rows.detach().cpu()
        first, second = [], []
        for f in range(self.n_fields):
            lo, hi = int(self.offsets_host[f]), int(self.offsets_host[f + 1])
            first.append(rows[lo:hi, self.kp:self.kp + 1].clone())
            second.append(rows[lo:hi, :self.k].clone())
        return first, second

    # ---- ftrl layout: (z, n) state; V and w are re-derived here exactly as the kernels do after an update ----
    def load_ftrl_state(self, zV, nV, zw, nw):
        assert self.layout == "ftrl"
        kp, k, zo, dev = self.kp, self.k, self.z_offset, self.device
        zV, nV = torch.as_tensor(zV, dtype=torch.float32).to(dev), torch.as_tensor(nV, dtype=torch.float32).to(dev)
        zw, nw = torch.as_tensor(zw, dtype=torch.float32).to(dev), torch.as_tensor(nw, dtype=torch.float32).to(dev)
        self.rows.zero_()
        self.rows[:, zo:zo + k] = zV
        self.rows[:, zo + kp:zo + kp + k] = nV
        self.rows[:, kp + 1] = zw
        self.rows[:, kp + 2] = nw
        self.rows[:, :k] = ftrl_weight_torch(zV, nV, self.ftrl)
        self.rows[:, kp] = ftrl_weight_torch(zw, nw, self.ftrl)

    def export_ftrl_state(self):
        assert self.layout == "ftrl"
        kp, k, zo = self.kp, self.k, self.z_offset
        r = self.rows.detach().cpu()
        return (r[:, zo:zo + k].clone(), r[:, zo + kp:zo + kp + k].clone(), r[:, kp + 1].clone(), r[:, kp + 2].clone())

    def bias_weight(self):
        """The bias as a 0-d device tensor (FTRL: derived from its (z, n) pair)."""
        if self.layout == "weights":
            return self.bias[0]
        return ftrl_weight_torch(self.bias[0], self.bias[1], self.ftrl)

    def set_bias_weight(self, value):
        value = float(value)
        if self.layout == "weights":
            self.bias[0] = value
        else:
            self.bias[0] = float(ftrl_z_for_weight_torch(torch.tensor(value), self.ftrl))
            self.bias[1] = 0.0
