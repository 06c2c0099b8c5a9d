# -*- coding:utf-8 -*-
"""Host-side input shaping: Criteo CSV / libsvm -> the Xi / Xv / Y nested lists the model classes take.

Same function names, argument meaning, return shapes and `random` consumption as the reference's
utils/data_preprocess.py (cited per function), re-implemented: the libsvm value -> index map is vectorised
(the reference does an O(N * vocab) list.index per cell, data_preprocess.py:102-108) and the file parsers are ours.
Indices stay per-field LOCAL (each field's table starts at 0); fmx.FlatTable adds the field offsets on the device.
"""
import random

import numpy as np

N_CRITEO_FIELDS = 39


def load_criteo_category_index(file_path):
    """`field,category,index` lines -> list of 39 dicts {category: index}   (reference :15-26)."""
    cate_dict = [dict() for _ in range(N_CRITEO_FIELDS)]
    with open(file_path, "r") as fh:
        for line in fh:
            field, category, index = line.strip().split(",")
            cate_dict[int(field)][category] = int(index)
    return cate_dict


def read_criteo_data(file_path, emb_file):
    """`label,i0..i38` lines -> {'size','label','index','value' (all 1),'feature_sizes'}   (reference :29-47)."""
    result = {"size": 0, "label": [], "index": [], "value": [], "feature_sizes": []}
    result["feature_sizes"] = [len(d) for d in load_criteo_category_index(emb_file)]
    with open(file_path, "r") as fh:
        for line in fh:
            cells = line.strip().split(",")
            result["label"].append(int(cells[0]))
            result["index"].append([int(c) for c in cells[1:]])
            result["value"].append([1] * N_CRITEO_FIELDS)
    result["size"] += len(result["value"])
    return result


def _split_by_label(labels):
    groups = {"0": [], "1": []}
    for i, lab in enumerate(labels):
        groups["0" if lab == 0 else "1"].append(i)
    return groups


def _balanced_order(labels):
    """Down-sample the negatives to the number of positives, then shuffle (two random.shuffle calls, :61-65)."""
    groups = _split_by_label(labels)
    random.shuffle(groups["0"])
    keep = groups["0"][:len(groups["1"])]
    keep.extend(groups["1"])
    random.shuffle(keep)
    return keep


def balance_criteo_data(file_path, emb_file):
    """reference :50-83"""
    result = read_criteo_data(file_path, emb_file)
    order = _balanced_order(result["label"])
    result["index"] = [result["index"][i] for i in order]
    result["value"] = [result["value"][i] for i in order]
    result["label"] = [result["label"][i] for i in order]
    result["size"] = len(result["value"])
    return result


def _parse_libsvm(file_path, n_features):
    labels, rows = [], []
    with open(file_path, "r") as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            cells = line.split()
            labels.append(float(cells[0]))
            row = np.zeros(n_features, dtype=np.float64)
            for cell in cells[1:]:
                j, v = cell.split(":")
                row[int(j) - 1] = float(v)
            rows.append(row)
    return np.asarray(rows, dtype=np.float64).reshape(-1, n_features), np.asarray(labels, dtype=np.float64)


def read_svm_file(file_path, permutation=False):
    """libsvm with 8 features -> per-column distinct-value indices in order of first appearance, raw values, labels
    with -1 -> 0   (reference :86-115)."""
    X, y = _parse_libsvm(file_path, 8)
    if permutation:
        order = np.random.permutation(X.shape[0])
        x_values, labels = np.asarray(X[order]), np.asarray(y[order])
    else:
        x_values, labels = np.asarray(X), np.asarray(y).astype(int)
    index = np.empty_like(x_values)
    sizes = []
    for j in range(x_values.shape[1]):
        col = x_values[:, j]
        uniq, first_pos, inverse = np.unique(col, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[np.argsort(first_pos, kind="stable")] = np.arange(len(uniq))
        index[:, j] = rank[inverse]
        sizes.append(len(uniq))
    return {"size": len(x_values), "label": np.where(labels == -1, 0, labels).astype(int), "index": index.astype(int),
            "value": x_values, "feature_sizes": np.asarray(sizes).astype(int)}


def balance_svm_data(file_path):
    """reference :118-151"""
    result = read_svm_file(file_path)
    order = _balanced_order(result["label"])
    result["index"] = np.asarray([result["index"][i] for i in order])
    result["value"] = np.asarray([result["value"][i] for i in order])
    result["label"] = np.asarray([result["label"][i] for i in order])
    result["size"] = len(result["value"])
    return result


def _construct_batch_criteo_data(train_dict, num_batchdata, num_batch):
    """Consecutive slices of num_batchdata samples   (reference :154-178)."""
    Xi, Xv, Y, ratios = [], [], [], []
    for i in range(num_batch):
        sl = slice(i * num_batchdata, (i + 1) * num_batchdata)
        yi = list(train_dict["label"][sl])
        if len(yi) < num_batchdata:
            raise IndexError("list index out of range")
        Xi.append(list(train_dict["index"][sl]))
        Xv.append(list(train_dict["value"][sl]))
        Y.append(yi)
        pos = sum(yi)
        ratios.append((len(yi) - pos, pos))
    return Xi, Xv, Y, ratios


def _find_pos_and_neg(file_path, emb_file):
    """reference :181-187"""
    result = read_criteo_data(file_path, emb_file)
    ratios = {"0": [], "1": []}
    for i, label in enumerate(result["label"]):
        ratios[str(label)].append(i)
    return result, ratios


def _draw_batches(result, ratios, num_batch, num_batchdata, num_pos_of, ratio_of):
    Xi, Xv, Y, ratio_list = [], [], [], []
    for i in range(num_batch):
        num_pos = num_pos_of(i)
        num_neg = num_batchdata - num_pos
        ratio_list.append(ratio_of(i, num_neg, num_pos))
        picked = ratios["1"][:num_pos] + ratios["0"][:num_neg]
        ratios["1"] = ratios["1"][num_pos:]
        ratios["0"] = ratios["0"][num_neg:]
        random.shuffle(picked)
        Xi.append([result["index"][j] for j in picked])
        Xv.append([result["value"][j] for j in picked])
        Y.append([result["label"][j] for j in picked])
    return Xi, Xv, Y, ratio_list


def create_ten_iter(file_path, emb_file, num_batch, num_batchdata):
    """Batch i holds int(num_batchdata / num_batch * (i + 1)) positives   (reference :190-226)."""
    result, ratios = _find_pos_and_neg(file_path, emb_file)
    return _draw_batches(result, ratios, num_batch, num_batchdata,
                         lambda i: int(num_batchdata / num_batch * (i + 1)), lambda i, neg, pos: (neg, pos))


def create_dataset(file_path, emb_file, batch_ratio, num_batch, num_batchdata):
    """Every batch holds int(num_batchdata / num_batch * batch_ratio) positives   (reference :229-264)."""
    result, ratios = _find_pos_and_neg(file_path, emb_file)
    return _draw_batches(result, ratios, num_batch, num_batchdata,
                         lambda i: int(num_batchdata / num_batch * batch_ratio),
                         lambda i, neg, pos: (batch_ratio, num_batch - batch_ratio))


def _make_user_post_dict(train_Xi, train_Y):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# the array / pinned-staging form of the same inputs (not in the reference: its models take nested lists, converted with
# torch.LongTensor(...) inside every forward(), reference fm_adam.py:35-36 -- ~10 ms per 4,096 samples, 300x a GPU step)
# ---------------------------------------------------------------------------------------------------------------------
def read_criteo_arrays(file_path, emb_file):
    """read_criteo_data (reference :29-47) into arrays: {'size', 'label' int64 [N], 'index' int32 [N, 39] (per-field LOCAL
    indices, the very numbers of the file), 'value': None (every Criteo value is 1), 'feature_sizes'}.  One C-speed parse
    instead of a Python loop per cell."""
    feature_sizes = [len(d) for d in load_criteo_category_index(emb_file)]
    raw = np.loadtxt(file_path, delimiter=",", dtype=np.int64, ndmin=2)
    if raw.shape[1] != N_CRITEO_FIELDS + 1:
        raise ValueError(f"{file_path}: expected label + {N_CRITEO_FIELDS} indices per line, found {raw.shape[1]} columns")
    return {"size": int(raw.shape[0]), "label": raw[:, 0].copy(), "index": np.ascontiguousarray(raw[:, 1:], dtype=np.int32),
            "value": None, "feature_sizes": feature_sizes}


class PinnedBatchStager:
    """Mini-batches of a host-resident stream as device tensors, staged through PINNED host buffers with non-blocking
    copies on a copy stream of their own, `depth` batches in flight (double-buffered by default): batch i + 1 crosses PCIe
    while batch i trains.

        for idx_d, xv_d, y_d in PinnedBatchStager(index, label, 4096, device):      # int32 [B, F], None or fp32 [B, F], fp32 [B]
            model.update_embedding(idx_d, xv_d, y_d)

    index: integer array [N, F] (per-field local indices); label: [N]; value: None (all ones, the Criteo case) or real [N, F].
    The tensors of a batch are valid until `depth` further batches have been drawn (their buffers are reused; the copy into
    a buffer waits for the work that was current on the consumer's stream when its previous batch was handed out plus
    everything enqueued until the next draw).  A last partial batch is yielded unless drop_last.  feature_sizes: if given,
    indices are range-checked on the host (vectorised), IndexError like nn.Embedding.  register_in_place=True: int32 index /
    fp32 value arrays (C-contiguous) are pinned where they are (hipHostRegister) and every batch is a DMA out of them -- no host pass,
    70 against 93 us per 4096 x 39 batch through FMAdam.update_embedding -- but the FIRST copy out of freshly pinned pages costs
    ~0.8 ms per batch (tools/class_surface_passes.py): worth it for a dataset that is passed over several times, a loss for a
    stream seen once, hence off by default.  The arrays must not be freed or resized while the stager lives (it keeps
    references); close() unpins them."""

    def __init__(self, index, label, batch_size, device=None, value=None, depth=2, drop_last=False, feature_sizes=None,
                 register_in_place=False, copy_stream=None):
        import torch
        self.torch = torch
        self.index = np.asarray(index)
        self.label = np.asarray(label)
        self.value = None if value is None else np.asarray(value)
        if self.index.ndim != 2 or self.label.shape[0] != self.index.shape[0]:
            raise ValueError(f"index {self.index.shape} must be [N, F] and label {self.label.shape} [N]")
        if feature_sizes is not None:
            sizes = np.asarray(feature_sizes, dtype=np.int64)[None, :]
            if (self.index < 0).any() or (self.index >= sizes).any():
                raise IndexError("index out of range in self")
        self.B, self.F, self.depth, self.drop_last = int(batch_size), self.index.shape[1], max(int(depth), 1), drop_last
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        pin = self.device.type == "cuda"
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=pin)
        self._host = [(mk((self.B, self.F), torch.int32), mk((self.B,), torch.float32),
                       None if self.value is None else mk((self.B, self.F), torch.float32)) for _ in range(self.depth)]
        self._dev = [(torch.empty((self.B, self.F), dtype=torch.int32, device=self.device),
                      torch.empty((self.B,), dtype=torch.float32, device=self.device),
                      None if self.value is None else torch.empty((self.B, self.F), dtype=torch.float32, device=self.device))
                     for _ in range(self.depth)]
        # copy_stream: reuse one across stagers -- every new stream of a process takes the next of a few hardware queues
        # (GPU_MAX_HW_QUEUES, 4 by default) and may land on the queue the consumer computes on, where the copies then run between the
        # kernels instead of beside them
        self._copy = (copy_stream if copy_stream is not None else torch.cuda.Stream(device=self.device)) if pin else None
        self._ready = [torch.cuda.Event() if pin else None for _ in range(self.depth)]
        self._free = [torch.cuda.Event() if pin else None for _ in range(self.depth)]
        # register_in_place: arrays that already have the device's types are PINNED WHERE THEY ARE (hipHostRegister through torch's
        # runtime binding): a batch is then a DMA straight out of the dataset -- no host pass at all (the typed memcpy into a pinned
        # buffer is 65 of the 75 us this class costs per 4096 x 39 batch: tools/class_surface_profile.py).  Other dtypes keep the
        # staged path below.
        self._registered = []
        self._src = None
        if pin and register_in_place:
            self._src = self._register_in_place()

    def _register_in_place(self):
        torch = self.torch
        if self.index.dtype != np.int32 or not self.index.flags["C_CONTIGUOUS"]:
            return None
        lab = self.label if (self.label.dtype == np.float32 and self.label.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(self.label, dtype=np.float32)
        val = self.value
        if val is not None and (val.dtype != np.float32 or not val.flags["C_CONTIGUOUS"]):
            return None
        arrays = [self.index, lab] + ([] if val is None else [val])
        rt = torch.cuda.cudart()
        done = []
        for a in arrays:
            if a.nbytes == 0 or int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) != 0:
                for b in done:
                    rt.cudaHostUnregister(b.ctypes.data)
                return None
            done.append(a)
        self._registered = done                              # kept alive (and unregistered in close()) with this object
        ts = [torch.from_numpy(a) for a in arrays]
        return (ts[0], ts[1], ts[2] if val is not None else None)

    def close(self):
        """Unpin arrays pinned in place (also run when the object is collected)."""
        if self._registered:
            if self._copy is not None:
                self._copy.synchronize()
            rt = self.torch.cuda.cudart()
            for a in self._registered:
                rt.cudaHostUnregister(a.ctypes.data)
            self._registered = []
            self._src = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = self.index.shape[0]
        return n // self.B if self.drop_last else (n + self.B - 1) // self.B

    def _stage(self, slot, lo, hi):
        torch = self.torch
        n = hi - lo
        hi_, hy_, hv_ = self._host[slot]
        di_, dy_, dv_ = self._dev[slot]
        if self._src is not None:                          # pinned in place: the copies read the dataset itself
            si_, sy_, sv_ = self._src
            cur = torch.cuda.current_stream(self.device)
            self._copy.wait_stream(cur)                    # the consumer is done with this slot's device tensors (see class doc)
            with torch.cuda.stream(self._copy):
                di_[:n].copy_(si_[lo:hi], non_blocking=True)
                dy_[:n].copy_(sy_[lo:hi], non_blocking=True)
                if sv_ is not None:
                    dv_[:n].copy_(sv_[lo:hi], non_blocking=True)
                self._ready[slot].record(self._copy)
            return
        if self._copy is not None:
            self._free[slot].synchronize()                 # the copy that last read this pinned buffer has completed
        hi_[:n].numpy()[...] = self.index[lo:hi]           # the only host pass over the batch: a typed memcpy into pinned memory
        hy_[:n].numpy()[...] = self.label[lo:hi]
        if hv_ is not None:
            hv_[:n].numpy()[...] = self.value[lo:hi]
        if self._copy is None:
            di_[:n].copy_(hi_[:n]); dy_[:n].copy_(hy_[:n])
            if hv_ is not None:
                dv_[:n].copy_(hv_[:n])
            return
        cur = torch.cuda.current_stream(self.device)
        self._copy.wait_stream(cur)                        # the consumer is done with this slot's device tensors (see class doc)
        with torch.cuda.stream(self._copy):
            di_[:n].copy_(hi_[:n], non_blocking=True)
            dy_[:n].copy_(hy_[:n], non_blocking=True)
            if hv_ is not None:
                dv_[:n].copy_(hv_[:n], non_blocking=True)
            self._free[slot].record(self._copy)
            self._ready[slot].record(self._copy)

    def __iter__(self):
        torch = self.torch
        n_batches = len(self)
        N = self.index.shape[0]
        bounds = [(i * self.B, min((i + 1) * self.B, N)) for i in range(n_batches)]
        for i in range(min(self.depth, n_batches)):
            self._stage(i % self.depth, *bounds[i])
        for i in range(n_batches):
            slot = i % self.depth
            lo, hi = bounds[i]
            n = hi - lo
            if self._copy is not None:
                torch.cuda.current_stream(self.device).wait_event(self._ready[slot])
            di_, dy_, dv_ = self._dev[slot]
            yield di_[:n], (None if dv_ is None else dv_[:n]), dy_[:n]
            if i + self.depth < n_batches:                  # refill the slot just consumed with batch i + depth
                self._stage(slot, *bounds[i + self.depth])
