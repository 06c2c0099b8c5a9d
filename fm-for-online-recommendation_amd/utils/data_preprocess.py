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
