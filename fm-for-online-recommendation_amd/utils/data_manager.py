"""Dataset readers feeding hot path B and the Frappe-shaped configuration -- same names, arguments and return
values as the reference's utils/data_manager.py (cited per function), re-implemented with vectorised parsing."""
import csv
from datetime import datetime

import numpy as np
from scipy.sparse import coo_matrix


def load_dataset_movielens(filename, lines, columns, nbUsers):
    """ml-100k `user \\t movie \\t rating \\t timestamp` -> one-hot sparse X [lines, columns] (float32), X2 with a
    constant column appended (float64), Y = (rating >= 3), Y2 = raw rating strings, timestamps (float32)
    (reference :18-48)."""
    users, movies, ratings, stamps = [], [], [], []
    with open(filename, "r") as fh:
        for user, movie, rating, stamp in csv.reader(fh, delimiter="\t"):
            users.append(int(user) - 1)
            movies.append(int(nbUsers) + int(movie) - 1)
            ratings.append(rating)
            stamps.append(stamp)
    n = len(users)
    r = np.arange(n)
    rows = np.concatenate([r, r])
    cols = np.concatenate([users, movies]).astype(np.int64)
    ones = np.ones(2 * n)
    X = coo_matrix((ones.astype(np.float32), (rows, cols)), shape=(lines, columns)).tolil().astype("float32")
    X2 = coo_matrix((np.ones(3 * n), (np.concatenate([rows, r]), np.concatenate([cols, np.full(n, columns)]))),
                    shape=(lines, columns + 1)).tolil().astype("float64")
    Y = np.array([1 if int(x) >= 3 else 0 for x in ratings]).astype("float64")
    return X, X2, Y, ratings, np.array(stamps).astype("float32")


def sort_dataset_movielens(X, Y, utc_time_stamp):
    """Rows in time order   (reference :51-59)."""
    time_order = np.array([datetime.utcfromtimestamp(float(t)) for t in utc_time_stamp])
    order = time_order.argsort()
    Y = np.array([float(v) for v in Y])
    return X[order], Y[order], time_order


def load_dataset_YearPredictionMSD(dataname, isTransformY=False, isRemoveEmpty=False):
    """libsvm -> min-max scaled dense X, y   (reference :62-80)."""
    from sklearn.datasets import load_svmlight_file
    X, y = load_svmlight_file(dataname)
    X = np.asarray(X.todense())
    print("Size of X is " + str(X.shape[0]) + "-by-" + str(X.shape[1]))
    print("Size of y is " + str(y.shape))
    lo, hi = X.min(axis=0), X.max(axis=0)
    X = (X - lo) / np.where(hi > lo, hi - lo, 1.0)
    if isTransformY:
        y = (y * 2) - 3
    if isRemoveEmpty:
        keep = np.sum(np.abs(X), axis=1) > 1e-6
        X, y = X[keep, :], y[keep]
    return X, y


def load_dataset_fappe(file, logloss_opt=False):
    """Frappe libfm lines `label tok tok ...` -> global token ids in order of first appearance, rows sorted by their
    number of tokens (stable)   (reference :83-136).  Returns (X, Y)."""
    features = {}
    X_, Y_ = [], []
    with open(file) as fh:
        lines = [ln.strip().split(" ") for ln in fh if ln.strip()]
    for items in lines:
        for tok in items[1:]:
            if tok not in features:
                features[tok] = len(features)
    for items in lines:
        Y_.append(1.0 * float(items[0]))
        X_.append([features[tok] for tok in items[1:]])
    order = np.argsort([len(r) for r in X_])
    rows = [X_[i] for i in order]
    ragged = len({len(r) for r in rows}) > 1
    X = np.empty(len(rows), dtype=object) if ragged else np.asarray(rows)
    if ragged:                      # rows of different lengths: an object array of lists (numpy >= 1.24 refuses to
        X[:] = rows                 # build it implicitly, where the reference's np.asarray raises ValueError)
    return X, np.asarray([Y_[i] for i in order])
