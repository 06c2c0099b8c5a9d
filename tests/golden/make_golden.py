#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *imported* reference.

Run in the build container only (the reference never travels to the GPU box):

    cd /root/repo && PYTHONPATH=/root/reference MPLBACKEND=Agg python tests/golden/make_golden.py

Everything is seeded (torch / numpy / random), inputs are synthetic and small, and
the outputs are plain arrays (npz) and JSON: data, not code.  The reference classes
are driven exactly through their public surface:

  G1  first_order / second_order / forward / forward_fm / predict at the initial state
  G2  the loss of update_embedding and of fit (both loss variants, duplicated rows in the batch)
  G3  the full state_dict after one update_embedding and after one fit
  G4  ONN: alpha trajectory and hidden weights over 16 B=1 fit steps; tables unchanged
  G5  run_experiment on a fixed 64-sample stream (4-tuple minus time) + final state
  G6  FM_FTRL (cls and reg) on a seeded 256x8 stream: pred, final w1, W2 (fp64)
  G7  data_preprocess readers / batchers on hand-made CSV + libsvm files (committed beside this script)
"""
import io
import json
import os
import random
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

from models.models_online_deep.fm_adam import FMAdam            # noqa: E402
from models.models_online_deep.deepfm_adam import DeepFMAdam    # noqa: E402
from models.models_online_deep.nfm_adam import NFMAdam          # noqa: E402
from models.models_online_deep.deepfm_onn import DeepFMOnn      # noqa: E402
from models.models_online_deep.nfm_onn import NFMOnn            # noqa: E402
from models.models_online.FM_FTRL import FM_FTRL                # noqa: E402
from utils import data_preprocess                               # noqa: E402

CLASSES = {"FMAdam": FMAdam, "DeepFMAdam": DeepFMAdam, "NFMAdam": NFMAdam,
           "DeepFMOnn": DeepFMOnn, "NFMOnn": NFMOnn}


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def put(out, prefix, d):
    for k, v in d.items():
        out[prefix + "/" + k] = np.asarray(v)


def build(name, feature_sizes, k, L, H, n, batch_size):
    cls = CLASSES[name]
    if name == "FMAdam":
        return cls(feature_sizes, embedding_size=k, n=n, use_cuda=False)
    if name == "NFMAdam":
        return cls(feature_sizes, embedding_size=k, num_hidden_layers=L, neuron_per_hidden_layer=H, n=n,
                   use_cuda=False)
    return cls(feature_sizes, embedding_size=k, num_hidden_layers=L, neuron_per_hidden_layer=H,
               batch_size=batch_size, n=n, use_cuda=False)


def make_inputs(rng, feature_sizes, B, real_values, dup=True):
    F = len(feature_sizes)
    Xi = np.stack([rng.integers(0, fs, size=B) for fs in feature_sizes], axis=1).astype(np.int64)
    if dup and B >= 3:
        # force duplicated rows inside the batch: sample 1 repeats sample 0 on the even fields,
        # sample 2 repeats sample 0 on every third field
        Xi[1, ::2] = Xi[0, ::2]
        Xi[2, ::3] = Xi[0, ::3]
    if real_values:
        Xv = rng.uniform(-1.0, 1.0, size=(B, F)).astype(np.float32)
    else:
        Xv = np.ones((B, F), dtype=np.float32)
    Y = rng.integers(0, 2, size=B).astype(np.int64)
    return Xi, Xv, Y


def model_fixture(name, tag, feature_sizes, k, L, H, n, seed):
    out = {}
    rng = np.random.default_rng(seed)
    B1, B2 = 8, 6
    Xi1, Xv1, Y1 = make_inputs(rng, feature_sizes, B1, real_values=False)
    Xi2, Xv2, Y2 = make_inputs(rng, feature_sizes, B2, real_values=True)
    meta = dict(name=name, tag=tag, feature_sizes=list(map(int, feature_sizes)), k=k, L=L, H=H, n=n, seed=seed,
                B1=B1, B2=B2)

    # ---- model A: batch ops (ONN needs batch_size == B of fit) ----
    seed_all(seed)
    m = build(name, feature_sizes, k, L, H, n, batch_size=B2)
    put(out, "A/sd0", sd_np(m))
    out["A/Xi1"], out["A/Xv1"], out["A/Y1"] = Xi1, Xv1, Y1
    out["A/Xi2"], out["A/Xv2"], out["A/Y2"] = Xi2, Xv2, Y2
    with torch.no_grad():
        for j, (Xi, Xv) in enumerate(((Xi1, Xv1), (Xi2, Xv2)), start=1):
            Xil, Xvl = Xi.tolist(), Xv.tolist()
            if name == "FMAdam":
                fwd = m.forward(Xil, Xvl)
                out[f"A/forward{j}"] = fwd.numpy()
            else:
                out[f"A/first_order{j}"] = m.first_order(Xil, Xvl).numpy()
                out[f"A/second_order{j}"] = m.second_order(Xil, Xvl).numpy()
                out[f"A/forward_fm{j}"] = m.forward_fm(Xil, Xvl).numpy()
                fwd = m.forward(Xil, Xvl)
                if isinstance(fwd, tuple):
                    out[f"A/forward{j}"] = fwd[0].numpy()
                    out[f"A/forward_layers{j}"] = fwd[1].numpy()
                else:
                    out[f"A/forward{j}"] = fwd.numpy()
            out[f"A/predict{j}"] = np.asarray(m.predict(Xil, Xvl))
    loss = m.update_embedding(Xi1.tolist(), Xv1.tolist(), Y1.tolist())
    out["A/loss_update_embedding"] = loss.detach().numpy()
    put(out, "A/sd1", sd_np(m))
    # second update_embedding with real-valued Xv
    loss = m.update_embedding(Xi2.tolist(), Xv2.tolist(), Y2.tolist())
    out["A/loss_update_embedding2"] = loss.detach().numpy()
    put(out, "A/sd2", sd_np(m))
    m.fit(Xi2.tolist(), Xv2.tolist(), Y2.tolist())
    put(out, "A/sd3", sd_np(m))

    # ---- model B: B=1 online stream through run_experiment ----
    seed_all(seed + 1)
    mb = build(name, feature_sizes, k, L, H, n, batch_size=1)
    put(out, "B/sd0", sd_np(mb))
    NS = 64
    Xis, Xvs, Ys = make_inputs(rng, feature_sizes, NS, real_values=False, dup=False)
    out["B/Xi"], out["B/Xv"], out["B/Y"] = Xis, Xvs, Ys
    if name.endswith("Onn"):
        alphas = [mb.alpha.detach().numpy().copy()]
        for i in range(16):
            mb.fit([Xis[i].tolist()], [Xvs[i].tolist()], [int(Ys[i])])
            alphas.append(mb.alpha.detach().numpy().copy())
        out["B/alpha_traj"] = np.stack(alphas)
        put(out, "B/sd_fit16", sd_np(mb))
        # restart from the same init for run_experiment
        seed_all(seed + 1)
        mb = build(name, feature_sizes, k, L, H, n, batch_size=1)
    _, acc, roc, cm = mb.run_experiment(Xis.tolist(), Xvs.tolist(), Ys.tolist())
    meta["run_experiment"] = dict(accuracy=float(acc), roc={k_: float(v) for k_, v in roc.items()},
                                  confusion_matrix={k_: int(v) for k_, v in cm.items()})
    meta["str"] = str(mb)
    put(out, "B/sd_end", sd_np(mb))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def fm_ftrl_fixture():
    out = {}
    rng = np.random.default_rng(77)
    N, d, m = 256, 8, 8
    X = rng.uniform(-1, 1, size=(N, d))
    for task in ("cls", "reg"):
        if task == "cls":
            y = np.where(rng.uniform(size=N) < 0.5, -1.0, 1.0)
        else:
            y = rng.normal(size=N)
        eta = 0.005
        torch.manual_seed(5)
        w1_0 = torch.randn(d, 1).type(torch.DoubleTensor).numpy().copy()
        W2_0 = torch.randn(2 * m, d - 1).type(torch.DoubleTensor).numpy().copy()
        torch.manual_seed(5)
        model = FM_FTRL(torch.DoubleTensor(X), torch.DoubleTensor(y), task, eta, m)
        with redirect_stdout(io.StringIO()):
            pred, real, _ = model.online_learning()
        out[f"{task}/X"], out[f"{task}/y"] = X, y
        out[f"{task}/w1_0"], out[f"{task}/W2_0"] = w1_0, W2_0
        out[f"{task}/pred"] = np.asarray([float(p) for p in pred])
        out[f"{task}/real"] = np.asarray([float(r) for r in real])
        out[f"{task}/w1"] = model.w1.numpy()
        out[f"{task}/W2"] = model.W2.numpy()
        out[f"{task}/eta"], out[f"{task}/m"] = np.float64(eta), np.int64(m)
    path = os.path.join(HERE, "FM_FTRL.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def write_handmade_files():
    """Our own small data files in the reference's on-disk formats (not copies of any reference file)."""
    rng = np.random.default_rng(2024)
    sizes = [int(s) for s in rng.integers(2, 9, size=39)]
    with open(os.path.join(HERE, "handmade_category_emb.csv"), "w") as f:
        for fld, s in enumerate(sizes):
            for c in range(s):
                f.write(f"{fld},cat{fld}_{c},{c}\n")
    with open(os.path.join(HERE, "handmade_train_input.csv"), "w") as f:
        for i in range(60):
            label = 1 if i % 3 != 0 else 0
            f.write(",".join([str(label)] + [str(int(rng.integers(0, s))) for s in sizes]) + "\n")
    with open(os.path.join(HERE, "handmade.libsvm"), "w") as f:
        vals = [-1.0, -0.5, 0.25, 0.75, 1.0]
        for i in range(24):
            lab = -1 if i % 2 else 1
            feats = [f"{j + 1}:{vals[int(rng.integers(0, len(vals)))]}" for j in range(8)]
            f.write(f"{lab} " + " ".join(feats) + "\n")
    return sizes


def preprocess_fixture():
    write_handmade_files()
    csv = os.path.join(HERE, "handmade_train_input.csv")
    emb = os.path.join(HERE, "handmade_category_emb.csv")
    svm = os.path.join(HERE, "handmade.libsvm")
    res = {}
    r = data_preprocess.read_criteo_data(csv, emb)
    res["read_criteo_data"] = dict(size=r["size"], label=r["label"], index=r["index"], value=r["value"],
                                   feature_sizes=r["feature_sizes"])
    random.seed(11)
    a = data_preprocess.create_ten_iter(csv, emb, 4, 8)
    res["create_ten_iter"] = dict(Xi=a[0], Xv=a[1], Y=a[2], ratio=[list(t) for t in a[3]])
    random.seed(12)
    a = data_preprocess.create_dataset(csv, emb, 3, 4, 8)
    res["create_dataset"] = dict(Xi=a[0], Xv=a[1], Y=a[2], ratio=[list(t) for t in a[3]])
    random.seed(13)
    r = data_preprocess.balance_criteo_data(csv, emb)
    res["balance_criteo_data"] = dict(size=r["size"], label=r["label"], index=r["index"], value=r["value"])
    r = data_preprocess.read_svm_file(svm)
    res["read_svm_file"] = dict(size=int(r["size"]), label=np.asarray(r["label"]).tolist(),
                                index=np.asarray(r["index"]).tolist(), value=np.asarray(r["value"]).tolist(),
                                feature_sizes=np.asarray(r["feature_sizes"]).tolist())
    random.seed(14)
    r = data_preprocess.balance_svm_data(svm)
    res["balance_svm_data"] = dict(size=int(r["size"]), label=np.asarray(r["label"]).tolist(),
                                   index=np.asarray(r["index"]).tolist(), value=np.asarray(r["value"]).tolist())
    path = os.path.join(HERE, "data_preprocess.json")
    with open(path, "w") as f:
        json.dump(res, f)
    print("wrote", path, os.path.getsize(path), "bytes")


def data_manager_fixture():
    """G8: utils/data_manager.py and utils/metric_manager.py on hand-made files / arrays."""
    from utils import data_manager, metric_manager
    rng = np.random.default_rng(8)
    ml = os.path.join(HERE, "handmade_movielens.tsv")
    n_users, n_movies, n_lines = 6, 9, 30
    with open(ml, "w") as f:
        for i in range(n_lines):
            f.write(f"{int(rng.integers(1, n_users + 1))}\t{int(rng.integers(1, n_movies + 1))}\t"
                    f"{int(rng.integers(1, 6))}\t{int(880000000 + rng.integers(0, 100000))}\n")
    fr = os.path.join(HERE, "handmade_frappe.libfm")
    with open(fr, "w") as f:
        for i in range(20):
            ntok = 4                      # equal-length rows (np.asarray refuses ragged ones in numpy >= 1.24)
            toks = [f"{int(rng.integers(0, 40))}:1" for _ in range(ntok)]
            f.write(("1" if i % 3 else "-1") + " " + " ".join(toks) + "\n")
    out = {}
    X, X2, Y, Y2, dt = data_manager.load_dataset_movielens(ml, n_lines, n_users + n_movies, n_users)
    out["ml/X"], out["ml/X2"], out["ml/Y"], out["ml/dt"] = X.toarray(), X2.toarray(), Y, dt
    out["ml/Y2"] = np.asarray([int(v) for v in Y2])
    # the reference hands numpy float32 stamps to datetime.utcfromtimestamp, which Python 3.10 rejects: feed floats
    sX, sY, order = data_manager.sort_dataset_movielens(X, Y2, [float(t) for t in dt])
    out["ml/sorted_X"], out["ml/sorted_Y"] = sX.toarray(), sY
    fx, fy = data_manager.load_dataset_fappe(fr)
    out["frappe/X"], out["frappe/Y"] = fx, fy
    pred, real = rng.normal(size=50), rng.normal(size=50)
    out["metric/pred"], out["metric/real"] = pred, real
    out["metric/reg"] = metric_manager.regression_metric(pred, real)
    sp, sr = np.sign(pred), np.sign(real)
    m, acc = metric_manager.classfication_metric(sp, sr)
    out["metric/cls"], out["metric/cls_acc"] = m, acc
    path = os.path.join(HERE, "data_manager.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def path_b_family_fixture():
    """G9: SFTRL_CCFM, SFTRL_Vanila, RRF_Online (cls and reg) on a seeded 300x8 stream."""
    from models.models_online.SFTRL_CCFM import SFTRL_CCFM
    from models.models_online.SFTRL_Vanila import SFTRL_Vanila
    from models.models_online.RRF_Online import RRF_Online
    out = {}
    rng = np.random.default_rng(91)
    N, d, m = 300, 8, 4
    X = rng.uniform(-1, 1, size=(N, d))
    for task in ("cls", "reg"):
        y = np.where(rng.uniform(size=N) < 0.5, -1.0, 1.0) if task == "cls" else rng.normal(size=N)
        out[f"{task}/X"], out[f"{task}/y"] = X, y
        for name, cls in (("SFTRL_CCFM", SFTRL_CCFM), ("SFTRL_Vanila", SFTRL_Vanila)):
            model = cls(torch.DoubleTensor(X), torch.DoubleTensor(y), task, 0.05, m)
            with redirect_stdout(io.StringIO()):
                pred, real, _ = model.online_learning()
            out[f"{task}/{name}/pred"] = np.asarray([np.asarray(p, dtype=np.float64) for p in pred])
            out[f"{task}/{name}/pred_shape"] = np.asarray(np.asarray(pred).shape)
            out[f"{task}/{name}/real"] = np.asarray([float(r) for r in real])
            out[f"{task}/{name}/BTP_gram"] = (model.BT_P.matmul(model.BT_P.t())).numpy()
            out[f"{task}/{name}/BTN_gram"] = (model.BT_N.matmul(model.BT_N.t())).numpy()
            out[f"{task}/{name}/counts"] = np.asarray([model.row_count_p, model.row_count_n])
            if name == "SFTRL_Vanila":
                out[f"{task}/{name}/w"] = model.w.numpy()
        # RRF's cls dynamics amplify rounding differences (1e-16 after 100 steps -> 1e-3 after 300): pin 100 steps
        seed_all(17)
        model = RRF_Online(torch.DoubleTensor(X[:100]), torch.DoubleTensor(y[:100]), task, num_sampled_spectral=6)
        with redirect_stdout(io.StringIO()):
            pred, real, _ = model.online_learning()
        out[f"{task}/RRF/pred"] = np.asarray([np.asarray(p, dtype=np.float64) for p in pred])
        out[f"{task}/RRF/pred_shape"] = np.asarray(np.asarray(pred).shape)
        out[f"{task}/RRF/real"] = np.asarray([float(r) for r in real])
        out[f"{task}/RRF/w"], out[f"{task}/RRF/gamma"] = model.w.numpy(), model.gamma.numpy()
    path = os.path.join(HERE, "path_b_family.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    if "--only-data-manager" in sys.argv:
        data_manager_fixture()
        return
    if "--only-path-b" in sys.argv:
        path_b_family_fixture()
        return
    assert any(p.rstrip("/") == "/root/reference" for p in sys.path), "run with PYTHONPATH=/root/reference"
    rng = np.random.default_rng(39)
    criteo39s = [int(s) for s in rng.integers(3, 24, size=39)]      # 39 fields, small vocabularies
    for i, name in enumerate(CLASSES):
        model_fixture(name, "criteo39s", criteo39s, k=10, L=5, H=10, n=0.01, seed=100 + 10 * i)
        model_fixture(name, "tiny4", [7, 5, 11, 3], k=4, L=2, H=32, n=0.01, seed=200 + 10 * i)
    fm_ftrl_fixture()
    preprocess_fixture()
    data_manager_fixture()
    path_b_family_fixture()


if __name__ == "__main__":
    main()
