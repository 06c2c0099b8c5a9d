"""Class-level parity on the GPU: the drop-in classes (models.models_online_deep.*) against the golden fixtures the
imported reference produced (tests/golden/make_golden.py) and against the oracle for the rules the reference lacks."""
import io
import pickle

import numpy as np
import pytest
import torch

from oracle import fm_oracle as orc
from helpers import (CLASS_NAMES, TAGS, assert_close, assert_ftrl_step_within_f64, assert_state_close, assert_within_f64, oracle_float64,
                     load_model_fixture, sub)

pytestmark = pytest.mark.gpu
RT = 1e-5


def build(name, meta, batch_size, **extra):
    from models.models_online_deep.fm_adam import FMAdam
    from models.models_online_deep.deepfm_adam import DeepFMAdam
    from models.models_online_deep.nfm_adam import NFMAdam
    from models.models_online_deep.deepfm_onn import DeepFMOnn
    from models.models_online_deep.nfm_onn import NFMOnn
    cls = dict(FMAdam=FMAdam, DeepFMAdam=DeepFMAdam, NFMAdam=NFMAdam, DeepFMOnn=DeepFMOnn, NFMOnn=NFMOnn)[name]
    fs, k, L, H, n = meta["feature_sizes"], meta["k"], meta["L"], meta["H"], meta["n"]
    if name == "FMAdam":
        return cls(fs, embedding_size=k, n=n, **extra)
    if name == "NFMAdam":
        return cls(fs, embedding_size=k, num_hidden_layers=L, neuron_per_hidden_layer=H, n=n, **extra)
    return cls(fs, embedding_size=k, num_hidden_layers=L, neuron_per_hidden_layer=H, batch_size=batch_size, n=n, **extra)


def sd_np(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_same_seed_same_model(name, tag):
    """The constructors draw from the torch RNG in the reference's order: same seed => identical parameters."""
    z, meta = load_model_fixture(name, tag)
    torch.manual_seed(meta["seed"])
    m = build(name, meta, meta["B2"])
    sd, ref = sd_np(m), sub(z, "A/sd0")
    assert set(sd) == set(ref)
    for k in ref:
        np.testing.assert_array_equal(sd[k], ref[k], err_msg=k)
    torch.manual_seed(meta["seed"] + 1)
    mb = build(name, meta, 1)
    assert str(mb) == meta["str"]


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_forward_pieces_vs_reference(name, tag):
    z, meta = load_model_fixture(name, tag)
    m = build(name, meta, meta["B2"])
    m.load_state_dict(sub(z, "A/sd0"))
    for j in (1, 2):
        Xi, Xv = z[f"A/Xi{j}"].tolist(), z[f"A/Xv{j}"].tolist()
        if name != "FMAdam":
            assert_close(m.first_order(Xi, Xv).cpu().numpy(), z[f"A/first_order{j}"], RT, 1e-7, "first_order")
            # bi = 0.5 (S^2 - sum e^2) cancels: fp32 leaves ~1e-6 of max|bi| whatever the summation order
            so = z[f"A/second_order{j}"]
            assert_close(m.second_order(Xi, Xv).cpu().numpy(), so, RT, 1e-6 * np.abs(so).max(), "second_order")
            assert_close(m.forward_fm(Xi, Xv).cpu().numpy(), z[f"A/forward_fm{j}"], RT, 2e-5, "forward_fm")
        out = m.forward(Xi, Xv)
        if isinstance(out, tuple):
            assert_close(out[0].cpu().numpy(), z[f"A/forward{j}"], RT, 1e-6, "forward")
            assert_close(out[1].cpu().numpy(), z[f"A/forward_layers{j}"], RT, 1e-6, "forward_layers")
        else:
            assert_close(out.cpu().numpy(), z[f"A/forward{j}"], RT, 2e-5, "forward")
        np.testing.assert_array_equal(np.asarray(m.predict(Xi, Xv)).reshape(-1), z[f"A/predict{j}"].reshape(-1))
    # a single sample may be passed 1-D (reference fm_adam.py:98)
    p1 = m.predict(z["A/Xi1"][0].tolist(), z["A/Xv1"][0].tolist())
    assert p1.shape == (1,) and bool(p1[0]) == bool(z["A/predict1"].reshape(-1)[0])


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_update_embedding_and_fit_vs_reference(name, tag):
    z, meta = load_model_fixture(name, tag)
    sd0 = sub(z, "A/sd0")
    m = build(name, meta, meta["B2"])
    m.load_state_dict(sd0)
    loss = m.update_embedding(z["A/Xi1"].tolist(), z["A/Xv1"].tolist(), z["A/Y1"].tolist())
    assert_close(float(loss.cpu().data), z["A/loss_update_embedding"], RT, 0, "loss1")
    sd1 = sub(z, "A/sd1")
    assert_state_close(sd_np(m), sd1, sd0, what="sd1")
    loss = m.update_embedding(z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist())
    assert_close(float(loss.cpu().data), z["A/loss_update_embedding2"], RT, 0, "loss2")
    sd2 = sub(z, "A/sd2")
    assert_state_close(sd_np(m), sd2, sd1, what="sd2")
    m.fit(z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist())
    # fit() of the MLP classes multiplies the row gradients by dz + dL/dbi, which can come out at the rule's eps (1e-8):
    # sign-like sensitivity there; 5e-11 = fp32 cancellation noise of a row gradient on this fixture (|dz S| ~ 1e-4)
    assert_state_close(sd_np(m), sub(z, "A/sd3"), sd2, what="sd3", sign_rule=(meta["n"], 1e-8, 5e-11))


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", ["DeepFMOnn", "NFMOnn"])
def test_hedge_trajectory_vs_reference(name, tag):
    z, meta = load_model_fixture(name, tag)
    sd0 = sub(z, "B/sd0")
    m = build(name, meta, 1)
    m.load_state_dict(sd0)
    traj = [m.alpha.cpu().numpy().copy()]
    for i in range(16):
        m.fit([z["B/Xi"][i].tolist()], [z["B/Xv"][i].tolist()], [int(z["B/Y"][i])])
        traj.append(m.alpha.cpu().numpy().copy())
    assert_close(np.stack(traj), z["B/alpha_traj"], 1e-5, 1e-7, "alpha")
    sd = sd_np(m)
    assert_state_close(sd, sub(z, "B/sd_fit16"), sd0, what="fit16")
    for k in sd0:
        if "embeddings" in k or k == "bias":
            np.testing.assert_array_equal(sd[k], sd0[k])          # ONN fit never trains the tables
    with pytest.raises(RuntimeError):
        m.fit(z["A/Xi1"].tolist(), z["A/Xv1"].tolist(), z["A/Y1"].tolist())   # B != batch_size, as in the reference


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_run_experiment_vs_reference(name, tag):
    z, meta = load_model_fixture(name, tag)
    m = build(name, meta, 1)
    m.load_state_dict(sub(z, "B/sd0"))
    t, acc, roc, cm = m.run_experiment(z["B/Xi"].tolist(), z["B/Xv"].tolist(), z["B/Y"].tolist())
    ref = meta["run_experiment"]
    assert isinstance(t, float) and t > 0
    assert cm == ref["confusion_matrix"]
    assert acc == pytest.approx(ref["accuracy"], rel=1e-12)
    assert roc["tpr"] == pytest.approx(ref["roc"]["tpr"], rel=1e-12)
    assert roc["fpr"] == pytest.approx(ref["roc"]["fpr"], rel=1e-12)
    sd, ref_sd = sd_np(m), sub(z, "B/sd_end")
    tot = bad = 0
    for k in ref_sd:
        d = np.abs(sd[k].astype(np.float64) - ref_sd[k])
        tot += d.size
        bad += int((d > 1e-5 * np.maximum(np.abs(ref_sd[k]), 1e-2)).sum())
    assert bad <= 0.002 * tot, f"{bad}/{tot} coordinates differ after 64 online steps"


@pytest.mark.parametrize("name", ["FMAdam", "DeepFMOnn"])
def test_pickle_and_state_dict_roundtrip(name):
    z, meta = load_model_fixture(name, "tiny4")
    m = build(name, meta, 1)
    m.load_state_dict(sub(z, "B/sd0"))
    m.fit([z["B/Xi"][0].tolist()], [z["B/Xv"][0].tolist()], [int(z["B/Y"][0])])
    buf = io.BytesIO()
    pickle.dump(m, buf)                                   # reference main_experiment.py:160-162
    m2 = pickle.loads(buf.getvalue())
    a, b = sd_np(m), sd_np(m2)
    assert set(a) == set(b)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    Xi, Xv = z["A/Xi1"].tolist(), z["A/Xv1"].tolist()
    o1, o2 = m.forward(Xi, Xv), m2.forward(Xi, Xv)
    o1, o2 = (o1[0], o2[0]) if isinstance(o1, tuple) else (o1, o2)
    np.testing.assert_array_equal(o1.cpu().numpy(), o2.cpu().numpy())
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bias": np.float32(0)})
    with pytest.raises(IndexError):
        m.predict([[99] * len(meta["feature_sizes"])], [[1.0] * len(meta["feature_sizes"])])


def test_pinned_stager_in_place_and_staged_give_the_same_batches():
    """PinnedBatchStager: int32 / fp32 arrays pinned where they are (every batch a DMA out of the dataset) against the staged path
    (typed copy into pinned buffers; taken for int64 indices) and against plain slicing: same batches, ragged tail, real values,
    more batches than slots; the arrays are usable and unpinned after close()."""
    from utils.data_preprocess import PinnedBatchStager
    rng = np.random.default_rng(2)
    N, F, B = 1000, 7, 96
    index = rng.integers(0, 50, size=(N, F)).astype(np.int32)
    label = (rng.uniform(size=N) < 0.5).astype(np.int64)
    value = rng.normal(size=(N, F)).astype(np.float32)
    for val in (None, value):
        a = PinnedBatchStager(index, label, B, value=val, depth=3, register_in_place=True)
        b = PinnedBatchStager(index.astype(np.int64), label, B, value=val, depth=2)
        assert a._src is not None and b._src is None
        n = 0
        for (ia, va, ya), (ib, vb, yb) in zip(a, b):
            lo, hi = n * B, min((n + 1) * B, N)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(ia.cpu().numpy(), index[lo:hi])
            np.testing.assert_array_equal(ib.cpu().numpy(), index[lo:hi])
            np.testing.assert_array_equal(ya.cpu().numpy(), label[lo:hi].astype(np.float32))
            np.testing.assert_array_equal(yb.cpu().numpy(), ya.cpu().numpy())
            if val is not None:
                np.testing.assert_array_equal(va.cpu().numpy(), value[lo:hi])
                np.testing.assert_array_equal(vb.cpu().numpy(), value[lo:hi])
            else:
                assert va is None and vb is None
            n += 1
        assert n == (N + B - 1) // B
        a.close()
        assert a._src is None and not a._registered
        assert int(index.sum()) > 0                      # still ordinary memory
    c = PinnedBatchStager(index, label, B)
    assert c._src is None and sum(1 for _ in c) == (N + B - 1) // B


@pytest.mark.parametrize("name", ["FMAdam", "DeepFMAdam"])
def test_tensor_inputs_equal_nested_lists(name):
    """The array / tensor fast path (no list conversion; CUDA int32 tensors used where they lie; Xv = None for all-ones)
    gives the same bits as the reference's nested lists: losses, parameters after update_embedding and fit, predictions.
    An out-of-range index in a tensor raises IndexError through the kernels' flag."""
    from utils.data_preprocess import PinnedBatchStager
    z, meta = load_model_fixture(name, "criteo39s")
    Xi, Y = z["A/Xi2"], z["A/Y2"]
    Xv = np.ones(Xi.shape, dtype=np.float32)              # the Criteo case: every value is 1 (None on the array path)
    models = []
    for _ in range(4):
        m = build(name, meta, meta["B2"])
        m.load_state_dict(sub(z, "A/sd0"))
        models.append(m)
    a, b, c, d = models
    la = a.update_embedding(Xi.tolist(), Xv.tolist(), Y.tolist())                       # the reference's convention
    lb = b.update_embedding(Xi.astype(np.int64), None, Y)                               # numpy, int64 like LongTensor
    lc = c.update_embedding(torch.from_numpy(Xi.astype(np.int32)).cuda(), torch.ones(Xi.shape, device="cuda"), torch.from_numpy(Y).cuda())
    (idx_d, xv_d, y_d), = list(PinnedBatchStager(Xi, Y, len(Y), feature_sizes=meta["feature_sizes"]))
    ld = d.update_embedding(idx_d, xv_d, y_d)                                            # pinned staging
    assert float(la) == float(lb) == float(lc) == float(ld)
    a.fit(Xi.tolist(), Xv.tolist(), Y.tolist())
    b.fit(Xi, None, Y)
    c.fit(torch.from_numpy(Xi).cuda(), None, torch.from_numpy(Y).cuda())
    d.fit(idx_d, None, y_d)
    ref = sd_np(a)
    for m in (b, c, d):
        got = sd_np(m)
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(a.predict(Xi.tolist(), Xv.tolist()), c.predict(torch.from_numpy(Xi).cuda(), None))
    bad = Xi.copy()
    bad[2, 5] = meta["feature_sizes"][5]
    with pytest.raises(IndexError):
        c.update_embedding(torch.from_numpy(bad).cuda(), None, torch.from_numpy(Y).cuda())
    c.strict_index_check = False                        # asynchronous: the flag is read when asked for
    c.update_embedding(torch.from_numpy(bad).cuda(), None, torch.from_numpy(Y).cuda())
    with pytest.raises(IndexError):
        c.check_index_flag()


def test_ftrl_pickle_resumes_bit_for_bit():
    """update_rule='ftrl': the pickle carries every coordinate's (z, n) and the bias pair, so N steps + pickle round trip
    + one more step equals the uninterrupted run bit for bit (a checkpoint of the derived weights alone would restart
    every per-coordinate learning rate from n = 0)."""
    z, meta = load_model_fixture("FMAdam", "criteo39s")
    ftrl = dict(alpha=0.1, beta=1.0, l1=0.001, l2=0.001)
    Xi, Xv, Y = z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist()

    def fresh():
        m = build("FMAdam", meta, 8, update_rule="ftrl", ftrl=ftrl)
        m.load_state_dict(sub(z, "A/sd0"))
        return m
    a, b = fresh(), fresh()
    for _ in range(3):
        a.update_embedding(Xi, Xv, Y)
        b.update_embedding(Xi, Xv, Y)
    buf = io.BytesIO()
    pickle.dump(b, buf)
    b = pickle.loads(buf.getvalue())
    sa, sb = a.ftrl_state_dict(), b.ftrl_state_dict()
    assert float(sa["nV"].abs().sum()) > 0                   # the state is not the n = 0 start
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    la, lb = a.update_embedding(Xi, Xv, Y), b.update_embedding(Xi, Xv, Y)
    assert float(la) == float(lb)
    sa, sb = a.ftrl_state_dict(), b.ftrl_state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    wa, wb = sd_np(a), sd_np(b)
    for k in wa:
        np.testing.assert_array_equal(wa[k], wb[k], err_msg=k)
    assert build("FMAdam", meta, 8).ftrl_state_dict() is None


@pytest.mark.parametrize("rule", ["sgd", "ftrl"])
def test_extension_rules_vs_oracle(rule):
    """SGD and FTRL-proximal are not in the shipped reference classes (parity unpinned): checked against the oracle."""
    z, meta = load_model_fixture("FMAdam", "criteo39s")
    sd0 = sub(z, "A/sd0")
    ftrl = dict(alpha=0.1, beta=1.0, l1=0.0, l2=0.001)
    m = build("FMAdam", meta, 8, update_rule=rule, ftrl=ftrl)
    m.load_state_dict(sd0)
    if rule == "ftrl":      # (z, n) start reproduces the given weights
        assert_state_close(sd_np(m), sd0, what="ftrl import")
    Xi, Xv, Y = z["A/Xi2"], z["A/Xv2"], z["A/Y2"]
    loss = m.update_embedding(Xi.tolist(), Xv.tolist(), Y.tolist())
    sizes = meta["feature_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    F = len(sizes)
    V = np.concatenate([sd0[f"second_order_embeddings.{i}.weight"] for i in range(F)]).astype(np.float32)
    w = np.concatenate([sd0[f"first_order_embeddings.{i}.weight"][:, 0] for i in range(F)]).astype(np.float32)
    rows = Xi + offs[:-1][None, :]
    sd = sd_np(m)
    Vh = np.concatenate([sd[f"second_order_embeddings.{i}.weight"] for i in range(F)])
    wh = np.concatenate([sd[f"first_order_embeddings.{i}.weight"][:, 0] for i in range(F)])
    # neither rule is in the reference: both are checked against the float64 evaluation of the step (pinned to the paper
    # by tests/test_ftrl_pin.py) at north_star's 1e-5 relative + the fp32 rounding floor of each element
    if rule == "sgd":
        st = dict(V=V.copy(), w=w.copy(), bias=np.float32(sd0["bias"]))
        ref = orc.flat_fm_step_f64(st, rows, Xv, Y, "logits", "sgd", dict(lr=meta["n"]))
        u, new, fl = ref["urows"], ref["new"], ref["floor"]
        assert_within_f64(float(loss.cpu().data), ref["loss"], fl["loss"], "loss")
        assert_within_f64(Vh[u], new["V"][u], fl["V"], "V")
        assert_within_f64(wh[u], new["w"][u], fl["w"], "w")
        assert_within_f64(sd["bias"], new["bias"], fl["bias"], "bias")
        mask = np.ones(len(V), dtype=bool)
        mask[u] = False
        np.testing.assert_array_equal(Vh[mask], V[mask])
        return
    st = dict(zV=orc.ftrl_z_for_weight(V, **ftrl), nV=np.zeros_like(V), zw=orc.ftrl_z_for_weight(w, **ftrl),
              nw=np.zeros_like(w), zb=orc.ftrl_z_for_weight(np.float32(sd0["bias"]), **ftrl), nb=np.float32(0))
    ref = orc.flat_fm_step_f64(st, rows, Xv, Y, "logits", "ftrl", ftrl)
    u, new, fl = ref["urows"], ref["new"], ref["floor"]
    assert_within_f64(float(loss.cpu().data), ref["loss"], fl["loss"], "loss")
    hs = m.ftrl_state_dict()
    assert_ftrl_step_within_f64(dict(zV=hs["zV"].numpy(), nV=hs["nV"].numpy(), zw=hs["zw"].numpy(), nw=hs["nw"].numpy(),
                                     zb=float(hs["bias_zn"][0]), nb=float(hs["bias_zn"][1])), ref)
    # the weights the class exposes (state_dict) are the ones derived from that state:  w = -(z - sgn z l1) / D,
    # D = (beta + sqrt n) / alpha + l2, so  dw = dz / D + |w| dn / (2 alpha sqrt(n) D)  on top of its own few roundings
    for got, zk, nk in ((Vh, "zV", "nV"), (wh, "zw", "nw")):
        z1, n1 = new[zk][u], new[nk][u]
        want = orc.ftrl_weight(z1, n1, dtype=np.float64, **ftrl)
        D = (ftrl["beta"] + np.sqrt(n1)) / ftrl["alpha"] + ftrl["l2"]
        floor = fl[zk] / D + np.abs(want) * fl[nk] / (2 * ftrl["alpha"] * np.sqrt(np.maximum(n1, 1e-30)) * D) + 4 * orc.EPS32 * np.abs(want)
        assert_within_f64(got[u], want, floor, "derived " + zk[1:])


@pytest.mark.parametrize("name", ["DeepFMAdam", "NFMAdam", "DeepFMOnn", "NFMOnn"])
def test_fused_mlp_kernel_equals_pytorch_path(name, monkeypatch):
    """Shapes within the fused MLP kernel's limits go through k_mlp_small, larger ones through PyTorch autograd: both
    must give the same step (checked by forcing the PyTorch path on the same inputs)."""
    import fmx
    z, meta = load_model_fixture(name, "criteo39s")
    B = meta["B2"]
    Xi, Xv, Y = z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist()
    results = []
    for force_torch in (False, True):
        m = build(name, meta, B)
        m.load_state_dict(sub(z, "A/sd0"))
        if force_torch:
            monkeypatch.setattr(fmx.FMEngine, "mlp_fits", staticmethod(lambda *a, **k: False))
        out = m.forward(Xi, Xv)
        out = out[1] if isinstance(out, tuple) else out
        m.fit(Xi, Xv, Y)
        results.append((out.cpu().numpy(), sd_np(m)))
        monkeypatch.undo()
    assert_close(results[0][0], results[1][0], 1e-5, 1e-6, "forward")
    assert_state_close(results[0][1], results[1][1], sub(z, "A/sd0"), what="fit: kernel vs pytorch")


@pytest.mark.parametrize("name", ["DeepFMAdam", "NFMAdam"])
def test_mfma_mlp_section_equals_pytorch_path_in_fit(name, monkeypatch):
    """Batches beyond the one-workgroup kernel go through fmx_mlp_section (fp32 MFMA GEMMs) + the closed form of the
    fresh-Adam step; with native_mlp = False through PyTorch autograd + a literal torch.optim.Adam: same step."""
    import fmx
    z, meta = load_model_fixture(name, "criteo39s")
    B = meta["B2"]
    Xi, Xv, Y = z["A/Xi2"].tolist(), z["A/Xv2"].tolist(), z["A/Y2"].tolist()
    monkeypatch.setattr(fmx.FMEngine, "mlp_fits", staticmethod(lambda *a, **k: False))
    results = []
    for native in (True, False):
        m = build(name, meta, B)
        m.load_state_dict(sub(z, "A/sd0"))
        m.native_mlp = native
        m.fit(Xi, Xv, Y)
        results.append(sd_np(m))
    assert_state_close(results[0], results[1], sub(z, "A/sd0"), what="fit: mfma section vs pytorch")


@pytest.mark.parametrize("name", ["DeepFMAdam", "NFMAdam"])
@pytest.mark.parametrize("native", [True, False])
def test_deep_trainer_step_vs_oracle(name, native):
    """fmx.DeepFMTrainer (device tensors, world size 1; MLP section through fmx_mlp_section or through PyTorch autograd)
    against the oracle's class step under SGD."""
    import fmx
    import torch.nn as nn
    z, meta = load_model_fixture(name, "criteo39s")
    sd0 = sub(z, "A/sd0")
    sizes, k, L, H, lr = meta["feature_sizes"], meta["k"], meta["L"], meta["H"], meta["n"]
    table = fmx.FlatTable(sizes, k, layout="weights")
    F_ = len(sizes)
    table.load_reference([sd0[f"first_order_embeddings.{i}.weight"] for i in range(F_)],
                         [sd0[f"second_order_embeddings.{i}.weight"] for i in range(F_)])
    table.set_bias_weight(float(sd0["bias"]))
    eng = fmx.FMEngine(table, max_batch=64)
    layers = [nn.Linear(k if j == 0 else H, H).cuda() for j in range(L)]
    with torch.no_grad():
        for j, layer in enumerate(layers):
            layer.weight.copy_(torch.from_numpy(sd0[f"hidden_layers.{j}.weight"]))
            layer.bias.copy_(torch.from_numpy(sd0[f"hidden_layers.{j}.bias"]))
    hyper = fmx.Hyper(lr=lr)
    loss_kind = orc.LOSS_KIND[(name, "fit")]
    tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, hyper, "sgd"), layers, k, table.kp, mlp_lr=lr,
                           fm_term=(name == "DeepFMAdam"), loss=loss_kind, native_mlp=native)
    assert tr.native == native
    Xi, Y = z["A/Xi1"], z["A/Y1"]                     # Xv == 1
    idx_d, _, y_d = eng.to_device(Xi.astype(np.int32), None, Y)
    tr.step(idx_d, y_d)
    torch.cuda.synchronize()
    om = orc.OracleModel(name, sd0, update_rule="sgd")
    om.fit(Xi, z["A/Xv1"], Y)
    ref = om.state_dict()
    first, second = table.export_reference()
    got = {"bias": table.bias_weight().cpu().numpy()}
    for i in range(F_):
        got[f"first_order_embeddings.{i}.weight"] = first[i].numpy()
        got[f"second_order_embeddings.{i}.weight"] = second[i].numpy()
    for j, layer in enumerate(layers):
        got[f"hidden_layers.{j}.weight"] = layer.weight.detach().cpu().numpy()
        got[f"hidden_layers.{j}.bias"] = layer.bias.detach().cpu().numpy()
    ref = {kk: v for kk, v in ref.items() if kk in got}
    assert_state_close(got, ref, {kk: sd0[kk] for kk in ref}, what=f"{name} sgd step")


@pytest.mark.parametrize("fm_term", [True, False])
def test_deepfm_stream_equals_trainer_steps(fm_term):
    """fmx_deepfm_stream (DeepFMTrainer.prepare_stream: the steps of a pool of batches issued from one foreign call, the sorts in
    groups on the side stream) against DeepFMTrainer.step on the same batches, DeepFM and NFM (fm_term=False: base = first-order +
    bias): tables, MLP parameters and per-step losses identical bits.  B = 256 takes the side-stream path, 13 steps over a pool of 5 batches cross a sort group and wrap the pool;
    the step path itself is pinned against the oracle by test_deep_trainer_step_vs_oracle."""
    import fmx
    import torch.nn as nn
    sizes, k, H, L, B, n_pool, n_steps, lr = [50, 7, 300, 2, 1200, 33], 16, 256, 3, 256, 5, 13, 0.01
    rng = np.random.default_rng(3)
    idx = np.stack([np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1) for _ in range(n_pool)]).astype(np.int32)
    y = (rng.uniform(size=(n_pool, B)) < 0.4).astype(np.float32)
    results = []
    for mode in ("steps", "stream"):
        torch.manual_seed(5)
        table = fmx.FlatTable(sizes, k, layout="weights")
        g = torch.Generator(device="cuda").manual_seed(9)
        table.rows[:, :k + 1] = torch.randn((table.n_rows, k + 1), generator=g, device="cuda") * 0.1
        eng = fmx.FMEngine(table, max_batch=B)
        layers = [nn.Linear(k if j == 0 else H, H).cuda() for j in range(L)]
        table.set_bias_weight(0.3)
        tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, fmx.Hyper(lr=lr), "sgd"), layers, k, table.kp, mlp_lr=lr, fm_term=fm_term)
        assert tr.native
        idx_d, y_d = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
        losses = torch.zeros(n_steps, device="cuda")
        if mode == "steps":
            for s in range(n_steps):
                losses[s] = tr.step(idx_d[s % n_pool], y_d[s % n_pool])
        else:
            run = tr.prepare_stream(idx_d, y_d, loss_out=losses)
            run(4)                                    # two calls: the second starts in the middle of the pool
            run2 = tr.prepare_stream(torch.roll(idx_d, -4, 0).contiguous(), torch.roll(y_d, -4, 0).contiguous(), loss_out=losses[4:])
            run2(n_steps - 4)
        torch.cuda.synchronize()
        eng.check_error_flag()
        results.append((table.rows.cpu().numpy().copy(), tr.flat.cpu().numpy().copy(), losses.cpu().numpy().copy()))
    for a, b, what in zip(results[0], results[1], ("tables", "MLP parameters", "losses")):
        assert np.array_equal(a, b), what
    assert np.all(np.isfinite(results[0][2])) and results[0][2].std() > 0


def test_deepfm_stream_rejects_what_it_does_not_take():
    """fmx_deepfm_stream / DeepFMTrainer.prepare_stream: NFM on FTRL-layout tables, a too-small workspace, per-sample records
    (sample_ld != 0) and null arguments end in a status code (or a ValueError before the call), never in a launch."""
    import ctypes as C
    import fmx
    import torch.nn as nn
    sizes, k, H, L, B = [50, 7, 300], 16, 256, 2, 64
    idx = torch.zeros((2, B, len(sizes)), dtype=torch.int32, device="cuda")
    y = torch.zeros((2, B), device="cuda")
    table = fmx.FlatTable(sizes, k, layout="ftrl", ftrl=dict(alpha=0.05, beta=1.0, l1=0.0, l2=0.0))
    eng = fmx.FMEngine(table, max_batch=B)
    layers = [nn.Linear(k if j == 0 else H, H).cuda() for j in range(L)]
    tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, fmx.Hyper(lr=0.01), "ftrl"), layers, k, table.kp, mlp_lr=0.01, fm_term=False)
    with pytest.raises(ValueError):
        tr.prepare_stream(idx, y)                                    # NFM needs the weights layout
    lib = fmx._lib.load()
    wt = fmx.FlatTable(sizes, k, layout="weights")
    e2 = fmx.FMEngine(wt, max_batch=B)
    tr2 = fmx.DeepFMTrainer(fmx.HipDeepBackend(e2, fmx.Hyper(lr=0.01), "sgd"), layers, k, wt.kp, mlp_lr=0.01)
    run = tr2.prepare_stream(idx, y)
    run(0)                                                           # no steps: no launch, no error
    out = e2._fwd_out(want_first=False, want_bi=True)
    m = e2._mlp_struct(tr2.flat, k, H, L)
    args = lambda **kw: [kw.get("table", wt.c_struct()), fmx.Hyper(lr=0.01).ref(), fmx._lib.RULES["sgd"], C.byref(m), fmx._lib.LOSSES["logits"], 1,
                         idx.data_ptr(), y.data_ptr(), 2, B, 1.0 / B, 1, e2.workspace.data_ptr(), kw.get("ws_bytes", e2._ws_bytes()),
                         e2._mlp_ws.data_ptr(), C.byref(kw.get("out", out)), kw.get("dz", e2._mlp_dz.data_ptr()), e2._mlp_gbi.data_ptr(),
                         tr2.gflat.data_ptr(), 0.01, None, None]
    assert lib.fmx_deepfm_stream(*args(ws_bytes=64)) == fmx._lib.ERR_SHAPE
    assert lib.fmx_deepfm_stream(*args(dz=None)) == fmx._lib.ERR_ARG
    rec = e2._fwd_out(want_first=False, want_bi=True)
    rec.sample_ld = 24
    assert lib.fmx_deepfm_stream(*args(out=rec)) == fmx._lib.ERR_ARG
    torch.cuda.synchronize()
    e2.check_error_flag()


def test_deepfm_stream_full_size_equals_trainer_steps_and_is_deterministic():
    """BASELINE configs[3] at its full size through fmx_deepfm_stream (Criteo vocabulary R = 1,006,628, k = 16, 3 x 256, B = 4096, SGD):
    19 steps over a pool of 6 batches -- two sort groups, the pool wrapped three times -- against DeepFMTrainer.step on the same
    batches (whose single step test_deep_trainer_full_size_step_vs_oracle pins against the oracle): tables, MLP parameters and losses
    identical bits; a second run of the loop from the same state gives the same bits again; rows no batch touches do not change."""
    import fmx
    import torch.nn as nn
    sizes = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887, 632, 3, 41738,
             5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86, 44549]
    k, L, H, B, lr, n_pool, n_steps = 16, 3, 256, 4096, 1e-3, 6, 19
    rng = np.random.default_rng(21)
    idx = np.stack([np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1) for _ in range(n_pool)]).astype(np.int32)
    y = (rng.uniform(size=(n_pool, B)) < 0.3).astype(np.float32)
    results = []
    for mode in ("steps", "stream", "stream"):
        torch.manual_seed(5)
        table = fmx.FlatTable(sizes, k, layout="weights")
        g = torch.Generator(device="cuda").manual_seed(9)
        table.rows[:, :k + 1] = torch.randn((table.n_rows, k + 1), generator=g, device="cuda") * 0.1
        rows0 = table.rows.clone()
        eng = fmx.FMEngine(table, max_batch=B)
        layers = [nn.Linear(k if j == 0 else H, H).cuda() for j in range(L)]
        tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, fmx.Hyper(lr=lr), "sgd"), layers, k, table.kp, mlp_lr=lr)
        idx_d, y_d = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
        losses = torch.zeros(n_steps, device="cuda")
        if mode == "steps":
            for s in range(n_steps):
                losses[s] = tr.step(idx_d[s % n_pool], y_d[s % n_pool])
        else:
            tr.prepare_stream(idx_d, y_d, loss_out=losses)(n_steps)
        torch.cuda.synchronize()
        eng.check_error_flag()
        results.append((table.rows.cpu().numpy().copy(), tr.flat.cpu().numpy().copy(), losses.cpu().numpy().copy()))
    for other in results[1:]:
        for a, b, what in zip(results[0], other, ("tables", "MLP parameters", "losses")):
            assert np.array_equal(a, b), what
    touched = np.zeros(table.n_rows, dtype=bool)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for f in range(len(sizes)):
        touched[offs[f] + np.unique(idx[:, :, f])] = True
    changed = (results[0][0] != rows0.cpu().numpy()).any(axis=1)
    assert not changed[~touched].any() and changed[touched].mean() > 0.99
    assert np.all(np.isfinite(results[0][2])) and results[0][2][-1] < results[0][2][0]


def test_deep_trainer_full_size_step_vs_oracle():
    """BASELINE configs[3] END TO END at its full size: one online DeepFM step (Criteo vocabulary R = 1,006,628, k = 16,
    3 x 256 relu MLP, B = 4096, SGD lr 1e-3) through fmx.DeepFMTrainer -- sort, forward, the MLP section (k_mlp_chain +
    wgrad + reduce), the row-reduced table update with dL/dbi, SGD on the MLP -- against the oracle's class step from the
    same parameters.  (The small-fixture form of this test is test_deep_trainer_step_vs_oracle.)"""
    import fmx
    import torch.nn as nn
    sizes = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887, 632, 3, 41738,
             5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86, 44549]
    k, L, H, B, lr = 16, 3, 256, 4096, 1e-3
    rng = np.random.default_rng(11)
    sd0 = {"bias": np.float32(0.1), "n": np.float32(lr)}
    for i, sz in enumerate(sizes):
        sd0[f"first_order_embeddings.{i}.weight"] = (rng.normal(size=(sz, 1)) * 0.1).astype(np.float32)
        sd0[f"second_order_embeddings.{i}.weight"] = (rng.normal(size=(sz, k)) * 0.1).astype(np.float32)
    for j in range(L):
        fan = k if j == 0 else H
        sd0[f"hidden_layers.{j}.weight"] = (rng.normal(size=(H, fan)) / np.sqrt(fan)).astype(np.float32)
        sd0[f"hidden_layers.{j}.bias"] = (rng.normal(size=H) * 0.05).astype(np.float32)
    Xi = np.stack([rng.integers(0, sz, size=B) for sz in sizes], axis=1)
    Y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    F_ = len(sizes)
    table = fmx.FlatTable(sizes, k, layout="weights")
    table.load_reference([sd0[f"first_order_embeddings.{i}.weight"] for i in range(F_)],
                         [sd0[f"second_order_embeddings.{i}.weight"] for i in range(F_)])
    table.set_bias_weight(float(sd0["bias"]))
    eng = fmx.FMEngine(table, max_batch=B)
    layers = [nn.Linear(k if j == 0 else H, H).cuda() for j in range(L)]
    with torch.no_grad():
        for j, layer in enumerate(layers):
            layer.weight.copy_(torch.from_numpy(sd0[f"hidden_layers.{j}.weight"]))
            layer.bias.copy_(torch.from_numpy(sd0[f"hidden_layers.{j}.bias"]))
    tr = fmx.DeepFMTrainer(fmx.HipDeepBackend(eng, fmx.Hyper(lr=lr), "sgd"), layers, k, table.kp, mlp_lr=lr, fm_term=True,
                           loss=orc.LOSS_KIND[("DeepFMAdam", "fit")], native_mlp=True)
    assert tr.native
    idx_d, _, y_d = eng.to_device(Xi.astype(np.int32), None, Y)
    tr.step(idx_d, y_d)
    torch.cuda.synchronize()
    eng.check_error_flag()
    om = orc.OracleModel("DeepFMAdam", sd0, update_rule="sgd")
    om.fit(Xi, np.ones(Xi.shape, dtype=np.float32), Y)
    ref = om.state_dict()
    first, second = table.export_reference()
    got = {"bias": table.bias_weight().cpu().numpy()}
    for i in range(F_):
        got[f"first_order_embeddings.{i}.weight"] = first[i].numpy()
        got[f"second_order_embeddings.{i}.weight"] = second[i].numpy()
    for j, layer in enumerate(layers):
        got[f"hidden_layers.{j}.weight"] = layer.weight.detach().cpu().numpy()
        got[f"hidden_layers.{j}.bias"] = layer.bias.detach().cpu().numpy()
    ref = {kk: v for kk, v in ref.items() if kk in got}
    assert_state_close(got, ref, {kk: sd0[kk] for kk in ref}, what="DeepFM 3x256 full-size sgd step")
    # the step did move every kind of parameter
    assert np.abs(got["hidden_layers.1.weight"] - sd0["hidden_layers.1.weight"]).max() > 0
    assert np.abs(got["second_order_embeddings.15.weight"] - sd0["second_order_embeddings.15.weight"]).max() > 0


def test_config4_nfm_hedge_on_a_frappe_shaped_10m_row_table():
    """BASELINE.json configs[4] as a parity case: online NFM + Hedge backprop (reference nfm_onn.py:111-156) on
    Frappe-shaped input -- 10 one-hot fields over 10 M embedding rows, k = 16, L = 3 -- in the reference's online protocol
    (predict, then fit, one sample at a time) against the oracle started from the same parameters.  The tables are not
    trained by Hedge (the reference's behaviour), so parity is on predictions, alpha and the hidden layers."""
    from models.models_online_deep.nfm_onn import NFMOnn
    torch.manual_seed(4)
    sizes = [1_000_000] * 10
    k, L, H, n_samples = 16, 3, 32, 120
    m = NFMOnn(sizes, embedding_size=k, num_hidden_layers=L, neuron_per_hidden_layer=H, batch_size=1, n=0.01)
    sd0 = sd_np(m)
    rng = np.random.default_rng(5)
    Xi = np.stack([np.minimum(rng.zipf(1.2, size=n_samples) - 1, s - 1) for s in sizes], axis=1)   # skewed ids, heavy head
    Xi[::7] = np.stack([rng.integers(0, s, size=len(Xi[::7])) for s in sizes], axis=1)               # and the long tail
    Xv = np.ones_like(Xi, dtype=np.float32)
    Y = (rng.uniform(size=n_samples) < 0.4).astype(np.float32)
    om = orc.OracleModel("NFMOnn", {kk: v.copy() for kk, v in sd0.items()}, batch_size=1)
    # every STEP of the trajectory at north_star's 1e-5: before each sample two oracles are restarted from the HIP model's own
    # hidden layers and alpha (the tables are not trained by Hedge: copied once) -- the fp32 oracle, and the same source evaluated
    # in FLOAT64 (helpers.oracle_float64).  The float64 value is the reference wherever fp32 does not SATURATE: with N(0, 1)
    # embeddings a layer's sigmoid is exactly 1.0f for many samples, where the reference's fp32 arithmetic (BCELoss's clamp,
    # autograd's (p - y) / max(p (1 - p), 1e-12) p (1 - p) = 0) is the specified behaviour and float64 is a different function;
    # there the restarted fp32 oracle is the reference.
    o64 = oracle_float64()
    om64 = o64.OracleModel("NFMOnn", {kk: v.copy() for kk, v in sd0.items()}, batch_size=1)
    om32 = orc.OracleModel("NFMOnn", {kk: v.copy() for kk, v in sd0.items()}, batch_size=1)
    eps32 = float(np.finfo(np.float32).eps)

    def small():    # alpha and the hidden layers only (state_dict() would export the 10 M-row tables at every step)
        out = {"alpha": m.alpha.detach().cpu().numpy().copy()}
        for j, layer in enumerate(m.hidden_layers):
            out[f"hidden_layers.{j}.weight"] = layer.weight.detach().cpu().numpy().copy()
            out[f"hidden_layers.{j}.bias"] = layer.bias.detach().cpu().numpy().copy()
        return out

    def restart(o, st, dt):
        o.alpha = st["alpha"].astype(dt)
        o.hidden = [[st[f"hidden_layers.{j}.weight"].astype(dt), st[f"hidden_layers.{j}.bias"].astype(dt)] for j in range(L)]

    def state(o):
        out = {"alpha": np.asarray(o.alpha, np.float64)}
        for j in range(L):
            out[f"hidden_layers.{j}.weight"], out[f"hidden_layers.{j}.bias"] = (np.asarray(a, np.float64) for a in o.hidden[j])
        return out
    preds_h, preds_o = [], []
    n_f64 = 0
    for i in range(n_samples):
        xi, xv = Xi[i].reshape(1, -1, 1).tolist(), Xv[i].reshape(1, -1).tolist()
        preds_h.append(bool(np.asarray(m.predict(xi, xv)).reshape(-1)[0]))
        preds_o.append(bool(np.asarray(om.predict(Xi[i], Xv[i])).reshape(-1)[0]))
        before = small()
        restart(om32, before, np.float32)
        restart(om64, before, np.float64)
        m.fit(xi, xv, [float(Y[i])])
        om.fit([Xi[i]], [Xv[i]], [Y[i]])
        om32.fit([Xi[i]], [Xv[i]], [Y[i]])
        om64.fit([Xi[i]], [Xv[i]], [Y[i]])
        after, r32, r64 = small(), state(om32), state(om64)
        # |HIP - ref| <= 1e-5 |ref's step| + floor on the STEP; floor = a few ulps of the tensor's largest element (every updated
        # element is rounded at that scale once, and its gradient is a sum whose terms are rounded there)
        unsat = True
        for key in r32:
            b0 = before[key].astype(np.float64)
            tol32 = 1e-5 * np.abs(r32[key] - b0) + 8 * eps32 * float(np.abs(r32[key]).max())
            unsat = unsat and bool((np.abs(r64[key] - r32[key]) <= tol32).all())
        for key in r32:
            b0 = before[key].astype(np.float64)
            d_got = after[key].astype(np.float64) - b0
            for ref_state, what in ((r32, "fp32 oracle"),) + (((r64, "float64"),) if unsat else ()):
                d_ref = ref_state[key] - b0
                tol = 1e-5 * np.abs(d_ref) + 8 * eps32 * float(np.abs(ref_state[key]).max())
                err = np.abs(d_got - d_ref)
                assert (err <= tol).all(), (f"step {i} {key} vs {what}: {int((err > tol).sum())}/{err.size} beyond 1e-5 rel + fp32 floor "
                                            f"on the step; worst err/tol {float((err / tol).max()):.2f}, max |step| {float(np.abs(d_ref).max()):.3e}")
        n_f64 += unsat
    assert n_f64 >= n_samples // 10, f"only {n_f64} of {n_samples} steps were unsaturated enough for the float64 reference"
    assert preds_h == preds_o
    got, ref = sd_np(m), om.state_dict()
    # the END of the 120-step trajectory against the fp32 oracle's own trajectory: two fp32 evaluations drift apart by their
    # rounding differences step after step (each step within 1e-5 of float64 above): 1e-4 after 120 dependent steps
    assert_close(got["alpha"], ref["alpha"], 1e-4, 1e-6, "alpha")
    for j in range(L):
        for part in ("weight", "bias"):
            key = f"hidden_layers.{j}.{part}"
            assert_close(got[key], ref[key], 1e-4, 1e-6 * np.abs(ref[key]).max(), key)
    for i in (0, 9):        # Hedge leaves the tables alone
        np.testing.assert_array_equal(got[f"second_order_embeddings.{i}.weight"], sd0[f"second_order_embeddings.{i}.weight"])


def test_driver_script_flow_end_to_end(tmp_path, capsys):
    """The reference's driver (main_experiment.py:60-162) on synthetic Criteo-shaped CSVs in its own file format: reading,
    batch schedule, pre-training through update_embedding, the online run_experiment loop over the five classes, the
    result pickle and the model pickles.  Small sizes; checks the flow and the artefacts, not accuracy."""
    import _experiment
    res = _experiment.run("Iteration", argv=["--synthetic", "--num-batchdata", "48", "--num-batch", "4", "--pretrain-iters", "3",
                                              "--out", str(tmp_path) + "/"])
    names = ["DeepFMAdam", "DeepFMOnn", "NFMAdam", "NFMOnn", "FMAdam"]
    assert sorted(res["accuracy"].keys()) == sorted(names)
    for nme in names:
        assert len(res["accuracy"][nme]) == 4 and all(0.0 <= a <= 100.0 for a in res["accuracy"][nme])
        assert all(set(r.keys()) == {"tpr", "fpr"} for r in res["roc"][nme])
    out = capsys.readouterr().out
    assert "i th iter 0 , loss :" in out and "confusion matrix :" in out
    models = sorted(p.name for p in (tmp_path / "save_model").iterdir())
    assert models == sorted(f"{nme}_Iteration.pickle" for nme in names)
    with open(tmp_path / "save_model" / "FMAdam_Iteration.pickle", "rb") as f:
        m = pickle.load(f)                                # our own file, written a moment ago
    assert str(m).startswith("FMAdam-")
    assert len(list((tmp_path / "save_log").iterdir())) == 1


@pytest.mark.parametrize("name", CLASS_NAMES)
def test_device_online_loop_equals_per_sample_loop(name):
    """run_experiment on the device (fmx_fm_online_run / fmx_online_run_mlp) against the literal per-sample loop of
    predict() + fit() calls: the same 4-tuple (minus the time) and bit-identical parameters."""
    z, meta = load_model_fixture(name, "criteo39s")
    Xi, Xv, Y = z["B/Xi"].tolist(), z["B/Xv"].tolist(), z["B/Y"].tolist()
    res = []
    for on_device in (True, False):
        m = build(name, meta, 1)
        m.load_state_dict(sub(z, "B/sd0"))
        m.device_online_loop = on_device
        assert m._device_loop_ok() == on_device
        t, acc, roc, cm = m.run_experiment(Xi, Xv, Y)
        res.append((acc, roc, cm, sd_np(m)))
    assert res[0][:3] == res[1][:3]
    for k in res[1][3]:
        np.testing.assert_array_equal(res[0][3][k], res[1][3][k], err_msg=k)


@pytest.mark.parametrize("name", ["DeepFMAdam", "NFMAdam", "DeepFMOnn", "NFMOnn"])
def test_batch_forward_gemm_path_equals_pytorch_path(name):
    """forward() / predict() on a whole batch (beyond the one-workgroup kernel's 16 samples) runs the MFMA forward chain;
    with native_mlp = False the PyTorch layers: same values."""
    z, meta = load_model_fixture(name, "criteo39s")
    rng = np.random.default_rng(3)
    B = 300
    Xi = np.stack([rng.integers(0, s, size=B) for s in meta["feature_sizes"]], axis=1).reshape(B, -1, 1).tolist()
    Xv = np.ones((B, len(meta["feature_sizes"])), dtype=np.float32).tolist()
    outs = []
    for native in (True, False):
        m = build(name, meta, 1)
        m.load_state_dict(sub(z, "A/sd0"))
        m.native_mlp = native
        out = m.forward(Xi, Xv)
        outs.append([o.cpu().numpy() for o in out] if isinstance(out, tuple) else [out.cpu().numpy()])
        pred = m.predict(Xi, Xv)
        assert pred.shape == (B,)
    for a, b in zip(outs[0], outs[1]):
        assert_close(a, b, 1e-5, 2e-6 * max(np.abs(b).max(), 1.0), "forward")


@pytest.mark.parametrize("name", ["DeepFMOnn", "NFMOnn"])
def test_hedge_mini_batch_gemm_path_equals_pytorch_path(name):
    """The ONN classes' fit() on a batch beyond the one-workgroup kernel's 16 samples: Hedge backprop on the MFMA GEMMs
    (fmx_mlp_hedge_section) against the PyTorch autograd form of the same step (native_mlp = False).  Plain SGD on the
    hidden layers, so the comparison is smooth: 1e-4 on the lr-sized deltas."""
    z, meta = load_model_fixture(name, "criteo39s")
    rng = np.random.default_rng(8)
    B = 200
    Xi = np.stack([rng.integers(0, s, size=B) for s in meta["feature_sizes"]], axis=1).reshape(B, -1, 1).tolist()
    Xv = np.ones((B, len(meta["feature_sizes"])), dtype=np.float32).tolist()
    Y = (rng.uniform(size=B) < 0.4).astype(np.float32).tolist()
    sd0 = sub(z, "A/sd0")
    res = []
    for native in (True, False):
        m = build(name, meta, B)
        m.load_state_dict(sd0)
        m.native_mlp = native
        for _ in range(3):
            m.fit(Xi, Xv, Y)
        res.append(sd_np(m))
    assert_state_close(res[0], res[1], sd0, what="hedge fit: mfma vs pytorch")
    assert not np.array_equal(res[0]["alpha"], sd0["alpha"])
    for i in (0, 5):
        np.testing.assert_array_equal(res[0][f"second_order_embeddings.{i}.weight"], sd0[f"second_order_embeddings.{i}.weight"])


def test_device_online_loop_with_ftrl_tables_under_hedge():
    """An ONN class over an FTRL-layout table (update_rule="ftrl" for its update_embedding): Hedge's run_experiment loop on
    the device reads the cached weights of that layout.  One workgroup walking the stream == the per-sample launches, bit
    for bit; the per-sample Python loop derives NFM's bias weight with torch ops instead of the kernels' 1-ulp rcp / sqrt,
    so it is compared at 1e-5."""
    import fmx
    lib = fmx._lib.load()
    z, meta = load_model_fixture("NFMOnn", "criteo39s")
    Xi, Xv, Y = z["B/Xi"].tolist(), z["B/Xv"].tolist(), z["B/Y"].tolist()
    res = []
    for on_device, persistent in ((True, 1), (True, 0), (False, 1)):
        prev = lib.fmx_set_option(b"online_persistent", persistent)
        try:
            m = build("NFMOnn", meta, 1, update_rule="ftrl", ftrl=dict(alpha=0.05, beta=1.0, l1=0.0, l2=0.0))
            m.load_state_dict(sub(z, "B/sd0"))
            m.device_online_loop = on_device
            assert m._device_loop_ok() == on_device
            t, acc, roc, cm = m.run_experiment(Xi, Xv, Y)
            res.append((acc, roc, cm, sd_np(m)))
        finally:
            lib.fmx_set_option(b"online_persistent", prev)
    assert res[0][:3] == res[1][:3] == res[2][:3]
    for k in res[1][3]:
        np.testing.assert_array_equal(res[0][3][k], res[1][3][k], err_msg=k)
        assert_close(res[0][3][k], res[2][3][k], 1e-5, 1e-7, k)
