"""Pins oracle/fm_oracle.py against the golden fixtures generated from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import fm_oracle as orc
from helpers import CLASS_NAMES, TAGS, assert_close, assert_state_close, load_model_fixture, sub

RT = 1e-5


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_forward_pieces(name, tag):
    z, meta = load_model_fixture(name, tag)
    m = orc.OracleModel(name, sub(z, "A/sd0"), batch_size=meta["B2"])
    for j in (1, 2):
        Xi, Xv = z[f"A/Xi{j}"], z[f"A/Xv{j}"]
        if name != "FMAdam":
            assert_close(m.first_order(Xi, Xv), z[f"A/first_order{j}"], RT, 1e-7, "first_order")
            assert_close(m.second_order(Xi, Xv), z[f"A/second_order{j}"], RT, 2e-6, "second_order")
            assert_close(m.forward_fm(Xi, Xv), z[f"A/forward_fm{j}"], RT, 2e-6, "forward_fm")
        out = m.forward(Xi, Xv)
        if isinstance(out, tuple):
            assert_close(out[0], z[f"A/forward{j}"], RT, 1e-6, "forward")
            assert_close(out[1], z[f"A/forward_layers{j}"], RT, 1e-6, "forward_layers")
        else:
            assert_close(out, z[f"A/forward{j}"], RT, 2e-6, "forward")
        np.testing.assert_array_equal(np.asarray(m.predict(Xi, Xv)).reshape(-1), z[f"A/predict{j}"].reshape(-1))


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_update_embedding_and_fit(name, tag):
    z, meta = load_model_fixture(name, tag)
    sd0 = sub(z, "A/sd0")
    m = orc.OracleModel(name, sd0, batch_size=meta["B2"])
    loss = m.update_embedding(z["A/Xi1"], z["A/Xv1"], z["A/Y1"])
    assert_close(loss, z["A/loss_update_embedding"], RT, 0, "loss1")
    sd1 = sub(z, "A/sd1")
    assert_state_close(m.state_dict(), sd1, sd0, what="sd1")
    loss = m.update_embedding(z["A/Xi2"], z["A/Xv2"], z["A/Y2"])
    assert_close(loss, z["A/loss_update_embedding2"], RT, 0, "loss2")
    sd2 = sub(z, "A/sd2")
    assert_state_close(m.state_dict(), sd2, sd1, what="sd2")
    m.fit(z["A/Xi2"], z["A/Xv2"], z["A/Y2"])
    assert_state_close(m.state_dict(), sub(z, "A/sd3"), sd2, what="sd3")


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", ["DeepFMOnn", "NFMOnn"])
def test_hedge_trajectory(name, tag):
    z, meta = load_model_fixture(name, tag)
    sd0 = sub(z, "B/sd0")
    m = orc.OracleModel(name, sd0, batch_size=1)
    traj = [m.alpha.copy()]
    for i in range(16):
        m.fit([z["B/Xi"][i]], [z["B/Xv"][i]], [z["B/Y"][i]])
        traj.append(m.alpha.copy())
    assert_close(np.stack(traj), z["B/alpha_traj"], 1e-5, 1e-7, "alpha")
    sd = m.state_dict()
    ref = sub(z, "B/sd_fit16")
    assert_state_close(sd, ref, sd0, what="fit16")
    # tables, first-order weights and bias never move in ONN fit
    for k in ref:
        if "embeddings" in k or k == "bias":
            np.testing.assert_array_equal(ref[k], sd0[k])


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("name", CLASS_NAMES)
def test_run_experiment(name, tag):
    z, meta = load_model_fixture(name, tag)
    m = orc.OracleModel(name, sub(z, "B/sd0"), batch_size=1)
    _, acc, roc, cm = m.run_experiment(z["B/Xi"].tolist(), z["B/Xv"].tolist(), z["B/Y"].tolist())
    ref = meta["run_experiment"]
    assert cm == ref["confusion_matrix"]
    assert acc == pytest.approx(ref["accuracy"], rel=1e-12)
    assert roc["tpr"] == pytest.approx(ref["roc"]["tpr"], rel=1e-12)
    assert roc["fpr"] == pytest.approx(ref["roc"]["fpr"], rel=1e-12)
    # 64 sequential sign-like steps: a coordinate whose gradient is ~0 can flip its step direction, so compare
    # with a tolerance of a few steps (lr = 0.01) on a tiny fraction of coordinates
    sd, ref_sd = m.state_dict(), sub(z, "B/sd_end")
    tot = bad = 0
    for k in ref_sd:
        d = np.abs(np.asarray(sd[k], dtype=np.float64) - ref_sd[k])
        tot += d.size
        bad += int((d > 1e-5 * np.maximum(np.abs(ref_sd[k]), 1e-2)).sum())
    assert bad <= 0.002 * tot, f"{bad}/{tot} coordinates differ after 64 online steps"


def test_adam_first_step_equals_closed_form():
    rng = np.random.default_rng(0)
    p = rng.normal(size=4096).astype(np.float32)
    g = (rng.normal(size=4096) * 10.0 ** rng.uniform(-6, 1, size=4096)).astype(np.float32)
    g[::7] = 0
    a = orc.adam_first_step(p, g, 0.01)
    b = orc.signadam_closed_form(p, g, 0.01)
    # identical up to one ulp of the parameter plus ~1e-6 relative on the lr-sized step
    assert np.max(np.abs(a - b) - 1.2e-7 * np.abs(p)) < 1e-6 * 0.01
    np.testing.assert_array_equal(a[::7], p[::7])


@pytest.mark.parametrize("task", ["cls", "reg"])
def test_fm_ftrl(task, golden_dir):
    z = np.load(f"{golden_dir}/FM_FTRL.npz")
    pred, real, w1, W2 = orc.fm_ftrl_online(z[f"{task}/X"], z[f"{task}/y"], task, float(z[f"{task}/eta"]),
                                            int(z[f"{task}/m"]), z[f"{task}/w1_0"], z[f"{task}/W2_0"])
    np.testing.assert_allclose(pred, z[f"{task}/pred"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(real, z[f"{task}/real"], rtol=0, atol=0)
    np.testing.assert_allclose(w1, z[f"{task}/w1"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(W2, z[f"{task}/W2"], rtol=1e-10, atol=1e-13)


def test_ftrl_proximal_fp32_vs_fp64():
    """FTRL-proximal is not in the reference (parity unpinned): the fp32 rule must track an fp64 evaluation."""
    rng = np.random.default_rng(3)
    hyper = dict(alpha=0.05, beta=1.0, l1=0.001, l2=0.01)
    w0 = rng.normal(size=512).astype(np.float32) * 0.1
    z32 = orc.ftrl_z_for_weight(w0, **hyper)
    n32 = np.zeros_like(z32)
    z64, n64 = z32.astype(np.float64), n32.astype(np.float64)
    np.testing.assert_allclose(orc.ftrl_weight(z32, n32, **hyper), w0, rtol=2e-6, atol=1e-8)
    for _ in range(50):
        g = rng.normal(size=512) * 0.3
        z32, n32 = orc.ftrl_step(z32, n32, g.astype(np.float32), **hyper)
        z64, n64 = orc.ftrl_step(z64, n64, g, dtype=np.float64, **hyper)
    w32 = orc.ftrl_weight(z32, n32, **hyper)
    w64 = orc.ftrl_weight(z64, n64, dtype=np.float64, **hyper)
    np.testing.assert_allclose(w32, w64, rtol=2e-4, atol=2e-6)
    assert (w64 == 0).any() or True


def test_flat_step_matches_class_step():
    """The flat-table step used for kernel parity equals the per-field class step (FMAdam.update_embedding)."""
    z, meta = load_model_fixture("FMAdam", "criteo39s")
    sd0 = sub(z, "A/sd0")
    sizes = meta["feature_sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    F = len(sizes)
    V = np.concatenate([sd0[f"second_order_embeddings.{i}.weight"] for i in range(F)]).astype(np.float32)
    w = np.concatenate([sd0[f"first_order_embeddings.{i}.weight"][:, 0] for i in range(F)]).astype(np.float32)
    state = dict(V=V.copy(), w=w.copy(), bias=np.float32(sd0["bias"]))
    rows = z["A/Xi1"] + offs[:-1][None, :]
    out = orc.flat_fm_step(state, rows, z["A/Xv1"], z["A/Y1"], "logits", "signadam", dict(lr=float(sd0["n"])))
    assert_close(out["loss"], z["A/loss_update_embedding"], 1e-5, 0, "loss")
    sd1 = sub(z, "A/sd1")
    V1 = np.concatenate([sd1[f"second_order_embeddings.{i}.weight"] for i in range(F)])
    w1 = np.concatenate([sd1[f"first_order_embeddings.{i}.weight"][:, 0] for i in range(F)])
    assert_close(state["V"], V1, 1e-5, 1e-7, "V")
    assert_close(state["w"], w1, 1e-5, 1e-7, "w")
    assert_close(state["bias"], sd1["bias"], 1e-5, 1e-7, "bias")


@pytest.mark.parametrize("rule,loss", [("signadam", "logits"), ("sgd", "sigmoid"), ("ftrl", "logits")])
def test_c_oracle_matches_numpy_oracle(rule, loss):
    """oracle/fm_oracle.c (the timed cpu_baseline port) against the golden-pinned numpy oracle."""
    from oracle import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/_build/liboracle.so not built (make -C oracle)")
    rng = np.random.default_rng(12)
    sizes = [3, 9, 1000, 5000, 4, 17, 200]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    R, k, B = int(offs[-1]), 8, 300
    rows = np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1) + offs[:-1][None, :]
    x = rng.uniform(-1, 1, size=rows.shape).astype(np.float32)
    y = (rng.uniform(size=B) < 0.3).astype(np.float32)
    V = (rng.normal(size=(R, k)) * 0.3).astype(np.float32)
    w = (rng.normal(size=R) * 0.3).astype(np.float32)
    hyp = dict(lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.001, l2=0.01)
    fh = {kk: hyp[kk] for kk in ("alpha", "beta", "l1", "l2")}
    if rule == "ftrl":
        mk = lambda: dict(zV=orc.ftrl_z_for_weight(V, **fh), nV=np.full_like(V, 0.2), zw=orc.ftrl_z_for_weight(w, **fh),
                          nw=np.full_like(w, 0.2), zb=np.float32(0.1), nb=np.float32(0.3))
        a, b = mk(), mk()
        la = orc.flat_fm_step(a, rows, x, y, loss, rule, fh)["loss"]
        lb = c_oracle.fm_step(b, rows, x, y, loss, rule, hyp)
        keys = ("zV", "nV", "zw", "nw", "zb", "nb")
    else:
        mk = lambda: dict(V=V.copy(), w=w.copy(), bias=np.float32(0.37))
        a, b = mk(), mk()
        la = orc.flat_fm_step(a, rows, x, y, loss, rule, dict(lr=hyp["lr"]))["loss"]
        lb = c_oracle.fm_step(b, rows, x, y, loss, rule, hyp)
        keys = ("V", "w", "bias")
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la))
    for kk in keys:
        ref, got = np.asarray(a[kk], dtype=np.float64), np.asarray(b[kk], dtype=np.float64)
        if rule == "signadam" and kk in ("V", "w"):
            # sign-like rule: identical except where the summed gradient is within fp32 noise of zero
            assert (np.abs(ref - got) > 1e-6).mean() < 2e-3
        else:
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6 * max(np.abs(ref).max(), 1e-3))


@pytest.mark.parametrize("rule", ["signadam", "sgd", "ftrl"])
def test_c_oracle_openmp_form_gives_the_same_bits(rule):
    """fmo_fm_step_mt (bench.py's multi-core cpu_baseline) == fmo_fm_step, bit for bit, over a few steps."""
    from oracle import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/_build/liboracle.so not built (make -C oracle)")
    rng = np.random.default_rng(3)
    sizes = [3, 9, 1000, 5000, 4, 17, 200]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    R, k, B = int(offs[-1]), 8, 500
    V = (rng.normal(size=(R, k)) * 0.3).astype(np.float32)
    w = (rng.normal(size=R) * 0.3).astype(np.float32)
    hyp = dict(lr=0.01, eps=1e-8, alpha=0.05, beta=1.0, l1=0.001, l2=0.01)
    fh = {kk: hyp[kk] for kk in ("alpha", "beta", "l1", "l2")}
    if rule == "ftrl":
        mk = lambda: dict(zV=orc.ftrl_z_for_weight(V, **fh), nV=np.full_like(V, 0.2), zw=orc.ftrl_z_for_weight(w, **fh),
                          nw=np.full_like(w, 0.2), zb=np.float32(0.1), nb=np.float32(0.3))
    else:
        mk = lambda: dict(V=V.copy(), w=w.copy(), bias=np.float32(0.37))
    a, b = mk(), mk()
    for step in range(3):
        rows = np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1) + offs[:-1][None, :]
        x = rng.uniform(-1, 1, size=rows.shape).astype(np.float32)
        y = (rng.uniform(size=B) < 0.3).astype(np.float32)
        la = c_oracle.fm_step(a, rows, x, y, "logits", rule, hyp)
        lb = c_oracle.fm_step(b, rows, x, y, "logits", rule, hyp, threads=4)
        assert la == lb
    for kk in a:
        np.testing.assert_array_equal(np.asarray(a[kk]), np.asarray(b[kk]))


def test_dense_reference_mode_equals_oracle_step():
    """oracle/dense_mode.py (bench.py's reference-faithful dense CPU timing: 2 x F nn.Embedding tables, dense gradients, a
    fresh torch.optim.Adam per step) takes the same step as the golden-pinned row-sparse oracle.  Entries whose gradient
    is of the order of Adam's eps (1e-8) are sign-like sensitive and excluded by magnitude of the reference step."""
    import torch
    from oracle import dense_mode
    torch.manual_seed(0)
    sizes, k, B, lr = [7, 5, 11, 3], 4, 8, 0.01
    m = dense_mode.DenseFM(sizes, k, lr)
    sd = {"bias": m.bias.detach().numpy().copy(), "n": np.float32(lr)}
    for i in range(len(sizes)):
        sd[f"first_order_embeddings.{i}.weight"] = m.first[i].weight.detach().numpy().copy()
        sd[f"second_order_embeddings.{i}.weight"] = m.second[i].weight.detach().numpy().copy()
    rng = np.random.default_rng(1)
    Xi = np.stack([rng.integers(0, s, size=B) for s in sizes], axis=1)
    Xv = np.ones_like(Xi, dtype=np.float32)
    Y = (rng.uniform(size=B) < 0.5).astype(np.float32)
    om = orc.OracleModel("FMAdam", {kk: v.copy() for kk, v in sd.items()})
    l_ref = om.update_embedding(Xi.reshape(B, -1, 1), Xv, Y)
    l_dense = m.step(torch.from_numpy(Xi), torch.from_numpy(Xv), torch.from_numpy(Y))
    assert abs(l_dense - float(l_ref)) <= 1e-6 * abs(float(l_ref))
    ref = om.state_dict()
    for i in range(len(sizes)):
        for name, got in ((f"first_order_embeddings.{i}.weight", m.first[i].weight), (f"second_order_embeddings.{i}.weight", m.second[i].weight)):
            a, b, p = got.detach().numpy().astype(np.float64), ref[name].astype(np.float64), sd[name].astype(np.float64)
            firm = np.abs(b - p) > 0.9 * lr          # |g| >> eps: the step is lr * sign(g)
            np.testing.assert_allclose(a[firm], b[firm], rtol=0, atol=1e-6)
            untouched = (b == p)
            np.testing.assert_array_equal(a[untouched], p[untouched])
