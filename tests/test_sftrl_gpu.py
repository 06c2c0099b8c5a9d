"""The sketched-FTRL family through fmx_sftrl_run (device="gpu") against the fixtures the imported reference produced
(tests/golden/path_b_family.npz, the same vectors and tolerances as the host path's test in test_host_logic.py) and
against the host fp64 path on wider sketches.  fp64; tolerance 1e-7 relative on predictions and on the sign-invariant
Gram matrices B B^T (the columns of B are defined up to sign by an SVD), counters exact."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def classes():
    from models.models_online.SFTRL_CCFM import SFTRL_CCFM
    from models.models_online.SFTRL_Vanila import SFTRL_Vanila
    return dict(SFTRL_CCFM=SFTRL_CCFM, SFTRL_Vanila=SFTRL_Vanila)


@pytest.mark.parametrize("task", ["cls", "reg"])
@pytest.mark.parametrize("name", ["SFTRL_CCFM", "SFTRL_Vanila"])
def test_gpu_sketched_ftrl_vs_reference_fixture(name, task, golden_dir, capsys):
    z = np.load(os.path.join(golden_dir, "path_b_family.npz"))
    m = classes()[name](torch.DoubleTensor(z[f"{task}/X"]), torch.DoubleTensor(z[f"{task}/y"]), task, 0.05, 4, device="gpu")
    pred, real, secs = m.online_learning()
    out = capsys.readouterr().out
    assert out.startswith("=" * 40 + f"\n{name}_0.05_4_start\n 0 th : pred ")
    assert tuple(pred.shape) == tuple(z[f"{task}/{name}/pred_shape"])
    np.testing.assert_allclose(pred, z[f"{task}/{name}/pred"].reshape(pred.shape), rtol=1e-7, atol=1e-9)
    np.testing.assert_array_equal(real, z[f"{task}/{name}/real"])
    assert [m.row_count_p, m.row_count_n] == list(z[f"{task}/{name}/counts"])
    np.testing.assert_allclose((m.BT_P @ m.BT_P.t()).numpy(), z[f"{task}/{name}/BTP_gram"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose((m.BT_N @ m.BT_N.t()).numpy(), z[f"{task}/{name}/BTN_gram"], rtol=1e-7, atol=1e-10)
    if name == "SFTRL_Vanila":
        np.testing.assert_allclose(m.w.numpy(), z[f"{task}/{name}/w"], rtol=1e-9, atol=1e-12)
    assert m.model_name == name and isinstance(secs, float)


def make_stream(n, D, seed, task):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, D)) / np.sqrt(D)
    wt = rng.standard_normal(D)
    s = X @ wt + 0.5 * (X[:, 0] * X[:, 1] - X[:, 2] * X[:, -1]) * D
    y = np.where(s >= 0, 1.0, -1.0) if task == "cls" else s
    return X, y


@pytest.mark.parametrize("name,task,D,m", [("SFTRL_CCFM", "cls", 32, 8), ("SFTRL_CCFM", "reg", 20, 32), ("SFTRL_Vanila", "cls", 33, 5),
                                            ("SFTRL_Vanila", "reg", 9, 64), ("SFTRL_CCFM", "cls", 6, 6), ("SFTRL_CCFM", "reg", 3, 1 + 1)])
def test_gpu_sketched_ftrl_equals_host_path(name, task, D, m, capsys):
    """Wider sketches than the fixture: sketch dim up to 32, m below, at and above it (both shrink branches), and a run
    continued from a previous run's state."""
    X, y = make_stream(700, D, 3 + D + m, task)
    cls = classes()[name]
    host = cls(torch.DoubleTensor(X[:400]), torch.DoubleTensor(y[:400]), task, 0.03, m)
    gpu = cls(torch.DoubleTensor(X[:400]), torch.DoubleTensor(y[:400]), task, 0.03, m, device="gpu")
    ph, _, _ = host.online_learning()
    pg, _, _ = gpu.online_learning()
    if task == "cls":
        # a +-1 prediction flips when y_hat is within rounding of 0: compare the sketches, allow no more than 1 flip
        assert (ph != pg).sum() <= 1
    else:
        np.testing.assert_allclose(pg, ph, rtol=1e-7, atol=1e-9)
    assert [gpu.row_count_p, gpu.row_count_n] == [host.row_count_p, host.row_count_n]
    for a, b in ((gpu.BT_P, host.BT_P), (gpu.BT_N, host.BT_N)):
        np.testing.assert_allclose((a @ a.t()).numpy(), (b @ b.t()).numpy(), rtol=1e-7, atol=1e-10)
    # second leg: both continue from the GPU run's state on the rest of the stream
    host2 = cls(torch.DoubleTensor(X[400:]), torch.DoubleTensor(y[400:]), task, 0.03, m)
    gpu2 = cls(torch.DoubleTensor(X[400:]), torch.DoubleTensor(y[400:]), task, 0.03, m, device="gpu")
    for dst in (host2, gpu2):
        dst.BT_P, dst.BT_N = gpu.BT_P.clone(), gpu.BT_N.clone()
        dst.row_count_p, dst.row_count_n = gpu.row_count_p, gpu.row_count_n
        if name == "SFTRL_Vanila":
            dst.w, dst.g_w = gpu.w.clone(), gpu.g_w.clone()
    ph, _, _ = host2.online_learning()
    pg, _, _ = gpu2.online_learning()
    capsys.readouterr()
    if task == "reg":
        np.testing.assert_allclose(pg, ph, rtol=1e-7, atol=1e-9)
    else:
        assert (ph != pg).sum() <= 1
    assert [gpu2.row_count_p, gpu2.row_count_n] == [host2.row_count_p, host2.row_count_n]
    for a, b in ((gpu2.BT_P, host2.BT_P), (gpu2.BT_N, host2.BT_N)):
        np.testing.assert_allclose((a @ a.t()).numpy(), (b @ b.t()).numpy(), rtol=1e-7, atol=1e-10)
    if name == "SFTRL_Vanila":
        np.testing.assert_allclose(gpu2.w.numpy(), host2.w.numpy(), rtol=1e-9, atol=1e-12)


def test_gpu_sketched_ftrl_limits_and_nan():
    """Outside the kernel's limits the gpu path raises (no silent host fallback); a NaN prediction raises like the reference."""
    from fmx import _lib
    X, y = make_stream(10, 40, 1, "reg")
    m = classes()["SFTRL_CCFM"](torch.DoubleTensor(X), torch.DoubleTensor(y), "reg", 0.05, 4, device="gpu")
    with pytest.raises(_lib.FmxError, match="sketch dim"):
        m.online_learning()
    X, y = make_stream(10, 8, 1, "reg")
    X[4, 2] = np.nan
    m = classes()["SFTRL_CCFM"](torch.DoubleTensor(X), torch.DoubleTensor(y), "reg", 0.05, 4, device="gpu")
    with pytest.raises(ValueError, match="Nan contained"):
        m.online_learning()
    with pytest.raises(ValueError):
        classes()["SFTRL_CCFM"](torch.DoubleTensor(X), torch.DoubleTensor(y), "reg", 0.05, 4, device="tpu")


@pytest.mark.parametrize("name,task", [("SFTRL_CCFM", "reg"), ("SFTRL_Vanila", "cls")])
def test_gpu_sketched_ftrl_grid_equals_single_runs(name, task, golden_dir, capsys):
    """fmx_sftrl_grid (Class.grid): every (learning_rate, m) setting of the grid over the reference's fixture stream equals its
    own single run bit for bit, and the fixture's own setting (0.05, 4) reproduces the reference's predictions."""
    z = np.load(os.path.join(golden_dir, "path_b_family.npz"))
    cls = classes()[name]
    X, y = torch.DoubleTensor(z[f"{task}/X"]), torch.DoubleTensor(z[f"{task}/y"])
    lrs, ms = [0.01, 0.05, 0.2], [2, 4, 7, 16]
    res = cls.grid(X, y, task, lrs, ms)
    assert len(res) == 12
    for (mdl, pred), (lr, m) in zip(res, [(a, b) for a in lrs for b in ms]):
        one = cls(X, y, task, lr, m, device="gpu")
        p1, _, _ = one.online_learning()
        np.testing.assert_array_equal(pred, p1)
        assert torch.equal(mdl.BT_P, one.BT_P) and torch.equal(mdl.BT_N, one.BT_N)
        assert [mdl.row_count_p, mdl.row_count_n] == [one.row_count_p, one.row_count_n]
        if name == "SFTRL_Vanila":
            assert torch.equal(mdl.w, one.w) and torch.equal(mdl.g_w, one.g_w)
        if (lr, m) == (0.05, 4):
            np.testing.assert_allclose(pred, z[f"{task}/{name}/pred"].reshape(pred.shape), rtol=1e-7, atol=1e-9)
    capsys.readouterr()


def test_grid_refuses_a_sketch_size_beyond_the_launch():
    """fmx_sftrl_grid sizes its LDS and the strides of B / w / pred for m_max; a setting whose m lies outside [1, m_max] (only a
    direct caller of the C ABI can pass one) is not run: status (2, m), its neighbours' state untouched."""
    import ctypes as C
    import fmx
    lib = fmx._lib.load()
    dev = torch.device("cuda")
    n, D, d, m_max = 64, 8, 8, 4
    rng = np.random.default_rng(0)
    X = torch.from_numpy(rng.normal(size=(n, D))).to(dev)
    y = torch.from_numpy(np.sign(rng.normal(size=n))).to(dev)
    ms = torch.tensor([4, 9, 2, 0], dtype=torch.int32, device=dev)                       # settings 1 and 3 are out of range
    etas = torch.full((4,), 0.1, dtype=torch.float64, device=dev)
    BP = torch.zeros((4, d * 2 * m_max), dtype=torch.float64, device=dev)
    BN = torch.zeros_like(BP)
    counts = torch.zeros((4, 2), dtype=torch.int32, device=dev)
    pred = torch.full((4, n), 7.0, dtype=torch.float64, device=dev)
    status = torch.zeros((4, 2), dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.fmx_sftrl_grid(p(X), p(y), n, D, d, 4, p(ms), p(etas), m_max, 1e-12, 0, p(BP), p(BN), p(counts), None, None, p(pred), p(status),
                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.fmx_last_error_string()
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert st[0, 0] == 0 and st[2, 0] == 0
    assert tuple(st[1]) == (2, 9) and tuple(st[3]) == (2, 0)
    pr = pred.cpu().numpy()
    assert (pr[1] == 7.0).all() and (pr[3] == 7.0).all()                                 # nothing was run for them
    assert (pr[0] != 7.0).any() and (pr[2] != 7.0).any()
    assert float(BP[1].abs().sum()) == 0.0 and float(BP[3].abs().sum()) == 0.0
