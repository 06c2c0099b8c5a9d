"""CPU-only tests of the host side: input readers / batchers against the golden fixture, input normalisation,
flat-table geometry, and that libfmx.so loads and exports every symbol of include/fmx.h (no compute calls)."""
import json
import os
import random
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "data_preprocess.json")) as fh:
        return json.load(fh)


def _paths(golden_dir):
    return (os.path.join(golden_dir, "handmade_train_input.csv"), os.path.join(golden_dir, "handmade_category_emb.csv"),
            os.path.join(golden_dir, "handmade.libsvm"))


def test_read_criteo_data(gold, golden_dir):
    from utils import data_preprocess as dp
    csv, emb, _ = _paths(golden_dir)
    r = dp.read_criteo_data(csv, emb)
    g = gold["read_criteo_data"]
    assert r["size"] == g["size"] and r["label"] == g["label"] and r["index"] == g["index"]
    assert r["value"] == g["value"] and r["feature_sizes"] == g["feature_sizes"]


def test_batchers_consume_random_like_the_reference(gold, golden_dir):
    from utils import data_preprocess as dp
    csv, emb, _ = _paths(golden_dir)
    random.seed(11)
    a = dp.create_ten_iter(csv, emb, 4, 8)
    g = gold["create_ten_iter"]
    assert a[0] == g["Xi"] and a[1] == g["Xv"] and a[2] == g["Y"] and [list(t) for t in a[3]] == g["ratio"]
    random.seed(12)
    a = dp.create_dataset(csv, emb, 3, 4, 8)
    g = gold["create_dataset"]
    assert a[0] == g["Xi"] and a[1] == g["Xv"] and a[2] == g["Y"] and [list(t) for t in a[3]] == g["ratio"]
    random.seed(13)
    r = dp.balance_criteo_data(csv, emb)
    g = gold["balance_criteo_data"]
    assert r["size"] == g["size"] and r["label"] == g["label"] and r["index"] == g["index"] and r["value"] == g["value"]


def test_read_svm_file(gold, golden_dir):
    from utils import data_preprocess as dp
    _, _, svm = _paths(golden_dir)
    r = dp.read_svm_file(svm)
    g = gold["read_svm_file"]
    assert int(r["size"]) == g["size"]
    np.testing.assert_array_equal(r["label"], g["label"])
    np.testing.assert_array_equal(r["index"], g["index"])
    np.testing.assert_array_equal(r["value"], g["value"])
    np.testing.assert_array_equal(r["feature_sizes"], g["feature_sizes"])
    assert r["index"].dtype.kind == "i" and r["label"].dtype.kind == "i"
    random.seed(14)
    r = dp.balance_svm_data(svm)
    g = gold["balance_svm_data"]
    assert int(r["size"]) == g["size"]
    np.testing.assert_array_equal(r["label"], g["label"])
    np.testing.assert_array_equal(r["index"], g["index"])
    np.testing.assert_array_equal(r["value"], g["value"])


def test_normalize_inputs():
    import fmx
    sizes = [7, 5, 11, 3]
    idx, xv = fmx.normalize_inputs([1, 2, 3, 0], [1, 1, 1, 1], 4, sizes)          # a 1-D single sample
    assert idx.shape == (1, 4) and idx.dtype == np.int32 and xv is None
    idx, xv = fmx.normalize_inputs([[1, 2, 3, 0], [6, 4, 10, 2]], [[1, .5, 1, 1], [1, 1, 1, 1]], 4, sizes)
    assert idx.shape == (2, 4) and xv.dtype == np.float32 and xv[0, 1] == 0.5
    for bad in ([[7, 0, 0, 0]], [[-1, 0, 0, 0]], [[0, 0, 0, 3]]):
        with pytest.raises(IndexError):
            fmx.normalize_inputs(bad, [[1, 1, 1, 1]], 4, sizes)
    with pytest.raises(ValueError):
        fmx.normalize_inputs([[1, 2, 3]], [[1, 1, 1, 1]], 4, sizes)


def test_flat_table_geometry():
    import fmx
    assert [fmx.padded_k(k) for k in (1, 4, 5, 10, 16, 17, 64)] == [4, 4, 8, 16, 16, 32, 64]
    with pytest.raises(ValueError):
        fmx.padded_k(65)
    t = fmx.FlatTable([7, 5, 11, 3], 10, device="cpu")
    assert (t.kp, t.row_stride, t.n_rows) == (16, 32, 26)
    np.testing.assert_array_equal(t.offsets_host, [0, 7, 12, 23, 26])
    t2 = fmx.FlatTable([7, 5, 11, 3], 16, layout="ftrl", device="cpu")
    assert (t2.z_offset, t2.row_stride, t2.bias.numel()) == (32, 64, 2)      # forward half | (z, n) half: 2 x 128 B
    with pytest.raises(ValueError):
        fmx.FlatTable([7, 0], 4, device="cpu")
    with pytest.raises(ValueError):
        fmx.FlatTable([7, 5], 16, device="cpu", row_stride=18)
    # reference-shaped import / export round trip on the flat buffer
    import torch
    first = [torch.randn(s, 1) for s in (7, 5, 11, 3)]
    second = [torch.randn(s, 10) for s in (7, 5, 11, 3)]
    t.load_reference(first, second)
    f2, s2 = t.export_reference()
    for a, b in zip(first + second, f2 + s2):
        assert torch.equal(a, b)
    assert float(t.rows[:, 10:16].abs().sum()) == 0 and float(t.rows[:, 17:].abs().sum()) == 0


def test_library_exports_every_declared_symbol():
    import fmx
    lib = fmx._lib.load()
    header = open(os.path.join(ROOT, "include", "fmx.h")).read()
    declared = set(re.findall(r"\b(fmx_[a-z_]+)\s*\(", header))
    assert declared == set(fmx._lib.EXPORTS), declared ^ set(fmx._lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.fmx_version() == 104
    assert (lib.fmx_sorted_width(1), lib.fmx_sorted_width(65), lib.fmx_sorted_width(4096)) == (64, 128, 4096)
    assert (lib.fmx_sorted_bbits(1), lib.fmx_sorted_bbits(4096), lib.fmx_sorted_bbits(4097)) == (6, 12, 13)


def test_byte_count_entry_points_return_int64():
    """Both *_workspace_bytes functions are int64_t in include/fmx.h; a 32-bit restype truncates sizes beyond 2 GiB (the
    MLP section's workspace at B = 32768, hidden = 2048, 8 layers is 6.4 GiB) and the caller would under-allocate."""
    import ctypes as C
    import fmx
    lib = fmx._lib.load()
    assert lib.fmx_workspace_bytes.restype is C.c_int64
    assert lib.fmx_mlp_section_workspace_bytes.restype is C.c_int64
    m = fmx._lib.Mlp(None, 8, 16, 2048, 0)
    big = lib.fmx_mlp_section_workspace_bytes(C.byref(m), 32768)          # host arithmetic only: no GPU needed
    assert big > (1 << 32), big


def test_models_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from models.models_online_deep.fm_adam import FMAdam
    with pytest.raises(RuntimeError, match="gfx950"):
        FMAdam([3, 4], embedding_size=4)
    import fmx
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        fmx.FMEngine(fmx.FlatTable([3, 4], 4, device="cpu"))


@pytest.mark.parametrize("task", ["cls", "reg"])
def test_fm_ftrl_class_vs_reference(task, golden_dir, capsys):
    """Hot path B (fp64, host): same seed => same init draws => same predictions and final weights as the reference."""
    import torch
    from models.models_online.FM_FTRL import FM_FTRL
    z = np.load(os.path.join(golden_dir, "FM_FTRL.npz"))
    torch.manual_seed(5)
    m = FM_FTRL(torch.DoubleTensor(z[f"{task}/X"]), torch.DoubleTensor(z[f"{task}/y"]), task, float(z[f"{task}/eta"]),
                int(z[f"{task}/m"]))
    pred, real, secs = m.online_learning()
    out = capsys.readouterr().out
    assert out.startswith("FM_FTRL_0.005_8_start\n 0 th : pred ") and "learning time : " in out
    assert pred.shape == ((256, 1) if task == "cls" else (256, 1, 1)) and real.shape == (256,)
    assert isinstance(secs, float) and m.model_name == "FM_FTRL" and m.eta == 0.005 and m.m == 8
    np.testing.assert_allclose(pred.reshape(-1), z[f"{task}/pred"], rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(real, z[f"{task}/real"])
    np.testing.assert_allclose(m.w1.numpy(), z[f"{task}/w1"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m.W2.numpy(), z[f"{task}/W2"], rtol=1e-10, atol=1e-13)
    assert m.w1.dtype == torch.float64 and tuple(m.W2.shape) == (16, 7)


def test_fm_ftrl_errors():
    import torch
    from models.models_online.FM_FTRL import FM_FTRL
    X = torch.DoubleTensor(np.full((3, 8), np.nan))
    with pytest.raises(ValueError, match="Nan contained"):
        FM_FTRL(X, torch.DoubleTensor([1, -1, 1]), "cls", 0.1, 2).online_learning()
    with pytest.raises(NotImplementedError):
        FM_FTRL(torch.DoubleTensor(np.ones((3, 8))), torch.DoubleTensor([1, -1, 1]), "rank", 0.1, 2).online_learning()


def test_data_manager_and_metrics_vs_reference(golden_dir, tmp_path):
    from utils import data_manager as dm
    from utils import metric_manager as mm
    z = np.load(os.path.join(golden_dir, "data_manager.npz"))
    X, X2, Y, Y2, dt = dm.load_dataset_movielens(os.path.join(golden_dir, "handmade_movielens.tsv"), 30, 15, 6)
    np.testing.assert_array_equal(X.toarray(), z["ml/X"])
    np.testing.assert_array_equal(X2.toarray(), z["ml/X2"])
    assert X.dtype == np.float32 and X2.dtype == np.float64 and dt.dtype == np.float32
    np.testing.assert_array_equal(Y, z["ml/Y"])
    np.testing.assert_array_equal([int(v) for v in Y2], z["ml/Y2"])
    np.testing.assert_array_equal(dt, z["ml/dt"])
    sX, sY, _ = dm.sort_dataset_movielens(X, Y2, dt)
    np.testing.assert_array_equal(sX.toarray(), z["ml/sorted_X"])
    np.testing.assert_array_equal(sY, z["ml/sorted_Y"])
    fx, fy = dm.load_dataset_fappe(os.path.join(golden_dir, "handmade_frappe.libfm"))
    np.testing.assert_array_equal(fx, z["frappe/X"])
    np.testing.assert_array_equal(fy, z["frappe/Y"])
    ragged = tmp_path / "ragged.libfm"
    ragged.write_text("1 3:1 4:1 5:1\n-1 3:1\n1 9:1 4:1\n")
    rx, ry = dm.load_dataset_fappe(str(ragged))
    assert [list(r) for r in rx] == [[0], [3, 1], [0, 1, 2]] and list(ry) == [-1.0, 1.0, 1.0]
    np.testing.assert_allclose(mm.regression_metric(z["metric/pred"], z["metric/real"]), z["metric/reg"], rtol=1e-12)
    sp, sr = np.sign(z["metric/pred"]), np.sign(z["metric/real"])
    m, acc = mm.classfication_metric(sp, sr)
    np.testing.assert_allclose(m, z["metric/cls"], rtol=1e-12)
    np.testing.assert_allclose(acc, z["metric/cls_acc"], rtol=1e-12)


@pytest.mark.parametrize("task", ["cls", "reg"])
@pytest.mark.parametrize("name", ["SFTRL_CCFM", "SFTRL_Vanila"])
def test_sketched_ftrl_vs_reference(name, task, golden_dir, capsys):
    """The sketched FTRL family (host fp64): predictions, sketch Gram matrices (sign-invariant) and counters."""
    import torch
    from models.models_online.SFTRL_CCFM import SFTRL_CCFM
    from models.models_online.SFTRL_Vanila import SFTRL_Vanila
    z = np.load(os.path.join(golden_dir, "path_b_family.npz"))
    cls = dict(SFTRL_CCFM=SFTRL_CCFM, SFTRL_Vanila=SFTRL_Vanila)[name]
    m = cls(torch.DoubleTensor(z[f"{task}/X"]), torch.DoubleTensor(z[f"{task}/y"]), task, 0.05, 4)
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


@pytest.mark.parametrize("task", ["cls", "reg"])
def test_rrf_online_vs_reference(task, golden_dir, capsys):
    import random
    import torch
    from models.models_online.RRF_Online import RRF_Online
    z = np.load(os.path.join(golden_dir, "path_b_family.npz"))
    torch.manual_seed(17)
    np.random.seed(17)
    random.seed(17)
    m = RRF_Online(torch.DoubleTensor(z[f"{task}/X"][:100]), torch.DoubleTensor(z[f"{task}/y"][:100]), task,
                   num_sampled_spectral=6)
    assert m.loss_type == ("logit" if task == "cls" else "l2")
    pred, real, _ = m.online_learning()
    capsys.readouterr()
    assert tuple(pred.shape) == tuple(z[f"{task}/RRF/pred_shape"])
    np.testing.assert_allclose(pred, z[f"{task}/RRF/pred"].reshape(pred.shape), rtol=1e-8, atol=1e-10)
    np.testing.assert_array_equal(real, z[f"{task}/RRF/real"])
    np.testing.assert_allclose(m.w.numpy(), z[f"{task}/RRF/w"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(m.gamma.numpy(), z[f"{task}/RRF/gamma"], rtol=1e-8, atol=1e-11)
    with pytest.raises(NotImplementedError):
        RRF_Online(torch.DoubleTensor(z["cls/X"]), torch.DoubleTensor(z["cls/y"]), "cls", loss_type="x").online_learning()


def test_exact_step_capacity_and_sub_steps():
    """The largest batch one exact step can take is the LDS merge width, whatever the vocabulary: a field whose indices
    would not fit a 32-bit (index, sample) composite is cut into sort pieces (FlatTable.ensure_sort_split).  How the
    multi-GPU wrappers split a global batch that exceeds a backend's cap."""
    import fmx
    from fmx.distributed import DataParallelFM, max_step_batch
    assert max_step_batch(3) == 32768 and max_step_batch(1 << 24) == 32768
    # the split: 18-bit fields stay whole up to 16,384 samples and fall into two pieces at 32,768
    t = fmx.FlatTable([3, 176373, 10, 131071, 131072], 4, device="cpu")
    t.ensure_sort_split(16384)
    assert t._sort_split is None
    t.ensure_sort_split(32768)
    so, sc, mx = t._sort_split
    assert sc.tolist() == [0, 1, 1, 2, 3, 4, 4] and mx == 131071              # 176373 -> 88187 + 88186, 131071 whole, 131072 -> 2 x 65536
    assert so.tolist() == [0, 3, 3 + 88187, 3 + 176373, 176386, 176386 + 131071, 176386 + 131071 + 65536, 176386 + 131071 + 131072]
    assert mx <= (0xFFFFFFFF >> 15)
    t.ensure_sort_split(4096)
    assert t._sort_split is None

    class Backend:
        max_global_batch = 16384

    dp = DataParallelFM(Backend())
    assert dp.world == 1 and dp._sub_steps(4096) == 1 and dp._sub_steps(16384) == 1 and dp._sub_steps(32768) == 2
    dp.world = 8
    assert dp._sub_steps(4096) == 2 and dp._sub_steps(2048) == 1 and dp._sub_steps(16384) == 8
    dp.world = 4
    assert dp._sub_steps(4096) == 1


def test_read_criteo_arrays_equals_read_criteo_data(golden_dir):
    """The array reader returns the same numbers as the reference-shaped list reader (reference data_preprocess.py:29-47)."""
    from utils import data_preprocess as dp
    csv, emb = os.path.join(golden_dir, "handmade_train_input.csv"), os.path.join(golden_dir, "handmade_category_emb.csv")
    a, b = dp.read_criteo_arrays(csv, emb), dp.read_criteo_data(csv, emb)
    assert a["size"] == b["size"] and a["feature_sizes"] == b["feature_sizes"] and a["value"] is None
    assert a["index"].dtype == np.int32 and a["index"].flags["C_CONTIGUOUS"]
    np.testing.assert_array_equal(a["index"], np.asarray(b["index"]))
    np.testing.assert_array_equal(a["label"], np.asarray(b["label"]))
    assert all(v == [1] * 39 for v in b["value"])


def test_pinned_batch_stager_on_cpu():
    """Batches come out in order, whole, with the right dtypes; the last partial batch too; buffers are reused after
    `depth` draws (documented); out-of-range indices raise IndexError when feature_sizes are given."""
    import torch
    from utils.data_preprocess import PinnedBatchStager
    rng = np.random.default_rng(0)
    sizes = [7, 5, 11, 3]
    N, B = 37, 8
    index = np.stack([rng.integers(0, s, size=N) for s in sizes], axis=1)
    label = rng.integers(0, 2, size=N)
    value = rng.uniform(size=(N, 4))
    for val in (None, value):
        for depth in (1, 2, 3):
            st = PinnedBatchStager(index, label, B, device="cpu", value=val, depth=depth, feature_sizes=sizes)
            assert len(st) == 5
            seen = 0
            for idx_d, xv_d, y_d in st:
                n = idx_d.shape[0]
                assert idx_d.dtype == torch.int32 and y_d.dtype == torch.float32 and (xv_d is None) == (val is None)
                np.testing.assert_array_equal(idx_d.numpy(), index[seen:seen + n])
                np.testing.assert_array_equal(y_d.numpy(), label[seen:seen + n].astype(np.float32))
                if val is not None:
                    np.testing.assert_array_equal(xv_d.numpy(), value[seen:seen + n].astype(np.float32))
                seen += n
            assert seen == N
    assert len(PinnedBatchStager(index, label, B, device="cpu", drop_last=True)) == 4
    bad = index.copy()
    bad[3, 1] = 5
    with pytest.raises(IndexError):
        PinnedBatchStager(bad, label, B, device="cpu", feature_sizes=sizes)
    with pytest.raises(ValueError):
        PinnedBatchStager(index[:, 0], label, B, device="cpu")
