"""SFTRL_CCFM -- drop-in for reference models/models_online/SFTRL_CCFM.py:18-121 (sketched FTRL, convex-concave FM).

y_hat = ||BP^T x||^2 - ||BN^T x||^2 (:42-44); the gradient sign s picks the sketch: s <= 0 appends sqrt(-eta s) x to
BP, s > 0 appends sqrt(eta s) x to BN (:77-121).  fp64 like the reference.  device="host" (the default, like the
reference's CPU path; see _sketch.py) or device="gpu": the whole stream in one fmx_sftrl_run launch (include/fmx.h;
one wavefront, sketches in LDS, the shrink as a Jacobi eigen-decomposition of B B^T -- same B B^T, predictions and
counts, columns of B up to sign).  The gpu path never falls back: outside the kernel's limits it raises."""
import time

import numpy as np
import torch

from models.models_online.FM_Base import FM_Base
from models.models_online._sketch import Sketch

Tensor_type = torch.DoubleTensor


class SFTRL_CCFM(FM_Base):
    _linear_term = False

    def __init__(self, inputs_matrix, outputs, task, learning_rate, num_feature, device="host"):
        super(SFTRL_CCFM, self).__init__(inputs_matrix, outputs, task, learning_rate, num_feature)
        if device not in ("host", "gpu"):
            raise ValueError("device must be 'host' or 'gpu'")
        self.device = device
        self.model_name = "SFTRL_CCFM"
        self.row_count_p = 0
        self.row_count_n = 0
        d = self._sketch_dim()
        self.BT_P = Tensor_type(np.zeros([d, 2 * self.m]))
        self.BT_N = Tensor_type(np.zeros([d, 2 * self.m]))

    def _sketch_dim(self):
        return self.num_feature

    def online_learning(self):
        start = time.time()
        print("==" * 20)
        print(self.model_name + "_" + str(self.eta) + "_" + str(self.m) + "_start")
        if self.task not in ("cls", "reg"):
            raise NotImplementedError
        cls = self.task == "cls"
        X = self.At.t().contiguous().numpy().astype(np.float64, copy=False)
        y = np.asarray(self.b.reshape(-1).numpy(), dtype=np.float64)
        d = self._sketch_dim()
        if self.device == "gpu":
            preds = self._online_learning_gpu(X, y, d, cls)
            end = time.time()
            print("learning time : %f " % (end - start))
            return preds, y.copy(), (end - start)
        P, N = Sketch(d, self.m, self._thres), Sketch(d, self.m, self._thres)
        P.B, P.count = self.BT_P.numpy().copy(), self.row_count_p
        N.B, N.count = self.BT_N.numpy().copy(), self.row_count_n
        w = g_w = None
        if self._linear_term:
            w, g_w = self.w.numpy().reshape(-1).copy(), self.g_w.numpy().reshape(-1).copy()
        preds = np.empty((self.num_data,) + self._pred_shape(cls), dtype=np.float64)
        for idx in range(self.num_data):
            x = X[idx]
            xs = x[:d]
            scalar = P.energy(xs) - N.energy(xs)
            if self._linear_term:
                scalar += float(w @ x)
            if np.isnan(scalar):
                raise ValueError("Nan contained")
            if cls:
                pred = 1.0 if scalar >= 0 else -1.0
                sign = (-1.0 / (1.0 + np.exp(scalar * y[idx]))) * y[idx]
            else:
                pred = scalar
                sign = 2.0 * (scalar - y[idx])
            if self._linear_term:
                g_w += sign * x
                w = -self.eta * g_w
            if sign <= 0:
                P.append(np.sqrt(-self.eta * sign) * xs)
            else:
                N.append(np.sqrt(self.eta * sign) * xs)
            preds[idx] = pred
            if idx % 1000 == 0:
                print(" %d th : pred %f , real %f " % (idx, pred, y[idx]))
        self.BT_P, self.row_count_p = torch.from_numpy(P.B.copy()), P.count
        self.BT_N, self.row_count_n = torch.from_numpy(N.B.copy()), N.count
        if self._linear_term:
            self.w = torch.from_numpy(w.reshape(-1, 1).copy())
            self.g_w = torch.from_numpy(g_w.reshape(-1, 1).copy())
        end = time.time()
        print("learning time : %f " % (end - start))
        return preds, y.copy(), (end - start)

    def _online_learning_gpu(self, X, y, d, cls):
        """The same stream through fmx_sftrl_run; state (sketches, counts, linear term) is read and written back."""
        import ctypes as C

        from fmx import _lib
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        n, D = X.shape
        Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
        BP, BN = self.BT_P.to(dev).contiguous(), self.BT_N.to(dev).contiguous()
        counts = torch.tensor([self.row_count_p, self.row_count_n], dtype=torch.int32, device=dev)
        w = g_w = None
        if self._linear_term:
            w, g_w = self.w.reshape(-1).to(dev).contiguous(), self.g_w.reshape(-1).to(dev).contiguous()
        pred = torch.empty(n, dtype=torch.float64, device=dev)
        status = torch.zeros(2, dtype=torch.int32, device=dev)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _lib.check(lib.fmx_sftrl_run(ptr(Xd), ptr(yd), n, D, d, self.m, float(self.eta), float(self._thres), 0 if cls else 1,
                                     ptr(BP), ptr(BN), ptr(counts), ptr(w), ptr(g_w), ptr(pred), ptr(status),
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        st = status.cpu()
        if int(st[0]) == 1:
            raise ValueError("Nan contained")
        c = counts.cpu()
        self.BT_P, self.row_count_p = BP.cpu(), int(c[0])
        self.BT_N, self.row_count_n = BN.cpu(), int(c[1])
        if self._linear_term:
            self.w, self.g_w = w.cpu().reshape(-1, 1), g_w.cpu().reshape(-1, 1)
        p = pred.cpu().numpy()
        for idx in range(0, n, 1000):
            print(" %d th : pred %f , real %f " % (idx, p[idx], y[idx]))
        return p.reshape((n,) + self._pred_shape(cls))

    @classmethod
    def grid(cls, inputs_matrix, outputs, task, learning_rates, num_features):
        """An extension the reference lacks (its notebooks try one (learning_rate, m) pair per run): every pair of
        `learning_rates` x `num_features` over the SAME stream in ONE launch (fmx_sftrl_grid: one wavefront per setting,
        up to 256 settings side by side).  -> list of (model, predictions) in the order of itertools.product; every model
        is what `cls(..., lr, m, device="gpu").online_learning()` leaves behind, bit for bit."""
        import ctypes as C
        import itertools

        from fmx import _lib
        if task not in ("cls", "reg"):
            raise NotImplementedError
        settings = list(itertools.product(learning_rates, num_features))
        models = [cls(inputs_matrix, outputs, task, lr, m, device="gpu") for lr, m in settings]
        if not settings:
            return []
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        m0 = models[0]
        X = m0.At.t().contiguous().numpy().astype(np.float64, copy=False)
        y = np.asarray(m0.b.reshape(-1).numpy(), dtype=np.float64)
        n, D = X.shape
        d, S, m_max = m0._sketch_dim(), len(settings), max(m for _, m in settings)
        Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
        ms = torch.tensor([m for _, m in settings], dtype=torch.int32, device=dev)
        etas = torch.tensor([float(lr) for lr, _ in settings], dtype=torch.float64, device=dev)
        BP = torch.zeros((S, d * 2 * m_max), dtype=torch.float64, device=dev)
        BN = torch.zeros_like(BP)
        counts = torch.zeros((S, 2), dtype=torch.int32, device=dev)
        w = g_w = None
        if cls._linear_term:
            w, g_w = torch.zeros((S, D), dtype=torch.float64, device=dev), torch.zeros((S, D), dtype=torch.float64, device=dev)
        pred = torch.empty((S, n), dtype=torch.float64, device=dev)
        status = torch.zeros((S, 2), dtype=torch.int32, device=dev)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _lib.check(lib.fmx_sftrl_grid(ptr(Xd), ptr(yd), n, D, d, S, ptr(ms), ptr(etas), m_max, float(m0._thres), 0 if task == "cls" else 1,
                                      ptr(BP), ptr(BN), ptr(counts), ptr(w), ptr(g_w), ptr(pred), ptr(status),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        st_h = status.cpu()
        if bool((st_h[:, 0] == 2).any()):
            raise ValueError("a setting's num_feature lies outside [1, max]: not run (fmx_sftrl_grid status 2)")
        if bool((st_h[:, 0] == 1).any()):
            raise ValueError("Nan contained")
        BPh, BNh, ch, ph = BP.cpu(), BN.cpu(), counts.cpu(), pred.cpu().numpy()
        out = []
        for s, (mdl, (_, m)) in enumerate(zip(models, settings)):
            mdl.BT_P = BPh[s, :d * 2 * m].reshape(d, 2 * m).clone()
            mdl.BT_N = BNh[s, :d * 2 * m].reshape(d, 2 * m).clone()
            mdl.row_count_p, mdl.row_count_n = int(ch[s, 0]), int(ch[s, 1])
            if cls._linear_term:
                mdl.w, mdl.g_w = w[s].cpu().reshape(-1, 1), g_w[s].cpu().reshape(-1, 1)
            out.append((mdl, ph[s].reshape((n,) + mdl._pred_shape(task == "cls"))))
        return out

    def _pred_shape(self, cls):
        return (1,) if cls else ()
