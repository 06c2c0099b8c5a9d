"""FM_FTRL -- drop-in for reference models/models_online/FM_FTRL.py:27-92 (hot path B, fp64, host side).

Per sample x (a column of At):  y_hat = w1^T x + ||W2 x'||^2 with x' = x without its last feature (:61-63);
cumulative gradients g_w1 += s x, g_W2 += 2 W2 x' x'^T -- without the factor s, a reference quirk kept here (:76-77);
then w1 = -eta g_w1, W2 = -eta g_W2 (:79-80).  The loop runs in numpy fp64 on the host (one matvec and one rank-1
update per sample instead of the reference's two matmuls), with the reference's return value and prints.
"""
import time

import numpy as np
import torch

from models.models_online.FM_Base import FM_Base

Tensor_type = torch.DoubleTensor
numpy_type = np.float64


class FM_FTRL(FM_Base):
    def __init__(self, inputs_matrix, outputs, task, learning_rate, num_feature):
        super(FM_FTRL, self).__init__(inputs_matrix, outputs, task, learning_rate, num_feature)
        self.model_name = "FM_FTRL"

    def _init_parameter(self):
        """randn init in the reference's draw order (:42-43)"""
        self.w1 = torch.randn(self.num_feature, 1).type(Tensor_type)
        self.W2 = torch.randn(2 * self.m, self.num_feature - 1).type(Tensor_type)

    def online_learning(self):
        start = time.time()
        self._init_parameter()
        print(self.model_name + "_" + str(self.eta) + "_" + str(self.m) + "_start")
        if self.task not in ("cls", "reg"):
            raise NotImplementedError
        cls = self.task == "cls"
        X = self.At.t().contiguous().numpy().astype(numpy_type, copy=False)      # [N, d]
        y = np.asarray(self.b.reshape(-1).numpy(), dtype=numpy_type)
        w1 = self.w1.numpy().reshape(-1).copy()
        W2 = self.W2.numpy().copy()
        g_w1 = np.zeros_like(w1)
        g_W2 = np.zeros_like(W2)
        eta = self.eta
        pred_list = np.empty((self.num_data, 1) if cls else (self.num_data, 1, 1), dtype=numpy_type)
        for idx in range(self.num_data):
            x = X[idx]
            xs = x[:-1]
            t = W2 @ xs
            scalar = w1 @ x + t @ t
            if np.isnan(scalar):
                raise ValueError("Nan contained")
            if cls:
                pred = 1.0 if scalar >= 0 else -1.0
                sign_idx = (-1.0 / (1.0 + np.exp(scalar * y[idx]))) * y[idx]
            else:
                pred = scalar
                sign_idx = 2.0 * (scalar - y[idx])
            g_w1 += sign_idx * x
            g_W2 += 2.0 * np.outer(t, xs)
            w1 = -eta * g_w1
            W2 = -eta * g_W2
            pred_list[idx] = pred
            if idx % 1000 == 0:
                print(" %d th : pred %f , real %f " % (idx, pred, y[idx]))
        self.w1 = torch.from_numpy(w1.reshape(-1, 1).copy())
        self.W2 = torch.from_numpy(W2.copy())
        end = time.time()
        print("learning time : %f " % (end - start))
        return pred_list, y.copy(), (end - start)
