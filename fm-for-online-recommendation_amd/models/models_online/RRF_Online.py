"""RRF_Online -- drop-in for reference models/models_online/RRF_Online.py:18-187 (reparameterised random Fourier
features, online).  phi(x) = [cos, sin](x (e^gamma * eps)) (:70-75); SGD on w and gamma with the reference's gradient
formulas (:88-123), including its `lr_w * exp(w)` term in d_w (:97) and, for the logit loss, the per-batch softmax
weight that equals 1 at batch size 1 (:101-102).  Host fp64; a kernel method, out of scope for the HIP kernels."""
import time

import numpy as np
import torch
from torch.nn import Module

Tensor_type = torch.DoubleTensor


class RRF_Online(Module):
    def __init__(self, inputs_matrix, outputs, task, loss_type=None, gamma=None, w=None, num_sampled_spectral=10,
                 random_seed=100, lr_RRF_w=0.05, lr_RRF_gamma=0.05):
        super(RRF_Online, self).__init__()
        self.X = inputs_matrix
        self.Y = outputs
        self.loss_type = loss_type
        self.num_feature = inputs_matrix.shape[1]
        self.model_name = "RRF_Online"
        self.task = task
        self.num_sampled_spectral = num_sampled_spectral
        self.lr_RRF_w = lr_RRF_w
        self.lr_RRF_gamma = lr_RRF_gamma
        self.random_seed = random_seed
        self._init_param(gamma, w, loss_type)

    def _init_param(self, gamma, w, loss_type):
        """Draw order of the reference (:47-67): numpy rand for gamma, then torch randn for w and for eps."""
        if self.task == "cls":
            self.loss_type = "logit" if loss_type is None else "hinge"
        elif self.task == "reg":
            self.loss_type = "l2" if loss_type is None else "l1"
        else:
            raise NotImplementedError("wrong task assigned")
        if gamma is None:
            self.gamma = Tensor_type(np.log(np.random.rand(self.num_feature, 1)))
        else:
            self.gamma = Tensor_type(np.log(gamma) * np.ones((self.num_feature, 1)))
        self.w = 0.1 * torch.randn(2 * self.num_sampled_spectral).type(Tensor_type) if w is None else w
        self.eps = torch.randn(self.num_feature, self.num_sampled_spectral).type(Tensor_type)

    def _compute_phi(self, x_t):
        z = x_t.matmul(self.gamma.exp().mul(self.eps))
        return torch.cat([z.cos(), z.sin()], 1)

    def _predict(self, phi):
        return phi.matmul(self.w)

    def online_learning(self):
        start = time.time()
        print("==" * 20)
        if self.loss_type not in ("logit", "l2"):
            raise NotImplementedError("wrong loss type in get_grad")
        X = self.X.numpy().astype(np.float64, copy=False)
        Y = np.asarray(self.Y.reshape(-1).numpy(), dtype=np.float64)
        gamma = self.gamma.numpy().reshape(-1).copy()
        w = self.w.numpy().copy()
        eps = self.eps.numpy()
        D = self.num_sampled_spectral
        cls = self.task == "cls"
        pred_list, real_list = [], []
        for t in range(X.shape[0]):
            x, y = X[t], Y[t]
            eg = np.exp(gamma)
            z = x @ (eg[:, None] * eps)
            cz, sz = np.cos(z), np.sin(z)
            phi = np.concatenate([cz, sz])
            scalar = float(phi @ w)
            if not np.isnan(scalar):
                coef = -y if self.loss_type == "logit" else (scalar - y)     # logit: -y * softmax over a batch of 1
                d_w = self.lr_RRF_w * np.exp(w) + coef * phi
                d_phi = coef * w
                # d phi / d gamma_n: -x_n eps_nd sin(z_d) e^gamma_n for the cos half, +x_n eps_nd cos(z_d) e^gamma_n for sin
                d_gamma = (x[:, None] * eps * (-sz * d_phi[:D] + cz * d_phi[D:])[None, :]).sum(axis=1) * eg
                w = w - self.lr_RRF_w * d_w
                gamma = gamma - self.lr_RRF_gamma * d_gamma
                pred_list.append([1.0 if scalar >= 0 else -1.0] if cls else [scalar])
                real_list.append(y)
            if t % 1000 == 0:
                print(" %d th : pred %f , real %f " % (t, (1.0 if scalar >= 0 else -1.0) if cls and not np.isnan(scalar)
                                                     else scalar, y))
        self.w = torch.from_numpy(w.copy())
        self.gamma = torch.from_numpy(gamma.reshape(-1, 1).copy())
        end = time.time()
        print("learning time : %f " % (end - start))
        return np.asarray(pred_list, dtype=np.float64), np.asarray(real_list, dtype=np.float64), (end - start)
