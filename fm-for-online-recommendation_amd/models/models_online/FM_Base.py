"""FM_Base -- drop-in for reference models/models_online/FM_Base.py:15-69 (hot path B, fp64, host side).

Path B is strictly sequential (each prediction depends on the previous update) on d = 8 features, so it stays on
the host like the reference's own CPU path (BASELINE.json configs[0]: "plumbing only"); no kernel is involved.
"""
import numpy as np
import torch
from torch.nn import Module

tensor_type = torch.DoubleTensor


class FM_Base(Module):
    def __init__(self, inputs_matrix, outputs, task, learning_rate, feature_m):
        super(FM_Base, self).__init__()
        self.At = inputs_matrix.t()          # [d, N]: column idx is sample idx (reference :21)
        self.b = outputs
        self._thres = 1e-12
        self.num_data = inputs_matrix.shape[0]
        self.num_feature = inputs_matrix.shape[1]
        self.task = task
        self.eta = learning_rate
        self.m = feature_m

    def _loss(self, x):
        """reference :34-41"""
        if self.task == "reg":
            return x ** 2
        if self.task == "cls":
            return 1 / (1 + torch.exp(x))
        return None

    def _grad_loss(self, x):
        """reg: 2x; cls: -1 / (1 + e^x)   (reference :44-51)"""
        if self.task == "reg":
            return 2.0 * x
        if self.task == "cls":
            return -1.0 / (1.0 + torch.exp(x))
        return None

    def _predict(self, scalar):
        """reg: (s, s); cls: (s, +-1 as a DoubleTensor of one element)   (reference :54-65)"""
        if self.task == "reg":
            return scalar, scalar
        if self.task == "cls":
            return scalar, torch.tensor([1.0 if scalar >= 0 else -1.0]).type(tensor_type)
        raise NotImplementedError

    def online_learning(self, logger=None):
        raise NotImplementedError
