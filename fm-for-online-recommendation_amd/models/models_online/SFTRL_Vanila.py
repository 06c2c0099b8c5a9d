"""SFTRL_Vanila -- drop-in for reference models/models_online/SFTRL_Vanila.py:16-130.

As SFTRL_CCFM plus a linear term w = -eta * sum s x (:59-60); the sketches see the feature vector without its last
entry (:43-44,62)."""
import numpy as np
import torch

from models.models_online.SFTRL_CCFM import SFTRL_CCFM

Tensor_type = torch.DoubleTensor


class SFTRL_Vanila(SFTRL_CCFM):
    _linear_term = True

    def __init__(self, inputs_matrix, outputs, task, learning_rate, num_feature, device="host"):
        super(SFTRL_Vanila, self).__init__(inputs_matrix, outputs, task, learning_rate, num_feature, device=device)
        self.model_name = "SFTRL_Vanila"
        self.w = Tensor_type(np.zeros([self.num_feature, 1]))
        self.g_w = Tensor_type(np.zeros([self.num_feature, 1]))

    def _sketch_dim(self):
        return self.num_feature - 1

    def _pred_shape(self, cls):
        return (1,) if cls else (1, 1)
