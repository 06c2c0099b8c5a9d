# -*- coding: utf-8 -*-
"""DeepFMOnn -- drop-in for reference models/models_online_deep/deepfm_onn.py:12-210.

forward returns (last, [L,B]) with per-layer sigmoid(forward_fm + sum x_l) (deepfm_onn.py:88-102); fit is Hedge
backprop on the hidden layers and alpha only (:109-154); update_embedding: BCEwl(forward_fm) (:164-166);
predict applies a second sigmoid (:173-175)."""
from ._base import OnlineFMBase


class DeepFMOnn(OnlineFMBase):
    _name = "DeepFMOnn"
    _has_mlp = True
    _onn = True
    _fm_term_in_forward = True
    _loss_update_embedding = "logits"

    def __init__(self, feature_sizes, embedding_size=4, num_hidden_layers=2, neuron_per_hidden_layer=32,
                 batch_size=1, num_classes=1, b=0.99, n=0.01, s=0.2, use_cuda=True, **fmx_options):
        super().__init__(feature_sizes, embedding_size=embedding_size, num_hidden_layers=num_hidden_layers,
                         neuron_per_hidden_layer=neuron_per_hidden_layer, batch_size=batch_size,
                         num_classes=num_classes, b=b, n=n, s=s, use_cuda=use_cuda, **fmx_options)

    def _logit_shape(self, B):
        return (B,)
