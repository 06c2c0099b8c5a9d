# -*- coding: utf-8 -*-
"""DeepFMAdam -- drop-in for reference models/models_online_deep/deepfm_adam.py:12-159.

forward = forward_fm + sum relu-MLP(bi) (deepfm_adam.py:79-89; second_order is evaluated twice there, so the table
gradient is the FM-term gradient plus the MLP-input gradient); update_embedding: BCEwl(forward_fm) (:99-101);
fit: BCEwl(sigmoid(forward)) (:115)."""
from ._base import OnlineFMBase


class DeepFMAdam(OnlineFMBase):
    _name = "DeepFMAdam"
    _has_mlp = True
    _fm_term_in_forward = True
    _loss_update_embedding = "logits"
    _loss_fit = "sigmoid"

    def __init__(self, feature_sizes, embedding_size=4, num_hidden_layers=2, neuron_per_hidden_layer=32,
                 batch_size=1, num_classes=1, b=0.99, n=0.01, use_cuda=True, **fmx_options):
        super().__init__(feature_sizes, embedding_size=embedding_size, num_hidden_layers=num_hidden_layers,
                         neuron_per_hidden_layer=neuron_per_hidden_layer, batch_size=batch_size,
                         num_classes=num_classes, b=b, n=n, use_cuda=use_cuda, **fmx_options)
