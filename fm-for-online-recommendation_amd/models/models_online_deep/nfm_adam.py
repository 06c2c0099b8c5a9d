# -*- coding: utf-8 -*-
"""NFMAdam -- drop-in for reference models/models_online_deep/nfm_adam.py:12-158.

forward = sum_f first + bias + sum relu-MLP(bi), WITHOUT the sum_d bi_d term (nfm_adam.py:78-88);
update_embedding: BCEwl(sigmoid(forward_fm)) (:100); fit: BCEwl(forward) (:114)."""
from ._base import OnlineFMBase


class NFMAdam(OnlineFMBase):
    _name = "NFMAdam"
    _has_mlp = True
    _fm_term_in_forward = False
    _loss_update_embedding = "sigmoid"
    _loss_fit = "logits"

    def __init__(self, feature_sizes, embedding_size=4, num_hidden_layers=2, neuron_per_hidden_layer=32,
                 num_classes=1, b=0.99, n=0.01, use_cuda=True, **fmx_options):
        super().__init__(feature_sizes, embedding_size=embedding_size, num_hidden_layers=num_hidden_layers,
                         neuron_per_hidden_layer=neuron_per_hidden_layer, num_classes=num_classes, b=b, n=n,
                         use_cuda=use_cuda, **fmx_options)
