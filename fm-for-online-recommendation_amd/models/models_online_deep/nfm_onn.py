# -*- coding: utf-8 -*-
"""NFMOnn -- drop-in for reference models/models_online_deep/nfm_onn.py:13-212.

As DeepFMOnn, but the per-layer logit has no sum_d bi_d term (nfm_onn.py:90-104) and update_embedding uses
BCEwl(sigmoid(forward_fm)) (:168).  Note the reference's argument order: num_classes before batch_size (:14-15)."""
from ._base import OnlineFMBase


class NFMOnn(OnlineFMBase):
    _name = "NFMOnn"
    _has_mlp = True
    _onn = True
    _fm_term_in_forward = False
    _loss_update_embedding = "sigmoid"

    def __init__(self, feature_sizes, embedding_size=4, num_hidden_layers=2, neuron_per_hidden_layer=32, num_classes=1,
                 batch_size=1, b=0.99, n=0.01, s=0.2, use_cuda=True, **fmx_options):
        super().__init__(feature_sizes, embedding_size=embedding_size, num_hidden_layers=num_hidden_layers,
                         neuron_per_hidden_layer=neuron_per_hidden_layer, batch_size=batch_size,
                         num_classes=num_classes, b=b, n=n, s=s, use_cuda=use_cuda, **fmx_options)
