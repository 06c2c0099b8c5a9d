# -*- coding: utf-8 -*-
"""FMAdam -- drop-in for reference models/models_online_deep/fm_adam.py:12-123 on the gfx950 kernels.

forward = sum_f first + sum_d bi_d + bias (fm_adam.py:53); update_embedding trains on BCEwl(logit) (:66),
fit on BCEwl(sigmoid(logit)) (:80); both with the fresh-Adam rule (:60,:75)."""
from ._base import OnlineFMBase


class FMAdam(OnlineFMBase):
    _name = "FMAdam"
    _has_mlp = False
    _loss_update_embedding = "logits"
    _loss_fit = "sigmoid"

    def __init__(self, feature_sizes, embedding_size=4, num_classes=1, b=0.99, n=0.01, use_cuda=True, **fmx_options):
        super().__init__(feature_sizes, embedding_size=embedding_size, num_classes=num_classes, b=b, n=n,
                         use_cuda=use_cuda, **fmx_options)
