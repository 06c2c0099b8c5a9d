"""Reference-faithful DENSE mode of the pure-FM mini-batch step on the CPU -- TEST
INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/): what the reference's FMAdam.update_embedding costs when it is
run the way the reference runs it (models/models_online_deep/fm_adam.py:26-33,56-69): 2 x F nn.Embedding tables, the
forward as F gathers and two F-term sums, loss.backward() materialising DENSE [feature_size, k] gradients for every table,
and a NEW torch.optim.Adam over ALL parameters each step (whose first step is p -= lr g / (|g| + 1e-8) on every entry,
touched or not).  The row-sparse port (fm_oracle.c) is the fair CPU algorithm; this one shows where the reference's
own time goes (Adam.step 63 %, embedding_dense_backward 23 %).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class DenseFM(nn.Module):
    def __init__(self, feature_sizes, k, lr):
        super().__init__()
        self.first = nn.ModuleList([nn.Embedding(s, 1) for s in feature_sizes])
        self.second = nn.ModuleList([nn.Embedding(s, k) for s in feature_sizes])
        self.bias = nn.Parameter(torch.tensor([0.99]))
        self.lr = lr

    def logits(self, Xi, Xv):
        # Xi [B, F] int64, Xv [B, F] float32
        first = [(emb(Xi[:, i]).sum(1) * Xv[:, i]) for i, emb in enumerate(self.first)]
        second = [(emb(Xi[:, i]) * Xv[:, i:i + 1]) for i, emb in enumerate(self.second)]
        s = sum(second)
        ss = sum(e * e for e in second)
        return sum(first) + 0.5 * (s * s - ss).sum(1) + self.bias

    def step(self, Xi, Xv, Y):
        opt = torch.optim.Adam(self.parameters(), lr=self.lr)      # a fresh optimizer per call, as in the reference
        opt.zero_grad()
        loss = F.binary_cross_entropy_with_logits(self.logits(Xi, Xv), Y)
        loss.backward()
        opt.step()
        return float(loss.detach())


def time_dense_steps(feature_sizes, k, idx_pool, y_pool, n_steps=3, lr=0.01, threads=None):
    """-> (samples/s, threads) over n_steps mini-batch steps of the pool (after one untimed step)."""
    import time
    if threads:
        torch.set_num_threads(threads)
    m = DenseFM(feature_sizes, k, lr)
    B = idx_pool.shape[1]
    Xv = torch.ones((B, len(feature_sizes)))
    batches = [(torch.from_numpy(idx_pool[j].astype(np.int64)), torch.from_numpy(y_pool[j])) for j in range(idx_pool.shape[0])]
    m.step(batches[0][0], Xv, batches[0][1])
    t0 = time.perf_counter()
    for s in range(n_steps):
        Xi, Y = batches[(s + 1) % len(batches)]
        m.step(Xi, Xv, Y)
    dt = time.perf_counter() - t0
    return n_steps * B / dt, torch.get_num_threads()
