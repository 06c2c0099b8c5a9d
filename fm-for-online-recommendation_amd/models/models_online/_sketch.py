"""Frequent-directions sketch shared by SFTRL_CCFM and SFTRL_Vanila (reference `_GFD`, SFTRL_CCFM.py:77-121).

A [d, 2m] buffer of directions; a new direction goes to column `count + 1` (the reference pre-increments, so column 0
stays zero); when column 2m-1 has been written the buffer is shrunk through the SVD of its 2m x 2m Gram matrix:
    Sigma = eigenvalues (squared singular values), those <= thres zeroed, nnz = how many remain
    V = B U[:, :nnz] diag(Sigma^-1/2)                                   (left singular vectors)
    nnz >= m : B <- V[:, :m-1] diag(sqrt(Sigma[:m-1] - Sigma[m])), count = m-1
    else     : B <- V[:, :nnz] diag(sqrt(Sigma[:nnz])),            count = nnz
fp64 on the host: path B is strictly sequential on d = 8 features (out of scope for kernels).
"""
import numpy as np


class Sketch:
    def __init__(self, dim, m, thres=1e-12):
        self.dim, self.m, self.thres = dim, m, thres
        self.B = np.zeros((dim, 2 * m), dtype=np.float64)
        self.count = 0

    def energy(self, x):
        """||B^T x||^2"""
        t = self.B.T @ x
        return float(t @ t)

    def append(self, vec):
        m = self.m
        self.count += 1
        self.B[:, self.count] = vec
        if self.count != 2 * m - 1:
            return
        U, sigma, _ = np.linalg.svd(self.B.T @ self.B)
        sigma = np.where(sigma <= self.thres, 0.0, sigma)
        nnz = int(np.count_nonzero(sigma))
        V = (self.B @ U[:, :nnz]) / np.sqrt(sigma[:nnz])[None, :]
        if nnz >= m:
            kept = V[:, :m - 1] * np.sqrt(sigma[:m - 1] - sigma[m])[None, :]
            self.count = m - 1
        else:
            kept = V[:, :nnz] * np.sqrt(sigma[:nnz])[None, :]
            self.count = nnz
        self.B = np.zeros((self.dim, 2 * m), dtype=np.float64)
        self.B[:, :kept.shape[1]] = kept
