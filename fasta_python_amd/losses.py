"""Smooth terms f(z) recognised by the device loop.

LogisticLoss(b): see the class.  LeastSquares(b):  f(z) = .5*||z - b||^2,  gradf(z) = z - b   (examples/sparse_least_squares.py:41-42,
same closures in lasso.py:42-43, nn_least_squares.py:39-40, tv_denoising.py:85-86 with b = M/mu).
Pass `ls.f` and `ls.gradf` as the `f` / `gradf` arguments of `fasta()`.
"""

import math

import numpy as np

__all__ = ["LeastSquares", "LogisticLoss"]


class LeastSquares:
    def __init__(self, b):
        self.b = np.ascontiguousarray(b, dtype=np.float64)

    def bind(self, ctx):
        ctx.set_loss_lsq(self.b)

    @staticmethod
    def f_from_device(s):
        """f1 from the device scalar sum (z-b)^2: .5*la.norm(z-b)**2 (sparse_least_squares.py:41)."""
        return .5 * np.float64(math.sqrt(s)) ** 2          # (math.sqrt: same IEEE result, a fraction of np.sqrt's call cost)

    # The device loop evaluates f inside its kernels and never calls these.  On host arrays they are the reference's
    # closures (sparse_least_squares.py:41-42), bit for bit, so the generic host loop -- or the reference -- can use them.
    def f(self, z):
        return .5 * np.linalg.norm((z - self.b).ravel()) ** 2

    def gradf(self, z):
        return z - self.b

    __call__ = f


class LogisticLoss:
    """f(z) = sum log(1+exp(z)) - (b==1)*z, gradf(z) = -b/(1+exp(b*z)) with labels b in {-1,+1}
    (examples/sparse_logistic.py:47-48).  Dense operators only."""

    def __init__(self, b):
        self.b = np.ascontiguousarray(b, dtype=np.float64)

    def bind(self, ctx):
        ctx.set_loss_logistic(self.b)

    @staticmethod
    def f_from_device(s):
        return np.float64(s)

    def f(self, z):
        return np.sum(np.log(1 + np.exp(z)) - (self.b == 1) * z)

    def gradf(self, z):
        return -self.b / (1 + np.exp(self.b * z))

    __call__ = f
