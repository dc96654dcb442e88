"""Smooth terms f(z) recognised by the device loop.

LeastSquares(b):  f(z) = .5*||z - b||^2,  gradf(z) = z - b   (examples/sparse_least_squares.py:41-42,
same closures in lasso.py:42-43, nn_least_squares.py:39-40, tv_denoising.py:85-86 with b = M/mu).
Pass `ls.f` and `ls.gradf` as the `f` / `gradf` arguments of `fasta()`.
"""

import numpy as np

__all__ = ["LeastSquares"]


class LeastSquares:
    def __init__(self, b):
        self.b = np.ascontiguousarray(b, dtype=np.float64)

    # The device loop evaluates f inside K-fwd/K-adj; these host forms exist so the object can be
    # inspected or handed to other code.  They are never called by fasta().
    def f(self, z):
        r = np.asarray(z, dtype=np.float64) - self.b
        return .5 * float(np.vdot(r, r))

    def gradf(self, z):
        return np.asarray(z, dtype=np.float64) - self.b

    __call__ = f
