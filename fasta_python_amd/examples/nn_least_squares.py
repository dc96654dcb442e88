"""min_x .5*||Ax - b||^2 subject to x >= 0 -- BASELINE config 3.
Recipe: fasta/examples/nn_least_squares.py:39-42 (closures), :46-72 (construct)."""

import numpy as np
from numpy import linalg as la

from .. import DenseMatrixMap, LeastSquares, NonNeg, fasta
from . import ExampleProblem, test_modes

__all__ = ["NNLeastSquaresProblem"]


class NNLeastSquaresProblem(ExampleProblem):
    def __init__(self, A, At, b, x=None):
        self.A = A if isinstance(A, DenseMatrixMap) else DenseMatrixMap(np.asarray(A))
        self.At = self.A.H
        self.b, self.x = b, x

    def solve(self, x0, fasta_options=None):
        loss, reg = LeastSquares(self.b), NonNeg()
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        c = fasta(self.A, self.At, loss.f, loss.gradf, reg.g, reg.prox, x0, **opts)
        return c.solution, c

    @staticmethod
    def construct(M=200, N=1000, K=10, sigma=0.005, seed=None):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        A = np.random.randn(M, N)
        A /= la.norm(A, 2)
        b = A @ x + sigma * np.random.randn(M)
        return NNLeastSquaresProblem(A, A.T, b, x=x), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = NNLeastSquaresProblem.construct()
    print("Constructed non-negative least squares problem.")
    test_modes(problem, x0)
    problem.close()
