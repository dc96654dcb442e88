"""min_x mu*||x||_1 + sum_i log(1+exp(a_i.x)) - [b_i==1] a_i.x  (sparse logistic regression).
Recipe: fasta/examples/sparse_logistic.py:47-50 (closures), :54-80 (construct; A is NOT normalised there)."""

import numpy as np

from .. import DenseMatrixMap, LogisticLoss, Shrink, fasta
from . import ExampleProblem, test_modes

__all__ = ["SparseLogisticProblem"]


class SparseLogisticProblem(ExampleProblem):
    def __init__(self, A, At, b, mu, x=None):
        self.A = A if isinstance(A, DenseMatrixMap) else DenseMatrixMap(np.asarray(A))
        self.At = self.A.H
        self.b, self.mu, self.x = b, mu, x

    def solve(self, x0, fasta_options=None):
        loss, reg = LogisticLoss(self.b), Shrink(self.mu)
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        c = fasta(self.A, self.At, loss.f, loss.gradf, reg.g, reg.prox, x0, **opts)
        return c.solution, c

    @staticmethod
    def construct(M=1000, N=2000, K=5, mu=40, seed=None):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        A = np.random.randn(M, N)
        p = 1 / (1 + np.exp(-A @ x))
        b = 2.0 * (np.random.rand(M) < p) - 1
        return SparseLogisticProblem(A, A.T, b, mu, x=x), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = SparseLogisticProblem.construct()
    print("Constructed sparse logistic problem.")
    test_modes(problem, x0)
    problem.close()
