"""min_x mu*||x||_1 + sum_i log(1+exp(a_i.x)) - [b_i==1] a_i.x  (sparse logistic regression).
Recipe: fasta/examples/sparse_logistic.py:47-50 (closures), :54-80 (construct; A is NOT normalised there)."""

import numpy as np
from numpy import linalg as la

from .. import DenseMatrixMap, LogisticLoss, Shrink, fasta, proximal
from . import ExampleProblem, cli_backend, test_modes

__all__ = ["SparseLogisticProblem"]


class SparseLogisticProblem(ExampleProblem):
    def __init__(self, A, At, b, mu, x=None, backend="hip"):
        self.A, self.At, self.b, self.mu, self.x, self.backend = A, At, b, mu, x, backend

    def solve(self, x0, fasta_options=None):
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        if self.backend == "numpy":                 # the reference's closures (sparse_logistic.py:47-50)
            f = lambda z: np.sum(np.log(1 + np.exp(z)) - (self.b == 1) * z)
            gradf = lambda z: -self.b / (1 + np.exp(self.b * z))
            g = lambda x: self.mu * la.norm(x.ravel(), 1)
            proxg = lambda x, t: proximal.shrink(x, t * self.mu)
            c = fasta(self.A, self.At, f, gradf, g, proxg, x0, **opts)
        else:
            op = self.A if isinstance(self.A, DenseMatrixMap) else self.device_operator(lambda: DenseMatrixMap(np.asarray(self.A)))
            loss, reg = LogisticLoss(self.b), Shrink(self.mu)
            c = fasta(op, op.H, loss.f, loss.gradf, reg.g, reg.prox, x0, backend="hip", **opts)
        return c.solution, c

    @staticmethod
    def construct(M=1000, N=2000, K=5, mu=40, seed=None, backend="hip"):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        A = np.random.randn(M, N)
        p = 1 / (1 + np.exp(-A @ x))
        b = 2.0 * (np.random.rand(M) < p) - 1
        return SparseLogisticProblem(A, A.T, b, mu, x=x, backend=backend), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = SparseLogisticProblem.construct(backend=cli_backend())
    print("Constructed sparse logistic problem.")
    test_modes(problem, x0)
    problem.close()
