"""min_x .5*||Ax - b||^2 subject to ||x||_1 <= mu (constrained LASSO), 6-argument call form.
Recipe: fasta/examples/lasso.py:42-45 (closures), :51-79 (construct; mu is scaled by ||x||_1)."""

import numpy as np
from numpy import linalg as la

from .. import L1Ball, LeastSquares, LinearMap, fasta
from . import ExampleProblem, test_modes

__all__ = ["LASSOProblem"]


class LASSOProblem(ExampleProblem):
    def __init__(self, A, b, mu, x=None):
        self.A, self.b, self.mu, self.x = A, b, mu, x

    def solve(self, x0, fasta_options=None):
        loss, reg = LeastSquares(self.b), L1Ball(self.mu)
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        c = fasta(self.A, loss.f, loss.gradf, reg.g, reg.prox, x0, **opts)
        return c.solution, c

    @staticmethod
    def construct(M=200, N=1000, K=10, sigma=0.01, mu=0.8, seed=None):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        mu = mu * la.norm(x, 1)
        A = np.random.randn(M, N)
        A /= la.norm(A, 2)
        b = A @ x + sigma * np.random.randn(M)
        return LASSOProblem(LinearMap.from_matrix(A), b, mu, x=x), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = LASSOProblem.construct()
    print("Constructed LASSO problem.")
    test_modes(problem, x0)
    problem.close()
