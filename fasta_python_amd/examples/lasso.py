"""min_x .5*||Ax - b||^2 subject to ||x||_1 <= mu (constrained LASSO), 6-argument call form.
Recipe: fasta/examples/lasso.py:42-45 (closures), :51-79 (construct; mu is scaled by ||x||_1)."""

import numpy as np
from numpy import linalg as la

from .. import DenseMatrixMap, L1Ball, LeastSquares, LinearMap, fasta, proximal
from . import ExampleProblem, cli_backend, test_modes

__all__ = ["LASSOProblem"]


class LASSOProblem(ExampleProblem):
    def __init__(self, A, b, mu, x=None, backend="hip"):
        self.A, self.b, self.mu, self.x, self.backend = A, b, mu, x, backend

    def solve(self, x0, fasta_options=None):
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        if self.backend == "numpy":                 # the reference's closures (lasso.py:42-45), 6-argument call
            f = lambda z: .5 * la.norm((z - self.b).ravel()) ** 2
            gradf = lambda z: z - self.b
            g = lambda x: 0
            proxg = lambda x, t: proximal.project_L1_ball(x, self.mu)
            c = fasta(self.A, f, gradf, g, proxg, x0, **opts)
        else:
            op = self.A if isinstance(self.A, DenseMatrixMap) else self.device_operator(lambda: DenseMatrixMap(np.asarray(self.A)))
            loss, reg = LeastSquares(self.b), L1Ball(self.mu)
            c = fasta(op, loss.f, loss.gradf, reg.g, reg.prox, x0, backend="hip", **opts)
        return c.solution, c

    @staticmethod
    def construct(M=200, N=1000, K=10, sigma=0.01, mu=0.8, seed=None, backend="hip"):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        mu = mu * la.norm(x, 1)
        A = np.random.randn(M, N)
        A /= la.norm(A, 2)
        b = A @ x + sigma * np.random.randn(M)
        return LASSOProblem(A, b, mu, x=x, backend=backend), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = LASSOProblem.construct(backend=cli_backend())
    print("Constructed LASSO problem.")
    test_modes(problem, x0)
    problem.close()
