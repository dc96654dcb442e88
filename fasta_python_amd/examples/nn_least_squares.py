"""min_x .5*||Ax - b||^2 subject to x >= 0 -- BASELINE config 3.
Recipe: fasta/examples/nn_least_squares.py:39-42 (closures), :46-72 (construct)."""

import numpy as np
from numpy import linalg as la

from .. import DenseMatrixMap, LeastSquares, NonNeg, fasta
from . import ExampleProblem, cli_backend, test_modes

__all__ = ["NNLeastSquaresProblem"]


class NNLeastSquaresProblem(ExampleProblem):
    def __init__(self, A, At, b, x=None, backend="hip"):
        self.A, self.At, self.b, self.x, self.backend = A, At, b, x, backend

    def solve(self, x0, fasta_options=None):
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        if self.backend == "numpy":                 # the reference's closures (nn_least_squares.py:39-42)
            f = lambda z: .5 * la.norm((z - self.b).ravel()) ** 2
            gradf = lambda z: z - self.b
            g = lambda x: 0
            proxg = lambda x, t: np.maximum(x, 0)
            c = fasta(self.A, self.At, f, gradf, g, proxg, x0, **opts)
        else:
            op = self.A if isinstance(self.A, DenseMatrixMap) else self.device_operator(lambda: DenseMatrixMap(np.asarray(self.A)))
            loss, reg = LeastSquares(self.b), NonNeg()
            c = fasta(op, op.H, loss.f, loss.gradf, reg.g, reg.prox, x0, backend="hip", **opts)
        return c.solution, c

    @staticmethod
    def construct(M=200, N=1000, K=10, sigma=0.005, seed=None, backend="hip"):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        A = np.random.randn(M, N)
        A /= la.norm(A, 2)
        b = A @ x + sigma * np.random.randn(M)
        return NNLeastSquaresProblem(A, A.T, b, x=x, backend=backend), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = NNLeastSquaresProblem.construct(backend=cli_backend())
    print("Constructed non-negative least squares problem.")
    test_modes(problem, x0)
    problem.close()
