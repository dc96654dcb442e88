"""min_x mu*||x||_1 + .5*||Ax - b||^2 (basis pursuit denoising) -- BASELINE configs 1, 2, 5.
Recipe: fasta/examples/sparse_least_squares.py:41-44 (closures), :50-76 (construct)."""

import numpy as np
from numpy import linalg as la

from .. import DenseMatrixMap, LeastSquares, ShardedDenseMatrixMap, Shrink, fasta, proximal
from . import ExampleProblem, cli_backend, cli_devices, test_modes

__all__ = ["SparseLeastSquaresProblem"]


class SparseLeastSquaresProblem(ExampleProblem):
    def __init__(self, A, At, b, mu, x=None, backend="hip", devices=None):
        """devices (backend "hip"): list of device ids -- A is row-sharded over them in this one process (ShardedDenseMatrixMap;
        BASELINE config 5's structure); None: one GPU."""
        self.A, self.At, self.b, self.mu, self.x, self.backend, self.devices = A, At, b, mu, x, backend, devices

    def solve(self, x0, fasta_options=None):
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        if self.backend == "numpy":                 # the reference's closures (sparse_least_squares.py:41-44)
            f = lambda z: .5 * la.norm((z - self.b).ravel()) ** 2
            gradf = lambda z: z - self.b
            g = lambda x: self.mu * la.norm(x.ravel(), 1)
            proxg = lambda x, t: proximal.shrink(x, t * self.mu)
            c = fasta(self.A, self.At, f, gradf, g, proxg, x0, **opts)
        else:
            make = ((lambda: ShardedDenseMatrixMap(np.asarray(self.A), devices=self.devices)) if self.devices
                    else (lambda: DenseMatrixMap(np.asarray(self.A))))
            op = self.A if isinstance(self.A, DenseMatrixMap) else self.device_operator(make)
            loss, reg = LeastSquares(self.b), Shrink(self.mu)
            c = fasta(op, op.H, loss.f, loss.gradf, reg.g, reg.prox, x0, backend="hip", **opts)
        return c.solution, c

    @staticmethod
    def construct(M=200, N=1000, K=10, sigma=0.01, mu=0.02, seed=None, backend="hip", devices=None):
        if seed is not None:
            np.random.seed(seed)
        x = np.zeros(N)
        x[np.random.permutation(N)[:K]] = 1
        A = np.random.randn(M, N)
        A /= la.norm(A, 2)
        b = A @ x + sigma * np.random.randn(M)
        return SparseLeastSquaresProblem(A, A.T, b, mu, x=x, backend=backend, devices=devices), np.zeros(N)


if __name__ == "__main__":
    problem, x0 = SparseLeastSquaresProblem.construct(backend=cli_backend(), devices=cli_devices())
    print("Constructed sparse least squares problem.")
    adaptive, accelerated, plain = test_modes(problem, x0)
    print("recovery error ||x - x_true||_inf = {:.3e}".format(np.abs(adaptive[0] - problem.x).max()))
    problem.close()
