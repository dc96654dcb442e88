"""min_X mu*TV(X) + .5*||X - M||^2 through its dual, min_Y .5*||div(Y) - M/mu||^2 with ||Y_ij|| <= 1
-- BASELINE config 4.  Recipe: fasta/examples/tv_denoising.py:85-103 (solve), :105-125 (construct).
`scipy.misc.ascent` no longer exists (and `scipy.datasets` needs a download), so the clean image is a
synthetic {0,1} checkerboard; noise level, mu and the zero dual start follow the reference."""

import numpy as np
from numpy import linalg as la

from .. import GradDivMap, LeastSquares, TVDualBall, fasta
from . import ExampleProblem, cli_backend, test_modes

__all__ = ["TVDenoisingProblem", "checkerboard", "grad", "div"]


def grad(X):
    """Periodic discrete gradient of an N-d array -> (N+1)-d (tv_denoising.py:26-40): roll(X, +1, axis) - X per axis."""
    out = np.zeros(X.shape + (X.ndim,))
    for axis in range(X.ndim):
        out[..., axis] = np.roll(X, 1, axis=axis) - X
    return out


def div(Y):
    """Adjoint of `grad` (tv_denoising.py:43-63): sum over axes of roll(Y[..., axis], -1, axis) - Y[..., axis]."""
    assert Y.shape[-1] == Y.ndim - 1
    out = np.zeros(Y.shape[:-1])
    for axis in range(Y.shape[-1]):
        comp = Y[..., axis]
        out += np.roll(comp, -1, axis=axis) - comp
    return out


def checkerboard(H, W, square):
    ii, jj = np.indices((H, W))
    return (((ii // square) + (jj // square)) % 2).astype(float)


class TVDenoisingProblem(ExampleProblem):
    def __init__(self, M, mu, backend="hip"):
        self.M, self.mu, self.backend = M, mu, backend

    def solve(self, Y0, fasta_options=None):
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        if self.backend == "numpy":                 # the reference's closures and bare-function operator pair (tv_denoising.py:85-99)
            f = lambda Z: .5 * la.norm((Z - self.M / self.mu).ravel()) ** 2
            gradf = lambda Z: Z - self.M / self.mu
            g = lambda Y: 0

            def proxg(Y, t):
                lengths = np.maximum(la.norm(Y, axis=Y.ndim - 1), 1)
                return Y / lengths[..., np.newaxis]

            c = fasta(div, grad, f, gradf, g, proxg, Y0, **opts)
            return self.M - self.mu * div(c.solution), c            # tv_denoising.py:101
        op = self.device_operator(lambda: GradDivMap(self.M.shape))
        loss, reg = LeastSquares(self.M / self.mu), TVDualBall()
        c = fasta(op, op.H, loss.f, loss.gradf, reg.g, reg.prox, Y0, backend="hip", **opts)
        return self.M - self.mu * op(c.solution), c

    @staticmethod
    def construct(sigma=0.1, mu=0.1, shape=(512, 512), square=64, seed=None, backend="hip"):
        if seed is not None:
            np.random.seed(seed)
        M = checkerboard(shape[0], shape[1], square)
        M /= np.max(M)
        M += sigma * np.random.randn(*M.shape)
        return TVDenoisingProblem(M, mu, backend=backend), np.zeros(M.shape + (2,))


if __name__ == "__main__":
    problem, Y0 = TVDenoisingProblem.construct(backend=cli_backend())
    print("Constructed total-variation denoising problem.")
    test_modes(problem, Y0, {"max_iters": 300})
    problem.close()
