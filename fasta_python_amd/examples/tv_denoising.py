"""min_X mu*TV(X) + .5*||X - M||^2 through its dual, min_Y .5*||div(Y) - M/mu||^2 with ||Y_ij|| <= 1
-- BASELINE config 4.  Recipe: fasta/examples/tv_denoising.py:85-103 (solve), :105-125 (construct).
`scipy.misc.ascent` no longer exists (and `scipy.datasets` needs a download), so the clean image is a
synthetic {0,1} checkerboard; noise level, mu and the zero dual start follow the reference."""

import numpy as np

from .. import GradDivMap, LeastSquares, TVDualBall, fasta
from . import ExampleProblem, test_modes

__all__ = ["TVDenoisingProblem", "checkerboard"]


def checkerboard(H, W, square):
    ii, jj = np.indices((H, W))
    return (((ii // square) + (jj // square)) % 2).astype(float)


class TVDenoisingProblem(ExampleProblem):
    def __init__(self, M, mu):
        self.M, self.mu = M, mu
        self.A = GradDivMap(M.shape)

    def solve(self, Y0, fasta_options=None):
        loss, reg = LeastSquares(self.M / self.mu), TVDualBall()
        opts = dict(verbose=False)
        opts.update(fasta_options or {})
        c = fasta(self.A, self.A.H, loss.f, loss.gradf, reg.g, reg.prox, Y0, **opts)
        X = self.M - self.mu * self.A(c.solution)          # tv_denoising.py:101
        return X, c

    @staticmethod
    def construct(sigma=0.1, mu=0.1, shape=(512, 512), square=64, seed=None):
        if seed is not None:
            np.random.seed(seed)
        M = checkerboard(shape[0], shape[1], square)
        M /= np.max(M)
        M += sigma * np.random.randn(*M.shape)
        return TVDenoisingProblem(M, mu), np.zeros(M.shape + (2,))


if __name__ == "__main__":
    problem, Y0 = TVDenoisingProblem.construct()
    print("Constructed total-variation denoising problem.")
    test_modes(problem, Y0, {"max_iters": 300})
    problem.close()
