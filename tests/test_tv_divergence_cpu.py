"""How sensitive is adaptive FBS on the TV dual to the ORDER of its float64 sums?  (the run backtracks every few
iterations -- fasta/__init__.py:195-217 -- and each backtracking decision compares two nearly equal numbers.)

The oracle is run against ITSELF on the TRANSPOSED problem: image transposed, the two components of the dual variable swapped.  TV is
symmetric under that map and the stencil and the prox are elementwise, so every iterate is the transpose of the original run's --
mathematically and, element by element, in floating point; only the ORDER in which the loop's dot products and norms add their
terms changes (L and tau0 are given, so no random probes are drawn).  The two runs agree for a while, then the step-size histories
part for good -- yet both converge to the same minimum.  tests/test_gpu_prox_tv.py measures the same thing for the HIP path
against the oracle; this test shows that the first-divergence index it reports is a property of the PROBLEM (any change of
summation order does it), not of the kernels."""
import warnings

import numpy as np
import pytest
from numpy import linalg as la

from oracle import fasta_np as fo
from oracle import problems as pr
from tests.helpers import first_divergence


def transposed(P):
    """the same TV problem seen through (i, j, c) -> (j, i, 1 - c)"""
    M, mu = P.data["M"], P.data["mu"]
    T = lambda Y: Y.transpose(1, 0, 2)[:, :, ::-1]               # (W,H,2) <-> (H,W,2), components swapped; its own inverse
    target = (M / mu).T.copy()
    A2 = lambda Y2: np.ascontiguousarray(P.A(np.ascontiguousarray(T(Y2))).T)
    At2 = lambda Z2: np.ascontiguousarray(T(P.At(np.ascontiguousarray(Z2.T))))
    f2 = lambda Z2: .5 * la.norm((Z2 - target).ravel()) ** 2
    gradf2 = lambda Z2: Z2 - target
    return A2, At2, f2, gradf2, P.g, P.proxg, np.zeros(M.T.shape + (2,)), T


@pytest.mark.parametrize("H,W,seed", [(32, 32, 21), (64, 80, 22)])
def test_oracle_against_itself_with_a_permuted_summation_order(H, W, seed):
    np.random.seed(seed)
    P = pr.tv_denoising(H=H, W=W, square=8)
    M, mu = P.data["M"], P.data["mu"]
    opts = dict(tolerance=1e-8, max_iters=3000, evaluate_objective=True, L=8.0, tau0=0.025)      # ||div||^2 <= 8; no RNG draws
    A2, At2, f2, gradf2, g2, proxg2, Y02, T = transposed(P)
    # the transposed operators ARE the originals, bit for bit, through the map
    Yr = np.random.RandomState(1).randn(H, W, 2)
    assert np.array_equal(A2(np.ascontiguousarray(T(Yr))).T, P.A(Yr)) and np.array_equal(proxg2(np.ascontiguousarray(T(Yr)), 1.0), np.ascontiguousarray(T(P.proxg(Yr, 1.0))))
    # (one BLAS thread: the loop makes ~10^5 norm / dot calls on 40-KiB arrays; OpenBLAS's worker threads spin between calls, and under a
    # CPU quota that turned these two solves from seconds into minutes)
    from threadpoolctl import threadpool_limits
    with warnings.catch_warnings(), threadpool_limits(limits=1, user_api="blas"):
        warnings.simplefilter("ignore")
        a = fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, P.x0, **opts)
        b = fo.fasta(A2, At2, f2, gradf2, g2, proxg2, Y02, **opts)
    k = min(a.iteration_count, b.iteration_count)
    first = first_divergence(a.stepsizes, b.stepsizes, k)
    print(f"\nTV {H}x{W}: oracle vs oracle on the transposed problem (same arithmetic, permuted sums): {a.iteration_count} / {b.iteration_count} "
          f"iterations, {a.backtracks} / {b.backtracks} backtracks, step sizes first differ (> 1e-6 relative) at iteration {first}")
    assert a.backtracks > 20 and b.backtracks > 20                        # the regime in question
    assert first >= 5                                                      # identical decisions to begin with
    assert first < k                                                       # ... but not for long: THAT is the problem's sensitivity
    # ... and the same minimum in the end, however different the paths
    fa_, fb_ = a.objectives[a.iteration_count], b.objectives[b.iteration_count]
    assert abs(fa_ - fb_) <= 1e-3 * abs(fb_)
    np.testing.assert_allclose(pr.tv_primal(M, mu, a.solution), pr.tv_primal(M, mu, np.ascontiguousarray(T(b.solution))), atol=2e-2)


@pytest.mark.parametrize("name", ["nnls_under_first40", "sparse_ls_unnormalised_backtracks"])
def test_dense_sensitive_runs_against_a_row_permuted_twin(name):
    """The same measurement for the dense fixtures that are pinned on a prefix (underdetermined NNLS with adaptive steps; the
    unnormalised matrix that backtracks 97 times): the oracle against itself with the ROWS of A and b permuted -- the same
    problem, every sum over the rows taken in another order.  Prints where the step sizes part."""
    from tests import helpers as H
    meta, z = H.load_case(name)
    d = H.case_data(meta, z)
    A, b = np.asarray(d["A"]), np.asarray(d["b"])
    perm = np.random.RandomState(1).permutation(A.shape[0])
    opts = dict(H.resolve_options(meta["options"], fo), max_iters=300, tolerance=0.0)
    runs = []
    for Ap, bp in ((A, b), (np.ascontiguousarray(A[perm]), b[perm])):
        P = pr.FROM_DATA[meta["kind"]](dict(d, A=Ap, b=bp))
        np.random.seed(meta["solver_seed"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runs.append(fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, P.x0, **opts))
    a, c = runs
    k = min(a.iteration_count, c.iteration_count)
    first = first_divergence(a.stepsizes, c.stepsizes, k)
    print(f"\n{name}: oracle vs oracle with permuted rows: {a.backtracks} / {c.backtracks} backtracks in {k} iterations, "
          f"step sizes first differ (> 1e-6 relative) at iteration {first}")
    assert first >= 5
