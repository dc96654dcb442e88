"""Randomised option/shape coverage: HIP solve vs the oracle loop on small ragged problems (every iteration,
every mode), plus the degenerate corners of the option space the reference allows."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from oracle import fasta_np as fo
from oracle import problems as pr

pytestmark = pytest.mark.gpu


def _pair(A, b, kind, mu, x0, opts, seed):
    if kind == "shrink":
        reg, P = fa.Shrink(mu), pr.sparse_least_squares_from(A, b, mu)
    elif kind == "nonneg":
        reg, P = fa.NonNeg(), pr.nn_least_squares_from(A, b)
    else:
        reg, P = fa.L1Ball(mu), pr.l1_ball_lasso_from(A, b, mu)
    ls = fa.LeastSquares(b)
    np.random.seed(seed)
    got = fa.fasta(A, A.T, ls.f, ls.gradf, reg.g, reg.prox, x0, verbose=False, backend="hip", **opts)
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, x0, **opts)
    return got, want


def _same(got, want, rtol=1e-6, rtol_steps=None):
    assert got.iteration_count == want.iteration_count
    assert got.backtracks == want.backtracks
    k = got.iteration_count
    for f in ("residuals", "norm_residuals", "stepsizes"):
        r = rtol_steps if (f == "stepsizes" and rtol_steps) else rtol
        np.testing.assert_allclose(getattr(got, f)[:k], getattr(want, f)[:k], rtol=r, atol=1e-14, err_msg=f)
    if want.objectives is not None:
        np.testing.assert_allclose(got.objectives[:k + 1], want.objectives[:k + 1], rtol=rtol, atol=1e-14)
    # the solution after up to 60 iterations: rtol 1e-5 per element, with an absolute floor of 1e-6 of the vector's largest entry -- an entry that the
    # prox has pulled close to zero carries the rounding of the whole column sum (order of summation: teams x rows), not 1e-5 of ITS OWN size
    np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-9 + 1e-6 * float(np.max(np.abs(want.solution), initial=0.0)))
    assert got.residuals.shape == want.residuals.shape and got.times.shape == want.times.shape     # untruncated histories


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_and_modes(seed):
    rng = np.random.RandomState(1000 + seed)
    m, n = int(rng.randint(1, 180)), int(rng.randint(1, 260))
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = rng.randn(m)
    x0 = rng.randn(n) * (0.1 if seed % 2 else 0.0)                 # zero and non-zero starts
    kind = ("shrink", "nonneg", "l1ball")[seed % 3]
    adaptive, accelerate = [(True, False), (False, True), (False, False), (True, True)][seed % 4]
    opts = dict(adaptive=adaptive, accelerate=accelerate, max_iters=60, tolerance=1e-7,
                evaluate_objective=bool(seed % 2), window=int(rng.randint(1, 12)),
                restart=bool((seed // 2) % 2))
    got, want = _pair(A, b, kind, 0.05, x0, opts, seed)
    _same(got, want)


def test_zero_iterations_returns_the_start():
    rng = np.random.RandomState(0)
    A, b, x0 = rng.randn(6, 9), rng.randn(6), rng.randn(9)
    got, want = _pair(A, b, "shrink", 0.1, x0, dict(max_iters=0), 0)
    assert got.iteration_count == want.iteration_count == 0
    assert np.array_equal(got.solution, x0) and got.residuals.shape == (0,)


def test_stop_at_first_iteration_and_no_backtrack_budget():
    rng = np.random.RandomState(1)
    A, b, x0 = rng.randn(20, 15), rng.randn(20), np.zeros(15)
    got, want = _pair(A, b, "shrink", 0.1, x0, dict(tolerance=1e30, max_iters=5), 1)
    assert got.iteration_count == want.iteration_count == 1
    # unnormalised A + explicit large step: the backtracking test fires but the budget is zero (fasta/__init__.py:201)
    got, want = _pair(A, b, "shrink", 0.1, x0, dict(L=1.0, tau0=5.0, max_backtracks=0, max_iters=4, tolerance=0.0), 1)
    assert got.backtracks == want.backtracks == 0
    np.testing.assert_allclose(got.residuals[:4], want.residuals[:4], rtol=1e-6)


def test_custom_stop_rule_and_matrix_valued_iterate_shapes():
    """stop_rule is an arbitrary host callable with the reference signature (fasta/stopping.py); x0 may be any
    shape the operator declares (here the (H,W,2) dual variable of TV)."""
    calls = []

    def rule(i, resid, norm_resid, max_resid, tol):
        calls.append((i, resid, norm_resid, max_resid, tol))
        return i >= 2
    np.random.seed(3)
    P = pr.tv_denoising(H=12, W=10, square=4)
    op = fa.GradDivMap((12, 10))
    try:
        ls, reg = fa.LeastSquares(P.data["M"] / P.data["mu"]), fa.TVDualBall()
        np.random.seed(4)
        c = fa.fasta(op, op.H, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", stop_rule=rule, tolerance=0.5)
    finally:
        op.close()
    assert c.iteration_count == 3 and [a[0] for a in calls] == [0, 1, 2] and calls[0][4] == 0.5
    assert c.solution.shape == (12, 10, 2)


@pytest.mark.parametrize("seed", range(10))
def test_random_stencil_shapes_and_modes(seed):
    """TV dual on random image shapes (1 x 1 up to a few strips wide), every mode, non-zero starts, record_iterates
    (which materialises the lazily-kept FISTA iterate every iteration): the default one-pass sweeps against the oracle's
    roll-based loop, 25 iterations (before the adaptive runs turn chaotic)."""
    rng = np.random.RandomState(2000 + seed)
    H_, W_ = int(rng.randint(1, 140)), int(rng.randint(1, 260))
    M = rng.randn(H_, W_)
    mu = 0.2 + rng.rand()
    Y0 = rng.randn(H_, W_, 2) * (0.4 if seed % 2 else 0.0)
    adaptive, accelerate = [(True, False), (False, True), (False, False), (True, True)][seed % 4]
    opts = dict(adaptive=adaptive, accelerate=accelerate, max_iters=25, tolerance=0.0, evaluate_objective=bool(seed % 3),
                record_iterates=bool(seed % 2), restart=bool((seed // 2) % 2), window=int(rng.randint(1, 12)))
    P = pr.tv_denoising_from(M, mu)
    no_prox = seed in (3, 8)                   # g = None: plain gradient descent on the stencil operator (fasta/__init__.py:88-90)
    op = fa.GradDivMap((H_, W_))
    try:
        ls, reg = fa.LeastSquares(M / mu), (fa.NoProx() if no_prox else fa.TVDualBall())
        solver = fa.FBSolver(op, ls, reg, Y0, verbose=False, **opts)
        np.random.seed(seed)
        got = solver.setup().run()
        assert solver.fused_steps == got.iteration_count + got.backtracks        # every launch was a one-pass sweep
    finally:
        op.close()
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(P.A, P.At, P.f, P.gradf, None if no_prox else P.g, None if no_prox else P.proxg, Y0, **opts)
    _same(got, want)
    if opts["record_iterates"]:
        np.testing.assert_allclose(got.iterates, want.iterates, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes_and_modes_in_float32_storage(seed):
    """Dense operator stored in float32 on random ragged shapes, every mode: the solve is the oracle's solve on the rounded
    matrix, to the float64 path's tolerances."""
    rng = np.random.RandomState(3000 + seed)
    m, n = int(rng.randint(1, 180)), int(rng.randint(1, 3000))
    A = (rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))).astype(np.float32).astype(np.float64)      # exactly representable
    b = rng.randn(m)
    x0 = rng.randn(n) * (0.1 if seed % 2 else 0.0)
    kind = ("shrink", "nonneg", "l1ball")[seed % 3]
    adaptive, accelerate = [(True, False), (False, True), (False, False), (True, True)][seed % 4]
    opts = dict(adaptive=adaptive, accelerate=accelerate, max_iters=60, tolerance=1e-7, evaluate_objective=bool(seed % 2),
                restart=bool((seed // 2) % 2))
    reg = {"shrink": fa.Shrink(0.05), "nonneg": fa.NonNeg(), "l1ball": fa.L1Ball(0.05)}[kind]
    P = {"shrink": lambda: pr.sparse_least_squares_from(A, b, 0.05), "nonneg": lambda: pr.nn_least_squares_from(A, b),
         "l1ball": lambda: pr.l1_ball_lasso_from(A, b, 0.05)}[kind]()
    op = fa.DenseMatrixMap(A, storage="f32")
    try:
        ls = fa.LeastSquares(b)
        np.random.seed(seed)
        got = fa.fasta(op, op.H, ls.f, ls.gradf, reg.g, reg.prox, x0, verbose=False, backend="hip", fused=bool(seed % 2) or "auto", **opts)
    finally:
        op.close()
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, x0, **opts)
    _same(got, want)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_modes_and_row_blocks_in_process(seed):
    """The same random corner of the option space, solved through the single-call multi-device form with a random number of row
    blocks on the one GPU (ShardedDenseMatrixMap, fh_create_ex ndev > 1): must equal the oracle like the unsharded solve does."""
    rng = np.random.RandomState(2000 + seed)
    shards = int(rng.randint(2, 9))
    m, n = int(rng.randint(shards, 200)), int(rng.randint(1, 300))
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = rng.randn(m)
    x0 = rng.randn(n) * (0.1 if seed % 2 else 0.0)
    kind = ("shrink", "nonneg", "l1ball")[seed % 3]
    adaptive, accelerate = [(True, False), (False, True), (False, False), (True, True)][seed % 4]
    opts = dict(adaptive=adaptive, accelerate=accelerate, max_iters=60, tolerance=1e-7,
                evaluate_objective=bool(seed % 2), window=int(rng.randint(1, 12)), restart=bool((seed // 2) % 2),
                fused=("auto", True, False)[seed % 3])
    mu = 0.05
    reg, P = {"shrink": (fa.Shrink(mu), pr.sparse_least_squares_from(A, b, mu)), "nonneg": (fa.NonNeg(), pr.nn_least_squares_from(A, b)),
              "l1ball": (fa.L1Ball(mu), pr.l1_ball_lasso_from(A, b, mu))}[kind]
    ls = fa.LeastSquares(b)
    op = fa.ShardedDenseMatrixMap(A, devices=[0] * shards)
    try:
        np.random.seed(seed)
        got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, x0, verbose=False, backend="hip", **opts)
    finally:
        op.close()
    oracle_opts = {k: v for k, v in opts.items() if k != "fused"}
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, x0, **oracle_opts)
    # (histories at the north-star tolerance: 60 adaptive iterations amplify the different summation order of the row blocks to a few
    # 1e-6; the Barzilai-Borwein step is a quotient ||Dx||^2 / <Dx, Dg> whose denominator cancels on these tiny random instances --
    # steps of 10..20 x the usual -- so an individual step size may differ in its 5th digit while the iterates stay within 1e-5)
    _same(got, want, rtol=1e-5, rtol_steps=2e-4)


@pytest.mark.parametrize("seed", range(16))
def test_random_shapes_modes_and_launch_lengths_with_the_loop_on_the_device(seed):
    """The same random corner of the option space with `device_iters` = a random launch length: the controller of csrc/fh_run.h (backtracking,
    restart, Barzilai-Borwein, residuals, best iterate, stop rule) against the oracle loop, zero and non-zero starts, ragged shapes up to the widest
    row the device loop takes (7168 columns), every built-in stop rule."""
    from fasta_python_amd import stopping
    rng = np.random.RandomState(3000 + seed)
    wide = seed % 4 == 3
    m = int(rng.randint(1, 260))
    n = int(rng.randint(4097, 7169)) if wide else int(rng.randint(1, 600))
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = rng.randn(m)
    x0 = rng.randn(n) * (0.1 if seed % 2 else 0.0)
    kind = ("shrink", "nonneg")[seed % 2]
    adaptive, accelerate = [(True, False), (False, True), (False, False), (True, True)][seed % 4]
    rule = ("hybrid_residual", "residual", "norm_residual", "ratio_residual")[(seed // 4) % 4]
    opts = dict(adaptive=adaptive, accelerate=accelerate, max_iters=50, tolerance=1e-7, evaluate_objective=bool(seed % 2),
                window=int(rng.randint(1, 12)), restart=bool((seed // 2) % 2), stop_rule=getattr(stopping, rule))
    mu = 0.05
    reg, P = {"shrink": (fa.Shrink(mu), pr.sparse_least_squares_from(A, b, mu)), "nonneg": (fa.NonNeg(), pr.nn_least_squares_from(A, b))}[kind]
    ls = fa.LeastSquares(b)
    np.random.seed(seed)
    # (FH_TUNE_RUN_MAX_N: by default the device loop is offered only where it beats the library's host-side loop -- not for a few hundred
    #  wide rows; the kernel is exercised on every width it has)
    op = fa.DenseMatrixMap(A, tuning={hip.TUNE_RUN_MAX_N: 7168})
    try:
        got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, x0, verbose=False, backend="hip", device_iters=int(rng.randint(1, 40)), **opts)
    finally:
        op.close()
    assert got.device_steps == got.iteration_count, "the device loop did not take this solve"
    oracle_opts = dict(opts, stop_rule=getattr(fo, rule))
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, x0, **oracle_opts)
    # Histories are compared while the solve is still moving: these tiny underdetermined instances reach machine precision within 20-30
    # iterations (residual 1e-8 of its start), after which step sizes and residuals are rounding noise in the oracle, in the per-iteration
    # path and here alike (seed 11: all three part at iteration 25).  Counts, the solution and the untruncated shapes are compared always.
    k = want.iteration_count
    assert got.iteration_count == k and got.backtracks == want.backtracks
    small = np.nonzero(want.residuals[:k] < 1e-6 * want.residuals[0])[0]
    kk = int(small[0]) if small.size else k
    for f, r in (("residuals", 1e-5), ("norm_residuals", 1e-5), ("stepsizes", 2e-4)):
        np.testing.assert_allclose(getattr(got, f)[:kk], getattr(want, f)[:kk], rtol=r, atol=1e-14, err_msg=f)
    if want.objectives is not None:
        np.testing.assert_allclose(got.objectives[:k + 1], want.objectives[:k + 1], rtol=1e-5, atol=1e-14)
    np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-8)
    assert got.residuals.shape == want.residuals.shape and got.times.shape == want.times.shape


@pytest.mark.parametrize("kind,adaptive,accelerate", [("shrink", True, False), ("shrink", False, True), ("shrink", False, False),
                                                      ("nonneg", True, False), ("nonneg", False, True), ("l1ball", True, False)])
@pytest.mark.parametrize("m,n", [(40000, 2048), (20000, 8192)])       # 256 teams of one / 128 teams of two members, 157 rows per team
def test_tall_problems_whose_rows_are_dealt_cyclically(m, n, kind, adaptive, accelerate):
    """Most cases above have few rows per team, i.e. the blocked dealing; from 128 rows per team on the one-pass kernel deals its rows cyclically
    (round 6, profiles/r06_placement.txt).  Whole solves in every mode against the oracle loop at such shapes, every iteration."""
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = rng.randn(m)
    x0 = rng.randn(n) * 0.1
    opts = dict(adaptive=adaptive, accelerate=accelerate, max_iters=30, tolerance=1e-8, evaluate_objective=True, window=5)
    got, want = _pair(A, b, kind, 0.05, x0, opts, 11)
    _same(got, want)
