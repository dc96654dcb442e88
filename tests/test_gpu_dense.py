"""GPU parity tests for the dense operator path, all through the C ABI (libfasta_hip.so).

Tolerances (north-star: iterate-for-iterate rtol 1e-5): matvecs rtol 1e-12 (float64, different
summation order than OpenBLAS); histories rtol 1e-6; iterates/solution rtol 1e-5 (+1e-9 abs).
"""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from oracle import fasta_np as fo
from oracle import problems as pr
from tests import gpu_util as G
from tests import helpers as H

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (1, 7), (3, 5), (16, 16), (17, 33), (64, 128), (128, 64), (200, 1000), (513, 1025), (1024, 96)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_apply_matches_numpy(m, n):
    rng = np.random.RandomState(m * 1000 + n)
    A = rng.randn(m, n)
    x, y = rng.randn(n), rng.randn(m)
    op = fa.DenseMatrixMap(A)
    try:
        np.testing.assert_allclose(op.device_apply(x), A @ x, rtol=1e-12, atol=1e-12)                  # K-fwd (fh_apply)
        np.testing.assert_allclose(op.device_apply(y, adjoint=True), A.T @ y, rtol=1e-12, atol=1e-12)    # K-adj
        assert np.array_equal(op(x), A @ x) and np.array_equal(op.H(y), A.T @ y)     # host arrays: the reference's closures (linalg.py:41)
        with pytest.raises(AssertionError):
            op(np.zeros(n + 1))                                     # fasta/linalg.py:58
    finally:
        op.close()


@pytest.mark.parametrize("cpt,rows,nt", [(1, 4, 1), (2, 8, 0), (4, 16, 1)])
def test_apply_all_kernel_variants(cpt, rows, nt):
    rng = np.random.RandomState(7)
    A = rng.randn(300, 1100)
    x, y = rng.randn(1100), rng.randn(300)
    op = fa.DenseMatrixMap(A, tuning={hip.TUNE_ADJ_CPT: cpt, hip.TUNE_FWD_ROWS: rows, hip.TUNE_NT_LOADS: nt,
                                      hip.TUNE_LD_PAD: 32, hip.TUNE_ADJ_SLAB_ROWS: 40, hip.TUNE_FWD_GRID_CAP: 7})
    try:
        np.testing.assert_allclose(op.device_apply(x), A @ x, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(op.device_apply(y, adjoint=True), A.T @ y, rtol=1e-12, atol=1e-12)
    finally:
        op.close()


def test_synthetic_generator_is_bit_identical_to_host_twin():
    m, n, seed, scale = 37, 101, 5, 0.125
    op = fa.DenseMatrixMap.synthetic(m, n, seed, scale, row0=11)
    try:
        got = op.host_rows(0, m)
        want = pr.synth_matrix(m, n, seed, scale, row0=11)
        assert np.array_equal(got, want)
        assert abs(want.std() - scale) < 0.05 * scale
    finally:
        op.close()


def test_single_step_scalars_match_numpy():
    """One K-fwd + K-adj against the NumPy expressions of fasta/__init__.py:181-188, 200, 254-274."""
    rng = np.random.RandomState(3)
    m, n, mu, tau = 96, 200, 0.05, 0.3
    A = rng.randn(m, n) / 10
    b = rng.randn(m)
    x0 = rng.randn(n) * 0.1
    op = fa.DenseMatrixMap(A)
    c = op.ctx
    try:
        c.set_loss_lsq(b)
        c.set_prox(hip.PROX_SHRINK, mu)
        c.set_vector(hip.VEC_X0, x0)
        s0 = c.init()
        g0 = A.T @ (A @ x0 - b)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G0, n), g0, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(s0[hip.S_FSQ], np.sum((A @ x0 - b) ** 2), rtol=1e-12)
        s = c.fwd(tau)
        xhat = x0 - tau * g0
        xp = fo.shrink(xhat, tau * mu)
        np.testing.assert_allclose(c.get_vector(hip.VEC_XHAT, n), xhat, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(c.get_vector(hip.VEC_XPROX, n), xp, rtol=1e-12, atol=1e-14)
        dx = xp - x0
        z = A @ xp
        want = {hip.S_FSQ: np.sum((z - b) ** 2), hip.S_DXG0: dx @ g0, hip.S_DX2: dx @ dx,
                hip.S_XH2: np.sum((xp - xhat) ** 2), hip.S_G02: g0 @ g0, hip.S_GSUM: np.abs(xp).sum(),
                hip.S_GMAX: np.abs(xp).max()}
        for k, v in want.items():
            np.testing.assert_allclose(s[k], v, rtol=1e-11, atol=1e-13, err_msg=str(k))
        a = c.adj(tau)
        g1 = A.T @ (z - b)
        dg = g1 + (xhat - x0) / tau
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), g1, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(a[hip.S_DXDG], dx @ dg, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(a[hip.S_DG2], dg @ dg, rtol=1e-10, atol=1e-13)
        # accelerated variant of the same step: extrapolated x1 / z1 (fasta/__init__.py:242-245)
        coef = 0.37
        a2 = c.adj(tau, accel=True, coef=coef)
        x1 = xp + coef * (xp - x0)                 # x_accel0 = x0 after init
        z1 = z + coef * (z - A @ x0)
        np.testing.assert_allclose(c.get_vector(hip.VEC_X1, n), x1, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(a2[hip.S_FSQ_ADJ], np.sum((z1 - b) ** 2), rtol=1e-11)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), A.T @ (z1 - b), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(a2[hip.S_XH2_ADJ], np.sum((x1 - xhat) ** 2), rtol=1e-11)
    finally:
        op.close()


DENSE_FULL = [n for n in H.golden_cases()
              if n.split("_")[0] in ("sparse", "nnls", "c1") and "unnormalised" not in n and "under" not in n
              and "window3" not in n]
DENSE_PREFIX = {"sparse_ls_unnormalised_backtracks": 25, "nnls_under_first40": 25, "sparse_ls_opt_window3_shrink": 25}


@pytest.mark.parametrize("name", DENSE_FULL)
def test_golden_parity_full_solve(name):
    """HIP solve vs the REFERENCE's recorded run: same iteration count, same backtracks, histories and
    iterates within tolerance on every iteration."""
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    c = G.run_hip(meta["kind"], data, meta["options"], meta["solver_seed"], g_none=meta["options"].get("g_none", False))
    assert c.iteration_count == int(z["iteration_count"])
    assert c.backtracks == int(z["backtracks"])
    get = lambda f: z[f] if f in z.files else None
    G.compare_histories(c, get, c.iteration_count, rtol=1e-6, atol=1e-14)
    np.testing.assert_allclose(c.solution, z["solution"], rtol=1e-5, atol=1e-9)
    if "iterates" in z.files:
        k = c.iteration_count + 1
        np.testing.assert_allclose(c.iterates[:k], z["iterates"][:k], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(c.function_hist[:k], z["function_hist"][:k], rtol=1e-6)


@pytest.mark.parametrize("name,k", sorted(DENSE_PREFIX.items()))
def test_golden_parity_prefix_of_sensitive_runs(name, k):
    """Backtracking-heavy / non-convergent runs amplify rounding: pin the first k
    iterations, including where the backtracks fall."""
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    opts = dict(meta["options"], max_iters=k, tolerance=0.0)
    c = G.run_hip(meta["kind"], data, opts, meta["solver_seed"])
    np.testing.assert_allclose(c.stepsizes[:k], z["stepsizes"][:k], rtol=1e-6)
    np.testing.assert_allclose(c.residuals[:k], z["residuals"][:k], rtol=1e-6)
    np.testing.assert_allclose(c.norm_residuals[:k], z["norm_residuals"][:k], rtol=1e-6)


@pytest.mark.parametrize("name,prefix", sorted(DENSE_PREFIX.items()))
def test_sensitive_runs_first_divergence_from_the_oracle_is_measured(name, prefix, capsys):
    """"report the first divergent iteration".  The prefix-pinned runs are continued to 300 iterations on
    both sides and the test MEASURES where the step sizes first differ by more than 1e-6 relative: not before the pinned prefix;
    identical backtracking decisions up to there.  (The oracle against itself with permuted rows parts at iteration 69 / 76:
    tests/test_tv_divergence_cpu.py -- the sensitivity is the problem's.)"""
    from tests.helpers import first_divergence
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    opts = dict(meta["options"], max_iters=300, tolerance=0.0)
    got = G.run_hip(meta["kind"], data, opts, meta["solver_seed"])
    P = H.oracle_problem(meta, z)
    o = H.resolve_options(opts, fo)
    np.random.seed(meta["solver_seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(P.A, P.At, P.f, P.gradf, P.g, P.proxg, P.x0, **o)
    k = min(got.iteration_count, want.iteration_count)
    first = first_divergence(got.stepsizes, want.stepsizes, k)
    with capsys.disabled():
        print(f"\n{name}: HIP {got.backtracks} backtracks, oracle {want.backtracks} in {k} iterations; step sizes first differ (> 1e-6 relative) at iteration {first}")
    assert first >= prefix
    np.testing.assert_allclose(got.residuals[:first], want.residuals[:first], rtol=1e-5)


def test_runs_are_bitwise_repeatable():
    meta, z = H.load_case("sparse_ls_64x128_adaptive")
    data = H.case_data(meta, z)
    a = G.run_hip(meta["kind"], data, meta["options"], meta["solver_seed"])
    b = G.run_hip(meta["kind"], data, meta["options"], meta["solver_seed"])
    for f in ("residuals", "norm_residuals", "stepsizes", "objectives", "solution"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_unrecognised_operands_fail_loudly():
    A = np.eye(4)
    ls, reg = fa.LeastSquares(np.zeros(4)), fa.Shrink(0.1)
    with pytest.raises(TypeError):
        fa.fasta(lambda x: x, lambda x: x, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(4), backend="hip")
    with pytest.raises(TypeError):
        fa.fasta(A, A.T, lambda z: 0.0, lambda z: z, reg.g, reg.prox, np.zeros(4), backend="hip")
    with pytest.raises(TypeError):
        fa.fasta(A, A.T, ls.f, ls.gradf, lambda x: 0, lambda x, t: x, np.zeros(4), backend="hip")
    with pytest.raises(AssertionError):
        fa.fasta(A, A.T, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(5), backend="hip")


@pytest.mark.parametrize("m,n", [(4096, 4096), (16384, 2048), (3000, 20000), (16384, 16384)])      # 4096^2 and 16384^2
def test_mid_size_matches_oracle_loop(m, n):
    """Synthetic LASSO: HIP loop vs oracle loop on the same (device-generated) matrix, every iteration."""
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap.synthetic(m, n, 0, scale)
    try:
        A = op.host_rows(0, m)
        x_true = pr.synth_sparse_signal(n, 1)
        b = A @ x_true + 0.01 * np.random.RandomState(2).randn(m)
        ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
        opts = dict(tolerance=1e-6, max_iters=40 if m * n > 2 ** 27 else 60, evaluate_objective=True, record_iterates=True)
        np.random.seed(3)
        got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", **opts)
        P = pr.sparse_least_squares_from(A, b, 0.02)
        np.random.seed(3)
        want = fo.fasta(*P.args7(), **opts)
        assert got.iteration_count == want.iteration_count and got.backtracks == want.backtracks
        k = got.iteration_count
        G.compare_histories(got, lambda f: getattr(want, f), k, rtol=1e-6, atol=1e-14)
        np.testing.assert_allclose(got.iterates[:k + 1], want.iterates[:k + 1], rtol=1e-5, atol=1e-9)
    finally:
        op.close()


@pytest.mark.parametrize("accelerate", [False, True])
def test_row_sharded_code_path_with_one_rank_communicator(accelerate):
    """The sharded adjoint (partial g1 -> RCCL all-reduce -> separate n-side epilogue) on a 1-rank
    communicator must reproduce the fused single-GPU launch (the 8-GPU run is the driver's)."""
    meta, z = H.load_case("sparse_ls_64x128_adaptive")
    data = H.case_data(meta, z)
    opts = dict(tolerance=1e-5, evaluate_objective=True, adaptive=not accelerate, accelerate=accelerate)
    ref = G.run_hip("sparse_ls", data, opts, 5)
    A, loss, reg, x0 = G.hip_operands("sparse_ls", data)
    try:
        A.ctx.comm_init(1, 0, hip.comm_unique_id())
        np.random.seed(5)
        got = fa.fasta(A, loss.f, loss.gradf, reg.g, reg.prox, x0, verbose=False, backend="hip", **opts)
    finally:
        A.close()
    assert got.iteration_count == ref.iteration_count and got.backtracks == ref.backtracks
    k = got.iteration_count
    # (both runs take the one-pass kernel -- the choice of fused="auto" at every size since round 5 --; the row-sharded one forms its n-side sums in
    # k_bb_epilogue's chunks after the all-reduce, the plain one in the kernel's finaliser: other summation orders, 1-2e-12 after 17 iterations)
    G.compare_histories(got, lambda f: getattr(ref, f), k, rtol=1e-10)
    np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-10, atol=1e-15)


def test_full_size_properties_65536():
    """BASELINE config 2 size (32 GiB of A): adjointness, linearity and row spot checks of the fused
    matvecs on the device-generated matrix -- size-independent properties instead of a host copy."""
    m = n = 65536
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap.synthetic(m, n, 0, scale)
    try:
        rng = np.random.RandomState(1)
        x, x2, y = rng.randn(n), rng.randn(n), rng.randn(m)
        Ax, Ax2, ATy = op(x), op(x2), op.H(y)
        assert abs(np.vdot(Ax, y) - np.vdot(x, ATy)) < 1e-9 * np.linalg.norm(Ax) * np.linalg.norm(y)
        np.testing.assert_allclose(op(2.0 * x - 3.0 * x2), 2.0 * Ax - 3.0 * Ax2, rtol=1e-9, atol=1e-11)
        rows = [0, 1, 7, 4095, 32768, 65535]
        for r in rows:
            a = pr.synth_matrix(1, n, 0, scale, row0=r, n_total=n)[0]
            assert np.array_equal(op.host_rows(r, 1)[0], a)
            np.testing.assert_allclose(Ax[r], a @ x, rtol=1e-10, atol=1e-12)
        cols = np.array([0, 3, 1000, 65535])
        for r in (0, 65535):
            pass
        # a column of A^T y from two full rows' worth of generator output is too costly on the host; check
        # instead that A^T e_r reproduces row r (exact: one non-zero term per sum)
        e = np.zeros(m); e[12345] = 1.0
        assert np.array_equal(op.H(e), pr.synth_matrix(1, n, 0, scale, row0=12345, n_total=n)[0])
    finally:
        op.close()


def test_c2_c3_full_size_first_iterations_match_oracle_loop():
    """BASELINE configs 2 (LASSO) and 3 (NNLS) at FULL size (65536^2, 32 GiB): the first FIVE iterations of the HIP solve
    against the oracle NumPy loop on ONE shared host copy of the same device-generated matrix (BASELINE.md section 4 parity
    gate: x rtol 1e-5, scalars rtol 1e-8, identical backtrack counts).  The default path at this size is the one-pass kernel.
    A third run forces the BACKTRACKING retry path at full size (fasta/__init__.py:195-217): L and tau0 are given, and tau0 is
    thousands of times too large, so the first iteration is re-launched with tau * 0.2 several times before it is accepted."""
    m = n = 65536
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    base = dict(tolerance=0.0, evaluate_objective=True, record_iterates=True)
    runs = (("lasso", 0.01, dict(max_iters=5)), ("nnls", 0.005, dict(max_iters=5)),          # nn_least_squares.py:49 uses 0.005
            ("lasso_forced_backtracking", 0.01, dict(max_iters=3, L=1.0, tau0=5000.0)))
    op = fa.DenseMatrixMap.synthetic(m, n, 0, scale)
    got, rhs = {}, {}
    try:
        x_true = pr.synth_sparse_signal(n, 1)
        Ax = op(x_true)
        for kind, sigma, extra in runs:
            b = Ax + sigma * np.random.RandomState(2).randn(m)
            ls, reg = fa.LeastSquares(b), (fa.NonNeg() if kind == "nnls" else fa.Shrink(0.02))
            solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, **base, **extra)
            np.random.seed(3)
            got[kind] = solver.setup().run()
            assert solver.fused_steps == extra["max_iters"] + got[kind].backtracks   # every launch of the loop was the one-pass kernel
            rhs[kind] = b
        assert got["lasso_forced_backtracking"].backtracks >= 4
        # [r4] the same LASSO run ROW-SHARDED: the matrix as 8 in-process row blocks of 8192 rows (all on this GPU), against the SAME oracle
        # run below -- the sharded path's full-size oracle leg (round 3 compared it with the unsharded HIP run only)
        op8 = fa.ShardedDenseMatrixMap.synthetic(m, n, seed=0, scale=scale, devices=[0] * 8)
        try:
            solver = fa.FBSolver(op8, fa.LeastSquares(rhs["lasso"]), fa.Shrink(0.02), np.zeros(n), verbose=False, **base, max_iters=5)
            np.random.seed(3)
            got["lasso_8_row_blocks"] = solver.setup().run()
            assert solver.fused_steps == 5 + got["lasso_8_row_blocks"].backtracks
            rhs["lasso_8_row_blocks"] = rhs["lasso"]
        finally:
            op8.close()
        A = op.host_rows(0, m)                       # 32 GiB host copy, D2H, shared by the oracle runs
    finally:
        op.close()
    runs = runs + (("lasso_8_row_blocks", 0.01, dict(max_iters=5)),)
    for kind, _, extra in runs:
        iters = extra["max_iters"]
        P = pr.nn_least_squares_from(A, rhs[kind]) if kind == "nnls" else pr.sparse_least_squares_from(A, rhs[kind], 0.02)
        np.random.seed(3)
        want = fo.fasta(*P.args7(), **base, **extra)
        g = got[kind]
        assert g.iteration_count == want.iteration_count == iters, kind
        assert g.backtracks == want.backtracks, kind
        for f in ("residuals", "norm_residuals", "stepsizes"):
            np.testing.assert_allclose(getattr(g, f)[:iters], getattr(want, f)[:iters], rtol=1e-8, err_msg=kind + " " + f)
        np.testing.assert_allclose(g.objectives[:iters + 1], want.objectives[:iters + 1], rtol=1e-8, err_msg=kind)
        np.testing.assert_allclose(g.iterates[:iters + 1], want.iterates[:iters + 1], rtol=1e-5, atol=1e-12, err_msg=kind)
        np.testing.assert_allclose(g.solution, want.solution, rtol=1e-5, atol=1e-12, err_msg=kind)


def test_box_prox_inside_a_solve_matches_the_oracle():
    """Box prox inside full solves on a dense operator: min_y .5||A y - b||^2 subject to 0 <= y <= C, the constraint of the
    SVM dual (svm.py:71 `np.minimum(np.maximum(y, 0), C)`), in all three modes against the oracle loop running the
    reference's clip closure."""
    rng = np.random.RandomState(12)
    m, n, C = 90, 140, 0.35
    A = rng.randn(m, n)
    A /= np.linalg.norm(A, 2)
    b = A @ rng.uniform(-0.2, 0.6, size=n) + 0.01 * rng.randn(m)
    for mode in (dict(adaptive=True, accelerate=False), dict(adaptive=False, accelerate=True), dict(adaptive=False, accelerate=False)):
        opts = dict(tolerance=1e-6, evaluate_objective=True, max_iters=400, **mode)
        ls, reg = fa.LeastSquares(b), fa.Box(0.0, C)
        np.random.seed(8)
        got = fa.fasta(A, A.T, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", **opts)
        np.random.seed(8)
        f = lambda z: .5 * np.linalg.norm((z - b).ravel()) ** 2
        want = fo.fasta(A, A.T, f, lambda z: z - b, lambda y: 0, lambda y, t: np.minimum(np.maximum(y, 0), C), np.zeros(n), **opts)
        assert got.iteration_count == want.iteration_count and got.backtracks == want.backtracks, mode
        k = got.iteration_count
        G.compare_histories(got, lambda fld: getattr(want, fld), k, rtol=1e-6, atol=1e-13)
        np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-9)
        assert got.solution.min() >= 0.0 and got.solution.max() <= C
        assert np.any(got.solution == 0.0) and np.any(got.solution == C)        # both faces of the box are active


@pytest.mark.parametrize("shape", [(300, 500), (64, 20000), (2000, 16384)])
def test_on_device_spectral_norm_matches_numpy(shape):
    """Opt-in Lipschitz estimate: power iteration on A^H A with device matvecs."""
    rng = np.random.RandomState(shape[0])
    A = rng.randn(*shape) / 10
    A += np.outer(rng.randn(shape[0]), rng.randn(shape[1])) / 40       # a clear top singular value: power iteration converges fast
    op = fa.DenseMatrixMap(A)
    try:
        L = op.spectral_norm_squared(iters=200, rtol=1e-13, seed=0)
        want = np.linalg.norm(A, 2) ** 2
        np.testing.assert_allclose(L, want, rtol=1e-8)
        assert op.spectral_norm_squared(iters=5, seed=1) <= want * (1 + 1e-12)      # a Rayleigh quotient never overshoots
        # and it is usable as fasta()'s L / tau0 (no RNG probes, fasta/__init__.py:100)
        b = rng.randn(shape[0])
        ls, reg = fa.LeastSquares(b), fa.Shrink(0.1)
        c = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(shape[1]), verbose=False, backend="hip", L=L, tau0=(2 / L) / 10,
                     max_iters=30, tolerance=0.0)
        assert c.iteration_count == 30 and np.all(np.isfinite(c.residuals[:30]))
    finally:
        op.close()


@pytest.mark.parametrize("m,n", [(37, 100), (512, 1024), (3000, 2000)])
def test_fwd_adj_under_one_sync_equals_fwd_then_adj(m, n):
    """fh_fwd_adj enqueues K-fwd and K-adj back to back: same launches, same bits, one synchronisation."""
    rng = np.random.RandomState(m)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx

        def state():
            c.set_loss_lsq(rng.randn(m) * 0 + 1.0); c.set_prox(hip.PROX_SHRINK, 0.03); c.set_vector(hip.VEC_X0, np.linspace(-1, 1, n)); c.init()

        state()
        s = c.fwd(0.3)
        a = c.adj(0.3)
        g = c.get_vector(hip.VEC_G1, n)
        state()
        p = c.fwd_adj(0.3)
        assert np.array_equal(p[:8], s[:8]) and np.array_equal(p[8:14], a[8:14])
        assert np.array_equal(c.get_vector(hip.VEC_G1, n), g)
        # fused="auto" takes the one-pass kernel at every size since round 5 (profiles/r05_crossover.txt) ...
        ls, reg = fa.LeastSquares(np.ones(m)), fa.Shrink(0.03)
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, max_iters=15, tolerance=0.0)
        np.random.seed(7)
        solver.setup().run()
        assert solver.mode == "speculative" and solver.fused_steps > 0
        ref = solver.result()
        # ... and this pair by itself where the one-pass kernel is not available (here: a co-residency probe that says no)
        c.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_PROBE_SAYS_NO)
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, max_iters=15, tolerance=0.0)
        np.random.seed(7)
        solver.setup().run()
        c.set_tuning(hip.TUNE_TEST_HOOKS, 0)
        assert solver.mode == "pair" and solver.pair_steps > 0 and solver.fused_steps == 0
        got = solver.result()
        assert got.backtracks == ref.backtracks
        np.testing.assert_allclose(got.residuals[:15], ref.residuals[:15], rtol=1e-8)
        np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-8, atol=1e-14)
    finally:
        op.close()


def test_a_large_allocation_reuses_the_kept_block_and_waits_for_the_clearing_of_a_large_free_only_when_asked():
    """A matrix of >= 1 GiB allocated while the driver still clears a large free is mapped less favourably for its lifetime (profiles/r06_alloc_settle.txt; the
    one-pass kernel with cyclically dealt rows, the default, does not depend on that: profiles/r06_placement.txt).  Default: the block a context gives up is
    kept, per device, and handed to the next matrix that fits it and fills at least half of it; no wait anywhere.  fh_alloc_settle(1): a matrix the kept block
    cannot serve waits ~35 ms per GiB freed; small matrices never do either."""
    import time
    hip.release_cached()
    time.sleep(0.3)                                       # (frees of earlier tests)
    gib2, gib1 = (16384, 16384), (8192, 16384)            # 2 GiB, 1 GiB
    def cycle(second, cache):
        hip.alloc_cache(cache)
        a = fa.DenseMatrixMap.synthetic(*gib2, 0, 1e-3)
        ref = a.host_rows(0, 2).copy()
        a.close()
        w0, h0, t0 = hip.alloc_settle_waited(), hip.alloc_cache_hits(), time.perf_counter()
        b = fa.DenseMatrixMap.synthetic(*second, 0, 1e-3)
        waited, hits, wall = hip.alloc_settle_waited() - w0, hip.alloc_cache_hits() - h0, time.perf_counter() - t0
        rows = b.host_rows(0, 2)
        b.close()
        hip.release_cached()
        time.sleep(0.3)
        return waited, hits, wall, ref, rows
    try:
        waited, hits, wall, ref, rows = cycle(gib2, True)              # same size: the kept block, no wait, the right matrix in it
        assert hits == 1 and waited == 0.0 and np.array_equal(ref, rows)
        waited, hits, wall, ref, rows = cycle(gib1, True)              # half the size: still the kept block
        assert hits == 1 and waited == 0.0 and np.array_equal(rows[:, :5], fa.DenseMatrixMap.synthetic(8, 16384, 0, 1e-3).host_rows(0, 2)[:, :5])
        waited, hits, wall, _, _ = cycle((4500, 16384), True)          # 0.55 GiB... below the 1 GiB threshold: the kept-block logic does not apply
        assert hits == 0 and waited == 0.0
        waited, hits, _, _, _ = cycle(gib2, False)                     # keeping off: the block is freed; by default nobody waits for its clearing
        assert hits == 0 and waited == 0.0
        hip.alloc_settle(True)
        waited, hits, wall, _, _ = cycle(gib2, False)                  # asked to: the next allocation waits
        assert hits == 0 and 0.02 < waited < 0.2 and wall >= waited, (waited, wall)
        waited, hits, wall, _, _ = cycle(gib2, True)                   # ... but not when the kept block serves it
        assert hits == 1 and waited == 0.0
    finally:
        hip.alloc_settle(False)
        hip.alloc_cache(True)
    a = fa.DenseMatrixMap.synthetic(4096, 4096, 0, 1e-3); a.close()          # 128 MiB: below the threshold
    w0, h0 = hip.alloc_settle_waited(), hip.alloc_cache_hits()
    b = fa.DenseMatrixMap.synthetic(4096, 4096, 0, 1e-3); b.close()
    assert hip.alloc_settle_waited() == w0 and hip.alloc_cache_hits() == h0
