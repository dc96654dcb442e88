"""The one-pass (fused) iteration kernel: same numbers as K-fwd + K-adj, same solves as the oracle."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from oracle import fasta_np as fo
from oracle import problems as pr

pytestmark = pytest.mark.gpu


def _state(op, b, mu, x0, prox=hip.PROX_SHRINK):
    c = op.ctx
    c.set_loss_lsq(b)
    c.set_prox(prox, mu)
    c.set_vector(hip.VEC_X0, x0)
    c.init()
    return c


# variant bits: 2 = team members on one XCD (default), 0 = consecutive blocks, 8 = n=65536 as 8 members x 16 pieces with the
# exchange in line (default there: 16 members x 8 pieces, posts two rows ahead); n in (65536, 131072] runs 16 members x 9..16 pieces
# with the x slice in LDS, posting one row ahead (variant bit 16: the round-1 shape, 16 x 16 in line)
@pytest.mark.parametrize("variant", [34, 2, 0, 10, 18, 42, 50])      # 34 = the default (2: members on one XCD, 32: rows dealt cyclically)
@pytest.mark.parametrize("m,n", [(1, 4096), (37, 4096), (300, 4096), (4097, 4096), (500, 8192), (200, 16384), (130, 32768),
                                 (70, 65536), (40, 131072),
                                 # ragged n: the next shape up with the surplus lanes masked
                                 (9, 100), (50, 5000), (120, 9001), (40, 20000), (33, 33000), (30, 50000), (20, 70000), (25, 75000), (21, 90000), (24, 100000), (23, 108000), (19, 120000),
                                 # 32 members (n in (131072, 262144])
                                 (14, 140000), (12, 200000), (10, 262144)])
def test_fused_step_equals_two_launch_step(m, n, variant):
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b, x0 = rng.randn(m), rng.randn(n) * 0.05
    tau, mu = 0.4, 0.03
    op = fa.DenseMatrixMap(A, tuning={hip.TUNE_FUSED_VARIANT: variant})   # 2: team members on one XCD, 0: consecutive blocks
    try:
        c = _state(op, b, mu, x0)
        assert c.fused_supported()
        s = c.fwd(tau)
        a = c.adj(tau)
        ref = {k: c.get_vector(k, n) for k in (hip.VEC_XHAT, hip.VEC_XPROX, hip.VEC_G1)}
        zref = c.get_vector(hip.VEC_Z, m)
        c = _state(op, b, mu, x0)
        f = c.step(tau)
        for k in (hip.S_FSQ, hip.S_DXG0, hip.S_DX2, hip.S_XH2, hip.S_G02, hip.S_GSUM, hip.S_GMAX):
            np.testing.assert_allclose(f[k], s[k], rtol=1e-12, atol=1e-300, err_msg=f"fwd scalar {k}")
        for k in (hip.S_DXDG, hip.S_DG2):
            np.testing.assert_allclose(f[k], a[k], rtol=1e-10, atol=1e-18, err_msg=f"adj scalar {k}")
        assert np.array_equal(c.get_vector(hip.VEC_XHAT, n), ref[hip.VEC_XHAT])
        assert np.array_equal(c.get_vector(hip.VEC_XPROX, n), ref[hip.VEC_XPROX])
        np.testing.assert_allclose(c.get_vector(hip.VEC_Z, m), zref, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), ref[hip.VEC_G1], rtol=1e-11, atol=1e-15)
        # against NumPy directly
        xp = fo.shrink(x0 - tau * (A.T @ (A @ x0 - b)), tau * mu)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), A.T @ (A @ xp - b), rtol=1e-10, atol=1e-14)
        # repeatable
        c = _state(op, b, mu, x0)
        f2 = c.step(tau)
        assert np.array_equal(f, f2)
    finally:
        op.close()


@pytest.mark.parametrize("m,n", [(1, 4096), (333, 4096), (2050, 8192)])
def test_fused_step_with_the_logistic_loss_equals_two_launch_step(m, n):
    """loss terms are summed after the row loop by each team's member 0; the gradient factor inside it."""
    rng = np.random.RandomState(m)
    A = rng.randn(m, n) / np.sqrt(n)
    b = np.where(rng.rand(m) < 0.5, 1.0, -1.0)
    x0, tau, mu = rng.randn(n) * 0.3, 0.7, 0.05
    op = fa.DenseMatrixMap(A)
    try:
        def state():
            c = op.ctx
            c.set_loss_logistic(b); c.set_prox(hip.PROX_SHRINK, mu); c.set_vector(hip.VEC_X0, x0); c.init()
            return c
        c = state()
        s, a = c.fwd(tau), c.adj(tau)
        g1 = c.get_vector(hip.VEC_G1, n)
        c = state()
        f = c.step(tau)
        np.testing.assert_allclose(f[hip.S_FSQ], s[hip.S_FSQ], rtol=1e-12)
        for k in (hip.S_DXDG, hip.S_DG2):
            np.testing.assert_allclose(f[k], a[k], rtol=1e-10, atol=1e-18)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), g1, rtol=1e-11, atol=1e-15)
        xp = fo.shrink(x0 - tau * (A.T @ (-b / (1 + np.exp(b * (A @ x0))))), tau * mu)
        z = A @ xp
        np.testing.assert_allclose(f[hip.S_FSQ], np.sum(np.log(1 + np.exp(z)) - (b == 1) * z), rtol=1e-11)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), A.T @ (-b / (1 + np.exp(b * z))), rtol=1e-9, atol=1e-13)
    finally:
        op.close()


@pytest.mark.parametrize("loss", ["lsq", "logistic"])
@pytest.mark.parametrize("m,n,restart", [(300, 4096, True), (300, 4096, False), (64, 65536, True), (90, 20000, True), (2500, 8192, False), (50, 100000, True), (45, 131072, False)])
def test_fused_accelerated_step_equals_two_launch_step(m, n, restart, loss):
    """fh_step_accel: restart dot before the first row, gradient at the extrapolated z, x1 extrapolated (FISTA)."""
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = np.where(rng.rand(m) < 0.5, 1.0, -1.0) if loss == "logistic" else rng.randn(m)
    x0 = rng.randn(n) * 0.05
    tau, mu = 0.4, 0.03
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx

        def warm():        # one committed accelerated iteration, so that x_accel0 / z_accel0 differ from x0 / z0
            (c.set_loss_logistic if loss == "logistic" else c.set_loss_lsq)(b)
            c.set_prox(hip.PROX_SHRINK, mu); c.set_vector(hip.VEC_X0, x0); c.init()
            c.fwd(tau); c.adj(tau, True, 0.0); c.commit()
            c.fwd(tau); c.adj(tau, True, 0.28); c.commit()

        coef = 0.43
        warm()
        s = c.fwd(tau)
        applied = 0.0 if (restart and s[hip.S_RDOT] > 1e-30) else coef
        a = c.adj(tau, True, applied)
        ref = {k: c.get_vector(k, n) for k in (hip.VEC_XHAT, hip.VEC_XPROX, hip.VEC_X1, hip.VEC_G1)}
        zref = c.get_vector(hip.VEC_Z, m)
        warm()
        f = c.step_accel(tau, coef, restart)
        np.testing.assert_allclose(f[hip.S_RDOT], s[hip.S_RDOT], rtol=1e-9, atol=1e-18)
        assert (f[hip.S_RDOT] > 1e-30) == (s[hip.S_RDOT] > 1e-30)
        for k in (hip.S_FSQ, hip.S_DXG0, hip.S_DX2, hip.S_XH2, hip.S_G02, hip.S_GSUM, hip.S_GMAX):
            np.testing.assert_allclose(f[k], s[k], rtol=1e-12, atol=1e-300, err_msg=f"fwd scalar {k}")
        for k in (hip.S_DXDG, hip.S_DG2, hip.S_FSQ_ADJ, hip.S_XH2_ADJ, hip.S_GSUM_ADJ, hip.S_GMAX_ADJ):
            np.testing.assert_allclose(f[k], a[k], rtol=1e-10, atol=1e-18, err_msg=f"adj scalar {k}")
        assert np.array_equal(c.get_vector(hip.VEC_XHAT, n), ref[hip.VEC_XHAT])
        assert np.array_equal(c.get_vector(hip.VEC_XPROX, n), ref[hip.VEC_XPROX])
        assert np.array_equal(c.get_vector(hip.VEC_X1, n), ref[hip.VEC_X1])
        np.testing.assert_allclose(c.get_vector(hip.VEC_Z, m), zref, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), ref[hip.VEC_G1], rtol=1e-10, atol=1e-15)
    finally:
        op.close()


def test_unsupported_shape_reports_and_auto_falls_back():
    n = 262144 + 16                                # a row no longer fits 32 members x 16 pieces x 256 lanes
    A = np.random.RandomState(0).randn(3, n) / 400
    op = fa.DenseMatrixMap(A)
    try:
        assert not op.ctx.fused_supported()
        ls, reg = fa.LeastSquares(np.ones(3)), fa.Shrink(0.1)
        with pytest.raises(ValueError):
            fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", fused=True, max_iters=2)
        c = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", max_iters=2, tolerance=0.0)
        assert c.iteration_count == 2
    finally:
        op.close()


@pytest.mark.parametrize("kind", ["shrink", "nonneg", "backtracking", "accelerated", "accelerated_norestart", "backtracking_n16384"])
def test_full_solve_with_fused_steps_matches_oracle(kind):
    rng = np.random.RandomState(5)
    m, n = (300, 16384) if kind == "backtracking_n16384" else (700, 4096)
    A = rng.randn(m, n)
    if not kind.startswith("backtracking"):
        A /= np.linalg.norm(A, 2)
    x_true = np.zeros(n); x_true[rng.permutation(n)[:20]] = 1
    b = A @ x_true + 0.01 * rng.randn(m)
    opts = dict(tolerance=1e-6, max_iters=80, evaluate_objective=True, record_iterates=True)
    if kind.startswith("backtracking"):
        opts.update(L=1.0, tau0=1.0, max_iters=40, tolerance=0.0)     # unnormalised A: forces backtracks -> fallback path
    if kind.startswith("accelerated"):
        opts.update(adaptive=False, accelerate=True, restart=kind == "accelerated", max_iters=120)
    mu = 0.02
    reg = fa.NonNeg() if kind == "nonneg" else fa.Shrink(mu)
    P = pr.nn_least_squares_from(A, b) if kind == "nonneg" else pr.sparse_least_squares_from(A, b, mu)
    ls = fa.LeastSquares(b)
    op = fa.DenseMatrixMap(A)
    try:
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, fused=True, **opts)
        np.random.seed(3)
        got = solver.setup().run()
        np.random.seed(3)
        two = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", fused=False, **opts)
    finally:
        op.close()
    np.random.seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    assert solver.fused_steps > 0
    assert got.iteration_count == want.iteration_count == two.iteration_count
    assert got.backtracks == want.backtracks == two.backtracks
    if kind == "backtracking":          # n = 4096: speculative use, abandoned for the retries and the cool-down iterations
        assert got.backtracks > 0 and solver.fused_steps < got.iteration_count
    if kind == "backtracking_n16384":   # from n = 16384 the one-pass kernel also serves every backtracking retry
        assert got.backtracks > 0 and solver.fused_steps == got.iteration_count + got.backtracks
    k = got.iteration_count
    rtol = 1e-6
    for f in ("residuals", "norm_residuals", "stepsizes"):
        np.testing.assert_allclose(getattr(got, f)[:k], getattr(want, f)[:k], rtol=rtol, err_msg=f)
        np.testing.assert_allclose(getattr(got, f)[:k], getattr(two, f)[:k], rtol=1e-6, err_msg=f)
    np.testing.assert_allclose(got.objectives[:k + 1], want.objectives[:k + 1], rtol=rtol)
    np.testing.assert_allclose(got.iterates[:k + 1], want.iterates[:k + 1], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("accelerate", [False, True])
def test_fused_step_on_the_row_sharded_path_with_one_rank(accelerate):
    """fh_step / fh_step_accel with a communicator: local one-pass kernel -> RCCL all-reduce of g1 and the loss sums ->
    n-side epilogue (with the FISTA coefficient the kernel decided on)."""
    rng = np.random.RandomState(11)
    m, n = 300, 4096
    A = rng.randn(m, n) / 30
    b = rng.randn(m)
    ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
    opts = dict(tolerance=1e-6, max_iters=30, evaluate_objective=True)
    if accelerate:
        opts.update(adaptive=False, accelerate=True, max_iters=60)
    op = fa.DenseMatrixMap(A)
    try:
        np.random.seed(1)
        ref = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", fused=True, **opts)
        op.ctx.comm_init(1, 0, hip.comm_unique_id())
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, fused=True, **opts)
        np.random.seed(1)
        got = solver.setup().run()
    finally:
        op.close()
    assert solver.fused_steps > 0
    assert got.iteration_count == ref.iteration_count and got.backtracks == ref.backtracks
    k = got.iteration_count
    # the sharded epilogue sums the n-side reductions over a different partition: last-digit differences
    np.testing.assert_allclose(got.residuals[:k], ref.residuals[:k], rtol=1e-9)
    np.testing.assert_allclose(got.objectives[:k + 1], ref.objectives[:k + 1], rtol=1e-9)
    np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-8, atol=1e-12)


def test_lost_partial_times_out_instead_of_hanging_and_the_solver_falls_back():
    """Safety net of the one-pass kernel: every spin is wall-clock bounded.  With a partial dot product deliberately
    withheld (fault-injection bit), the launch must finish within about a second, report the timeout, and the
    driver must carry on with the two-launch path and still produce the right solve."""
    import time
    rng = np.random.RandomState(2)
    m, n = 96, 4096
    A = rng.randn(m, n) / 40
    b = rng.randn(m)
    ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
    opts = dict(tolerance=1e-6, max_iters=12, evaluate_objective=True)
    op = fa.DenseMatrixMap(A)
    try:
        np.random.seed(4)
        ref = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", fused=False, **opts)
        op.ctx.set_tuning(hip.TUNE_FUSED_VARIANT, 2 | 8)            # bit 8: the 8-member shape (n = 4096 runs without any exchange by default)
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_WITHHOLD_PARTIAL)
        c = _state(op, b, 0.02, np.zeros(n))
        t0 = time.time()
        with pytest.raises(hip.HipError):
            c.step(0.3)
        assert time.time() - t0 < 5.0
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, fused=True, **opts)
        np.random.seed(4)
        with pytest.warns(UserWarning, match="one-pass kernel disabled"):
            got = solver.setup().run()
    finally:
        op.close()
    assert not solver.use_fused
    assert got.iteration_count == ref.iteration_count
    k = got.iteration_count
    np.testing.assert_allclose(got.residuals[:k], ref.residuals[:k], rtol=1e-12)
    np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-12, atol=1e-15)


def _pair_reference(op, b, mu, x0, tau, accel):
    """fh_fwd + fh_adj from a fresh state: scalars and vectors the one-pass launch must reproduce"""
    c = _state(op, b, mu, x0)
    s = c.fwd(tau)
    coef = 0.25 if accel else 0.0
    a = c.adj(tau, accel, coef)
    n, m = x0.size, b.size
    return s, a, {k: c.get_vector(k, n) for k in (hip.VEC_XPROX, hip.VEC_G1, hip.VEC_X1)}, c.get_vector(hip.VEC_Z, m)


def _assert_step_matches(c, got, ref, n, m):
    s, a, vecs, z = ref
    for k in (hip.S_FSQ, hip.S_DXG0, hip.S_DX2, hip.S_XH2, hip.S_G02, hip.S_GSUM, hip.S_GMAX):
        np.testing.assert_allclose(got[k], s[k], rtol=1e-12, atol=1e-300, err_msg=f"fwd scalar {k}")
    for k in (hip.S_DXDG, hip.S_DG2, hip.S_FSQ_ADJ):
        np.testing.assert_allclose(got[k], a[k], rtol=1e-10, atol=1e-18, err_msg=f"adj scalar {k}")
    assert np.array_equal(c.get_vector(hip.VEC_XPROX, n), vecs[hip.VEC_XPROX])
    np.testing.assert_allclose(c.get_vector(hip.VEC_Z, m), z, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), vecs[hip.VEC_G1], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(c.get_vector(hip.VEC_X1, n), vecs[hip.VEC_X1], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("after", ["same_shape", "team_of_one", "other_team_count"])
def test_one_pass_kernel_recovers_on_the_same_context_after_a_timed_out_launch(after):
    """a timed-out launch leaves hand-off slots un-posted / un-armed and the barrier counters armed; the host must refill
    them before the next one-pass launch on that context.  Sabotage one launch (fault-injection bit 64), clear the bit, then run
    the one-pass kernel again -- at BOTH slot parities (two consecutive launches), plain and accelerated -- in the same team shape,
    in a team-of-one shape (which exchanges nothing through the slots but shares the counters) and with a different number of
    teams; every launch must equal fh_fwd + fh_adj."""
    import time
    rng = np.random.RandomState(11)
    m, n = 96, 4096
    A = rng.randn(m, n) / 40
    b, x0 = rng.randn(m), rng.randn(n) * 0.05
    tau, mu = 0.3, 0.02
    op = fa.DenseMatrixMap(A)
    try:
        refs = {acc: _pair_reference(op, b, mu, x0, tau, acc) for acc in (False, True)}
        op.ctx.set_tuning(hip.TUNE_FUSED_VARIANT, 2 | 8)              # 8 members per team at n = 4096
        c = _state(op, b, mu, x0)
        _assert_step_matches(c, c.step(tau), refs[False], n, m)       # healthy launch first (slot parity flips)
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_WITHHOLD_PARTIAL)
        c = _state(op, b, mu, x0)
        t0 = time.time()
        with pytest.raises(hip.HipError):
            c.step(tau)
        assert time.time() - t0 < 5.0
        variant = {"same_shape": 2 | 8, "team_of_one": 2, "other_team_count": (2 | 8) | (4 << 16)}[after]     # high half: rows-per-team floor
        op.ctx.set_tuning(hip.TUNE_FUSED_VARIANT, variant)
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, 0)
        for launch in range(4):                                       # parities 0, 1, 0, 1; plain, plain, accelerated, accelerated
            accel = launch >= 2
            c = _state(op, b, mu, x0)
            got = c.step_accel(tau, 0.25, False) if accel else c.step(tau)
            _assert_step_matches(c, got, refs[accel], n, m)
    finally:
        op.close()


def test_setup_passes_use_the_one_pass_kernel_at_large_n():
    """fh_init / fh_gradient_at: z = A x and g = A^T(z - b) from one read of A when n >= 32768."""
    rng = np.random.RandomState(8)
    m, n = 260, 32768
    A = rng.randn(m, n) / 100
    b, x = rng.randn(m), rng.randn(n)
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx
        c.set_loss_lsq(b)
        c.set_vector(hip.VEC_X0, x)
        c.timing_enable(True)
        s = c.init()
        c.set_vector(hip.VEC_T0, x * 0.5)
        c.gradient_at(hip.VEC_T0, hip.VEC_T2)
        assert c.timing_get(hip.K_FUSED)[1] == 2                   # init + gradient_at each took the one-pass kernel
        np.testing.assert_allclose(s[hip.S_FSQ], np.sum((A @ x - b) ** 2), rtol=1e-12)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G0, n), A.T @ (A @ x - b), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(c.get_vector(hip.VEC_T2, n), A.T @ (A @ (0.5 * x) - b), rtol=1e-10, atol=1e-13)
    finally:
        op.close()


def test_coresidency_probe_says_yes_for_the_cus_and_no_beyond():
    """fh_fused_supported consults no environment variable for hidden CUs: a probe launch of one whole-CU workgroup per
    reported CU must see all of them running at once.  One workgroup MORE can never be co-resident: the probe must say so, quickly."""
    import time
    op = fa.DenseMatrixMap(np.ones((64, 20000)))
    try:
        c = op.ctx
        assert c.fused_supported() == 1                     # probes on first use
        ncu, used = c.cu_count()                            # read from the device, not assumed
        assert used == ncu and ncu >= 8
        assert c.coresident_probe(ncu)
        t0 = time.time()
        assert not c.coresident_probe(ncu + 1)
        assert time.time() - t0 < 1.0
        assert c.coresident_probe(ncu) and c.coresident_probe(8)      # and the counters are left clean
        s = _state(op, np.ones(64), 0.01, np.zeros(20000)).step(0.1)
        assert np.isfinite(s).all()
    finally:
        op.close()


@pytest.mark.parametrize("m,n,cus", [(300, 4096, 128), (200, 16384, 64), (70, 65536, 128), (40, 131072, 32), (130, 32768, 96), (23, 108000, 64)])
def test_one_pass_kernel_on_a_capped_number_of_cus(m, n, cus):
    """FH_TUNE_FUSED_CUS: the one-pass launch (and its co-residency probe) on `cus` workgroups instead of one per CU of the device --
    what lets two one-pass grids share a device.  Same iterates and sums to rounding (the number of teams, hence the summation order, changes), plain and accelerated; and TWO contexts whose
    caps add up to the device, stepped in turn (each step() waits for its own launch, so these launches run one after the other: the
    CONCURRENT case is test_two_capped_contexts_run_their_launches_at_the_same_time below), agree bit for bit."""
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b, x0 = rng.randn(m), rng.randn(n) * 0.05
    tau, mu = 0.4, 0.03
    op = fa.DenseMatrixMap(A)
    op2 = fa.DenseMatrixMap(A, tuning={hip.TUNE_FUSED_CUS: cus})
    try:
        ref = _state(op, b, mu, x0)
        f0 = ref.step(tau)
        xp0, g0 = ref.get_vector(hip.VEC_XPROX, n), ref.get_vector(hip.VEC_G1, n)
        c = _state(op2, b, mu, x0)
        assert c.cu_count() == (ref.cu_count()[0], cus) and c.fused_supported() in (1, 3)
        f1 = c.step(tau)
        # (not bit-equal: g0 = A^T(A x0 - b) of fh_init is itself summed over another number of teams)
        np.testing.assert_allclose(c.get_vector(hip.VEC_XPROX, n), xp0, rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), g0, rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(f1[:15], f0[:15], rtol=1e-10, atol=1e-18)
        c.commit(False)
        ref.commit(False)
        a0, a1 = ref.step_accel(tau, 0.3, True), c.step_accel(tau, 0.3, True)
        np.testing.assert_allclose(a1[:15], a0[:15], rtol=1e-10, atol=1e-18)
        # two capped contexts stepped alternately (step() is synchronous: one launch at a time)
        total = ref.cu_count()[0]
        if 2 * cus <= total:
            op.ctx.set_tuning(hip.TUNE_FUSED_CUS, cus)
            ra, rb = _state(op, b, mu, x0), _state(op2, b, mu, x0)
            for _ in range(20):
                sa, sb = ra.step(tau), rb.step(tau)                   # a HipTimeout here would fail the test
                assert np.array_equal(sa, sb)                         # same cap, same shape: same bits
    finally:
        op.close()
        op2.close()


def test_two_capped_contexts_run_their_launches_at_the_same_time():
    """Two solves on ONE device, each one-pass grid capped to half of the CUs (FH_TUNE_FUSED_CUS), driven from one host thread through
    fh_step_begin / fh_step_end: both launches are issued before either is waited for.  The HIP events that bracket each launch on its
    own stream must show the two intervals OVERLAPPING (fh_timing_overlap) -- the test fails if the launches are serialised --, no
    launch may time out, and every scalar block must equal, bit for bit, that of the same context stepped alone."""
    from fasta_python_amd import synthetic
    m, n = 8192, 16384                                  # 1 GiB per context: a launch on half the device takes ~0.3 ms
    ops = [fa.DenseMatrixMap.synthetic(m, n, seed=5, scale=synthetic.lasso_scale(m, n)) for _ in range(2)]
    try:
        dev_cus = ops[0].ctx.cu_count()[0]
        cus = dev_cus // 2 // 32 * 32
        assert cus >= 32
        rng = np.random.RandomState(3)
        b, x0 = rng.randn(m), rng.randn(n) * 0.01
        tau, mu, steps = 0.4, 0.02, 12
        for op in ops:
            op.ctx.set_tuning(hip.TUNE_FUSED_CUS, cus)

        def fresh(op):
            c = _state(op, b, mu, x0)
            assert c.fused_supported() in (1, 3) and c.cu_count() == (dev_cus, cus)
            return c
        # each context alone: the reference bits (and a calling sequence check: begin/end == step)
        alone = []
        c = fresh(ops[0])
        for _ in range(steps):
            c.step_begin(tau)
            with pytest.raises(hip.HipError, match="fh_step_end"):
                c.commit(False)                                       # the context is busy until step_end
            alone.append(c.step_end())
            c.commit(False)
        c = fresh(ops[0])
        for k in range(steps):
            assert np.array_equal(c.step(tau), alone[k])
            c.commit(False)
        # both at once
        ca, cb = fresh(ops[0]), fresh(ops[1])
        for c in (ca, cb):
            c.timing_reset()
            c.timing_enable(True)
        overlaps = []
        for k in range(steps):
            ca.step_begin(tau)
            cb.step_begin(tau)
            sa, sb = ca.step_end(), cb.step_end()                     # a HipTimeout here would fail the test
            assert np.array_equal(sa, alone[k]) and np.array_equal(sb, alone[k])
            overlaps.append(ca.timing_overlap(cb, hip.K_FUSED))
            ca.commit(False)
            cb.commit(False)
        for c in (ca, cb):
            c.timing_enable(False)
        frac = [both / min(a_ms, b_ms) for a_ms, b_ms, both in overlaps]
        print("\nlaunch a / launch b / both running (ms):", [tuple(round(v, 3) for v in o) for o in overlaps[:4]], "...")
        assert min(frac) > 0.0, f"two launches did not overlap at all: {overlaps}"
        assert sorted(frac)[len(frac) // 2] > 0.5, f"the launches overlap for less than half of the shorter one: {overlaps}"
    finally:
        for op in ops:
            op.close()


@pytest.mark.parametrize("m,n,expect", [(40000, 4096, "cyclic"),      # one member per team, 256 teams: 157 rows each
                                        (20000, 8192, "cyclic"),      # two members, 128 teams: 157 rows each
                                        (6000, 16384, "blocked"),     # four members, 64 teams: 94 rows each
                                        (3000, 4096, "blocked")])     # 12 rows each
def test_rows_are_dealt_cyclically_from_128_rows_per_team_on(m, n, expect):
    """Round 6 (profiles/r06_placement.txt): team t of the one-pass kernel takes rows t, t + T, ... (FH_TUNE_FUSED_VARIANT bit 32) where a team has at least 128 rows,
    a contiguous block below; a caller's word is passed on unchanged.  The dealing decides the order in which g1 = A^T r is summed, so the default must equal the explicit
    word of the expected dealing BIT FOR BIT and the other one to rounding; both must equal the two-launch step and NumPy."""
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b, x0 = rng.randn(m), rng.randn(n) * 0.05
    tau, mu = 0.4, 0.03
    op = fa.DenseMatrixMap(A)
    try:
        ref = _pair_reference(op, b, mu, x0, tau, False)
        c = _state(op, b, mu, x0)
        default = c.step(tau)
        _assert_step_matches(c, default, ref, n, m)
        g_default, z_default = c.get_vector(hip.VEC_G1, n), c.get_vector(hip.VEC_Z, m)
        got = {}
        other = "blocked" if expect == "cyclic" else "cyclic"
        for name, word in (("cyclic", 2 | 32), ("blocked", 2)):
            op.ctx.set_tuning(hip.TUNE_FUSED_VARIANT, word)
            ref_w = _pair_reference(op, b, mu, x0, tau, False)                    # (fh_init's gradient g0 comes from the one-pass kernel too: a reference per dealing)
            c = _state(op, b, mu, x0)
            _assert_step_matches(c, c.step(tau), ref_w, n, m)
            got[name] = c.get_vector(hip.VEC_G1, n)
            if name == expect:
                assert np.array_equal(c.get_vector(hip.VEC_Z, m), z_default)
            ref_acc = _pair_reference(op, b, mu, x0, tau, True)                   # FISTA's extrapolation and restart dot under the same dealing
            c = _state(op, b, mu, x0)
            _assert_step_matches(c, c.step_accel(tau, 0.25, False), ref_acc, n, m)
        assert np.array_equal(g_default, got[expect])
        assert not np.array_equal(got[expect], got[other])                        # (the two dealings do sum in different orders)
        np.testing.assert_allclose(got[other], got[expect], rtol=1e-9, atol=1e-13)
        xp = fo.shrink(x0 - tau * (A.T @ (A @ x0 - b)), tau * mu)
        np.testing.assert_allclose(g_default, A.T @ (A @ xp - b), rtol=1e-10, atol=1e-14)
    finally:
        op.close()
