"""fasta(..., device_iters=K): the FBS loop itself on the device (fh_run, csrc/fh_run.h) -- backtracking test and retries, FISTA restart
and alpha recursion, Barzilai-Borwein step, residuals, best iterate and the four built-in stop rules decided inside ONE persistent
launch per K iterations -- against the reference's recorded runs, the oracle, and the per-iteration path of the same library."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip, stopping
from oracle import fasta_np as fo
from oracle import problems as pr
from tests import gpu_util as G
from tests import helpers as H

pytestmark = pytest.mark.gpu

DENSE = [n for n in H.golden_cases() if n.split("_")[0] in ("sparse", "nnls", "c1", "logistic")]
PREFIX = {"sparse_ls_unnormalised_backtracks": 25, "nnls_under_first40": 25, "sparse_ls_opt_window3_shrink": 25}


def _needs_the_host(options):
    return bool(options.get("record_iterates") or options.get("func"))


@pytest.mark.parametrize("K", [8, 1])
@pytest.mark.parametrize("name", DENSE)
def test_golden_parity_with_the_loop_on_the_device(name, K):
    """Every dense fixture captured from the reference, run with device_iters=K: identical iteration and backtrack counts, histories
    rtol 1e-6, solution rtol 1e-5 -- and the iterations really ran inside persistent launches (device_steps), except where an option
    needs the host between iterations (record_iterates / func), which must fall back to the per-iteration path with the same result."""
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    opts = dict(meta["options"], device_iters=K)
    k = PREFIX.get(name)
    if k:
        opts.update(max_iters=k, tolerance=0.0)
    c = G.run_hip(meta["kind"], data, opts, meta["solver_seed"], g_none=meta["options"].get("g_none", False))
    assert c.device_steps == (0 if _needs_the_host(meta["options"]) else c.iteration_count)
    get = lambda f: z[f] if f in z.files else None
    if k:
        G.compare_histories(c, get, k, rtol=1e-6, atol=1e-14, fields=("residuals", "norm_residuals", "stepsizes"))
        return
    assert c.iteration_count == int(z["iteration_count"])
    assert c.backtracks == int(z["backtracks"])
    G.compare_histories(c, get, c.iteration_count, rtol=1e-6, atol=1e-14)
    np.testing.assert_allclose(c.solution, z["solution"], rtol=1e-5, atol=1e-9)


def _solve(A, b, reg, x0, tuning=None, **opts):
    op = fa.DenseMatrixMap(A, tuning=tuning)
    try:
        ls = fa.LeastSquares(b)
        np.random.seed(5)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g, prox = (None, None) if reg is None else (reg.g, reg.prox)
            return fa.fasta(op, ls.f, ls.gradf, g, prox, x0, verbose=False, backend="hip", **opts)
    finally:
        op.close()


MODES = {"adaptive": dict(), "accelerated": dict(adaptive=False, accelerate=True), "plain": dict(adaptive=False),
         "accel_adaptive": dict(accelerate=True), "no_restart": dict(adaptive=False, accelerate=True, restart=False),
         "forced_backtracking": dict(L=1.0, tau0=5000.0), "no_backtrack": dict(backtrack=False), "window3": dict(window=3, stepsize_shrink=0.4)}


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("m,n", [(300, 4096), (96, 160), (700, 2000), (1500, 1000), (40, 3000),
                                 (300, 4100), (520, 5000), (130, 6000), (64, 7000), (600, 7168)])      # n > 4096: 10 / 10 / 12 / 14 / 14 pieces per lane, g0 and x_accel0 from L2
def test_device_loop_equals_the_per_iteration_path_and_the_oracle(m, n, mode):
    # (rows wider than 4096 columns are offered to the persistent launch only inside its measured window -- FH_TUNE_RUN_MAX_N lifts it)
    _device_loop_against_host_and_oracle(m, n, mode, {hip.TUNE_RUN_MAX_N: 7168} if n > 4096 else None)


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("m,n", [(300, 7000), (100, 6500), (520, 8192), (64, 9000), (200, 16384), (9000, 4096), (37, 12345)])
def test_chained_launches_equal_the_per_iteration_path_and_the_oracle(m, n, mode):
    """Round 6: outside the persistent launch's window `device_iters=K` runs as a CHAIN of K one-pass launches (k_fused_chain: step size and
    buffer roles from a device state block, the controller in each launch's finaliser; opt-in, FH_TUNE_RUN_CHAIN) -- teams of 1 (a tall
    4096-column matrix), 2 and 4 members; every mode; same bar as the persistent launch, except with window = 3 (the backtracking-heavy mode:
    30+ backtracks in ~95 iterations), where the solve may end a few iterations apart (95 vs 97 at 520 x 8192): the chain runs the one-pass
    kernel in every attempt, the per-iteration path K-fwd / K-adj for eight iterations after each backtrack -- other summation orders in
    a regime where the oracle parts from its own row-permuted twin (fixture sparse_ls_opt_window3_shrink is pinned
    on a 25-iteration prefix for the same reason)."""
    _device_loop_against_host_and_oracle(m, n, mode, {hip.TUNE_RUN_CHAIN: 1})


def _device_loop_against_host_and_oracle(m, n, mode, tuning):
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 50)]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    opts = dict(tolerance=1e-7, max_iters=120, evaluate_objective=True, **MODES[mode])
    host = _solve(A, b, fa.Shrink(0.02), np.zeros(n), fused=True, **opts)
    dev = _solve(A, b, fa.Shrink(0.02), np.zeros(n), tuning=tuning, device_iters=16, **opts)
    assert host.device_steps == 0 and dev.device_steps == dev.iteration_count
    k = host.iteration_count
    sensitive = bool(tuning and tuning.get(hip.TUNE_RUN_CHAIN)) and mode == "window3"
    if sensitive:      # (see the chain test's docstring) the same minimum a few iterations apart: the first 40 iterations are compared, and where the solves end
        assert abs(dev.iteration_count - k) <= 4 and abs(dev.backtracks - host.backtracks) <= 3
        kk = min(k, dev.iteration_count, 40)
        for f in ("residuals", "norm_residuals", "stepsizes"):
            np.testing.assert_allclose(getattr(dev, f)[:kk], getattr(host, f)[:kk], rtol=1e-6, atol=1e-300, err_msg=f)
        np.testing.assert_allclose(dev.objectives[dev.iteration_count], host.objectives[k], rtol=1e-9)
        np.testing.assert_allclose(dev.solution, host.solution, rtol=1e-4, atol=1e-6)
        return
    assert dev.iteration_count == k and dev.backtracks == host.backtracks
    # Histories are pinned over the first 40 iterations: late in a solve the residuals are ~1e-8 of their start and the adaptive step sizes
    # amplify summation-order rounding (the per-iteration path itself takes K-fwd / K-adj for a few iterations after every backtrack, i.e.
    # other summation orders than the one-pass arithmetic of the device loop); counts and the solution are compared for the whole solve.
    kk = min(k, 40)
    for f in ("residuals", "norm_residuals", "stepsizes"):
        np.testing.assert_allclose(getattr(dev, f)[:kk], getattr(host, f)[:kk], rtol=1e-6, atol=1e-300, err_msg=f)
    np.testing.assert_allclose(dev.objectives[:k + 1], host.objectives[:k + 1], rtol=1e-8)
    # (atol: the signal's entries are 1; entries of 1e-4 that sit next to the shrink threshold after 120 iterations move by ~1e-8 with the summation
    # order -- the per-iteration path sums n = 8192 in teams of two, the device loop in one workgroup -- in the backtracking-heavy modes)
    np.testing.assert_allclose(dev.solution, host.solution, rtol=1e-5, atol=1e-7 if n > 4096 else 1e-9)
    P = pr.sparse_least_squares_from(A, b, 0.02)
    np.random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    if mode == "forced_backtracking":
        assert want.backtracks >= 4
    assert dev.iteration_count == want.iteration_count and dev.backtracks == want.backtracks
    np.testing.assert_allclose(dev.residuals[:kk], want.residuals[:kk], rtol=1e-6)
    np.testing.assert_allclose(dev.stepsizes[:kk], want.stepsizes[:kk], rtol=1e-6)
    np.testing.assert_allclose(dev.objectives[:k + 1], want.objectives[:k + 1], rtol=1e-8)
    np.testing.assert_allclose(dev.solution, want.solution, rtol=1e-5, atol=1e-7 if n > 4096 else 1e-9)


@pytest.mark.parametrize("reg", ["nonneg", "box", "none"])
def test_device_loop_with_the_other_separable_prox_kinds(reg):
    rng = np.random.RandomState(12)
    m, n = 400, 1200
    A = rng.randn(m, n) / 40
    b = rng.rand(m)
    tag = {"nonneg": fa.NonNeg(), "box": fa.Box(-0.01, 0.02), "none": None}[reg]
    opts = dict(tolerance=1e-6, max_iters=80, evaluate_objective=True)
    host = _solve(A, b, tag, np.zeros(n), **opts)
    dev = _solve(A, b, tag, np.zeros(n), device_iters=7, **opts)
    k = host.iteration_count
    assert dev.device_steps == dev.iteration_count == k and dev.backtracks == host.backtracks
    np.testing.assert_allclose(dev.residuals[:k], host.residuals[:k], rtol=2e-6)
    np.testing.assert_allclose(dev.objectives[:k + 1], host.objectives[:k + 1], rtol=1e-8)
    np.testing.assert_allclose(dev.solution, host.solution, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("rule", hip.STOP_RULES)
def test_the_launch_length_does_not_change_a_single_bit(rule):
    """K = 1, 3, 1000 iterations per launch: the state carried between launches (step size, alpha, f window, best iterate, maximal
    residual, buffer roles) must make the solve independent of where the launches are cut."""
    rng = np.random.RandomState(3)
    m, n = 500, 2500
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = rng.randn(m)
    runs = []
    for K in (1, 3, 1000):
        for acc in (False, True):
            runs.append((acc, _solve(A, b, fa.Shrink(0.05), np.zeros(n), device_iters=K, tolerance=1e-4, max_iters=70, accelerate=acc,
                                     adaptive=not acc, stop_rule=getattr(stopping, rule), evaluate_objective=acc)))
    for acc in (False, True):
        same = [r for a, r in runs if a == acc]
        for r in same[1:]:
            assert r.iteration_count == same[0].iteration_count and r.backtracks == same[0].backtracks and r.device_steps == r.iteration_count
            for f in ("residuals", "norm_residuals", "stepsizes", "solution"):
                assert np.array_equal(getattr(r, f), getattr(same[0], f)), f


def test_options_that_need_the_host_between_iterations_keep_the_per_iteration_path():
    rng = np.random.RandomState(4)
    m, n = 200, 600
    A = rng.randn(m, n) / 30
    b = rng.randn(m)
    for extra in (dict(func=lambda x: float(np.abs(x).sum())), dict(record_iterates=True), dict(stop_rule=lambda i, r, nr, mr, tol: i >= 9)):
        got = _solve(A, b, fa.Shrink(0.02), np.zeros(n), device_iters=8, max_iters=30, **extra)
        ref = _solve(A, b, fa.Shrink(0.02), np.zeros(n), max_iters=30, **extra)
        assert got.device_steps == 0 and got.iteration_count == ref.iteration_count
        assert np.array_equal(got.residuals, ref.residuals) and np.array_equal(got.solution, ref.solution)
    wide = _solve(rng.randn(50, 20000) / 100, rng.randn(50), fa.Shrink(0.02), np.zeros(20000), device_iters=8, max_iters=10)     # n > 16384: neither the persistent launch nor the chain
    assert wide.device_steps == 0 and wide.library_steps == wide.iteration_count == 10
    # outside the persistent launch's window nothing runs on the device by default; with FH_TUNE_RUN_CHAIN = 1 the chained form does (n <= 16384)
    on_ = {hip.TUNE_RUN_CHAIN: 1}
    for shape, chained in (((50, 7000), True), ((64, 5000), True), ((4096, 4100), False)):
        b_ = rng.randn(shape[0])
        A_ = rng.randn(*shape) / 100
        default = _solve(A_, b_, fa.Shrink(0.02), np.zeros(shape[1]), device_iters=8, max_iters=10)
        assert default.device_steps == (0 if chained else 10) and default.library_steps == (10 if chained else 0)
        opted = _solve(A_, b_, fa.Shrink(0.02), np.zeros(shape[1]), tuning=on_, device_iters=8, max_iters=10)
        assert opted.device_steps == 10


def test_operators_without_a_device_loop_keep_the_per_iteration_path():
    """device_iters on operators fh_run has no kernel for -- float32 storage, in-process row blocks, the l1-ball prox (its level search is a launch
    of its own), the stencil -- must run the ordinary loop and produce its bits."""
    rng = np.random.RandomState(6)
    m, n = 240, 900
    A = rng.randn(m, n) / 30
    b = rng.randn(m)
    x0 = np.zeros(n)

    def both(make_op, reg):
        out = []
        for kw in (dict(device_iters=8), {}):
            op = make_op()
            try:
                np.random.seed(5)
                ls = fa.LeastSquares(b)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out.append(fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, x0, verbose=False, max_iters=25, tolerance=0.0, backend="hip", **kw))
            finally:
                op.close()
        got, ref = out
        assert got.device_steps == 0 and got.iteration_count == ref.iteration_count == 25
        assert np.array_equal(got.residuals, ref.residuals) and np.array_equal(got.solution, ref.solution)

    both(lambda: fa.DenseMatrixMap(A, storage="f32"), fa.Shrink(0.02))
    both(lambda: fa.ShardedDenseMatrixMap(A, devices=[0, 0, 0]), fa.Shrink(0.02))
    both(lambda: fa.DenseMatrixMap(A), fa.L1Ball(3.0))
    both(lambda: fa.DenseMatrixMap(A), fa.LinfProx(0.05))


@pytest.mark.parametrize("n", [8192, 12000])
def test_the_chain_length_does_not_change_a_single_bit(n):
    """1, 3 and 1000 launches per chain: the state block that travels between the chains makes the solve independent of where they are cut."""
    rng = np.random.RandomState(n)
    m = 300
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b = rng.randn(m)
    for acc in (False, True):
        runs = [_solve(A, b, fa.Shrink(0.05), np.zeros(n), tuning={hip.TUNE_RUN_CHAIN: 1}, device_iters=K, tolerance=1e-4, max_iters=60, accelerate=acc,
                       adaptive=not acc, evaluate_objective=acc) for K in (1, 3, 1000)]
        for r in runs[1:]:
            assert r.iteration_count == runs[0].iteration_count and r.backtracks == runs[0].backtracks and r.device_steps == r.iteration_count
            for f in ("residuals", "norm_residuals", "stepsizes", "solution"):
                assert np.array_equal(getattr(r, f), getattr(runs[0], f)), f


@pytest.mark.parametrize("mode", ["adaptive", "accelerated", "plain"])
def test_full_length_histories_of_the_device_loop_against_the_oracle(mode):
    """the random-problem tests above pin histories on the first 40 iterations.  Here the WHOLE solve, per mode, on a problem outside
    the sensitive regime (overdetermined, m > n: SURVEY section 7 -- re-ordered sums stay within 1e-13 over entire solves there): every iteration of
    every history at rtol 1e-6 against the oracle, to the stop rule's last iteration, through persistent launches of 16."""
    rng = np.random.RandomState(77)
    m, n = 1500, 1000
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:20]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    opts = dict(tolerance=1e-7, max_iters=400, evaluate_objective=True, **MODES[mode])
    dev = _solve(A, b, fa.Shrink(0.02), np.zeros(n), device_iters=16, stop_rule=stopping.residual, **opts)      # (the absolute rule: runs on well past the hybrid rule's stop)
    P = pr.sparse_least_squares_from(A, b, 0.02)
    np.random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), stop_rule=fo.residual, **opts)
    k = want.iteration_count
    assert dev.device_steps == dev.iteration_count == k and dev.backtracks == want.backtracks and 15 < k < 400
    for f in ("residuals", "norm_residuals", "stepsizes"):
        np.testing.assert_allclose(getattr(dev, f)[:k], getattr(want, f)[:k], rtol=1e-6, atol=1e-300, err_msg=f)
    np.testing.assert_allclose(dev.objectives[:k + 1], want.objectives[:k + 1], rtol=1e-10)
    np.testing.assert_allclose(dev.solution, want.solution, rtol=1e-6, atol=1e-12)

