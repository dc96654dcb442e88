"""The bounded in-launch hand-offs that arrived in round 5, made to fail on purpose (FH_TUNE_TEST_HOOKS, csrc/fh_experimental.h): from a spin's
expiry there must be no way to a `Convergence` that is not either the un-sabotaged result or an exception.
  * the multi-workgroup clipping-level search (csrc/fh_prox.h, n in (16384, 262144]): a workgroup whose peers' records do not arrive searches
    alone -- same sums in the same order, the same level to the bit; without that fall-back the level is NaN, which the prox PROPAGATES and the
    library reports as FH_E_TIMEOUT (round 5: prox_scalar mapped NaN to "level <= 0": zeros / identity, finite and wrong, no error);
  * fh_run's persistent launch (csrc/fh_run.h): a grid barrier that times out ends the launch with the state of the last completed iteration
    in place; the driver adopts it and carries on with the library's host-side loop (round 5: FH_E_STATE, the solve thrown away)."""
import time
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip

pytestmark = pytest.mark.gpu


def _level_problem(n=40000, m=48, seed=0):
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:n // 80]] = rng.choice([-1.0, 1.0], n // 80)
    return A, A @ xt + 0.01 * rng.randn(m), xt


@pytest.mark.parametrize("kind", ["l1ball", "linf"])
@pytest.mark.parametrize("driver", ["library", "python"])
def test_level_search_whose_hand_off_times_out_finishes_alone_with_the_same_level(kind, driver):
    A, b, xt = _level_problem()
    n = A.shape[1]
    reg = fa.L1Ball(0.7 * np.abs(xt).sum()) if kind == "l1ball" else fa.LinfProx(0.05)
    ls = fa.LeastSquares(b)
    opts = dict(verbose=False, backend="hip", max_iters=4, tolerance=0.0, evaluate_objective=True, driver=driver)
    op = fa.DenseMatrixMap(A)
    try:
        np.random.seed(3)
        want = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), **opts)
        assert op.ctx.recovered_count(hip.RECOVERED_LEVEL_FALLBACK) == 0
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_LEVEL_WITHHOLD)       # the last workgroup's record never arrives: 0.2 s per search, then alone
        np.random.seed(3)
        t0 = time.time()
        got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), **opts)
        launches = got.iteration_count + got.backtracks
        assert time.time() - t0 < 2.0 + 0.6 * launches
        assert op.ctx.recovered_count(hip.RECOVERED_LEVEL_FALLBACK) >= launches and op.ctx.recovered_count(hip.RECOVERED_LEVEL_FAILED) == 0
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, 0)
    finally:
        op.close()
    assert got.iteration_count == want.iteration_count == 4 and got.backtracks == want.backtracks
    for f in ("residuals", "norm_residuals", "stepsizes", "objectives"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), f          # the SAME level, not a close one
    assert np.array_equal(got.solution, want.solution)
    assert np.all(np.isfinite(got.solution)) and np.any(got.solution != 0.0)
    if kind == "l1ball":                     # (a projection onto the ball: the iterate sits on it or inside it)
        assert np.abs(got.solution).sum() <= reg.mu * (1 + 1e-9)


@pytest.mark.parametrize("driver", ["library", "python"])
def test_level_search_without_its_fall_back_raises_and_never_returns_a_wrong_prox(driver):
    A, b, xt = _level_problem(seed=1)
    n = A.shape[1]
    ls = fa.LeastSquares(b)
    op = fa.DenseMatrixMap(A)
    try:
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_LEVEL_WITHHOLD | hip.HOOK_LEVEL_NO_FALLBACK)
        for reg in (fa.L1Ball(0.7 * np.abs(xt).sum()), fa.LinfProx(0.05)):
            np.random.seed(3)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")         # (the one-pass launch is dropped first: "one-pass kernel disabled"; then K-fwd fails the same way)
                with pytest.raises(hip.HipTimeout, match="clipping-level search"):
                    fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, backend="hip", max_iters=4, tolerance=0.0, driver=driver)
        assert op.ctx.recovered_count(hip.RECOVERED_LEVEL_FAILED) >= 2
        # the entry points themselves: typed status, NaN -- not zeros, not the identity -- in the prox output
        c = op.ctx
        c.set_prox(hip.PROX_LINF, 0.05)
        c.set_vector(hip.VEC_X0, np.linspace(-1, 1, n))
        c.init()
        for call in (lambda: c.fwd(0.3), lambda: c.step(0.3), lambda: c.fwd_adj(0.3)):
            with pytest.raises(hip.HipTimeout):
                call()
            assert np.all(np.isnan(c.get_vector(hip.VEC_XPROX, n)))
        c.set_tuning(hip.TUNE_TEST_HOOKS, 0)            # the context is usable again at once (counters left clean by the failed launches)
        s = c.fwd(0.3)
        assert np.isfinite(s[hip.S_ALPHA]) and np.all(np.isfinite(c.get_vector(hip.VEC_XPROX, n)))
    finally:
        op.close()


def _run_problem(m, n, seed=4):
    rng = np.random.RandomState(seed)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 50)]] = 1
    return A, A @ xt + 0.01 * rng.randn(m)


RUN_MODES = {"adaptive": dict(), "accelerated": dict(adaptive=False, accelerate=True), "forced_backtracking": dict(L=1.0, tau0=5000.0)}


@pytest.mark.parametrize("attempt", [1, 2, 9])
@pytest.mark.parametrize("mode", sorted(RUN_MODES))
@pytest.mark.parametrize("m,n", [(300, 2000), (260, 5000)])          # a workgroup owns whole rows in LDS / 10 pieces per lane, g0 and x_accel0 from L2
def test_a_grid_barrier_timeout_of_the_device_loop_keeps_the_solve(m, n, mode, attempt):
    """The last workgroup stays away from the first grid barrier of attempt `attempt` of EVERY persistent launch: the launch ends after the bounded
    spins (0.5 s) with stopped = 3, the host adopts the state of the last completed iteration -- none at attempt 1 --, the driver warns, carries
    on with the library's host-side loop and tries the device loop again later (which fails again: the hook stays on).  Iteration and backtrack
    counts and the solution must be those of the un-sabotaged run."""
    A, b = _run_problem(m, n)
    ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
    opts = dict(verbose=False, backend="hip", max_iters=90, tolerance=1e-7, evaluate_objective=True, device_iters=32, **RUN_MODES[mode])
    op = fa.DenseMatrixMap(A, tuning={hip.TUNE_RUN_MAX_N: 7168})          # (the wide shape with few rows lies outside the window fh_run is offered in by default)
    try:
        np.random.seed(5)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), **opts)
        assert want.device_steps == want.iteration_count and op.ctx.recovered_count(hip.RECOVERED_RUN_TIMEOUT) == 0
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, attempt << hip.HOOK_RUN_ATTEMPT_SHIFT)
        np.random.seed(5)
        t0 = time.time()
        with pytest.warns(UserWarning, match="fh_run.*timed out"):
            got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), **opts)
        assert time.time() - t0 < 8.0
        assert 1 <= op.ctx.recovered_count(hip.RECOVERED_RUN_TIMEOUT) <= 2
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, 0)
    finally:
        op.close()
    assert got.device_steps < got.iteration_count and got.device_steps + got.library_steps == got.iteration_count
    if mode == "forced_backtracking":
        assert want.backtracks >= 4
    assert got.iteration_count == want.iteration_count and got.backtracks == want.backtracks
    k = min(got.iteration_count, 40)
    for f in ("residuals", "norm_residuals", "stepsizes"):
        np.testing.assert_allclose(getattr(got, f)[:k], getattr(want, f)[:k], rtol=1e-6, atol=1e-300, err_msg=f)
    np.testing.assert_allclose(got.objectives[:got.iteration_count + 1], want.objectives[:got.iteration_count + 1], rtol=1e-8)
    np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-7 if n > 4096 else 1e-9)


def test_fh_run_reports_the_timeout_with_a_typed_status_and_the_completed_history():
    """The C ABI side of the same path: FH_E_TIMEOUT (not FH_E_STATE), state.stopped = 3, steps_done = the completed iterations, and the context
    continues with fh_iterate from exactly there."""
    A, b = _run_problem(200, 1500, seed=6)
    n = A.shape[1]
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx
        c.set_loss_lsq(b)
        c.set_prox(hip.PROX_SHRINK, 0.02)
        o = hip.RunOpts()
        o.adaptive, o.backtrack, o.stop_rule, o.window, o.max_backtracks, o.stepsize_shrink, o.tolerance = 1, 1, 3, 10, 20, 0.2, 0.0
        o.launch_mode = hip.LAUNCH_ONEPASS_SPECULATIVE

        def fresh():
            c.set_vector(hip.VEC_X0, np.zeros(n))
            s = c.init()
            st = hip.RunState()
            st.tau_next, st.alpha1, st.max_residual, st.best_quality = 0.3, 1.0, -np.inf, np.inf
            st.f_window[0] = .5 * np.float64(np.sqrt(s[hip.S_FSQ])) ** 2
            st.onepass_off_until, st.onepass_backoff = -1, 64
            return st

        st = fresh()
        ref = c.run(12, o, st)
        assert len(ref) == 12 and st.stopped == 0
        x_ref = c.get_vector(hip.VEC_X0, n)
        c.set_tuning(hip.TUNE_TEST_HOOKS, 6 << hip.HOOK_RUN_ATTEMPT_SHIFT)
        st = fresh()
        h = c.run(12, o, st)                                   # (hip.run() does not raise FH_E_TIMEOUT: it hands the partial block back)
        done = len(h)
        assert st.stopped == 3 and done == int(st.iteration) and done < 12
        attempts = np.cumsum(1 + ref[:, 5])                      # attempts (iterations + their retries) used up to and including iteration j
        assert (attempts[done - 1] if done else 0) <= 5 < attempts[done]       # five attempts completed, the sixth was interrupted
        assert np.array_equal(h, ref[:done])
        assert st.tau_next == (ref[done, 2] if ref[done, 5] == 0 else st.tau_next)      # the step the interrupted iteration began with
        lib = c.lib
        c.set_tuning(hip.TUNE_TEST_HOOKS, 1 << hip.HOOK_RUN_ATTEMPT_SHIFT)
        hist3, done3 = np.empty((3, hip.RUN_HIST)), hip._i32(7)
        status = lib.fh_run(c._h, 3, hip.C.byref(o), hip.C.byref(st), hist3.ctypes.data_as(hip._pd), hip.C.byref(done3))
        assert status == hip.E_TIMEOUT and b"timed out" in lib.fh_last_error() and done3.value == 0 and st.stopped == 3 and st.iteration == done
        c.set_tuning(hip.TUNE_TEST_HOOKS, 0)
        st.stopped = 0
        rest = c.iterate(12 - done, o, st)                      # per-iteration launches from the adopted state
        assert len(rest) == 12 - done and st.iteration == 12
        np.testing.assert_allclose(np.vstack([h, rest])[:, :4], ref[:, :4], rtol=1e-6)
        np.testing.assert_allclose(c.get_vector(hip.VEC_X0, n), x_ref, rtol=1e-6, atol=1e-10)
    finally:
        op.close()


def test_a_hand_off_timeout_inside_a_chain_of_launches_keeps_the_solve():
    """The chained form of the device loop (opt-in, FH_TUNE_RUN_CHAIN; k_fused_chain): a team member withholds its first partial in EVERY launch
    (fault-injection bit 1), so the first launch of the chain reports a hand-off timeout -- its finaliser marks the state `stopped = 3`, the
    rest of the chain returns at once, the host adopts the (unchanged) state and the driver carries on with the library loop, whose
    one-pass launches time out the same way and fall back to K-fwd / K-adj.  The solve must equal the two-launch solve."""
    rng = np.random.RandomState(2)
    m, n = 96, 8192
    A = rng.randn(m, n) / 40
    b = rng.randn(m)
    ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
    opts = dict(verbose=False, backend="hip", tolerance=1e-6, max_iters=12, evaluate_objective=True)
    op = fa.DenseMatrixMap(A, tuning={hip.TUNE_RUN_CHAIN: 1})
    try:
        np.random.seed(4)
        ref = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), fused=False, **opts)
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_WITHHOLD_PARTIAL)
        np.random.seed(4)
        with pytest.warns(UserWarning) as rec:
            got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), device_iters=8, **opts)
        op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, 0)
        assert op.ctx.recovered_count(hip.RECOVERED_RUN_TIMEOUT) == 1
    finally:
        op.close()
    text = " | ".join(str(w.message) for w in rec)
    assert "fh_run" in text and "timed out" in text and "one-pass kernel disabled" in text
    assert got.device_steps == 0 and got.library_steps == got.iteration_count == ref.iteration_count
    k = got.iteration_count
    np.testing.assert_allclose(got.residuals[:k], ref.residuals[:k], rtol=1e-12)
    np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-12, atol=1e-15)

