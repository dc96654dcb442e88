"""fh_iterate (csrc/fh_host_iterate.h): the FBS loop of fasta/__init__.py:171-312 driven from the host side of the LIBRARY -- the default
driver of fasta() since round 6 -- against the Python driver of rounds 1-5 (FBSolver.step, driver="python").  Both issue the same launches
and read the same scalar blocks, and the library takes the reference's decisions with the same float64 expressions (down to calling the C
library's pow for NumPy's `x ** 2`), so every history must be EQUAL, bit for bit, on every operator, loss, prox and sharding form --
wherever the calls are cut.  Parity with the reference itself is what the other -m gpu modules check (they run through this driver
by default); here the reference-captured fixtures are additionally compared at rtol 1e-6 with calls of 8 iterations."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip, stopping
from tests import gpu_util as G
from tests import helpers as H

pytestmark = pytest.mark.gpu

FIELDS = ("residuals", "norm_residuals", "stepsizes", "objectives", "function_hist")
PREFIX = {"tv_32x32_adaptive": 40, "linf_96x96_adaptive": 40, "linf_96x96_accelerated": 100, "linf_96x96_plain": 100,
          "nnls_under_first40": 25, "sparse_ls_unnormalised_backtracks": 25, "sparse_ls_opt_window3_shrink": 25}


def _needs_python(options):
    return bool(options.get("record_iterates") or options.get("func"))


def _assert_identical(a, b, what=""):
    assert a.iteration_count == b.iteration_count and a.backtracks == b.backtracks, what
    k = a.iteration_count
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None) == (y is None), (what, f)
        if x is not None:
            hi = k + 1 if f in ("objectives", "function_hist") else k
            assert np.array_equal(np.asarray(x)[:hi], np.asarray(y)[:hi], equal_nan=True), (what, f, np.flatnonzero(np.asarray(x)[:hi] != np.asarray(y)[:hi])[:4])
    assert np.array_equal(a.solution, b.solution, equal_nan=True), what


@pytest.mark.parametrize("name", H.golden_cases())
def test_library_loop_is_bit_identical_to_the_python_driver_on_every_fixture(name):
    """All 36 reference-captured fixtures (dense / l-infinity / l1-ball / stencil / logistic; every mode, stop rule and option): driver="python"
    against the default driver, and against calls of 8 iterations -- EQUAL histories, counts and solutions; and the fixture itself at rtol 1e-6
    through calls of 8."""
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    g_none = meta["options"].get("g_none", False)
    py = G.run_hip(meta["kind"], data, dict(meta["options"], driver="python"), meta["solver_seed"], g_none=g_none)
    lib = G.run_hip(meta["kind"], data, meta["options"], meta["solver_seed"], g_none=g_none)
    lib8 = G.run_hip(meta["kind"], data, dict(meta["options"], driver="library", device_iters=8), meta["solver_seed"], g_none=g_none)
    assert py.library_steps == 0 and py.device_steps == 0
    for c in (lib, lib8):
        assert c.device_steps == 0 and c.library_steps == (0 if _needs_python(meta["options"]) else c.iteration_count)
        _assert_identical(c, py, name)
    get = lambda f: z[f] if f in z.files else None
    k = PREFIX.get(name)
    if k:       # long backtracking-heavy runs are pinned on their first k iterations, as everywhere in the suite (tests/test_gpu_prox_tv.py, test_gpu_run.py)
        G.compare_histories(lib8, get, min(k, lib8.iteration_count), rtol=1e-6, atol=1e-14, fields=("residuals", "stepsizes"))
        return
    assert lib8.iteration_count == int(z["iteration_count"]) and lib8.backtracks == int(z["backtracks"])
    G.compare_histories(lib8, get, lib8.iteration_count, rtol=1e-6, atol=1e-13)
    np.testing.assert_allclose(lib8.solution, z["solution"], rtol=1e-5, atol=1e-9)


def _problem(kind, rng):
    if kind == "tv":
        M = np.kron(rng.randint(0, 2, (4, 6)).astype(float), np.ones((8, 8))) + 0.1 * rng.randn(32, 48)
        return fa.GradDivMap(M.shape), fa.LeastSquares(M / 0.1), fa.TVDualBall(), np.zeros(M.shape + (2,))
    m, n = {"l1ball": (80, 20000), "linf": (64, 17000), "wide": (48, 9000), "f32": (300, 5000), "blocks": (400, 3000), "logistic": (120, 300)}[kind]
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:max(1, n // 60)]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    if kind == "logistic":
        return fa.DenseMatrixMap(A), fa.LogisticLoss(np.sign(rng.randn(m))), fa.Shrink(0.05), np.zeros(n)
    if kind == "blocks":
        return fa.ShardedDenseMatrixMap(A, devices=[0, 0, 0, 0]), fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n)
    op = fa.DenseMatrixMap(A, storage="f32") if kind == "f32" else fa.DenseMatrixMap(A)
    reg = {"l1ball": fa.L1Ball(0.8 * np.abs(xt).sum()), "linf": fa.LinfProx(0.02)}.get(kind, fa.Shrink(0.02))
    return op, fa.LeastSquares(b), reg, np.zeros(n)


MODES = {"adaptive": dict(), "accelerated": dict(adaptive=False, accelerate=True), "plain": dict(adaptive=False),
         "forced_backtracking": dict(L=1.0, tau0=3000.0), "accel_adaptive_window3": dict(accelerate=True, window=3, stepsize_shrink=0.4)}


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("kind", ["tv", "l1ball", "linf", "wide", "f32", "blocks", "logistic"])
def test_the_call_length_does_not_change_a_single_bit(kind, mode):
    """1, 3 and 1000 iterations per library call, the time-sized default and the Python driver: the state that travels between calls (step size,
    alpha, f window, maximal residual, best quality, the launch policy's cool-down) makes the solve independent of where the calls are cut --
    on the stencil, both level-search prox kinds (n > 16384: several workgroups), a team-of-two width, float32 storage, four row blocks in
    this process and the logistic loss.  The launch counters agree too: the same kernels produced the same numbers."""
    if kind == "tv" and mode == "forced_backtracking":
        pytest.skip("L / tau0 of the dense recipe")
    rng = np.random.RandomState(17)
    op, loss, reg, x0 = _problem(kind, rng)
    opts = dict(tolerance=1e-7, max_iters=45, evaluate_objective=True, verbose=False, **MODES[mode])
    runs = {}
    try:
        for tag, kw in (("python", dict(driver="python")), ("auto", {}), ("1", dict(driver="library", device_iters=1)),
                        ("3", dict(driver="library", device_iters=3)), ("1000", dict(driver="library", device_iters=1000))):
            np.random.seed(5)
            solver = fa.FBSolver(op, loss, reg, x0, **opts, **kw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                runs[tag] = (solver.setup().run(), solver.fused_steps, solver.pair_steps, solver.mode)
    finally:
        op.close()
    ref, fused_steps, pair_steps, mode_name = runs["python"]
    assert ref.iteration_count > 5
    if mode == "forced_backtracking":
        assert ref.backtracks >= 3
    for tag in ("auto", "1", "3", "1000"):
        c, fs, ps, mn = runs[tag]
        assert c.library_steps == c.iteration_count
        _assert_identical(c, ref, (kind, mode, tag))
        assert (fs, ps, mn) == (fused_steps, pair_steps, mode_name), (kind, mode, tag)


@pytest.mark.parametrize("rule", hip.STOP_RULES)
def test_the_four_stop_rules_fire_at_the_same_iteration(rule):
    rng = np.random.RandomState(3)
    op, loss, reg, x0 = _problem("wide", rng)
    try:
        res = []
        for kw in (dict(driver="python"), dict(driver="library", device_iters=7)):
            np.random.seed(2)
            res.append(fa.fasta(op, loss.f, loss.gradf, reg.g, reg.prox, x0, verbose=False, tolerance=3e-3, max_iters=200,
                                stop_rule=getattr(stopping, rule), backend="hip", **kw))
    finally:
        op.close()
    assert 1 < res[0].iteration_count < 200
    _assert_identical(res[1], res[0], rule)


@pytest.mark.parametrize("mode", ["adaptive", "accelerated"])
def test_verbose_output_is_the_reference_text_whoever_drives_the_loop(mode, capsys):
    """verbose=True is the reference's default (fasta/__init__.py:42): header (:118-120), one line per iteration (:302-306) and "Restarted
    acceleration." (:233) -- the library and device loops print them from their history records after each call: same text, same order."""
    rng = np.random.RandomState(9)
    m, n = 200, 700
    A = rng.randn(m, n) / 30
    b = rng.randn(m)
    ls, reg = fa.LeastSquares(b), fa.Shrink(0.05)
    op = fa.DenseMatrixMap(A)
    texts = {}
    try:
        for tag, kw in (("python", dict(driver="python")), ("library", dict(driver="library", device_iters=8)), ("device", dict(device_iters=8))):
            np.random.seed(1)
            c = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), backend="hip", max_iters=150, tolerance=0.0, evaluate_objective=True,
                         **MODES[mode], **kw)
            texts[tag] = (capsys.readouterr().out, c)
    finally:
        op.close()
    out, c = texts["python"]
    assert out.startswith("Initializing FASTA...\n\nIteration #\tResidual\tStepsize\tAccel. param\tBacktracks\tObjective\n")
    assert out.count("\n[") == c.iteration_count
    if mode == "accelerated":
        assert "Restarted acceleration." in out
    assert texts["library"][0] == out and texts["library"][1].library_steps == c.iteration_count
    # the device loop sums in another order (rtol 1e-6 histories): same lines, same restarts, numbers equal to the printed digits' tolerance
    dev_out, dev = texts["device"]
    assert dev.device_steps == dev.iteration_count == c.iteration_count
    assert [ln.split("\t")[0] for ln in dev_out.splitlines()] == [ln.split("\t")[0] for ln in out.splitlines()]
    for a, b_ in zip(dev_out.splitlines()[3:43], out.splitlines()[3:43]):      # (numbers: the first 40 iterations, as everywhere for the device loop)
        if a.startswith("["):
            np.testing.assert_allclose([float(v) for v in a.split("\t")[1:]], [float(v) for v in b_.split("\t")[1:]], rtol=1e-5)


def test_a_one_pass_timeout_inside_a_library_call_falls_back_and_warns():
    """The bounded spins of the one-pass kernel run out inside fh_iterate (fault-injection bit): the call itself must carry on with K-fwd / K-adj,
    the driver must hear about it (a warning, use_fused off, the back-off doubled) and the solve must equal the two-launch solve."""
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
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, fused=True, driver="library", device_iters=5, **opts)
        np.random.seed(4)
        with pytest.warns(UserWarning, match="one-pass kernel disabled"):
            got = solver.setup().run()
    finally:
        op.close()
    assert not solver.use_fused and solver._fused_backoff == 128 and solver.fused_steps == 0 and got.library_steps == got.iteration_count
    assert got.iteration_count == ref.iteration_count
    k = got.iteration_count
    np.testing.assert_allclose(got.residuals[:k], ref.residuals[:k], rtol=1e-12)
    np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-12, atol=1e-15)


def test_fh_iterate_reports_completed_iterations_when_a_later_one_fails():
    """C ABI contract: on an error the completed iterations stay committed and are reported (state, history, steps_done)."""
    rng = np.random.RandomState(5)
    m, n = 64, 20000
    A = rng.randn(m, n) / 100
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx
        c.set_loss_lsq(rng.randn(m))
        c.set_prox(hip.PROX_L1BALL, 5.0)
        c.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
        s = c.init()
        o = hip.RunOpts()
        o.adaptive, o.backtrack, o.stop_rule, o.window, o.max_backtracks, o.stepsize_shrink, o.tolerance = 1, 1, 3, 10, 20, 0.2, 0.0
        o.launch_mode = hip.LAUNCH_ONEPASS_SPECULATIVE
        st = hip.RunState()
        st.tau_next, st.alpha1, st.max_residual, st.best_quality = 0.05, 1.0, -np.inf, np.inf
        st.f_window[0] = .5 * s[hip.S_FSQ]
        st.onepass_off_until, st.onepass_backoff = -1, 64
        h = c.iterate(3, o, st)
        assert len(h) == 3 and st.iteration == 3 and st.stopped == 0
        c.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_LEVEL_WITHHOLD | hip.HOOK_LEVEL_NO_FALLBACK)      # the level search now finds no level at all
        with pytest.raises(hip.HipTimeout) as exc:
            c.iterate(5, o, st)
        assert len(exc.value.partial) == 0 and st.iteration == 3
        c.set_tuning(hip.TUNE_TEST_HOOKS, 0)
        x_before = c.get_vector(hip.VEC_X0, n)
        h = c.iterate(2, o, st)                                   # the failed iteration changed nothing: the solve simply goes on
        assert len(h) == 2 and st.iteration == 5 and np.all(np.isfinite(h[:, :4]))
        assert not np.array_equal(c.get_vector(hip.VEC_X0, n), x_before)
    finally:
        op.close()


@pytest.mark.parametrize("kind", ["dense_one_pass", "dense_two_launches", "dense_pair", "tv", "tv_accelerated", "l1ball"])
def test_scalars_by_sequence_number_are_the_scalars_after_a_stream_synchronisation(kind):
    """Round 6: a single-device step no longer waits for the launch's completion signal but for a sequence number the launch's finaliser
    writes behind the host-mapped scalar block (csrc/fh_device.h:publish_seq, FH_TUNE_SEQ_POLL).  Whatever the host reads after that number
    must be what it reads after hipStreamSynchronize: a few hundred iterations of every kind of launch, both ways, EQUAL histories --
    a stale entry of the block (the first form of the hand-off had one) changes a decision or a residual somewhere."""
    rng = np.random.RandomState(23)
    if kind.startswith("tv"):
        op, loss, reg, x0 = _problem("tv", rng)
        extra = dict(adaptive=False, accelerate=True) if kind == "tv_accelerated" else {}
    else:
        m, n = (64, 20000) if kind == "l1ball" else ((700, 1100) if kind != "dense_pair" else (300, 900))
        A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
        b = rng.randn(m)
        op, loss, x0 = fa.DenseMatrixMap(A), fa.LeastSquares(b), np.zeros(n)
        reg = fa.L1Ball(4.0) if kind == "l1ball" else fa.Shrink(0.03)
        extra = dict(fused=False) if kind == "dense_two_launches" else {}
    runs = []
    try:
        if kind == "dense_pair":
            op.ctx.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_PROBE_SAYS_NO)          # no one-pass kernel: K-fwd + K-adj under one wait
        for poll in (1, 0, 1):
            op.ctx.set_tuning(hip.TUNE_SEQ_POLL, poll)
            np.random.seed(8)
            solver = fa.FBSolver(op, loss, reg, x0, verbose=False, max_iters=300, tolerance=0.0, evaluate_objective=True, **extra)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                runs.append((solver.setup().run(), solver.mode))
    finally:
        op.close()
    assert runs[0][1] == {"dense_one_pass": "speculative", "dense_two_launches": None, "dense_pair": "pair"}.get(kind, "always")      # (l1ball: n >= 16384)
    assert runs[0][0].iteration_count == 300
    _assert_identical(runs[1][0], runs[0][0], kind)
    _assert_identical(runs[2][0], runs[0][0], kind)

