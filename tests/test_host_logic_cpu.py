"""CPU tier for the product's HOST logic: fasta_python_amd.solver driven through a NumPy stand-in for the
device context (tests/fake_ctx.py) on the reference-captured golden fixtures -- backtracking, FISTA restart,
Barzilai-Borwein, stop rules, histories, best iterate, and the speculative one-pass path with its fallback."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import stopping as fstop
from tests import helpers as H
from tests.fake_ctx import FakeDenseMap, FakeStencilMap
from tests.gpu_util import TAGS

CASES = [n for n in H.golden_cases() if not n.startswith("c1_")]


def _run(name, fused, dense_kind=1, **extra):
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    kind = meta["kind"]
    if kind == "tv":
        A = FakeStencilMap(data["M"].shape, fused_kind=2 if fused else 0)
        loss, x0 = fa.LeastSquares(data["M"] / float(data["mu"])), np.zeros(data["M"].shape + (2,))
    else:
        A = FakeDenseMap(data["A"], fused_kind=dense_kind if fused else 0)
        loss = fa.LogisticLoss(data["b"]) if kind == "logistic" else fa.LeastSquares(data["b"])
        x0 = np.zeros(data["A"].shape[1])
    reg = TAGS[kind](data)
    o = H.resolve_options(meta["options"], fstop)
    g, proxg = (None, None) if o.pop("g_none", False) else (reg.g, reg.prox)
    np.random.seed(meta["solver_seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # kind 3 (one-pass kernel available but not recommended at this size) is only used when forced
        c = fa.fasta(A, A.H, loss.f, loss.gradf, g, proxg, x0, verbose=extra.pop("verbose", False),
                     fused=True if (fused and dense_kind == 3 and kind != "tv") else "auto", **dict(o, **extra))
    return meta, z, c, A.ctx


@pytest.mark.parametrize("fused", [False, True, 3])
@pytest.mark.parametrize("name", CASES)
def test_driver_reproduces_reference_run(name, fused):
    """fused: False = two launches; True = one-pass kernel wherever it costs no more than K-fwd (also for the backtracking
    retries); 3 = dense one-pass kernel used speculatively (sizes where a rejected step wastes the A^T half)."""
    meta, z, c, ctx = _run(name, bool(fused), dense_kind=3 if fused == 3 else 1)
    long_chaotic = int(z["backtracks"]) > 50                       # rounding is amplified there
    if not long_chaotic:
        assert c.iteration_count == int(z["iteration_count"])
        assert c.backtracks == int(z["backtracks"])
    k = min(c.iteration_count, int(z["iteration_count"]), 25 if long_chaotic else 10 ** 9)
    for f in ("residuals", "norm_residuals", "stepsizes"):
        np.testing.assert_allclose(getattr(c, f)[:k], z[f][:k], rtol=1e-6, atol=1e-13, err_msg=f)
    if "objectives" in z.files:
        np.testing.assert_allclose(c.objectives[:k + 1], z["objectives"][:k + 1], rtol=1e-6)
    if not long_chaotic:
        np.testing.assert_allclose(c.solution, z["solution"], rtol=1e-5, atol=1e-9)
        if "iterates" in z.files:
            np.testing.assert_allclose(c.iterates[:k + 1], z["iterates"][:k + 1], rtol=1e-5, atol=1e-9)
    accelerated = bool(meta["options"].get("accelerate", False))
    if fused:                                                      # (FISTA in one pass: dense and stencil operators alike)
        assert ctx.calls["step"] > 0                               # the one-pass path was taken ...
        if fused == 3 and meta["kind"] != "tv" and c.backtracks:
            assert ctx.calls["fwd"] > 0 and ctx.calls["adj"] > 0   # ... and abandoned for the backtracking retries
        if fused is True:
            assert ctx.calls["fwd"] == 0 and ctx.calls["adj"] == 0 # ... for every launch of the loop
    else:
        assert ctx.calls["step"] == 0
    if not fused and meta["kind"] != "tv" and not accelerated:
        assert ctx.calls["pair"] > 0                               # short launches: K-fwd + K-adj under one synchronisation
        if c.backtracks:
            assert ctx.calls["fwd"] > 0 and ctx.calls["adj"] > 0   # retries and the cool-down iterations go the plain way
    else:
        assert ctx.calls["pair"] == 0


@pytest.mark.parametrize("fused", [False, True, 3])
@pytest.mark.parametrize("name", CASES)
def test_library_loop_plumbing_equals_the_python_driver(name, fused):
    """Round 6: by default the decisions between two launches are taken by the library's host-side loop (fh_iterate) in calls of several
    iterations; `FBSolver._library_call` slices its history records into the reference's arrays and carries solver state and launch policy
    from call to call.  With the NumPy twin of fh_iterate on the stand-in context (tests/fake_ctx.py) that plumbing runs on the CPU tier:
    every fixture, calls of 7 iterations and time-sized calls, against the Python driver -- EQUAL histories, counts, solutions and launch
    counters (the same launches produced the same numbers); options that need Python between iterations keep `step()`."""
    meta, z, py, ctx_py = _run(name, bool(fused), dense_kind=3 if fused == 3 else 1, driver="python")
    needs_python = bool(meta["options"].get("record_iterates") or meta["options"].get("func"))
    for kw in (dict(driver="library", device_iters=7), {}):
        _, _, lib, ctx = _run(name, bool(fused), dense_kind=3 if fused == 3 else 1, **kw)
        assert lib.library_steps == (0 if needs_python else lib.iteration_count) and py.library_steps == 0
        assert lib.iteration_count == py.iteration_count and lib.backtracks == py.backtracks
        for f in ("residuals", "norm_residuals", "stepsizes", "objectives", "function_hist", "iterates", "solution"):
            a, b = getattr(lib, f), getattr(py, f)
            assert (a is None) == (b is None), f
            if a is not None:
                assert np.array_equal(a, b, equal_nan=True), f
        assert ctx.calls == ctx_py.calls


@pytest.mark.parametrize("name", ["sparse_ls_64x128_accelerated", "sparse_ls_unnormalised_backtracks", "tv_32x32_adaptive"])
def test_verbose_lines_from_history_records_are_the_reference_text(name, capsys):
    """verbose=True (the reference's default): header, one line per iteration, "Restarted acceleration." -- printed per iteration by the Python
    driver and per call, from the history records, by the library loop: the same text."""
    _run(name, True, driver="python", verbose=True)
    want = capsys.readouterr().out
    _run(name, True, driver="library", device_iters=5, verbose=True)
    got = capsys.readouterr().out
    assert want.startswith("Initializing FASTA...") and want.count("\n[") > 5
    assert got == want


def test_speculation_backs_off_after_a_backtrack():
    meta, z, c, ctx = _run("sparse_ls_unnormalised_backtracks", True, dense_kind=3)
    # 105 backtracks in 200 iterations: most iterations must have used the two-launch path
    assert ctx.calls["step"] < c.iteration_count // 2


def test_operand_recognition_errors_without_a_gpu():
    A = FakeDenseMap(np.eye(4))
    ls, reg = fa.LeastSquares(np.zeros(4)), fa.Shrink(0.1)
    other = fa.Shrink(0.2)
    # backend="hip" insists on the device loop: operands it cannot take are a TypeError, not a detour over the host
    with pytest.raises(TypeError):
        fa.fasta(A, lambda z: 0.0, ls.gradf, reg.g, reg.prox, np.zeros(4), verbose=False, backend="hip")
    with pytest.raises(TypeError):
        fa.fasta(A, ls.f, ls.gradf, other.g, reg.prox, np.zeros(4), verbose=False, backend="hip")      # g and proxg from different tags
    with pytest.raises(TypeError):
        fa.fasta(A, ls.f, fa.LeastSquares(np.zeros(4)).gradf, reg.g, reg.prox, np.zeros(4), verbose=False, backend="hip")
    with pytest.raises(TypeError):
        fa.fasta(lambda x: x, lambda x: x, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(4), verbose=False, backend="hip")
    with pytest.raises(AssertionError):
        fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(5), verbose=False)
    with pytest.raises(TypeError):
        fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, verbose=False)                     # wrong arity
    with pytest.raises(ValueError):
        fa.fasta(FakeDenseMap(np.eye(4), fused_kind=0), ls.f, ls.gradf, reg.g, reg.prox, np.zeros(4), verbose=False, fused=True)


def test_linear_map_algebra_on_the_host():
    """LinearMap composition / scaling / sums (fasta/linalg.py:71-136) stay host-side callables."""
    rng = np.random.RandomState(0)
    M, N = rng.randn(4, 4), rng.randn(4, 4)
    A, B = FakeDenseMap(M), FakeDenseMap(N)
    x = rng.randn(4)
    np.testing.assert_allclose((A @ B)(x), M @ (N @ x))
    np.testing.assert_allclose((A @ B).H(x), N.T @ (M.T @ x))
    np.testing.assert_allclose((2.5 * A)(x), 2.5 * (M @ x))
    np.testing.assert_allclose((A - B)(x), (M - N) @ x)
    np.testing.assert_allclose((-A).H(x), -(M.T @ x))
    np.testing.assert_allclose((A ** 3)(x), M @ (M @ (M @ x)))
    assert A.is_operator and fa.LinearMap.identity((3,))(np.ones(3)).sum() == 3
    L = fa.LinearOperator(lambda v: 2 * v, lambda v: 2 * v, (5,))       # 3-argument legacy form
    assert L.Vshape == L.Wshape == (5,)
    with pytest.raises(AssertionError):
        A(np.zeros(5))


def test_one_pass_kernel_is_retried_with_backoff_after_a_timeout():
    """A hand-off timeout (a co-tenant on the GPU: the launch needs every CU at once) makes the driver fall back to K-fwd / K-adj --
    but not for the rest of the solve: it tries the one-pass kernel again after `_fused_backoff` iterations and doubles the wait
    each time the launch fails again.  The iterates do not depend on which kernels ran."""
    from fasta_python_amd import hip
    meta, z = H.load_case("sparse_ls_64x128_adaptive")
    d = H.case_data(meta, z)
    runs = {}
    for fail_at in ((), (2, 6)):                          # indices of the one-pass launches that time out
        op = FakeDenseMap(d["A"], fused_kind=1)
        real_step, launches = op.ctx.step, []

        def step(tau, real_step=real_step, launches=launches, fail_at=fail_at):
            launches.append(len(launches))
            out = real_step(tau)                          # (the collectives of a timed-out launch still complete)
            if launches[-1] in fail_at:
                raise hip.HipTimeout("fused one-pass kernel: team hand-off timed out (injected)")
            return out
        op.ctx.step = step
        ls, reg = fa.LeastSquares(d["b"]), fa.Shrink(float(d["mu"]))
        np.random.seed(meta["solver_seed"])
        solver = fa.FBSolver(op, ls, reg, np.zeros(d["A"].shape[1]), verbose=False, max_iters=40, tolerance=0.0).setup()
        solver._fused_backoff = 3
        disabled = []
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            while not solver.step():
                disabled.append(not solver.use_fused)
        runs[fail_at] = (solver.result(), disabled, len(launches), op.ctx.calls["fwd"])
    (ref, dis0, n0, fwd0), (got, dis1, n1, fwd1) = runs[()], runs[(2, 6)]
    assert not any(dis0) and fwd0 == 0
    # launch 2 = iteration 2 fails: iterations 2, 3, 4 run two launches, iteration 5 retries (launch 3), ... launch 6 fails -> 6 more
    assert dis1[2] and dis1[3] and dis1[4] and not dis1[5]
    assert sum(dis1) == 3 + 6 and fwd1 >= 9 and n1 < n0
    assert got.iteration_count == ref.iteration_count == 40 and got.backtracks == ref.backtracks
    np.testing.assert_allclose(got.residuals, ref.residuals, rtol=1e-12)
    np.testing.assert_allclose(got.solution, ref.solution, rtol=1e-12, atol=1e-15)


def test_only_a_typed_timeout_is_recovered_from():
    """VERDICT r3 / 'typed timeout': the driver falls back to K-fwd / K-adj for `hip.HipTimeout` alone (the launch succeeded, its hand-off
    spins ran out).  Any other non-zero status of the one-pass launch -- a device fault, an RCCL error on one rank -- is NOT a reason to
    change the launch (and, in a sharded run, the collective) sequence: it must reach the caller."""
    from fasta_python_amd import hip
    assert issubclass(hip.HipTimeout, hip.HipError)
    meta, z = H.load_case("sparse_ls_64x128_adaptive")
    d = H.case_data(meta, z)
    for exc_type, recovered in ((hip.HipTimeout, True), (hip.HipError, False)):
        for accelerate in (False, True):
            op = FakeDenseMap(d["A"], fused_kind=1)
            name = "step_accel" if accelerate else "step"
            real, count = getattr(op.ctx, name), [0]

            def launch(*args, real=real, count=count, exc_type=exc_type):
                count[0] += 1
                out = real(*args)
                if count[0] == 3:
                    raise exc_type("[700] injected: an illegal address was encountered" if exc_type is hip.HipError else "injected timeout")
                return out
            setattr(op.ctx, name, launch)
            ls, reg = fa.LeastSquares(d["b"]), fa.Shrink(float(d["mu"]))
            np.random.seed(meta["solver_seed"])
            solver = fa.FBSolver(op, ls, reg, np.zeros(d["A"].shape[1]), verbose=False, max_iters=8, tolerance=0.0,
                                 accelerate=accelerate).setup()
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                if recovered:
                    while not solver.step():
                        pass
                    assert solver.i == 8 and op.ctx.calls["fwd"] >= 1          # fell back, finished the solve
                else:
                    with pytest.raises(hip.HipError, match="illegal address") as info:
                        while not solver.step():
                            pass
                    assert not isinstance(info.value, hip.HipTimeout)
                    assert solver.use_fused and op.ctx.calls["fwd"] == 0       # no silent switch to the two-launch path
