"""GPU parity tests: prox operators (known answers from the reference), the l-inf / l1-ball level
search, and the TV stencil operator pair -- all through the C ABI."""
import os
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip, proximal
from oracle import fasta_np as fo
from oracle import problems as pr
from tests import gpu_util as G
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_prox_functions_match_reference_known_answers(golden_dir):
    k = np.load(os.path.join(golden_dir, "kat_prox.npz"))
    x, xr = k["x"], k["xr"]
    """The DEVICE prox kernels (through proximal.device_prox / tag.prox_on_device) against the reference's known answers."""
    dev = proximal.device_prox
    shrink1, linf1, l1ball = fa.Shrink(1.0), fa.LinfProx(1.0), (lambda t: fa.L1Ball(t))
    # elementwise prox: bit-exact (same IEEE operations in the same order)
    assert np.array_equal(dev(shrink1, x, 1.0), k["shrink_t1"])
    assert np.array_equal(np.signbit(dev(shrink1, x, 1.0)), np.signbit(k["shrink_t1"]))   # the -0.0 of P1
    assert np.array_equal(dev(shrink1, xr, 0.3), k["shrink_r"])
    assert np.array_equal(fa.NonNeg().prox_on_device(xr, 0.5), np.maximum(xr, 0))
    assert np.array_equal(fa.Box(-0.25, 0.5).prox_on_device(xr, 0.5), np.clip(xr, -0.25, 0.5))
    # level search replaces the sort: equal up to the rounding of the sums
    for t, key in ((1.0, "linf_t1"), (4.0, "linf_t4"), (10.5, "linf_t10p5"), (11.0, "linf_t11")):
        np.testing.assert_allclose(dev(linf1, x, t), k[key], rtol=0, atol=1e-14, err_msg=key)
    for t, key in ((4.0, "l1_t4"), (1.0, "l1_t1"), (10.5, "l1_t10p5")):
        np.testing.assert_allclose(dev(l1ball(t), x, 1.0), k[key], rtol=0, atol=1e-14, err_msg=key)
    np.testing.assert_allclose(dev(linf1, xr, 7.0), k["linf_r"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(dev(l1ball(7.0), xr, 1.0), k["l1_r"], rtol=0, atol=1e-14)
    # the scratch context is cached per shape: a second shape, then the first again, and the host forms agree with the device
    assert len(proximal._scratch) == 2
    assert np.array_equal(dev(shrink1, x, 1.0), proximal.shrink(x, 1.0))
    assert np.array_equal(dev(shrink1, xr, 0.3), proximal.shrink(xr, 0.3))
    assert len(proximal._scratch) == 2
    proximal.release_scratch()
    assert not proximal._scratch


@pytest.mark.parametrize("n", [1, 2, 63, 1024, 1025, 5000, 20000, 70001])
def test_level_search_equals_sort_based_level(n):
    rng = np.random.RandomState(n)
    x = rng.randn(n) * rng.choice([0.01, 1.0, 30.0], size=n)
    for t in (1e-3, 0.7 * np.abs(x).sum(), 0.999 * np.abs(x).sum(), 2.0 * np.abs(x).sum()):
        # the level is a quotient of n-term sums: allow n * ulp(level) of absolute slack on the outputs
        atol = 4e-16 * n * max(1.0, np.abs(x).max())
        np.testing.assert_allclose(proximal.device_prox(fa.LinfProx(1.0), x, t), fo.prox_linf(x, t), rtol=1e-12, atol=atol)
        np.testing.assert_allclose(proximal.device_prox(fa.L1Ball(t), x, 1.0), fo.project_l1(x, t), rtol=1e-12, atol=atol)


@pytest.mark.parametrize("n", [700, 6000, 16384, 16385, 40000, 65536, 100000, 262144])
def test_level_search_warm_started_from_any_previous_level(n):
    """round 5: the search starts from the level the previous launch left behind (one workgroup up to n = 16384, several beyond).  Whatever
    that guess is -- far below the new root, far above it, above EVERY entry, left over from a vector with ||x||_1 <= t -- the level
    must equal the sort-based one of the reference (fasta/proximal.py:22-26) and be the same as from a cold start."""
    rng = np.random.RandomState(n)
    base = rng.randn(n) * rng.choice([0.01, 1.0, 30.0], size=n)
    l1 = np.abs(base).sum()
    # (scale of the vector, radius): consecutive launches on ONE cached context, so each search is warm-started from the one before
    sequence = [(1.0, 0.5 * l1), (1.02, 0.5 * l1), (0.97, 0.52 * l1), (1e-3, 1e-4 * l1), (1.0, 1e-3), (50.0, 0.9 * 50 * l1), (1.0, 3.0 * l1), (1.0, 0.2 * l1),
                (1e-6, 1e-9 * l1)]
    for scale, t in sequence:
        x = base * scale
        atol = 4e-16 * n * max(1.0, np.abs(x).max())
        np.testing.assert_allclose(proximal.device_prox(fa.L1Ball(t), x, 1.0), fo.project_l1(x, t), rtol=1e-12, atol=atol)
    cold = {}
    for scale, t in sequence[:3]:                       # the same launches from a cold start (a fresh context each): bit-identical
        proximal.release_scratch()
        cold[(scale, t)] = proximal.device_prox(fa.L1Ball(t), base * scale, 1.0)
    proximal.release_scratch()
    for scale, t in sequence[:3]:
        assert np.array_equal(proximal.device_prox(fa.L1Ball(t), base * scale, 1.0), cold[(scale, t)])


TV_SHAPES = [(1, 1), (1, 5), (5, 1), (2, 2), (16, 128), (17, 129), (33, 300), (40, 257), (96, 96)]


@pytest.mark.parametrize("H_,W_", TV_SHAPES)
def test_stencil_pair_matches_numpy_rolls(H_, W_):
    rng = np.random.RandomState(H_ * 1000 + W_)
    Y, X = rng.randn(H_, W_, 2), rng.randn(H_, W_)
    op = fa.GradDivMap((H_, W_))
    try:
        assert np.array_equal(op(Y), pr.div(Y))               # same subtractions/additions, same order
        assert np.array_equal(op.H(X), pr.grad(X))
        assert abs(np.vdot(op(Y), X) - np.vdot(Y, op.H(X))) < 1e-10 * max(1.0, H_ * W_)
    finally:
        op.close()


def test_tv_ball_prox_is_bit_exact():
    rng = np.random.RandomState(4)
    Y = rng.randn(37, 45, 2) * 1.5
    assert np.array_equal(fa.TVDualBall().prox_on_device(Y, 0.3), fo.tv_dual_ball(Y))
    assert np.array_equal(fa.TVDualBall().prox(Y, 0.3), fo.tv_dual_ball(Y))        # the host form, same bits


@pytest.mark.parametrize("name", ["tv_32x32_accelerated", "tv_32x32_plain", "l1ball_64x128_adaptive",
                                  "l1ball_64x128_accelerated", "l1ball_64x128_plain",
                                  "logistic_100x160_adaptive", "logistic_100x160_accelerated", "logistic_100x160_plain"])
def test_golden_parity_full_solve(name):
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    c = G.run_hip(meta["kind"], data, meta["options"], meta["solver_seed"])
    assert c.iteration_count == int(z["iteration_count"])
    assert c.backtracks == int(z["backtracks"])
    G.compare_histories(c, lambda f: z[f] if f in z.files else None, c.iteration_count, rtol=1e-6, atol=1e-13)
    np.testing.assert_allclose(c.solution, z["solution"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name,k", [("tv_32x32_adaptive", 40), ("linf_96x96_adaptive", 40),
                                    ("linf_96x96_accelerated", 100), ("linf_96x96_plain", 100)])
def test_golden_parity_prefix(name, k):
    """Long backtracking-heavy runs (97-334 backtracks) are pinned on their first k iterations."""
    meta, z = H.load_case(name)
    data = H.case_data(meta, z)
    c = G.run_hip(meta["kind"], data, dict(meta["options"], max_iters=k, tolerance=0.0), meta["solver_seed"])
    np.testing.assert_allclose(c.stepsizes[:k], z["stepsizes"][:k], rtol=1e-6)
    np.testing.assert_allclose(c.residuals[:k], z["residuals"][:k], rtol=1e-6)
    np.testing.assert_allclose(c.objectives[:k + 1], z["objectives"][:k + 1], rtol=1e-7)


def test_tv_denoising_recovers_piecewise_constant_image():
    """End-to-end config-4 recipe at 256^2: primal image from the dual solution (tv_denoising.py:101)."""
    np.random.seed(9)
    P = pr.tv_denoising(H=256, W=256, square=32)
    M, mu = P.data["M"], P.data["mu"]
    op = fa.GradDivMap(M.shape)
    try:
        ls, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        opts = dict(max_iters=60, tolerance=1e-4, evaluate_objective=True)
        np.random.seed(2)
        got = fa.fasta(op, op.H, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", **opts)
        np.random.seed(2)
        want = fo.fasta(*P.args7(), **opts)
        assert got.iteration_count == want.iteration_count and got.backtracks == want.backtracks
        k = got.iteration_count
        G.compare_histories(got, lambda f: getattr(want, f), k, rtol=1e-6, atol=1e-13)
        np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-9)
        X = M - mu * op(got.solution)
        clean = pr.checkerboard(256, 256, 32)
        assert np.abs(X - clean).mean() < np.abs(M - clean).mean()      # denoised is closer to the clean image
    finally:
        op.close()


@pytest.mark.parametrize("image", [(8192, 8192)])
def test_c4_full_size_first_iterations_match_oracle_loop(image):
    """BASELINE config 4 at FULL size (8192^2): the first iterations of the step kernels -- the default one-pass sweep
    (k_fused_tv_step) and the two-launch pair (k_fwd_tv_step / k_adj_tv_step, fused=False), adaptive and accelerated --
    against the oracle's NumPy roll-based div/grad loop.  tau0 = 1 is far above 2/L = 1/4 (||div||^2 <= 8), so the first
    iteration BACKTRACKS (multi-chunk halo rows and the retry path at full size); with L and tau0 given no RNG is drawn."""
    Hh, Ww = image
    iters = 4
    np.random.seed(7)
    M = pr.checkerboard(Hh, Ww, Hh // 32)
    M += 0.1 * np.random.standard_normal(M.shape)
    mu = 0.1
    P = pr.tv_denoising_from(M, mu)
    base = dict(max_iters=iters, tolerance=0.0, evaluate_objective=True, L=8.0, tau0=1.0)
    want = {}
    for name, mode in (("adaptive", dict(adaptive=True, accelerate=False)), ("accelerated", dict(adaptive=False, accelerate=True))):
        want[name] = (mode, fo.fasta(*P.args7(), **base, **mode))
        assert want[name][1].backtracks >= 1
    op = fa.GradDivMap(M.shape)
    try:
        ls, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        for name, (mode, w) in want.items():
            for fused in ("auto", False):
                solver = fa.FBSolver(op, ls, reg, P.x0, verbose=False, fused=fused, **base, **mode)
                got = solver.setup().run()
                tag = f"{name} fused={fused}"
                if fused == "auto":
                    assert solver.fused_steps == iters + got.backtracks, tag      # every launch was the one-pass sweep
                else:
                    assert solver.fused_steps == 0, tag
                assert got.iteration_count == w.iteration_count == iters and got.backtracks == w.backtracks, tag
                for f in ("residuals", "norm_residuals", "stepsizes"):
                    np.testing.assert_allclose(getattr(got, f)[:iters], getattr(w, f)[:iters], rtol=1e-8, err_msg=tag + " " + f)
                np.testing.assert_allclose(got.objectives[:iters + 1], w.objectives[:iters + 1], rtol=1e-8, err_msg=tag)
                np.testing.assert_allclose(got.solution, w.solution, rtol=1e-5, atol=1e-12, err_msg=tag)
    finally:
        op.close()


def test_tv_full_size_adjointness_8192():
    """BASELINE config 4 size: <div Y, X> == <Y, grad X> on the 8192^2 stencil (size-independent property)."""
    Hh = Ww = 8192
    rng = np.random.RandomState(0)
    X = rng.standard_normal((Hh, Ww))
    Y = rng.standard_normal((Hh, Ww, 2))
    op = fa.GradDivMap((Hh, Ww))
    try:
        lhs = np.vdot(op(Y), X)
        rhs = np.vdot(Y, op.H(X))
        assert abs(lhs - rhs) <= 1e-9 * np.sqrt(X.size) * 10
        # periodic wrap spot check on the image border
        Z = op(Y)
        assert Z[-1, -1] == (Y[0, -1, 0] - Y[-1, -1, 0]) + (Y[-1, 0, 1] - Y[-1, -1, 1])
    finally:
        op.close()


ONE_PASS_SHAPES = [(1, 1), (2, 3), (5, 64), (33, 61), (40, 257), (97, 130), (3, 59), (4, 60), (130, 121), (70, 500)]
ONE_PASS_ACCEL_SHAPES = [(1, 1), (2, 3), (5, 64), (33, 61), (40, 257), (97, 130), (64, 1000), (130, 121)]


@pytest.mark.parametrize("H_,W_", ONE_PASS_SHAPES)
def test_one_pass_tv_step_equals_two_launch_step(H_, W_):
    one_pass_tv_step_equals_two_launch_step(H_, W_, None)


def one_pass_tv_step_equals_two_launch_step(H_, W_, zfree):
    """fh_step on the stencil operator against fh_fwd + fh_adj on the same state.  zfree None / 1: the shipped kernel, which
    recomputes z in flight (k_tv_onepass); 0: the round-1 kernel that streams it (k_fused_tv_step; only in the experimental
    library: tests/test_gpu_experimental.py)."""
    rng = np.random.RandomState(H_ * 7 + W_)
    M = rng.randn(H_, W_)
    Y0 = rng.randn(H_, W_, 2) * 0.8
    tau = 0.11
    op = fa.GradDivMap((H_, W_))
    try:
        c = op.ctx
        if zfree is not None:
            c.set_tuning(hip.TUNE_TV_ZFREE, zfree)
        assert c.fused_supported() == 2

        def fresh():
            c.set_loss_lsq(M)
            c.set_prox(hip.PROX_TVBALL)
            c.set_vector(hip.VEC_X0, Y0)
            c.init()
        fresh()
        s = c.fwd(tau)
        a = c.adj(tau)
        xp_ref = c.get_vector(hip.VEC_XPROX, Y0.size)
        z_ref = c.get_vector(hip.VEC_Z, M.size)
        fresh()
        f = c.step(tau)
        assert np.array_equal(c.get_vector(hip.VEC_XPROX, Y0.size), xp_ref)
        assert np.array_equal(c.get_vector(hip.VEC_Z, M.size), z_ref)
        for k in (hip.S_FSQ, hip.S_DXG0, hip.S_DX2, hip.S_XH2, hip.S_G02):
            np.testing.assert_allclose(f[k], s[k], rtol=1e-12, atol=1e-300, err_msg=f"fwd scalar {k}")
        for k in (hip.S_DXDG, hip.S_DG2):
            np.testing.assert_allclose(f[k], a[k], rtol=1e-11, atol=1e-300, err_msg=f"adj scalar {k}")
    finally:
        op.close()


def test_tv_solve_identical_with_and_without_the_one_pass_kernel():
    np.random.seed(9)
    P = pr.tv_denoising(H=96, W=130, square=16)
    M, mu = P.data["M"], P.data["mu"]
    op = fa.GradDivMap(M.shape)
    try:
        ls, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        opts = dict(max_iters=80, tolerance=1e-5, evaluate_objective=True)
        out = []
        for fused in (True, False):
            np.random.seed(2)
            out.append(fa.fasta(op, op.H, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", fused=fused, **opts))
    finally:
        op.close()
    a, b = out
    assert a.iteration_count == b.iteration_count and a.backtracks == b.backtracks and a.backtracks > 0
    k = a.iteration_count
    np.testing.assert_allclose(a.residuals[:k], b.residuals[:k], rtol=1e-7)
    np.testing.assert_allclose(a.objectives[:k + 1], b.objectives[:k + 1], rtol=1e-9)
    np.testing.assert_allclose(a.solution, b.solution, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("prox", ["tvball", "identity"])
@pytest.mark.parametrize("restart", [True, False])
@pytest.mark.parametrize("H_,W_", ONE_PASS_ACCEL_SHAPES)
def test_one_pass_accelerated_tv_steps_equal_two_launch_steps(H_, W_, restart, prox):
    one_pass_accelerated_tv_steps_equal_two_launch_steps(H_, W_, restart, prox, None)


def one_pass_accelerated_tv_steps_equal_two_launch_steps(H_, W_, restart, prox, zfree):
    """fh_step_accel on the stencil operator (k_fused_tv_accel: both coefficient candidates in one sweep, iterate kept in
    extrapolated-on-the-fly form) in LOCKSTEP with fh_fwd + fh_adj(accel) on a second context: same alpha recursion, same
    backtracking-style retry (a launch repeated with a smaller tau before the commit), several restarts along the way.
    Materialised vectors must be bit-identical (every value is produced by the same IEEE expressions); scalars are sums."""
    rng = np.random.RandomState(H_ * 11 + W_)
    M = rng.randn(H_, W_)
    Y0 = rng.randn(H_, W_, 2) * 0.8
    one, two = fa.GradDivMap((H_, W_)), fa.GradDivMap((H_, W_))
    try:
        if zfree is not None:
            one.ctx.set_tuning(hip.TUNE_TV_ZFREE, zfree)
        for c in (one.ctx, two.ctx):
            c.set_loss_lsq(M)
            c.set_prox(hip.PROX_TVBALL if prox == "tvball" else hip.PROX_IDENTITY)
            c.set_vector(hip.VEC_X0, Y0)
            c.init()
        alpha, restarts, best_it = 1.0, 0, None
        for it in range(25):
            # With the unit-ball prox the restart rule (:231) never fires on this problem (nor in the reference's own TV runs).
            # With the identity prox it can be provoked: iterations 10-13 step UPHILL (a negative tau is plain arithmetic to
            # the kernels) against the downhill momentum, then small downhill steps run against the uphill momentum.
            tau = -0.05 if 10 <= it <= 13 else (0.01 if it in (14, 15, 16) else 0.1)
            if prox == "tvball" and tau > 0:
                tau *= 2.4
            for attempt in range(2 if it in (1, 4, 14) else 1):        # iterations 1, 4, 14: retry with a smaller step, then commit
                if attempt:
                    tau *= 0.5
                a1 = (1 + np.sqrt(1 + 4 * alpha ** 2)) / 2
                s1 = one.ctx.step_accel(tau, (alpha - 1) / a1, restart)
                f2 = two.ctx.fwd(tau)
                alpha0 = 1.0 if (restart and f2[hip.S_RDOT] > 1e-30) else alpha
                coef = (alpha0 - 1) / ((1 + np.sqrt(1 + 4 * alpha0 ** 2)) / 2)
                s2 = two.ctx.adj(tau, True, coef)
                np.testing.assert_allclose(s1[hip.S_RDOT], f2[hip.S_RDOT], rtol=1e-9, atol=1e-13 * Y0.size)
                assert (s1[hip.S_RDOT] > 1e-30) == (f2[hip.S_RDOT] > 1e-30)
                for k in (hip.S_FSQ, hip.S_DXG0, hip.S_DX2, hip.S_XH2, hip.S_G02):
                    np.testing.assert_allclose(s1[k], f2[k], rtol=1e-11, atol=1e-300, err_msg=f"it {it} fwd scalar {k}")
                for k in (hip.S_DXDG, hip.S_DG2, hip.S_FSQ_ADJ, hip.S_XH2_ADJ, hip.S_GSUM_ADJ):
                    np.testing.assert_allclose(s1[k], s2[k], rtol=1e-10, atol=1e-300, err_msg=f"it {it} adj scalar {k}")
                assert s1[hip.S_GMAX_ADJ] == s2[hip.S_GMAX_ADJ]
                assert np.array_equal(one.ctx.get_vector(hip.VEC_XPROX, Y0.size), two.ctx.get_vector(hip.VEC_XPROX, Y0.size))
                assert np.array_equal(one.ctx.get_vector(hip.VEC_Z, M.size), two.ctx.get_vector(hip.VEC_Z, M.size))
            restarts += int(restart and f2[hip.S_RDOT] > 1e-30)
            alpha = (1 + np.sqrt(1 + 4 * alpha0 ** 2)) / 2
            save = it in (0, 2, 5)
            one.ctx.commit(save_best=save)
            two.ctx.commit(save_best=save)
            # the extrapolated iterate is never stored by the one-pass kernel: materialised on demand, same bits
            assert np.array_equal(one.ctx.get_vector(hip.VEC_X0, Y0.size), two.ctx.get_vector(hip.VEC_X0, Y0.size)), it
            assert np.array_equal(one.ctx.get_vector(hip.VEC_BEST, Y0.size), two.ctx.get_vector(hip.VEC_BEST, Y0.size)), it
        if restart and prox == "identity" and H_ * W_ >= 1000:
            assert 0 < restarts < 25                     # both branches of the restart rule were exercised
        with pytest.raises(hip.HipError):
            one.ctx.fwd(0.1)                             # no silent mixing of the two state representations
        one.ctx.set_vector(hip.VEC_X0, Y0)
        one.ctx.init()
        one.ctx.fwd(0.1)                                 # ... a fresh fh_init lifts it
    finally:
        one.close()
        two.close()


@pytest.mark.parametrize("restart", [True, False])
def test_accelerated_tv_solve_identical_with_and_without_the_one_pass_kernel(restart):
    np.random.seed(9)
    P = pr.tv_denoising(H=96, W=130, square=16)
    M, mu = P.data["M"], P.data["mu"]
    op = fa.GradDivMap(M.shape)
    try:
        ls, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        opts = dict(max_iters=120, tolerance=1e-5, evaluate_objective=True, adaptive=False, accelerate=True, restart=restart,
                    record_iterates=True)
        out, used = [], []
        for fused in (True, False):
            solver = fa.FBSolver(op, ls, reg, P.x0, verbose=False, fused=fused, **opts)
            np.random.seed(2)
            out.append(solver.setup().run())
            used.append(solver.fused_steps)
        np.random.seed(2)
        want = fo.fasta(*P.args7(), **opts)
    finally:
        op.close()
    a, b = out
    assert used[0] == a.iteration_count + a.backtracks and used[1] == 0
    assert a.iteration_count == b.iteration_count == want.iteration_count and a.backtracks == b.backtracks == want.backtracks
    k = a.iteration_count
    np.testing.assert_allclose(a.residuals[:k], b.residuals[:k], rtol=1e-7)
    np.testing.assert_allclose(a.objectives[:k + 1], b.objectives[:k + 1], rtol=1e-9)
    assert np.array_equal(a.iterates[:k + 1], b.iterates[:k + 1]) or np.allclose(a.iterates[:k + 1], b.iterates[:k + 1], rtol=1e-6, atol=1e-10)
    G.compare_histories(a, lambda f: getattr(want, f), k, rtol=1e-6, atol=1e-13)
    np.testing.assert_allclose(a.iterates[:k + 1], want.iterates[:k + 1], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(a.solution, want.solution, rtol=1e-5, atol=1e-9)


def test_z_free_steps_can_be_followed_by_two_launch_steps():
    """The default one-pass stencil step never stores z; when a caller of the C ABI then switches to fh_fwd / fh_adj, the
    stored image of x0 is brought up to date first (one plain div pass) -- same results as two-launch steps throughout."""
    rng = np.random.RandomState(5)
    H_, W_ = 70, 150
    M = rng.randn(H_, W_)
    Y0 = rng.randn(H_, W_, 2) * 0.8
    a, b = fa.GradDivMap((H_, W_)), fa.GradDivMap((H_, W_))
    try:
        for c in (a.ctx, b.ctx):
            c.set_loss_lsq(M)
            c.set_prox(hip.PROX_TVBALL)
            c.set_vector(hip.VEC_X0, Y0)
            c.init()
        for it, tau in enumerate((0.2, 0.15, 0.22, 0.1, 0.2, 0.18)):
            zfree_step = it in (0, 1, 3)
            if zfree_step:
                sa = a.ctx.step(tau)
            else:
                sa = a.ctx.fwd(tau)
                sa2 = a.ctx.adj(tau)
            sb = b.ctx.fwd(tau)
            sb2 = b.ctx.adj(tau)
            np.testing.assert_allclose(sa[hip.S_FSQ], sb[hip.S_FSQ], rtol=1e-12)
            np.testing.assert_allclose((sa if zfree_step else sa2)[hip.S_DXDG], sb2[hip.S_DXDG], rtol=1e-10, atol=1e-300)
            assert np.array_equal(a.ctx.get_vector(hip.VEC_XPROX, Y0.size), b.ctx.get_vector(hip.VEC_XPROX, Y0.size)), it
            assert np.array_equal(a.ctx.get_vector(hip.VEC_Z, M.size), b.ctx.get_vector(hip.VEC_Z, M.size)), it
            a.ctx.commit(save_best=(it == 2))
            b.ctx.commit(save_best=(it == 2))
        assert np.array_equal(a.ctx.get_vector(hip.VEC_BEST, Y0.size), b.ctx.get_vector(hip.VEC_BEST, Y0.size))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("H_,W_,seed", [(32, 32, 21), (64, 80, 22)])
def test_tv_adaptive_runs_to_convergence_and_the_first_divergence_from_the_oracle_is_late(H_, W_, seed, capsys):
    """Adaptive FBS on the TV dual backtracks every few iterations and is sensitive to the last digit of its sums: the oracle run
    against itself with merely a permuted summation order parts ways after ~115 iterations (tests/test_tv_divergence_cpu.py).  So HIP
    and oracle are run to convergence and the test MEASURES where their step sizes first differ by more than 1e-6 relative: that
    must not be early (>= 40 iterations of identical decisions, the prefix the fixtures pin), and both runs must end at the same
    minimum (objective within 1e-3 relative, the same denoised image)."""
    from tests.helpers import first_divergence
    np.random.seed(seed)
    P = pr.tv_denoising(H=H_, W=W_, square=8)
    M, mu = P.data["M"], P.data["mu"]
    opts = dict(tolerance=1e-8, max_iters=3000, evaluate_objective=True, L=8.0, tau0=0.025)      # ||div||^2 <= 8: no random probes
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    op = fa.GradDivMap((H_, W_))
    try:
        ls, reg = fa.LeastSquares(M / mu), fa.TVDualBall()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", **opts)
    finally:
        op.close()
    k = min(got.iteration_count, want.iteration_count)
    first = first_divergence(got.stepsizes, want.stepsizes, k)
    with capsys.disabled():
        print(f"\nTV {H_}x{W_} adaptive: HIP {got.iteration_count} iterations / {got.backtracks} backtracks, oracle {want.iteration_count} / "
              f"{want.backtracks}; step sizes first differ (> 1e-6 relative) at iteration {first}")
    assert want.backtracks > 20
    assert first >= 40
    # identical decisions up to there: same backtracking pattern, histories to the usual tolerance
    np.testing.assert_allclose(got.residuals[:first], want.residuals[:first], rtol=1e-5)
    np.testing.assert_allclose(got.objectives[:first + 1], want.objectives[:first + 1], rtol=1e-8)
    fg, fw = got.objectives[got.iteration_count], want.objectives[want.iteration_count]
    assert abs(fg - fw) <= 1e-3 * abs(fw)
    np.testing.assert_allclose(pr.tv_primal(M, mu, got.solution), pr.tv_primal(M, mu, want.solution), atol=2e-2)


def test_experimental_stencil_keys_are_refused_by_the_shipped_library():
    """round 5: the forms of the stencil sweep that measured flat twice (LDS-DMA ring, persistent chunk walk, occupancy limiter) and the
    superseded z-streaming one-pass kernels are compiled only into libfasta_hip_experimental.so; the shipped library names the reason."""
    if "experimental" in os.path.basename(hip.LIB_PATH):
        return                                          # (a run of this file against the experimental library: nothing to refuse)
    op = fa.GradDivMap((8, 8))
    try:
        for key in hip.EXPERIMENTAL_KEYS:
            with pytest.raises(hip.HipError, match="experimental"):
                op.ctx.set_tuning(key, 1)
    finally:
        op.close()
