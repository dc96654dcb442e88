"""fh_setup: the solver's whole set-up -- the two Lipschitz probes (fasta/__init__.py:100-113) and the initial pass (:135-137) -- from ONE
read of a dense A (csrc/fh_setup.h: two dot products and two rank-1 updates per row buffer: x0, and the probes' difference, because
grad(x1) - grad(x2) = A^T A (x1 - x2) for least squares), against the three separate passes fh_gradient_at x 2 + fh_init it replaces.
The x0 column is summed in the order the one-pass kernel sums it, so z, f and g0 must come out BIT-identical; the norm of the
gradient difference is formed another way (no cancellation between independent probes: rtol 1e-12)."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from oracle import fasta_np as fo
from oracle import problems as pr

pytestmark = pytest.mark.gpu


def _three_pass(c, n, m):
    c.timing_reset(); c.timing_enable(True)
    c.gradient_at(hip.VEC_T0, hip.VEC_T2)
    c.gradient_at(hip.VEC_T1, hip.VEC_T3)
    dg, dx = c.diff_norm(hip.VEC_T2, hip.VEC_T3), c.diff_norm(hip.VEC_T0, hip.VEC_T1)
    s = c.init()
    c.timing_enable(False)
    return dict(s=s, dg=dg, dx=dx, t2=c.get_vector(hip.VEC_T2, n), t3=c.get_vector(hip.VEC_T3, n), g0=c.get_vector(hip.VEC_G0, n),
                z=c.get_vector(hip.VEC_Z, m), one_pass_launches=c.timing_get(hip.K_FUSED)[1])


def _one_call(c, n, m):
    c.timing_reset(); c.timing_enable(True)
    s = c.setup()
    c.timing_enable(False)
    return dict(s=s, dg=np.sqrt(s[hip.S_DG2]), dx=np.sqrt(s[hip.S_DX2]), t2=c.get_vector(hip.VEC_T2, n), t3=c.get_vector(hip.VEC_T3, n),
                g0=c.get_vector(hip.VEC_G0, n), z=c.get_vector(hip.VEC_Z, m), one_pass_launches=c.timing_get(hip.K_FUSED)[1])


def _load(c, rng, m, n, loss):
    p1, p2, x0 = rng.randn(n), rng.randn(n), rng.randn(n) * 0.05
    if loss == "logistic":
        c.set_loss_logistic(np.sign(rng.randn(m)))
    else:
        c.set_loss_lsq(rng.randn(m))
    c.set_prox(hip.PROX_SHRINK, 0.02)
    for which, v in ((hip.VEC_T0, p1), (hip.VEC_T1, p2), (hip.VEC_X0, x0)):
        c.set_vector(which, v)
    return p1, p2, x0


# teams of 1, 2, 4, 8, 16 members (n <= 4096, 8192, 16384, 32768, 65536), full and ragged widths, few and many rows per team
SHAPES = [(2100, 4096), (4200, 2000), (9000, 1000), (1100, 8192), (1500, 6000), (600, 16384), (700, 12000), (300, 32768), (330, 20000),
          (150, 65536), (260, 50000), (37, 65536), (1, 40000),
          # at least 128 rows per team: rows dealt cyclically (round 6), in the set-up kernel as in the step kernel
          (40000, 4096), (20000, 8192), (12000, 16384), (3000, 65536)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_one_read_of_A_serves_the_two_probes_and_the_initial_pass(m, n, loss="lsq"):
    rng = np.random.RandomState(m * 3 + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx
        assert c.fused_supported() == 1
        p1, p2, x0 = _load(c, rng, m, n, loss)
        want = _three_pass(c, n, m)
        for which, v in ((hip.VEC_T2, np.zeros(n)), (hip.VEC_T3, np.zeros(n))):          # nothing may survive from the first run
            c.set_vector(which, v)
        c.set_vector(hip.VEC_X0, x0)
        got = _one_call(c, n, m)
        assert want["one_pass_launches"] == 3 and got["one_pass_launches"] == 1           # three reads of A -> one
        # 256-thread shapes sum every right-hand side exactly as the one-pass step kernel does: bit-identical.  The two 512-thread shapes (full 8-piece
        # widths of 8 / 16 members: n in (28672, 32768] and (57344, 65536]) hold other pieces per lane: equal to summation-order rounding.
        wide = 28672 < n <= 32768 or 57344 < n <= 65536
        for key in ("g0", "z"):
            if wide:
                np.testing.assert_allclose(got[key], want[key], rtol=1e-12, atol=1e-13 * np.abs(want[key]).max(), err_msg=key)
            else:
                assert np.array_equal(got[key], want[key]), key
        for k in (hip.S_GSUM, hip.S_GMAX, hip.S_FSQ):
            if wide:
                np.testing.assert_allclose(got["s"][k], want["s"][k], rtol=1e-12)
            else:
                assert got["s"][k] == want["s"][k]
        b = c.get_vector(hip.VEC_B, m)
        np.testing.assert_allclose(got["s"][hip.S_FSQ], np.sum((A @ x0 - b) ** 2), rtol=1e-11)
        np.testing.assert_allclose(got["t2"], A.T @ (A @ (p1 - p2)), rtol=1e-9, atol=1e-12 * np.abs(want["t2"]).max())     # A^T A (x1 - x2)
        np.testing.assert_allclose([got["dg"], got["dx"]], [want["dg"], want["dx"]], rtol=1e-12)
        np.testing.assert_allclose(got["dg"], np.linalg.norm(A.T @ (A @ p1 - b) - A.T @ (A @ p2 - b)), rtol=1e-11)         # the reference's expression (:106-110)
        np.testing.assert_allclose(got["dx"], np.linalg.norm(p1 - p2), rtol=1e-12)
        # and the state fh_init leaves: the first iteration after either set-up is the same launch
        s1 = c.step(0.3)
        c.set_vector(hip.VEC_X0, x0)
        c.init()
        if wide:
            np.testing.assert_allclose(c.step(0.3)[:14], s1[:14], rtol=1e-10, atol=1e-14)
        else:
            assert np.array_equal(c.step(0.3), s1)
    finally:
        op.close()


@pytest.mark.parametrize("m,n", [(300, 16384), (2100, 8192), (4300, 2000), (9000, 1000), (600, 30000), (150, 65536), (37, 65536), (260, 50000)])
def test_float32_storage_takes_one_read_of_A_for_the_set_up(m, n):
    """Round 6: k_setup_dense<..., F32 = 1> -- four-column pieces, at most four per lane (two right-hand sides cost 16 accumulator registers per
    piece), teams of 1-16 members, n <= 65536: ONE read of the float32 matrix instead of three.  Against the three passes on the same (rounded)
    matrix: rtol 1e-11 (other team shapes than the step kernel's, so not bit for bit), and against NumPy on A.astype(float32)."""
    rng = np.random.RandomState(m * 3 + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    A32 = A.astype(np.float32).astype(np.float64)
    op = fa.DenseMatrixMap(A, storage="f32")
    try:
        c = op.ctx
        assert c.fused_supported() == 1
        p1, p2, x0 = _load(c, rng, m, n, "lsq")
        want = _three_pass(c, n, m)
        for which, v in ((hip.VEC_T2, np.zeros(n)), (hip.VEC_T3, np.zeros(n))):
            c.set_vector(which, v)
        c.set_vector(hip.VEC_X0, x0)
        got = _one_call(c, n, m)
        assert want["one_pass_launches"] == 3 and got["one_pass_launches"] == 1
        for key in ("g0", "z"):
            np.testing.assert_allclose(got[key], want[key], rtol=1e-11, atol=1e-13 * np.abs(want[key]).max(), err_msg=key)
        np.testing.assert_allclose([got["s"][k] for k in (hip.S_GSUM, hip.S_GMAX, hip.S_FSQ)], [want["s"][k] for k in (hip.S_GSUM, hip.S_GMAX, hip.S_FSQ)], rtol=1e-12)
        b = c.get_vector(hip.VEC_B, m)
        np.testing.assert_allclose(got["s"][hip.S_FSQ], np.sum((A32 @ x0 - b) ** 2), rtol=1e-11)
        np.testing.assert_allclose(got["t2"], A32.T @ (A32 @ (p1 - p2)), rtol=1e-9, atol=1e-12 * np.abs(want["t2"]).max())
        np.testing.assert_allclose(got["g0"], A32.T @ (A32 @ x0 - b), rtol=1e-9, atol=1e-12 * np.abs(want["g0"]).max())
        np.testing.assert_allclose([got["dg"], got["dx"]], [want["dg"], want["dx"]], rtol=1e-12)
        s1 = c.step(0.3)
        c.set_vector(hip.VEC_X0, x0)
        c.init()
        np.testing.assert_allclose(c.step(0.3)[:14], s1[:14], rtol=1e-9, atol=1e-13)
    finally:
        op.close()


@pytest.mark.parametrize("m,n,storage,loss", [(40, 100000, "f64", "lsq"), (30, 200000, "f64", "lsq"), (64, 100000, "f32", "lsq"), (96, 160, "f64", "lsq"),
                                               (64, 4096, "f64", "lsq"), (600, 16384, "f64", "logistic"), (2100, 4096, "f64", "logistic")])
def test_shapes_without_a_one_read_kernel_take_the_three_passes_inside_the_call(m, n, storage, loss):
    """rows wider than 65536 columns (either storage), the logistic loss (its gradient is not linear in x: no difference trick), matrices
    too small for the one-pass kernel to pay: fh_setup is the three passes, with their results bit for bit."""
    rng = np.random.RandomState(n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap(A, storage=storage)
    try:
        c = op.ctx
        _load(c, rng, m, n, loss)
        want = _three_pass(c, n, m)
        got = _one_call(c, n, m)
        assert got["one_pass_launches"] == want["one_pass_launches"]
        for key in ("t2", "t3", "g0", "z"):
            assert np.array_equal(got[key], want[key]), key
        for k in (hip.S_FSQ, hip.S_GSUM, hip.S_GMAX):
            assert got["s"][k] == want["s"][k]
        np.testing.assert_allclose([got["dg"], got["dx"]], [want["dg"], want["dx"]], rtol=1e-14)
    finally:
        op.close()


def test_a_timed_out_set_up_launch_falls_back_to_the_three_passes():
    """fault injection (a withheld team partial): the launch ends with its timeout word set, fh_setup redoes the set-up pass by pass --
    which time out as well and end in K-fwd / K-adj -- and still delivers the right vectors."""
    import time
    rng = np.random.RandomState(5)
    m, n = 1100, 8192
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap(A)
    try:
        c = op.ctx
        p1, p2, x0 = _load(c, rng, m, n, "lsq")
        b = c.get_vector(hip.VEC_B, m)
        c.set_tuning(hip.TUNE_TEST_HOOKS, hip.HOOK_WITHHOLD_PARTIAL)
        t0 = time.time()
        s = c.setup()
        assert time.time() - t0 < 10.0
        c.set_tuning(hip.TUNE_TEST_HOOKS, 0)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G0, n), A.T @ (A @ x0 - b), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(s[hip.S_DG2], np.sum((A.T @ (A @ (p1 - p2))) ** 2), rtol=1e-9)
        got = c.step(0.3)                                                 # the hand-off slots are usable again
        assert np.isfinite(got).all() and got[15] == 0.0
    finally:
        op.close()


@pytest.mark.parametrize("mode", ["adaptive", "accelerated"])
def test_full_solve_through_the_one_read_set_up_matches_the_oracle(mode):
    np.random.seed(2)
    P = pr.sparse_least_squares(M=2100, N=4096, K=40)
    opts = dict(tolerance=1e-6, max_iters=60, evaluate_objective=True, adaptive=(mode == "adaptive"), accelerate=(mode != "adaptive"))
    np.random.seed(4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    ls, reg = fa.LeastSquares(P.data["b"]), fa.Shrink(P.data["mu"])
    op = fa.DenseMatrixMap(P.data["A"])
    try:
        op.ctx.timing_enable(True)
        np.random.seed(4)
        got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", **opts)
        launches = op.ctx.timing_get(hip.K_FUSED)[1]
    finally:
        op.close()
    k = want.iteration_count
    assert got.iteration_count == k and got.backtracks == want.backtracks
    assert launches == 1 + k + got.backtracks                            # ONE launch for the whole set-up, one per iteration / retry
    np.testing.assert_allclose(got.residuals[:k], want.residuals[:k], rtol=1e-6)
    np.testing.assert_allclose(got.stepsizes[:k], want.stepsizes[:k], rtol=1e-8)
    np.testing.assert_allclose(got.objectives[:k + 1], want.objectives[:k + 1], rtol=1e-8)
    np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("blocks", [2, 4, 8])
@pytest.mark.parametrize("rows_per_block,n", [(2100, 4096), (1100, 8192), (80, 16384), (40, 40000), (30, 65536)])
def test_row_blocks_take_one_read_of_A_each_for_the_set_up(rows_per_block, n, blocks):
    """Round 6: least squares is linear in the rows, so a multi-device context (here: all blocks on the one GPU) launches the two-right-hand-side
    kernel once per row block and sums A_k^T A_k (x1 - x2), the gradients and the loss sums in ONE exchange -- instead of K-fwd + K-adj three
    times per block (six reads of A).  g0 / z / f agree with that to summation-order rounding, L to rtol 1e-12, and against NumPy; the first step
    after the one-read set-up equals the first step after fh_init."""
    m = rows_per_block * blocks
    rng = np.random.RandomState(m + n + blocks)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    op = fa.ShardedDenseMatrixMap(A, devices=[0] * blocks)
    try:
        c = op.ctx
        assert c.fused_supported() == 1
        p1, p2, x0 = _load(c, rng, m, n, "lsq")
        want = _three_pass(c, n, m)
        passes_before = (c.timing_get(hip.K_FWD)[1], c.timing_get(hip.K_ADJ)[1])      # row blocks took K-fwd + K-adj for each of the three: SIX reads of A per block
        for which, v in ((hip.VEC_T2, np.zeros(n)), (hip.VEC_T3, np.zeros(n))):
            c.set_vector(which, v)
        c.set_vector(hip.VEC_X0, x0)
        got = _one_call(c, n, m)
        assert want["one_pass_launches"] == 0 and passes_before == (3 * blocks, 3 * blocks)
        assert got["one_pass_launches"] == blocks and (c.timing_get(hip.K_FWD)[1], c.timing_get(hip.K_ADJ)[1]) == (0, 0)      # ... now ONE
        wide = 28672 < n <= 32768 or 57344 < n <= 65536
        for key in ("g0", "z"):                                     # (K-fwd / K-adj sum in another order than the one-read kernel)
            np.testing.assert_allclose(got[key], want[key], rtol=1e-11, atol=1e-13 * np.abs(want[key]).max(), err_msg=key)
        np.testing.assert_allclose([got["s"][k] for k in (hip.S_GSUM, hip.S_GMAX, hip.S_FSQ)], [want["s"][k] for k in (hip.S_GSUM, hip.S_GMAX, hip.S_FSQ)], rtol=1e-12)
        b = c.get_vector(hip.VEC_B, m)
        np.testing.assert_allclose(got["s"][hip.S_FSQ], np.sum((A @ x0 - b) ** 2), rtol=1e-11)
        np.testing.assert_allclose(got["t2"], A.T @ (A @ (p1 - p2)), rtol=1e-9, atol=1e-12 * np.abs(want["t2"]).max())
        np.testing.assert_allclose([got["dg"], got["dx"]], [want["dg"], want["dx"]], rtol=1e-12)
        np.testing.assert_allclose(got["g0"], A.T @ (A @ x0 - b), rtol=1e-9, atol=1e-12 * np.abs(want["g0"]).max())
        for k in range(blocks):                                    # replicated vectors: the same on every block
            shard, _, _ = c.shard(k)
            assert np.array_equal(shard.get_vector(hip.VEC_G0, n), got["g0"])
        s1 = c.step(0.3)
        c.set_vector(hip.VEC_X0, x0)
        c.init()
        np.testing.assert_allclose(c.step(0.3)[:14], s1[:14], rtol=1e-9, atol=1e-13)
    finally:
        op.close()


def test_full_solve_on_row_blocks_through_the_one_read_set_up_matches_the_oracle():
    np.random.seed(2)
    P = pr.sparse_least_squares(M=8400, N=4096, K=40)
    opts = dict(tolerance=1e-6, max_iters=60, evaluate_objective=True)
    np.random.seed(4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(*P.args7(), **opts)
    ls, reg = fa.LeastSquares(P.data["b"]), fa.Shrink(P.data["mu"])
    op = fa.ShardedDenseMatrixMap(P.data["A"], devices=[0, 0, 0, 0])
    try:
        op.ctx.timing_enable(True)
        np.random.seed(4)
        got = fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", **opts)
        launches = op.ctx.timing_get(hip.K_FUSED)[1]
    finally:
        op.close()
    k = want.iteration_count
    assert got.iteration_count == k and got.backtracks == want.backtracks
    assert launches == 4 * (1 + k + got.backtracks)                      # per block: ONE launch for the whole set-up, one per iteration / retry
    np.testing.assert_allclose(got.residuals[:k], want.residuals[:k], rtol=1e-6)
    np.testing.assert_allclose(got.objectives[:k + 1], want.objectives[:k + 1], rtol=1e-8)
    np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-9)
