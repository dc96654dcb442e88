"""Opt-in float32-STORAGE mode of the dense operator (`DenseMatrixMap(A, storage="f32")`, C ABI fh_create_ex with
FH_DTYPE_F32_STORAGE): the device copy of A is float32 (half the bytes per pass), every vector, accumulation and scalar
stays float64.  What is checked:
  * the operator IS the rounded matrix: A.astype(float32) exactly (upload, read-back, synthetic generator);
  * parity: the solve equals the oracle's solve ON THE ROUNDED MATRIX iterate for iterate, to the same tolerances as the
    float64 path (histories rtol 1e-6, iterates rtol 1e-5) -- the mode adds no arithmetic error of its own;
  * the stated tolerance against the float64-matrix run (<= 3e-7 on the iterates away from the
    chaotic regime) on a reference-captured fixture;
  * one-pass kernel == two-launch kernels, bitwise repeatability, every team size of the float32 shapes."""
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


def rounded(A):
    return np.asarray(A, dtype=np.float64).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("m,n", [(1, 1), (3, 5), (17, 33), (64, 128), (100, 1000), (37, 4100), (300, 5121), (50, 20000), (20, 70001)])
def test_operator_is_the_rounded_matrix(m, n):
    rng = np.random.RandomState(m * 31 + n)
    A = rng.randn(m, n)
    A32 = rounded(A)
    x, y = rng.randn(n), rng.randn(m)
    op = fa.DenseMatrixMap(A, storage="f32")
    try:
        assert np.array_equal(op.host_rows(0, m), A32)                       # upload rounds to nearest, read-back widens exactly
        np.testing.assert_allclose(op.device_apply(x), A32 @ x, rtol=1e-12, atol=1e-12 * np.abs(A32).max() * np.abs(x).sum())
        np.testing.assert_allclose(op.device_apply(y, adjoint=True), A32.T @ y, rtol=1e-12, atol=1e-12 * np.abs(A32).max() * np.abs(y).sum())
        assert np.array_equal(op(x), A32 @ x)                                # on host arrays: the rounded matrix as well
        with pytest.raises(AssertionError):
            op(np.zeros(n + 1))
    finally:
        op.close()


def test_float32_host_matrix_goes_in_without_a_float64_detour():
    rng = np.random.RandomState(3)
    A32 = rng.randn(45, 300).astype(np.float32)
    x = rng.randn(300)
    op = fa.LinearMap.from_matrix(A32, storage="f32")
    try:
        assert np.array_equal(op.host_rows(0, 45), A32.astype(np.float64))
        np.testing.assert_allclose(op.device_apply(x), A32.astype(np.float64) @ x, rtol=1e-12, atol=1e-12)
    finally:
        op.close()
    import ctypes as C
    op = fa.DenseMatrixMap(np.zeros((2, 2)))                                  # float64 context refuses the float32 entry point
    try:
        buf = np.zeros((2, 2), dtype=np.float32)
        assert op.ctx.lib.fh_set_matrix_f32(op.ctx._h, buf.ctypes.data_as(C.POINTER(C.c_float)), 2, 2, 2) != 0
    finally:
        op.close()


def test_synthetic_generator_rounds_like_astype_float32():
    m, n = 70, 333
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    full = pr.synth_matrix(m, n, 0, scale)
    op = fa.DenseMatrixMap.synthetic(m, n, 0, scale, storage="f32")
    shard = fa.DenseMatrixMap.synthetic(30, n, 0, scale, row0=40, m_total=m, storage="f32")
    try:
        assert np.array_equal(op.host_rows(0, m), rounded(full))
        assert np.array_equal(shard.host_rows(0, 30), rounded(full[40:70]))
    finally:
        op.close()
        shard.close()


def _state(op, b, mu, x0, g0=None):
    c = op.ctx
    c.set_loss_lsq(b)
    c.set_prox(hip.PROX_SHRINK, mu)
    c.set_vector(hip.VEC_X0, x0)
    c.init()
    return c


@pytest.mark.parametrize("m,n", [(1, 40), (37, 512), (60, 1024), (300, 2048), (90, 5120), (70, 6000), (50, 7000), (64, 8192),   # one member: 1, 2, 4, 5..8 pieces per lane
                                 (130, 9000), (50, 12000), (40, 15000), (33, 16384),              # 2 members
                                 (60, 20000), (35, 32768), (30, 50000), (25, 65536),               # 4 and 8 members
                                 (20, 81920), (18, 100000), (16, 120000), (15, 131072),            # 16 members
                                 # full 8-piece widths: twice the members x 4 pieces, two workgroups per CU (512 co-resident workgroups)
                                 (300, 30000), (1030, 32768), (40, 57345), (600, 60000), (700, 65536)])
def test_fused_step_equals_two_launch_step_in_float32_storage(m, n):
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    A32 = rounded(A)
    b, x0 = rng.randn(m), rng.randn(n) * 0.05
    tau, mu = 0.4, 0.03
    op = fa.DenseMatrixMap(A, storage="f32")
    try:
        c = _state(op, b, mu, x0)
        assert c.fused_supported() in (1, 3)
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
        # against NumPy on the rounded matrix
        g0 = A32.T @ (A32 @ x0 - b)
        xp = fo.shrink(x0 - tau * g0, tau * mu)
        np.testing.assert_allclose(c.get_vector(hip.VEC_XPROX, n), xp, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), A32.T @ (A32 @ xp - b), rtol=1e-9, atol=1e-13)
        c = _state(op, b, mu, x0)
        assert np.array_equal(c.step(tau), f)                                # repeatable
        # FISTA form of the one-pass launch against fwd + adj(accel)
        c = _state(op, b, mu, x0)
        c.step(tau); c.commit()
        s1 = c.fwd(tau)
        a1 = c.adj(tau, True, 0.0 if s1[hip.S_RDOT] > 1e-30 else 0.28)
        x1_ref = c.get_vector(hip.VEC_X1, n)
        c = _state(op, b, mu, x0)
        c.step(tau); c.commit()
        f1 = c.step_accel(tau, 0.28, True)
        np.testing.assert_allclose(f1[hip.S_DXDG], a1[hip.S_DXDG], rtol=1e-9, atol=1e-18)
        np.testing.assert_allclose(f1[hip.S_FSQ_ADJ], a1[hip.S_FSQ_ADJ], rtol=1e-11)
        assert np.array_equal(c.get_vector(hip.VEC_X1, n), x1_ref)
    finally:
        op.close()


@pytest.mark.parametrize("name", ["sparse_ls_64x128_adaptive", "sparse_ls_64x128_accelerated", "nnls_128x64_adaptive", "l1ball_64x128_plain",
                                  "linf_96x96_accelerated", "logistic_100x160_adaptive", "c1_sparse_ls_512x1024_adaptive"])
def test_solve_equals_the_oracle_on_the_rounded_matrix(name):
    """Parity of the float32-storage mode: same iteration and backtrack counts, histories rtol 1e-6, iterates rtol 1e-5
    against the oracle loop run on A.astype(float32) -- exactly the gate of the float64 path."""
    meta, z = H.load_case(name)
    d = dict(H.case_data(meta, z))
    d["A"] = rounded(d["A"])
    want_P = pr.FROM_DATA[meta["kind"]](d)
    o = H.resolve_options(meta["options"], fo)
    np.random.seed(meta["solver_seed"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = fo.fasta(want_P.A, want_P.At, want_P.f, want_P.gradf, want_P.g, want_P.proxg, want_P.x0, **o)
    A = fa.DenseMatrixMap(d["A"], storage="f32")
    try:
        loss = fa.LogisticLoss(d["b"]) if meta["kind"] == "logistic" else fa.LeastSquares(d["b"])
        reg = G.TAGS[meta["kind"]](d)
        np.random.seed(meta["solver_seed"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = fa.fasta(A, A.H, loss.f, loss.gradf, reg.g, reg.prox, np.zeros(d["A"].shape[1]), verbose=False, backend="hip",
                           **H.resolve_options(meta["options"], fa.stopping))
    finally:
        A.close()
    assert got.iteration_count == want.iteration_count and got.backtracks == want.backtracks
    k = got.iteration_count
    G.compare_histories(got, lambda f: getattr(want, f), k, rtol=1e-6, atol=1e-13)
    np.testing.assert_allclose(got.solution, want.solution, rtol=1e-5, atol=1e-9)


def test_stated_tolerance_against_the_float64_matrix():
    """What the opt-in costs: against the reference's own run on the float64 matrix (fixture captured from the reference), the
    float32-storage solve takes the same number of iterations and lands within 3e-7 (relative to the largest entry) of its
    solution, objective within 1e-6 relative (SURVEY.md section 7 probe)."""
    meta, z = H.load_case("sparse_ls_64x128_adaptive")
    d = H.case_data(meta, z)
    A = fa.DenseMatrixMap(d["A"], storage="f32")
    try:
        ls, reg = fa.LeastSquares(d["b"]), fa.Shrink(float(d["mu"]))
        np.random.seed(meta["solver_seed"])
        got = fa.fasta(A, A.H, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(d["A"].shape[1]), verbose=False, backend="hip",
                       **H.resolve_options(meta["options"], fa.stopping))
    finally:
        A.close()
    assert got.iteration_count == int(z["iteration_count"]) and got.backtracks == int(z["backtracks"])
    k = got.iteration_count
    assert np.abs(got.solution - z["solution"]).max() <= 3e-7 * np.abs(z["solution"]).max()
    np.testing.assert_allclose(got.objectives[:k + 1], z["objectives"][:k + 1], rtol=1e-6)
    assert not np.array_equal(got.solution, z["solution"])                   # it IS a different matrix


def test_mid_size_solve_in_float32_storage_matches_oracle_on_rounded_matrix():
    """4096 x 16384 (8 members x 2 pieces... the one-pass kernel on every launch): full solve, iterate for iterate."""
    m, n = 2048, 16384
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    op = fa.DenseMatrixMap.synthetic(m, n, 0, scale, storage="f32")
    try:
        x_true = pr.synth_sparse_signal(n, 1)
        b = op(x_true) + 0.01 * np.random.RandomState(2).randn(m)
        ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
        opts = dict(max_iters=40, tolerance=1e-6, evaluate_objective=True, record_iterates=True)
        solver = fa.FBSolver(op, ls, reg, np.zeros(n), verbose=False, **opts)
        np.random.seed(3)
        got = solver.setup().run()
        assert solver.fused_steps == got.iteration_count + got.backtracks
        A32 = op.host_rows(0, m)
    finally:
        op.close()
    assert np.array_equal(A32, rounded(pr.synth_matrix(m, n, 0, scale)))
    P = pr.sparse_least_squares_from(A32, b, 0.02)
    np.random.seed(3)
    want = fo.fasta(*P.args7(), **opts)
    assert got.iteration_count == want.iteration_count and got.backtracks == want.backtracks
    k = got.iteration_count
    G.compare_histories(got, lambda f: getattr(want, f), k, rtol=1e-6, atol=1e-13)
    np.testing.assert_allclose(got.iterates[:k + 1], want.iterates[:k + 1], rtol=1e-5, atol=1e-9)


def test_create_ex_rejects_what_it_cannot_do():
    import ctypes as C
    lib = hip.load_library()
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 1)
    if hip.device_count() < 2:
        assert lib.fh_create_ex(2, ids, hip.DTYPE_F64, C.byref(h)) != 0        # device 1 does not exist on a one-GPU box
        assert b"out of range" in lib.fh_last_error()
    assert lib.fh_create_ex(0, ids, hip.DTYPE_F64, C.byref(h)) != 0
    assert lib.fh_create_ex(1, ids, 7, C.byref(h)) != 0
    assert lib.fh_create_ex(1, ids, hip.DTYPE_F64, C.byref(h)) == 0
    assert lib.fh_destroy(h) == 0
    with pytest.raises(ValueError):
        hip.HipContext(0, storage="bf16")
