"""In-process row sharding (fh_create_ex with ndev > 1) on ONE GPU: `ShardedDenseMatrixMap(A, devices=[0] * p)`
puts all p row blocks on device 0, where the sum over the blocks is the in-library fixed-order kernel instead of RCCL -- every
other line of the sharded code path (row split of A / b / z, local launches in mode 2, packed loss sums and timeout word, the
separate n-side epilogue, one synchronisation, scalars from shard 0) is the one a multi-GPU run executes.

Full solves through `fasta(ShardedDenseMatrixMap(...), ls.f, ls.gradf, reg.g, reg.prox, x0)` are compared
  * with the oracle (NumPy restatement of fasta/__init__.py:95-320): identical iteration and backtrack counts, histories rtol 1e-6,
    iterates rtol 1e-5 (the north-star tolerance);
  * with the unsharded HIP run of the same problem: histories rtol 1e-9 (only the order of the float64 sums over rows differs).
The operator being sharded is `A @ x` / `A.T @ x` of fasta/linalg.py:41."""
import warnings

import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from oracle import fasta_np as fo
from oracle import problems as pr

pytestmark = pytest.mark.gpu


def _solve(op, P, reg, loss=None, seed=9, **opts):
    ls = loss or fa.LeastSquares(P.data["b"])
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fa.fasta(op, ls.f, ls.gradf, reg.g, reg.prox, P.x0, verbose=False, backend="hip", **opts)


def _oracle(P, seed=9, **opts):
    np.random.seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return fo.fasta(*P.args7(), **opts)


def _check_against(got, want, rtol_hist, rtol_x, atol_x=1e-9):
    assert got.iteration_count == want.iteration_count, (got.iteration_count, want.iteration_count)
    assert got.backtracks == want.backtracks, (got.backtracks, want.backtracks)
    k = want.iteration_count
    for field in ("residuals", "norm_residuals", "stepsizes"):
        np.testing.assert_allclose(getattr(got, field)[:k], getattr(want, field)[:k], rtol=rtol_hist, err_msg=field)
    if want.objectives is not None:
        np.testing.assert_allclose(got.objectives[:k + 1], want.objectives[:k + 1], rtol=rtol_hist)
    if want.iterates is not None:
        np.testing.assert_allclose(got.iterates[:k + 1], want.iterates[:k + 1], rtol=rtol_x, atol=atol_x)
    np.testing.assert_allclose(got.solution, want.solution, rtol=rtol_x, atol=atol_x)


MODES = {
    "adaptive": dict(adaptive=True, accelerate=False),
    "fista": dict(adaptive=False, accelerate=True),
    "plain": dict(adaptive=False, accelerate=False),
    # L and tau0 given and far too optimistic: the first iterations backtrack several times each (fasta/__init__.py:195-217)
    "forced_backtracking": dict(adaptive=True, accelerate=False, L=1.0, tau0=5000.0),
}


@pytest.mark.parametrize("shards", [2, 8])
@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("fused", ["auto", True, False])
def test_sharded_lasso_solve_matches_oracle_and_unsharded_run(shards, mode, fused):
    np.random.seed(5)
    P = pr.sparse_least_squares(M=96, N=160, K=6)
    opts = dict(tolerance=1e-6, evaluate_objective=True, record_iterates=True, max_iters=300, **MODES[mode])
    want = _oracle(P, **opts)
    reg = fa.Shrink(P.data["mu"])
    whole = fa.DenseMatrixMap(P.data["A"])
    op = fa.ShardedDenseMatrixMap(P.data["A"], devices=[0] * shards)
    try:
        assert op.ctx.shard_count() == shards and op.ctx.comm_count() == shards
        assert sum(r for _, r in op.row_blocks()) == 96 and op.row_blocks()[0] == (0, 96 // shards)
        got = _solve(op, P, reg, fused=fused, **opts)
        ref = _solve(whole, P, reg, fused=fused, **opts)
    finally:
        op.close()
        whole.close()
    if mode == "forced_backtracking":
        assert want.backtracks >= 4
    _check_against(got, want, rtol_hist=1e-6, rtol_x=1e-5)
    _check_against(got, ref, rtol_hist=1e-9, rtol_x=1e-9, atol_x=1e-13)


@pytest.mark.parametrize("m,n,shards", [(600, 4096, 3), (256, 16384, 4), (100, 33000, 8), (64, 70000, 2)])
@pytest.mark.parametrize("mode", ["adaptive", "fista"])
def test_sharded_solve_on_team_shapes(m, n, shards, mode):
    """Wider rows: the one-pass kernel runs as teams of 1 / 4 / 16 members per row block (x slice in LDS at n = 70000); uneven
    row blocks (600 over 3 is even, 100 over 8 gives 13,13,13,13,12,12,12,12)."""
    rng = np.random.RandomState(m + n)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    xt = np.zeros(n)
    xt[rng.permutation(n)[:20]] = 1
    b = A @ xt + 0.01 * rng.randn(m)
    P = pr.sparse_least_squares_from(A, b, 0.02)
    opts = dict(tolerance=1e-7, evaluate_objective=True, max_iters=25, **MODES[mode])
    want = _oracle(P, **opts)
    op = fa.ShardedDenseMatrixMap(A, devices=[0] * shards)
    try:
        blocks = op.row_blocks()
        assert [r0 for r0, _ in blocks] == list(np.cumsum([0] + [r for _, r in blocks[:-1]]))
        assert max(r for _, r in blocks) - min(r for _, r in blocks) <= 1 and sum(r for _, r in blocks) == m
        assert op.ctx.fused_supported() in (1, 3)
        got = _solve(op, P, fa.Shrink(0.02), fused=True, **opts)
    finally:
        op.close()
    _check_against(got, want, rtol_hist=1e-6, rtol_x=1e-5)


def test_every_shard_holds_the_same_replicated_state():
    """After sharded steps the n-side vectors and the scalar block are bit-identical on every shard (each entry is either a sum
    over all shards or computed from replicated data), and the shards' z blocks tile z."""
    m, n, shards = 301, 5000, 4
    rng = np.random.RandomState(0)
    A = rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n))
    b, x0 = rng.randn(m), rng.randn(n) * 0.05
    op = fa.ShardedDenseMatrixMap(A, devices=[0] * shards)
    try:
        c = op.ctx
        c.set_loss_lsq(b)
        c.set_prox(hip.PROX_SHRINK, 0.03)
        c.set_vector(hip.VEC_X0, x0)
        c.init()
        for step in (lambda: c.step(0.4), lambda: (c.fwd(0.3), c.adj(0.3))[1], lambda: c.step_accel(0.35, 0.2, True), lambda: c.fwd_adj(0.2)):
            s = step()
            views = [c.shard(k) for k in range(shards)]
            for vec in (hip.VEC_X0, hip.VEC_XPROX, hip.VEC_X1, hip.VEC_G1, hip.VEC_XHAT):
                v0 = views[0][0].get_vector(vec, n)
                for view, _, _ in views[1:]:
                    assert np.array_equal(view.get_vector(vec, n), v0), vec
            z = c.get_vector(hip.VEC_Z, m)
            for view, r0, rows in views:
                assert np.array_equal(view.get_vector(hip.VEC_Z, rows), z[r0:r0 + rows])
            # against NumPy: g1 = A^T (A xprox - b) (fasta/__init__.py:248) with the all-shard sum
            xp = c.get_vector(hip.VEC_XPROX, n)
            x1 = c.get_vector(hip.VEC_X1, n)
            assert np.isfinite(s).all()
            np.testing.assert_allclose(z, A @ xp, rtol=1e-11, atol=1e-14)
            c.commit(False)
            del x1
    finally:
        op.close()


def test_sharded_apply_and_adjoint_match_numpy():
    m, n = 203, 3000
    rng = np.random.RandomState(1)
    A = rng.randn(m, n)
    x, y = rng.randn(n), rng.randn(m)
    op = fa.ShardedDenseMatrixMap(A, devices=[0, 0, 0])
    try:
        np.testing.assert_allclose(op.device_apply(x), A @ x, rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(op.device_apply(y, adjoint=True), A.T @ y, rtol=1e-12, atol=1e-11)
        assert np.array_equal(op.host_rows(60, 90), A[60:150])          # rows gathered across block boundaries (68, 136)
        with pytest.raises(AssertionError):
            op(np.zeros(n + 1))                                         # fasta/linalg.py:58
    finally:
        op.close()


@pytest.mark.parametrize("kind", ["nnls", "l1ball", "linf", "logistic"])
def test_sharded_solves_with_the_other_prox_and_loss_kinds(kind):
    np.random.seed(3)
    if kind == "nnls":
        P, reg = pr.nn_least_squares(M=80, N=60, K=6), fa.NonNeg()
    elif kind == "l1ball":
        P = pr.l1_ball_lasso(M=64, N=128, K=5)
        reg = fa.L1Ball(P.data["mu"])
    elif kind == "linf":
        P = pr.linf_regularised(M=48, N=48)
        reg = fa.LinfProx(P.data["mu"])
    else:
        P = pr.sparse_logistic(M=100, N=160, K=4, mu=4)
        reg = fa.Shrink(P.data["mu"])
    loss = fa.LogisticLoss(P.data["b"]) if kind == "logistic" else None
    opts = dict(tolerance=1e-6, evaluate_objective=True, max_iters=40)
    want = _oracle(P, **opts)
    op = fa.ShardedDenseMatrixMap(P.data["A"], devices=[0] * 4)
    try:
        got = _solve(op, P, reg, loss=loss, **opts)
    finally:
        op.close()
    _check_against(got, want, rtol_hist=1e-6, rtol_x=1e-5)


def test_sharded_float32_storage_matches_the_oracle_on_the_rounded_matrix():
    np.random.seed(5)
    P0 = pr.sparse_least_squares(M=96, N=160, K=6)
    A32 = P0.data["A"].astype(np.float32).astype(np.float64)
    P = pr.sparse_least_squares_from(A32, P0.data["b"], P0.data["mu"])
    opts = dict(tolerance=1e-6, evaluate_objective=True, max_iters=200)
    want = _oracle(P, **opts)
    op = fa.ShardedDenseMatrixMap(P0.data["A"], devices=[0, 0], storage="f32")
    try:
        got = _solve(op, P, fa.Shrink(P.data["mu"]), **opts)
    finally:
        op.close()
    _check_against(got, want, rtol_hist=1e-6, rtol_x=1e-5)


def test_synthetic_sharded_matrix_is_the_unsharded_one():
    m, n, shards = 999, 777, 5
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    whole = fa.DenseMatrixMap.synthetic(m, n, 0, scale)
    op = fa.ShardedDenseMatrixMap.synthetic(m, n, 0, scale, devices=[0] * shards)
    try:
        assert np.array_equal(op.host_rows(0, m), whole.host_rows(0, m))
        assert np.array_equal(op.host_rows(0, m), pr.synth_matrix(m, n, 0, scale))
    finally:
        op.close()
        whole.close()


@pytest.mark.parametrize("mode", ["adaptive", "fista", "forced_backtracking"])
@pytest.mark.parametrize("fused", [True, False])
def test_the_rccl_branch_of_the_in_process_form_with_one_device(mode, fused):
    """Distinct device ids make the exchange a grouped ncclAllReduce on communicators from ncclCommInitAll.  A one-GPU box cannot have
    two ids, but FH_CREATE_RCCL_SHELL builds that form over ONE device: dlopen of librccl, ncclCommInitAll, ncclGroupStart /
    ncclAllReduce(n + 3) / ncclGroupEnd, the device-memory scalar block and the separate epilogue all run for real (a 1-rank sum is
    the identity, so the solve must equal the plain context's)."""
    np.random.seed(5)
    P = pr.sparse_least_squares(M=96, N=160, K=6)
    opts = dict(tolerance=1e-6, evaluate_objective=True, record_iterates=True, max_iters=300, **MODES[mode])
    want = _oracle(P, **opts)
    reg = fa.Shrink(P.data["mu"])
    op = fa.ShardedDenseMatrixMap(P.data["A"], devices=[0], _rccl_shell=True)
    whole = fa.DenseMatrixMap(P.data["A"])
    try:
        assert op.ctx.shard_count() == 1 and op.ctx.comm_count() == 1
        got = _solve(op, P, reg, fused=fused, **opts)
        ref = _solve(whole, P, reg, fused=fused, **opts)
        comm_ms, comm_launches = op.ctx.timing_get(hip.K_COMM)
    finally:
        op.close()
        whole.close()
    _check_against(got, want, rtol_hist=1e-6, rtol_x=1e-5)
    _check_against(got, ref, rtol_hist=1e-9, rtol_x=1e-9, atol_x=1e-13)       # (the separate epilogue kernel sums in another order)


def test_bench_in_process_mode_on_one_gpu():
    """`bench.py --gpus 2 --inproc --devices 0,0`: the benchmark's single-process multi-device mode, both row blocks on this GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--inproc", "--devices", "0,0", "--rows", "4096",
                          "--cols", "8192", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["roofline"]["ranks_seen"] == 2 and out["roofline"]["comm_launches"] >= 4
    assert "in-process" in out["config"]["parallelism"] and out["value"] > 0


def test_refusals():
    lib = hip.load_library()
    with pytest.raises(hip.HipError, match="all different"):
        hip.HipContext(devices=[0, 0, 1])                               # a mixture of repeated and distinct ids
    with pytest.raises(hip.HipError):
        hip.HipContext(devices=[0] * 65)
    c = hip.HipContext(devices=[0, 0, 0])
    try:
        with pytest.raises(hip.HipError, match="cannot be split"):
            c.set_matrix(np.ones((2, 8)))                               # fewer rows than shards
        with pytest.raises(hip.HipError, match="dense operator only"):
            c.set_stencil(8, 8)
        with pytest.raises(hip.HipError, match="already shards"):
            c.comm_init(1, 0, hip.comm_unique_id())
        c.set_matrix(np.ones((7, 8)))
        assert [c.shard(k)[1:] for k in range(3)] == [(0, 3), (3, 2), (5, 2)]
        assert c.shape() == (7, 8)
    finally:
        c.close()
    del lib


@pytest.mark.parametrize("workload,m", [("lasso", 65536), ("nnls", 65536), ("lasso", 262144)])
def test_config2_and_config5_as_eight_row_blocks_first_iterations(workload, m):
    """BASELINE config 2's matrix (65536 x 65536 float64, 32 GiB) as 8 x (8192 x 65536) row blocks on one GPU, and BASELINE config
    5's matrix ITSELF (262144 x 65536, 128 GiB: it fits the 288 GB of one MI355X) as its 8 per-GPU shards of 32768 x 65536: the
    first three iterations equal the unsharded run of the same matrix (scalars rtol 1e-10, iterate rtol 1e-9) and take the
    one-pass kernel on every block."""
    from fasta_python_amd import synthetic
    n = 65536
    scale = synthetic.lasso_scale(m, n)
    x_true = synthetic.sparse_signal(n, seed=1)
    runs = {}
    for name in ("sharded", "whole"):
        op = (fa.ShardedDenseMatrixMap.synthetic(m, n, 0, scale, devices=[0] * 8) if name == "sharded"
              else fa.DenseMatrixMap.synthetic(m, n, 0, scale))
        try:
            b = synthetic.lasso_observation(op, x_true, seed_noise=2, sigma=0.01)
            reg = fa.Shrink(0.02) if workload == "lasso" else fa.NonNeg()
            solver = fa.FBSolver(op, fa.LeastSquares(b), reg, np.zeros(n), verbose=False, max_iters=3, tolerance=0.0,
                                 evaluate_objective=True)
            np.random.seed(3)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                solver.setup()
                while not solver.step():
                    pass
                res = solver.result()
            if name == "sharded":
                assert op.row_blocks() == [(k * (m // 8), m // 8) for k in range(8)]
            runs[name] = (res, solver.fused_steps, b)
        finally:
            op.close()
    (got, fused_s, b_s), (ref, fused_w, b_w) = runs["sharded"], runs["whole"]
    np.testing.assert_allclose(b_s, b_w, rtol=1e-12, atol=1e-14)
    assert fused_s == 3 and fused_w == 3
    _check_against(got, ref, rtol_hist=1e-10, rtol_x=1e-9, atol_x=1e-13)
