"""Soak run of the one-pass kernels: thousands of consecutive launches at full size, checking that the bounded-spin
hand-offs never time out, that per-iteration time stays flat and that two identical solves are bitwise equal (GPU box)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

def soak(name, make, iters):
    sols = []
    for rep in range(2):
        A, loss, reg, x0, opts = make()
        solver = fa.FBSolver(A, loss, reg, x0, verbose=False, max_iters=iters, tolerance=0.0, **opts)
        np.random.seed(3)
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            solver.setup()
            t0 = time.perf_counter()
            c = solver.run()
            dt = time.perf_counter() - t0
        per = np.diff(c.times[:c.iteration_count + 1]) * 1e3
        assert solver.use_fused and solver.fused_steps == c.iteration_count + c.backtracks, "a launch fell back"
        sols.append((c.solution.copy(), c.residuals.copy()))
        print(f"{name} run {rep}: {c.iteration_count} iterations, {c.backtracks} backtracks, {c.iteration_count / dt:8.1f} it/s; per-iteration ms: "
              f"min {per.min():.3f} median {np.median(per):.3f} p99 {np.percentile(per, 99):.3f} max {per.max():.3f}; one-pass launches {solver.fused_steps}", flush=True)
        A.close()
    same = np.array_equal(sols[0][0], sols[1][0]) and np.array_equal(sols[0][1], sols[1][1], equal_nan=True)
    print(f"{name}: two runs bitwise identical: {same}", flush=True)
    assert same

def dense(storage):
    def make():
        m = n = 65536
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), storage=storage)
        b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
        return A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), dict(adaptive=True)
    return make

def dense_blocks(m, blocks):
    """the single-call multi-device form with all row blocks on this GPU (ShardedDenseMatrixMap, repeated device id)"""
    def make():
        n = 65536
        A = fa.ShardedDenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), devices=[0] * blocks)
        b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
        return A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), dict(adaptive=True)
    return make

def tv(accel):
    def make():
        from fasta_python_amd.examples.tv_denoising import checkerboard
        np.random.seed(7)
        M = checkerboard(8192, 8192, 256) + 0.1 * np.random.standard_normal((8192, 8192))
        A = fa.GradDivMap(M.shape)
        return A, fa.LeastSquares(M / 0.1), fa.TVDualBall(), np.zeros(M.shape + (2,)), dict(adaptive=not accel, accelerate=accel)
    return make

def tv_form(tuning):
    """round 4: the LDS-DMA ring / the persistent chunk walk of the one-pass TV sweep"""
    inner = tv(False)
    def make():
        A, loss, reg, x0, opts = inner()
        for k, v in tuning.items():
            A.ctx.set_tuning(k, v)
        return A, loss, reg, x0, opts
    return make

def side_by_side(cus, iters):
    """round 4: TWO solves at once on this GPU, each on its own thread and context, each one-pass grid capped to `cus` CUs (FH_TUNE_FUSED_CUS):
    the launches overlap in time; neither may ever time out, and both must produce the bits of a solve that ran alone with the same cap"""
    import threading
    m, n = 32768, 65536
    def one(out, key):
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), tuning={hip.TUNE_FUSED_CUS: cus})
        try:
            b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
            solver = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=iters, tolerance=0.0)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                with np_seed_lock:          # setup() draws its two Lipschitz probes from the GLOBAL RNG: seed + draws must not interleave between threads
                    np.random.seed(3)
                    solver.setup()
                barrier.wait()
                t0 = time.perf_counter()
                c = solver.run()
                dt = time.perf_counter() - t0
            assert solver.use_fused and solver._fused_backoff == 64 and solver.fused_steps == c.iteration_count + c.backtracks, "a launch timed out / fell back"
            out[key] = (c.solution.copy(), c.residuals.copy(), c.iteration_count / dt)
        finally:
            A.close()
    res = {}
    np_seed_lock = threading.Lock()
    barrier = threading.Barrier(1)
    one(res, "alone")
    barrier = threading.Barrier(2)
    ts = [threading.Thread(target=one, args=(res, k)) for k in ("a", "b")]
    for t in ts: t.start()
    for t in ts: t.join()
    same = all(np.array_equal(res[k][0], res["alone"][0]) and np.array_equal(res[k][1], res["alone"][1], equal_nan=True) for k in ("a", "b"))
    print(f"two solves side by side, {cus} CUs each ({m}x{n}, {iters} iterations): alone {res['alone'][2]:.1f} it/s; together {res['a'][2]:.1f} + {res['b'][2]:.1f} it/s; "
          f"no timeout; bitwise identical to the solve that ran alone: {same}", flush=True)
    assert same

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["dense", "f32", "tv", "tvacc", "blocks8", "config5", "tvring", "tvslots", "pair128"]
if "pair128" in which:
    side_by_side(128, 600)
if "tvring" in which:
    soak("TV 8192^2 adaptive, FH_TUNE_TV_RING = 2", tv_form({hip.TUNE_TV_RING: 2}), 800)
if "tvslots" in which:
    soak("TV 8192^2 adaptive, FH_TUNE_TV_SLOTS = 5 x 32-row chunks", tv_form({hip.TUNE_TV_SLOTS: 5, hip.TUNE_TV_ROWS: 32}), 800)
if "blocks8" in which:
    soak("LASSO 65536^2 as 8 in-process row blocks", dense_blocks(65536, 8), 1500)
if "config5" in which:
    soak("LASSO 262144x65536 (BASELINE config 5) as 8 in-process row blocks on one GPU", dense_blocks(262144, 8), 400)
if "dense" in which:
    soak("LASSO 65536^2 f64", dense("f64"), 1500)
if "f32" in which:
    soak("LASSO 65536^2 f32-storage", dense("f32"), 1500)
if "tv" in which:
    soak("TV 8192^2 adaptive", tv(False), 1500)
if "tvacc" in which:
    soak("TV 8192^2 FISTA", tv(True), 1500)
