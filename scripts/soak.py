"""Soak run of the kernels with cross-workgroup hand-offs: thousands of consecutive launches at full size (one-pass dense and stencil kernels, the
multi-workgroup level search, the set-up kernel) and tens of thousands of iterations inside persistent launches (the device loop), checking that the
bounded spins never time out, that per-iteration time stays flat and that two identical solves are bitwise equal (GPU box)."""
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

def dense_prox(kind):
    """round 5: the l1-ball / l-infinity prox -- every forward launch is preceded by the multi-workgroup level search (csrc/fh_prox.h)"""
    def make():
        m = n = 65536
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
        x_true = synthetic.sparse_signal(n, 1)
        b = synthetic.lasso_observation(A, x_true, 2, 0.01)
        reg = fa.L1Ball(0.8 * float(np.abs(x_true).sum())) if kind == "l1ball" else fa.LinfProx(0.02)
        return A, fa.LeastSquares(b), reg, np.zeros(n), dict(adaptive=True)
    return make

def device_loop(m, n, iters, per_launch):
    """round 5: the FBS loop on the device (fh_run): `iters` iterations in persistent launches of `per_launch`, twice; every iteration must have run
    inside a persistent launch (no timeout, no fall-back) and the two solves must be bitwise identical"""
    sols = []
    for rep in range(2):
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
        try:
            b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
            np.random.seed(3)
            t0 = time.perf_counter()
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
                c = fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, max_iters=iters, tolerance=0.0, device_iters=per_launch, backend="hip")
            dt = time.perf_counter() - t0
            assert c.device_steps == c.iteration_count == iters, (c.device_steps, c.iteration_count)
            sols.append((c.solution.copy(), c.residuals.copy(), c.stepsizes.copy()))
            print(f"device loop {m}x{n} run {rep}: {c.iteration_count} iterations in launches of {per_launch}, {c.backtracks} backtracks, {c.iteration_count / dt:9.1f} it/s (whole call)", flush=True)
        finally:
            A.close()
    same = all(np.array_equal(a, b_, equal_nan=True) for a, b_ in zip(sols[0], sols[1]))
    print(f"device loop {m}x{n}: two runs bitwise identical: {same}", flush=True)
    assert same

def setups(calls):
    """round 5: the one-read set-up (fh_setup) over and over on one context: one one-pass launch per call, never the three-pass fall-back, same scalars every time"""
    m = n = 65536
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    try:
        c = A.ctx
        b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
        c.set_loss_lsq(b); c.set_prox(hip.PROX_SHRINK, 0.02)
        rng = np.random.RandomState(0)
        c.set_vector(hip.VEC_T0, rng.randn(n)); c.set_vector(hip.VEC_T1, rng.randn(n)); c.set_vector(hip.VEC_X0, np.zeros(n))
        first, t0 = None, time.perf_counter()
        c.timing_reset(); c.timing_enable(True)
        for _ in range(calls):
            s = c.setup().copy()
            if first is None: first = s
            assert np.array_equal(s, first), "fh_setup is not repeatable"
        c.timing_enable(False)
        ms, cnt = c.timing_get(hip.K_FUSED)
        assert cnt == calls, f"{cnt} one-pass launches for {calls} calls: some took the three-pass fall-back"
        print(f"fh_setup x {calls} at {m}x{n}: {cnt} one-pass launches ({ms / cnt:.3f} ms each), {(time.perf_counter() - t0) / calls * 1e3:.3f} ms per call, scalars identical every time", flush=True)
    finally:
        A.close()

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

def seq_wait(name, make_op, loss_reg, iters, extra=None):
    """round 6: a step waits for its scalar block by the sequence number the launch writes behind it (FH_TUNE_SEQ_POLL) and the decisions are taken
    by the library's host-side loop: tens of thousands of iterations, three ways -- library loop + sequence number (default), library loop +
    hipStreamSynchronize, Python driver + sequence number -- must give EQUAL histories: one stale entry of one scalar block would change a decision."""
    runs = {}
    for tag, poll, driver in (("library, sequence number", 1, "library"), ("library, stream synchronisation", 0, "library"), ("python, sequence number", 1, "python")):
        A = make_op()
        try:
            A.ctx.set_tuning(hip.TUNE_SEQ_POLL, poll)
            loss, reg, x0 = loss_reg(A)
            np.random.seed(3)
            solver = fa.FBSolver(A, loss, reg, x0, verbose=False, max_iters=iters, tolerance=0.0, driver=driver, **(extra or {}))
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                solver.setup()
                t0 = time.perf_counter()
                c = solver.run()
                dt = time.perf_counter() - t0
            runs[tag] = (c.residuals.copy(), c.stepsizes.copy(), c.solution.copy(), c.backtracks)
            print(f"{name} [{tag}]: {c.iteration_count} iterations, {c.backtracks} backtracks, {c.iteration_count / dt:9.1f} it/s", flush=True)
        finally:
            A.close()
    ref = runs["library, sequence number"]
    same = all(all(np.array_equal(a, b_, equal_nan=True) for a, b_ in zip(r[:3], ref[:3])) and r[3] == ref[3] for r in runs.values())
    print(f"{name}: the three ways bitwise identical: {same}", flush=True)
    assert same

def lasso_op(m, n):
    return lambda: fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))

def lasso_terms(A):
    n = A.Vshape[0]
    b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
    return fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n)

def tv_small_op(side):
    return lambda: fa.GradDivMap((side, side))

def tv_small_terms(A):
    from fasta_python_amd.examples.tv_denoising import checkerboard
    side = A.image_shape[0]
    np.random.seed(7)
    M = checkerboard(side, side, max(1, side // 32)) + 0.1 * np.random.standard_normal((side, side))
    return fa.LeastSquares(M / 0.1), fa.TVDualBall(), np.zeros(M.shape + (2,))

# (tvring / tvslots: the experimental library only -- FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_experimental.so python scripts/soak.py tvring,tvslots)
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["seq", "dense", "f32", "l1ball", "linf", "devloop", "setup", "tv", "tvacc", "blocks8", "config5", "pair128"]
if "seq" in which:
    seq_wait("LASSO 8192^2", lasso_op(8192, 8192), lasso_terms, 30000)
    seq_wait("LASSO 2048x3000, FISTA", lasso_op(2048, 3000), lasso_terms, 30000, dict(adaptive=False, accelerate=True))
    seq_wait("LASSO 600x1000, two launches per iteration", lasso_op(600, 1000), lasso_terms, 30000, dict(fused=False))
    seq_wait("TV 512^2 adaptive", tv_small_op(512), tv_small_terms, 30000)
    seq_wait("TV 512^2 FISTA", tv_small_op(512), tv_small_terms, 30000, dict(adaptive=False, accelerate=True))
if "devloop" in which:
    device_loop(4096, 4096, 20000, 64)
    device_loop(512, 1024, 50000, 512)
    device_loop(2048, 3000, 20000, 7)
if "setup" in which:
    setups(300)
if "l1ball" in which:
    soak("l1-ball constrained LASSO 65536^2 (level search every launch)", dense_prox("l1ball"), 1500)
if "linf" in which:
    soak("least squares + l-infinity prox 65536^2 (level search every launch)", dense_prox("linf"), 1500)
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
