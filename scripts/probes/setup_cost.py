"""What the solver's set-up costs (fasta/__init__.py:100-113 + :135-137): the three passes fh_gradient_at x 2 + fh_init against the ONE
call fh_setup (csrc/fh_setup.h: one read of A for the three right-hand sides), HIP-event time of the launches and wall clock of the
calls, across sizes; and the whole natural solve (tolerance 1e-5) of the LASSO bench problem: loop seconds and whole-call seconds.
Usage: python scripts/probes/setup_cost.py [sizes...]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536]
for n in sizes:
    m = n
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    c = A.ctx
    x_true = synthetic.sparse_signal(n, seed=1)
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
    c.set_loss_lsq(b); c.set_prox(hip.PROX_SHRINK, 0.02)
    rng = np.random.RandomState(0)
    for which in (hip.VEC_T0, hip.VEC_T1):
        c.set_vector(which, rng.randn(n))
    c.set_vector(hip.VEC_X0, np.zeros(n))
    res = {}
    for name in ("three passes", "fh_setup", "three passes", "fh_setup"):
        c.sync(); c.timing_reset(); c.timing_enable(True)
        t0 = time.perf_counter()
        if name == "fh_setup":
            s = c.setup()
        else:
            c.gradient_at(hip.VEC_T0, hip.VEC_T2); c.gradient_at(hip.VEC_T1, hip.VEC_T3)
            dg, dx = c.diff_norm(hip.VEC_T2, hip.VEC_T3), c.diff_norm(hip.VEC_T0, hip.VEC_T1)
            s = c.init()
        wall = (time.perf_counter() - t0) * 1e3
        c.timing_enable(False)
        ms, cnt = c.timing_get(hip.K_FUSED)
        res[name] = (wall, ms, cnt)
    print(f"n={n:6d}  " + "  |  ".join(f"{k}: wall {v[0]:8.3f} ms, {v[2]} one-pass launch(es) {v[1]:8.3f} ms" for k, v in res.items()), flush=True)
    ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
    for rep in range(2):
        np.random.seed(3)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, tolerance=1e-5, backend="hip")
        wall = time.perf_counter() - t0
        k = r.iteration_count
        loop = r.times[k] - r.times[0]
        print(f"          natural LASSO run: {k} iterations, loop {loop * 1e3:8.3f} ms, whole call {wall * 1e3:8.3f} ms, set-up + rest {1e3 * (wall - loop):7.3f} ms", flush=True)
    A.close()
