"""Iterations/s of the dense LASSO bench problem with the loop on the host (one launch per iteration) and on the device
(fasta(..., device_iters=K): K iterations per persistent launch), sizes up to n = 4096.  Usage: python scripts/probes/run_cost.py [sizes...]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(512, 1024), (2048, 2048), (4096, 4096), (16384, 4096), (65536, 4096)]
for m, n in shapes:
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    x_true = synthetic.sparse_signal(n, seed=1)
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
    loss, reg = fa.LeastSquares(b), fa.Shrink(0.02)
    out = []
    for name, kw in (("host loop", {}), ("device_iters=8", dict(device_iters=8)), ("device_iters=64", dict(device_iters=64)), ("device_iters=512", dict(device_iters=512))):
        best = 0.0
        for rep in range(3):
            np.random.seed(3)
            solver = fa.FBSolver(A, loss, reg, np.zeros(n), verbose=False, max_iters=1024, tolerance=0.0, backtrack=True, **kw)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                solver.setup()
                A.ctx.sync()
                t0 = time.perf_counter()
                solver.run()
                A.ctx.sync()
                dt = time.perf_counter() - t0
            best = max(best, solver.i / dt)
        out.append(f"{name} {best:9.0f} it/s ({1e6 / best:6.1f} us, {m * n * 8 / (1e3 / best) / 1e9 if False else m * n * 8 * best / 1e12:5.2f} TB/s, device steps {solver.device_steps})")
    print(f"{m:6d} x {n:5d}: " + " | ".join(out), flush=True)
    A.close()
