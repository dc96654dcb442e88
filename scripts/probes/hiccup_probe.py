"""Where does the 40-60 ms stall in the first iterations of a second solve come from?  (GPU box)"""
import os, sys, time, gc, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

m = n = 65536
A = fa.DenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n))
ctx = A.ctx
x_true = synthetic.sparse_signal(n, seed=1)
b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "nogc":
    gc.disable()
for rep in range(4):
    solver = fa.FBSolver(A, ls, reg, np.zeros(n), verbose=False, tolerance=0.0, max_iters=12)
    np.random.seed(3)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        solver.setup()
        ctx.timing_reset(); ctx.timing_enable(True)
        walls = []
        for _ in range(12):
            t0 = time.perf_counter()
            solver.step()
            walls.append((time.perf_counter() - t0) * 1e3)
        ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(hip.K_FUSED)
    print(f"[{mode}] solve {rep}: wall per step {np.round(walls, 1).tolist()}  kernel avg {ms / cnt:.3f} ms over {cnt} launches", flush=True)
    if mode == "sleep":
        time.sleep(0.5)
    if mode in ("d2h", "both"):
        ctx.get_vector(hip.VEC_BEST, n)                    # what fasta() does at the end of a solve
    if mode in ("h2d", "both"):
        ctx.set_loss_lsq(b + rep)                          # what the next solve does first
    if mode == "apply":
        A(np.ones(n))                                      # H2D + K-fwd + D2H (lasso_observation)
A.close()
