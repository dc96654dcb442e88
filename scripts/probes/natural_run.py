"""the natural run of BASELINE config 2 / 3 (tolerance 1e-5, default options) -- iterations to converge and wall time.

Run with OPENBLAS_NUM_THREADS=1: this script's own np.linalg.norm wakes OpenBLAS's 64 worker threads, which then busy-wait for
~0.1 s; under the GPU box's CPU quota that stalls the main thread for 40-60 ms a few iterations into the NEXT solve (seen as one
slow iteration followed by a clock ramp).  The product's host loop makes no BLAS call."""
import os, sys, time, warnings
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import synthetic

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
A = fa.DenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n))
x_true = synthetic.sparse_signal(n, seed=1)
cases = (("LASSO (shrink, mu=0.02)", fa.Shrink(0.02), 0.01), ("NNLS (non-negativity)", fa.NonNeg(), 0.005))
if len(sys.argv) > 2:      # any extra argument: repeat the pair in reverse order (is a slow first iteration tied to the problem or to the call?)
    cases = cases + cases[::-1]
for name, reg, sigma in cases:
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=sigma)
    ls = fa.LeastSquares(b)
    np.random.seed(3)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, tolerance=1e-5, evaluate_objective=True)
    wall = time.perf_counter() - t0
    k = c.iteration_count
    loop = c.times[k] - c.times[0]
    err = np.linalg.norm(c.solution - x_true) / np.linalg.norm(x_true)
    dts = np.diff(c.times[:k + 1]) * 1e3
    print(f"   per-iteration ms: median {np.median(dts):.3f}, max {dts.max():.3f} (iteration {int(dts.argmax())}), first five {np.round(dts[:5], 3).tolist()}")
    print(f"{name} {m}x{n}: {k} iterations, {c.backtracks} backtracks, loop {loop:.3f} s ({k / loop:.1f} it/s), whole call {wall:.3f} s "
          f"(setup passes included), final residual {c.residuals[k - 1]:.3e}, objective {c.objectives[k]:.6e}, ||x - x_true||/||x_true|| = {err:.3e}", flush=True)
A.close()
