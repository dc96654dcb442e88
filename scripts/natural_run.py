"""the natural run of BASELINE config 2 / 3 (tolerance 1e-5, default options) -- iterations to converge and wall time."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import synthetic

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
A = fa.DenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n))
x_true = synthetic.sparse_signal(n, seed=1)
for name, reg, sigma in (("LASSO (shrink, mu=0.02)", fa.Shrink(0.02), 0.01), ("NNLS (non-negativity)", fa.NonNeg(), 0.005)):
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
    print(f"{name} {m}x{n}: {k} iterations, {c.backtracks} backtracks, loop {loop:.3f} s ({k / loop:.1f} it/s), whole call {wall:.3f} s "
          f"(setup passes included), final residual {c.residuals[k - 1]:.3e}, objective {c.objectives[k]:.6e}, ||x - x_true||/||x_true|| = {err:.3e}", flush=True)
A.close()
