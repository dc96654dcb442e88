"""Under `rocprofv3 --kernel-trace`: one LASSO solve of 300 iterations at m = n = argv[1] driven by argv[2] in {library, device}.  The trace's dispatch
time stamps give the GPU-side gap between consecutive launches (scripts/probes/chain_gaps_report.py)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import synthetic

n, driver = int(sys.argv[1]), sys.argv[2]
A = fa.DenseMatrixMap.synthetic(n, n, seed=0, scale=synthetic.lasso_scale(n, n))
x_true = synthetic.sparse_signal(n, seed=1)
b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
np.random.seed(3)
solver = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=300, tolerance=0.0, driver=driver, device_iters=64)
with warnings.catch_warnings(), np.errstate(all="ignore"):
    warnings.simplefilter("ignore")
    solver.setup().run()
print(driver, "device steps", solver.device_steps, "library steps", solver.library_steps)
A.close()
