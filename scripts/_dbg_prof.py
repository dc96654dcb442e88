import os, sys, warnings
sys.path.insert(0, os.getcwd())
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
for m, n in ((8192, 8192), (4096, 4096)):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, seed=1), seed_noise=2, sigma=0.01)
    np.random.seed(3)
    s = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=256, tolerance=0.0, device_iters=128)
    print(m, n, flush=True)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        s.setup().run()
    A.ctx.sync()
    A.close()
