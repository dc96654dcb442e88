import os, sys, warnings
sys.path.insert(0, os.getcwd())
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
m, n = int(sys.argv[1]), int(sys.argv[2])
A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
x_true = synthetic.sparse_signal(n, seed=1)
b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
loss, reg = fa.LeastSquares(b), fa.Shrink(0.02)
np.random.seed(3)
s = fa.FBSolver(A, loss, reg, np.zeros(n), verbose=False, max_iters=64, tolerance=0.0, device_iters=8)
with warnings.catch_warnings(), np.errstate(all="ignore"):
    warnings.simplefilter("ignore")
    s.setup()
    c, st = s.ctx, hip.RunState()
    for call in range(6):
        i = s.i
        st.tau_next, st.alpha1, st.max_residual, st.best_quality = s.tau_next, s.alpha1, s.max_residual, s.best_quality
        st.iteration, st.backtracks, st.stopped = i, s.total_backtracks, 0
        for j in range(max(i - s.window + 1, 0), i + 1):
            st.f_window[j % 64] = s.f_hist[j]
        h = c.run(8, s._run_opts, st)
        print("call", call, "len(h)", len(h), "iteration", st.iteration, "backtracks", st.backtracks, "stopped", st.stopped, "tau", st.tau_next, flush=True)
        print("   resid", h[:, 0], flush=True)
        k = len(h)
        s.f_hist[i + 1:i + k + 1] = h[:, 3]
        s.tau_next, s.alpha1, s.max_residual, s.best_quality = st.tau_next, st.alpha1, st.max_residual, st.best_quality
        s.total_backtracks = int(st.backtracks); s.i = int(st.iteration)
A.close()
