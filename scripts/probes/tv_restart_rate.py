import sys, warnings
sys.path.insert(0, "/root/repo")
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip
from fasta_python_amd.examples.tv_denoising import checkerboard
side = 8192
np.random.seed(7)
M = checkerboard(side, side, side // 32); M += 0.1 * np.random.standard_normal(M.shape)
A = fa.GradDivMap(M.shape)
solver = fa.FBSolver(A, fa.LeastSquares(M / 0.1), fa.TVDualBall(), np.zeros(M.shape + (2,)), adaptive=False, accelerate=True, verbose=False, max_iters=400, tolerance=0.0)
np.random.seed(3)
orig = A.ctx.step_accel
rd = []
def wrapped(tau, coef, restart):
    s = orig(tau, coef, restart); rd.append(float(s[hip.S_RDOT])); return s
A.ctx.step_accel = wrapped
with warnings.catch_warnings(), np.errstate(all="ignore"):
    warnings.simplefilter("ignore")
    solver.setup()
    for _ in range(400): solver.step()
r = np.array(rd) > 1e-30
print("launches", len(rd), "restarts", int(r.sum()), "in first 160:", int(r[:160].sum()), "backtracks", solver.total_backtracks, "restart iterations:", np.nonzero(r)[0][:40].tolist())
A.close()
