"""Where the non-loop time of a natural solve goes (GPU box): wall clock of every stage of fasta() outside the iteration loop at BASELINE config 2's size.
Usage: python scripts/probes/natural_run_stages.py [n]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic, solver as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = n
A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, seed=1), seed_noise=2, sigma=0.01)
ls, reg = fa.LeastSquares(b), fa.Shrink(0.02)
stamps = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); stamps.append((name, (time.perf_counter() - t0) * 1e3)); return r
    setattr(obj, name, g)
c = A.ctx
for name in ("set_vector", "setup", "get_vector", "set_prox", "fused_agree", "set_loss_lsq"):
    if hasattr(c, name): wrap(c, name)
orig_randn = np.random.randn
def randn(*a):
    t0 = time.perf_counter(); r = orig_randn(*a); stamps.append(("randn", (time.perf_counter() - t0) * 1e3)); return r
np.random.randn = randn
for rep in range(3):
    stamps.clear()
    np.random.seed(3)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = fa.fasta(A, ls.f, ls.gradf, reg.g, reg.prox, np.zeros(n), verbose=False, tolerance=1e-5, backend="hip")
    wall = (time.perf_counter() - t0) * 1e3
    k = r.iteration_count
    loop = (r.times[k] - r.times[0]) * 1e3
    print(f"whole call {wall:8.3f} ms, loop {loop:8.3f} ms ({k} iterations), outside the loop {wall - loop:7.3f} ms: " + ", ".join(f"{nm} {ms:.3f}" for nm, ms in stamps)
          + f" | unaccounted {wall - loop - sum(ms for _, ms in stamps):.3f}", flush=True)
A.close()
