"""Same-box A/B of two builds of libfasta_hip.so: python scripts/probes/ab_lib.py <lib.so> [n]  (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fasta_python_amd import hip
lib = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
import ctypes
probe = ctypes.CDLL(lib)
for name in list(hip.SIGNATURES):
    if not hasattr(probe, name):
        del hip.SIGNATURES[name]            # an older build lacks newer entry points
if hasattr(probe, "fh_abi_sizes"):
    hip.load_library(lib)
else:                                       # a build from before the layout check: bind what it has (the probe below passes no struct by pointer)
    for name, (res, args) in hip.SIGNATURES.items():
        fn = getattr(probe, name); fn.restype, fn.argtypes = res, args
    hip._lib = probe
import fasta_python_amd as fa
from fasta_python_amd import synthetic
m = n
A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
ctx = A.ctx
rng = np.random.RandomState(0)
ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
ctx.init()
def timed(fn, kid, reps=8 if n > 16384 else 200):
    fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt
for rnd in range(3):
    print(f"{os.path.basename(lib):28s} n={n} round {rnd}: one-pass {timed(lambda: ctx.step(0.2), hip.K_FUSED):.3f} ms   K-fwd {timed(lambda: ctx.fwd(0.2), hip.K_FWD, 3):.3f}  K-adj {timed(lambda: ctx.adj(0.2), hip.K_ADJ, 3):.3f}", flush=True)
A.close()
