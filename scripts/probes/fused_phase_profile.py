"""Per-phase s_memtime ticks of the one-pass kernel's row loop (block 0, wave 0).  Needs the debug library:
    make -C fasta_python_amd/csrc prof      (then run this on the GPU box)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fasta_python_amd import hip
import ctypes
_lib = os.environ.get("FASTA_PROF_LIB") or os.path.join(os.path.dirname(hip.__file__), "libfasta_hip_prof.so")
_probe = ctypes.CDLL(_lib)
for _name in list(hip.SIGNATURES):
    if not hasattr(_probe, _name):
        del hip.SIGNATURES[_name]            # an older build lacks newer entry points
hip.load_library(_lib)
import fasta_python_amd as fa
from fasta_python_amd import synthetic
for n, m in ((4096, 4096), (8192, 8192), (16384, 16384), (16384, 8192), (65536, 8192)):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    for v in (2,):
        ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
        print(f"--- n={n} m={m} variant {v}", flush=True)
        ctx.step(0.2); ctx.step(0.2)
    A.close()
