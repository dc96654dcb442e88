"""One-pass kernel on small problems: does a smaller grid (fewer teams when m is small) cut its fixed cost?  (GPU box)
FH_TUNE_FUSED_VARIANT high half = rows-per-team floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

for m, n in ((512, 1024), (2048, 2048), (4096, 4096), (1024, 8192), (2048, 16384), (512, 65536), (64, 65536)):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    res = []
    for floor in (0, 4, 8, 16, 32):
        ctx.set_tuning(hip.TUNE_FUSED_VARIANT, 2 | ((floor or 0xFFFF) << 16))
        for _ in range(5):
            ctx.step(0.2)
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.step(0.2)
        res.append(f"floor {floor:2d}: {(time.perf_counter() - t0) / 50 * 1e6:6.1f} us")
    for _ in range(5):
        ctx.fwd_adj(0.2)
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.fwd_adj(0.2)
    print(f"m={m:5d} n={n:6d}  " + "  ".join(res) + f"   pair {(time.perf_counter() - t0) / 50 * 1e6:6.1f} us", flush=True)
    A.close()
