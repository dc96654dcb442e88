import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from fasta_python_amd import hip
import fasta_python_amd as fa
from fasta_python_amd import synthetic
def t_step(m, n):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01); ctx.init()
    best = 1e9
    for _ in range(3):
        for _ in range(5): ctx.step(0.2)
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(40): ctx.step(0.2)
        ctx.timing_enable(False)
        ms, cnt = ctx.timing_get(hip.K_FUSED)
        best = min(best, ms / cnt)
    A.close()
    return best
for n in (8192, 16384, 65536):
    ms_ = []
    sizes = [m for m in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768) if m * n * 8 <= 20e9]
    for m in sizes:
        ms_.append(t_step(m, n))
    # fit on the 4 largest
    x = np.array(sizes[-4:], float); y = np.array(ms_[-4:])
    b, a = np.polyfit(x, y, 1)
    print(f"n={n}: " + "  ".join(f"m={m}: {t*1e3:.1f} us" for m, t in zip(sizes, ms_)))
    print(f"   fit on the 4 largest: T = {a*1e3:.1f} us + m x {b*1e6:.3f} ns  => stream rate {n*8/(b*1e-3)/1e12:.2f} TB/s, fixed {a*1e3:.1f} us", flush=True)
