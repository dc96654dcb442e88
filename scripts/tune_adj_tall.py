"""K-adj slabs taller than the LDS stage (multi-piece) at the large sizes.  GPU box only."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic


def time_kernel(ctx, kid, fn, reps=5):
    fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


for m, n in ((32768, 32768), (65536, 65536)):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init(); ctx.fwd(0.2)
    for cpt, slab in itertools.product((2, 4), (1024, 2048, 4096, 8192, 16384)):
        ctx.set_tuning(hip.TUNE_ADJ_CPT, cpt); ctx.set_tuning(hip.TUNE_ADJ_SLAB_ROWS, slab)
        t = time_kernel(ctx, hip.K_ADJ, lambda: ctx.adj(0.2))
        print(f"{m}x{n} cpt={cpt} slab={slab:5d}: {t:8.4f} ms  {m * n * 8 / t / 1e6:6.0f} GB/s", flush=True)
    A.close()
