"""Does a padded leading dimension help the one-pass kernel at n=65536 (power-of-two row stride)?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = n
for pad in (0, 16, 32, 64, 128, 256, 512, 2048):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), tuning={hip.TUNE_LD_PAD: pad})
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    out = []
    for v in (2, 34, 10):
        ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
        ctx.step(0.2)
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(5):
            ctx.step(0.2)
        ctx.timing_enable(False)
        ms, cnt = ctx.timing_get(hip.K_FUSED)
        out.append(f"v{v}: {ms / cnt:6.3f} ms ({m * n * 8 / (ms / cnt) / 1e6:5.0f} GB/s)")
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(3):
        ctx.fwd(0.2); ctx.adj(0.2)
    ctx.timing_enable(False)
    tf = ctx.timing_get(hip.K_FWD); ta = ctx.timing_get(hip.K_ADJ)
    print(f"n={n} pad={pad:5d}  " + "  ".join(out) + f"   fwd {tf[0] / tf[1]:.3f} adj {ta[0] / ta[1]:.3f}", flush=True)
    A.close()
