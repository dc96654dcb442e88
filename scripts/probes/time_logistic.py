"""Logistic loss at m=n=32768/65536: one-pass kernel vs K-fwd + K-adj (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

for n in (32768, 65536):
    m = n
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_logistic(np.where(rng.rand(m) < 0.5, 1.0, -1.0)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()

    def timed(fn, kid, reps=6):
        fn()
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(reps):
            fn()
        ctx.timing_enable(False)
        ms, cnt = ctx.timing_get(kid)
        return ms / cnt

    t = timed(lambda: ctx.step(0.2), hip.K_FUSED)
    tf = timed(lambda: ctx.fwd(0.2), hip.K_FWD)
    ta = timed(lambda: ctx.adj(0.2), hip.K_ADJ)
    print(f"logistic n={n}: one-pass {t:.3f} ms   two-launch {tf:.3f} + {ta:.3f} = {tf + ta:.3f} ms", flush=True)
    A.close()
