"""A/B the fused one-pass kernel's scheduling variants in one process (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
ctx = A.ctx
rng = np.random.RandomState(0)
ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
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


for rnd in range(2):
    for variant in (2, 10):
        ctx.set_tuning(hip.TUNE_FUSED_VARIANT, variant)
        t = timed(lambda: ctx.step(0.2), hip.K_FUSED)
        print(f"round {rnd} variant {variant} (same-XCD teams={(variant >> 1) & 1} legacy 8-member shape={(variant >> 3) & 1} cyclic rows={(variant >> 5) & 1}): "
              f"{t:7.3f} ms  moves {m * n * 8 / t / 1e6:6.0f} GB/s  algorithmic {2 * m * n * 8 / t / 1e6:6.0f} GB/s", flush=True)
ctx.set_tuning(hip.TUNE_FUSED_VARIANT, 2)
ctx.step(0.2); ctx.commit()
t = timed(lambda: ctx.step_accel(0.2, 0.4, True), hip.K_FUSED)
print(f"accelerated one-pass step: {t:7.3f} ms", flush=True)
tf = timed(lambda: ctx.fwd(0.2), hip.K_FWD)
ta = timed(lambda: ctx.adj(0.2), hip.K_ADJ)
print(f"two-launch: fwd {tf:.3f} + adj {ta:.3f} = {tf + ta:.3f} ms")
A.close()
