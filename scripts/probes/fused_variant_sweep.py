"""One-pass kernel: scheduling variants (FH_TUNE_FUSED_VARIANT bits 2 = members on one XCD, 4 = no sleep between polls, 32 = rows dealt cyclically) across sizes.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
def make(m, n, seed, f32=False):
    A = fa.DenseMatrixMap.synthetic(m, n, seed, synthetic.lasso_scale(m, n), storage="f32" if f32 else "f64")
    rng = np.random.RandomState(0); ctx = A.ctx
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02); ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01); ctx.init(); ctx.sync()
    return A
def timed(ctx, fn, kid, reps):
    fn(); ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps): fn()
    ctx.timing_enable(False); ms, cnt = ctx.timing_get(kid); return ms / cnt
shapes = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1:]] or [(65536, 65536), (32768, 32768), (16384, 16384), (8192, 8192)]
for (m, n) in shapes:
    reps = max(6, min(200, int(2e9 / (m * n))))
    A = make(m, n, 1); ctx = A.ctx
    (ppt, pipe, team, xlds, nbo), inst = hip.fused_shape(n, "f64", 34, 256)
    for rnd in range(2):
        out = []
        for v in (34, 32, 2, 0, 38):
            ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
            out.append(f"v{v} {timed(ctx, lambda: ctx.step(0.2), hip.K_FUSED, reps):.4f}")
        print(f"{m}x{n} (team of {team}, {ppt} pieces): " + "  ".join(out), flush=True)
    A.close(); time.sleep(1)
