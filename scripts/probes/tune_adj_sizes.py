"""K-adj (CPT, slab rows) sweep across matrix sizes, to derive the auto-tuning rule.  GPU box only."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic


def time_kernel(ctx, kid, fn, reps=10):
    fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


for m, n in ((2048, 2048), (4096, 4096), (8192, 8192), (16384, 16384), (512, 1024), (16384, 2048), (2048, 16384), (32768, 32768), (65536, 65536)):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    ctx.fwd(0.2)
    res = []
    for cpt, slab in itertools.product((1, 2, 4), (32, 64, 128, 256, 512, 1024, 2048)):
        if slab > max(32, m):
            continue
        ctx.set_tuning(hip.TUNE_ADJ_CPT, cpt); ctx.set_tuning(hip.TUNE_ADJ_SLAB_ROWS, slab)
        t = time_kernel(ctx, hip.K_ADJ, lambda: ctx.adj(0.2))
        res.append((t, cpt, slab))
    ctx.set_tuning(hip.TUNE_ADJ_CPT, 0); ctx.set_tuning(hip.TUNE_ADJ_SLAB_ROWS, 0)
    auto = time_kernel(ctx, hip.K_ADJ, lambda: ctx.adj(0.2))
    res.sort()
    best = ", ".join(f"cpt={c} slab={s}: {t * 1e3:.1f}us" for t, c, s in res[:5])
    print(f"{m}x{n}: auto {auto * 1e3:.1f}us ({m * n * 8 / auto / 1e6:.0f} GB/s) | best: {best}", flush=True)
    fres = []
    for rows, cap in itertools.product((4, 8, 16), (0, 256, 512, 768, 1024, 2048)):
        ctx.set_tuning(hip.TUNE_FWD_ROWS, rows); ctx.set_tuning(hip.TUNE_FWD_GRID_CAP, cap if cap else 1 << 30)
        t = time_kernel(ctx, hip.K_FWD, lambda: ctx.fwd(0.2))
        fres.append((t, rows, cap))
    ctx.set_tuning(hip.TUNE_FWD_ROWS, 0); ctx.set_tuning(hip.TUNE_FWD_GRID_CAP, 0)
    autof = time_kernel(ctx, hip.K_FWD, lambda: ctx.fwd(0.2))
    fres.sort()
    bestf = ", ".join(f"rows={r} cap={c}: {t * 1e3:.1f}us" for t, r, c in fres[:4])
    print(f"   fwd auto {autof * 1e3:.1f}us ({m * n * 8 / autof / 1e6:.0f} GB/s) | best: {bestf}", flush=True)
    A.close()
