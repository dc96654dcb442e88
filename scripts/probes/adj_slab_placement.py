"""K-adj at 65536^2: slab height (FH_TUNE_ADJ_SLAB_ROWS; default 2048 = 32 slabs, all 1024 workgroups resident at once = 32 read windows spread over the matrix)
against smaller slabs (more workgroups than fit: dispatched in order, i.e. a moving window), cyclic dealing, column chunk width -- on the first and the last of
seven held 32 GiB matrices (profiles/r06_placement.txt).   Usage: python scripts/probes/adj_slab_placement.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

hip.alloc_cache(False)
N = 65536
rng = np.random.RandomState(0)
b = rng.randn(N); x0 = rng.randn(N) * 0.01


def timed(ctx, fn, kid, reps):
    fn(); ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False); ms, cnt = ctx.timing_get(kid); return ms / cnt


held = []
for i in range(7):
    A = fa.DenseMatrixMap.synthetic(N, N, i, synthetic.lasso_scale(N, N)); held.append(A)
    ctx = A.ctx
    ctx.set_loss_lsq(b); ctx.set_prox(hip.PROX_SHRINK, 0.02); ctx.set_vector(hip.VEC_X0, x0); ctx.init(); ctx.fwd(0.2)
    if i not in (0, 3, 6):
        continue
    out = []
    for cpt in (4, 2):
        ctx.set_tuning(hip.TUNE_ADJ_CPT, cpt)
        for cyc in (2, 1):
            ctx.set_tuning(hip.TUNE_ADJ_CYCLIC, cyc)
            for slab in (2048, 1024, 512, 256, 128):
                ctx.set_tuning(hip.TUNE_ADJ_SLAB_ROWS, slab)
                out.append(f"cpt{cpt}{'c' if cyc == 1 else 'b'}{slab}: {timed(ctx, lambda: ctx.adj(0.2), hip.K_ADJ, 3):.3f}")
    print(f"matrix {i}: " + "  ".join(out), flush=True)
