"""K-adj with contiguous slabs against rows dealt cyclically to the slabs (FH_TUNE_ADJ_CYCLIC), by where the matrix lies (seven 65536^2 matrices held) and over shapes.
Usage: python scripts/probes/adj_cyclic.py   -> profiles/r06_placement.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

hip.alloc_cache(False)


def make(m, n, seed, f32=False):
    A = fa.DenseMatrixMap.synthetic(m, n, seed, synthetic.lasso_scale(m, n), storage="f32" if f32 else "f64")
    rng = np.random.RandomState(0)
    ctx = A.ctx
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02); ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01); ctx.init(); ctx.sync()
    return A


def timed(ctx, fn, kid, reps):
    fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


def both(ctx, reps):
    out = {2: [], 1: []}
    ctx.fwd(0.2)
    for _ in range(2):
        for v in (2, 1):
            ctx.set_tuning(hip.TUNE_ADJ_CYCLIC, v)
            out[v].append(timed(ctx, lambda: ctx.adj(0.2), hip.K_ADJ, reps))
    ctx.set_tuning(hip.TUNE_ADJ_CYCLIC, 0)
    return out


held = []
for i in range(7):
    A = make(65536, 65536, i); held.append(A)
    o = both(A.ctx, 4)
    fwd = timed(A.ctx, lambda: A.ctx.fwd(0.2), hip.K_FWD, 3)
    print(f"65536^2 matrix {i} ({(i + 1) * 32:3d} GiB resident): K-adj contiguous slabs {o[2][0]:.3f} {o[2][1]:.3f}   cyclic {o[1][0]:.3f} {o[1][1]:.3f} ms   (K-fwd {fwd:.3f})", flush=True)
for A in held:
    A.close()
time.sleep(9.0)
for (m, n, f32, reps) in ((65536, 65536, True, 8), (32768, 131072, False, 6), (65536, 32768, False, 8), (32768, 32768, False, 20), (16384, 16384, False, 40), (8192, 8192, False, 100),
                          (4096, 4096, False, 200), (1024, 2048, False, 200), (20000, 30000, False, 30), (100000, 10000, False, 30), (200, 100000, False, 100)):
    A = make(m, n, 1, f32)
    o = both(A.ctx, reps)
    print(f"{m} x {n}{' float32 storage' if f32 else ''}: K-adj contiguous slabs {o[2][0]:.4f} {o[2][1]:.4f}   cyclic {o[1][0]:.4f} {o[1][1]:.4f} ms", flush=True)
    A.close(); time.sleep(1.5)
