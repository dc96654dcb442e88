"""Blocked against row-cyclic dealing of the rows to the teams of the one-pass kernel (FH_TUNE_FUSED_VARIANT bit 32), by WHERE the matrix lies:
seven 65536^2 matrices allocated one after the other and all held (placement_spread.py: the later ones run 6-12 % slow with the blocked dealing), both
dealings timed on each, interleaved twice; then other shapes (fresh allocations).   Usage: python scripts/probes/placement_cyclic.py  -> profiles/r06_placement.txt"""
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
    out = {2: [], 34: []}
    for _ in range(2):
        for v in (2, 34):
            ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
            out[v].append(timed(ctx, lambda: ctx.step(0.2), hip.K_FUSED, reps))
    return out


held = []
for i in range(7):
    A = make(65536, 65536, i); held.append(A)
    o = both(A.ctx, 6)
    adj = timed(A.ctx, lambda: A.ctx.adj(0.2), hip.K_ADJ, 3)
    print(f"65536^2 matrix {i} ({(i + 1) * 32:3d} GiB resident): blocked {o[2][0]:.3f} {o[2][1]:.3f}   cyclic {o[34][0]:.3f} {o[34][1]:.3f} ms   (K-adj {adj:.3f})", flush=True)
for A in held:
    A.close()
time.sleep(9.0)
for (m, n, f32, reps) in ((65536, 65536, True, 8), (32768, 131072, False, 6), (65536, 32768, False, 8), (32768, 32768, False, 20), (16384, 16384, False, 40), (8192, 8192, False, 100),
                          (4096, 4096, False, 200), (20000, 30000, False, 30), (100000, 10000, False, 30)):
    A = make(m, n, 1, f32)
    o = both(A.ctx, reps)
    print(f"{m} x {n}{' float32 storage' if f32 else ''}: blocked {o[2][0]:.4f} {o[2][1]:.4f}   cyclic {o[34][0]:.4f} {o[34][1]:.4f} ms", flush=True)
    A.close(); time.sleep(1.5)
