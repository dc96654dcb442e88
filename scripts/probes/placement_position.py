"""Follow-up to placement_spread.py: is the slow half of the device a matter of POSITION (which physical memory), and does the row stride matter there?
 1. four ballast matrices (128 GiB) are held, so that whatever is allocated next comes from the part of the device where round 0/1 of placement_spread.py ran slow;
 2. there: the 65536^2 matrix with leading-dimension pads 0 / 16 / 64 / 512 / 2048 doubles, each with three shapes of the one-pass kernel, K-fwd, K-adj, stream;
 3. two ballast matrices are released, the device settles, a new matrix is allocated (it should land where they were) and timed; then once more behind the ballast.
Usage: python scripts/probes/placement_position.py   -> profiles/r06_placement.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

N = 65536
hip.alloc_cache(False)
rng = np.random.RandomState(0)
b = rng.randn(N); x0 = rng.randn(N) * 0.01


def make(seed, pad=0):
    A = fa.DenseMatrixMap.synthetic(N, N, seed, synthetic.lasso_scale(N, N), tuning={hip.TUNE_LD_PAD: pad} if pad else None)
    ctx = A.ctx
    ctx.set_loss_lsq(b); ctx.set_prox(hip.PROX_SHRINK, 0.02); ctx.set_vector(hip.VEC_X0, x0); ctx.init(); ctx.sync()
    return A


def timed(ctx, fn, kid, reps):
    fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


def report(tag, A, variants=(0,)):
    ctx = A.ctx
    out = []
    for v in variants:
        if v:
            ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
        out.append(f"one-pass{'' if not v else ' v%d' % v} {timed(ctx, lambda: ctx.step(0.2), hip.K_FUSED, 6):.3f}")
    fwd = timed(ctx, lambda: ctx.fwd(0.2), hip.K_FWD, 3); adj = timed(ctx, lambda: ctx.adj(0.2), hip.K_ADJ, 3)
    ms, nbytes = ctx.stream_read_ms(3)
    print(f"{tag}: " + "  ".join(out) + f"   K-fwd {fwd:.3f}  K-adj {adj:.3f}  stream {nbytes / ms / 1e6:.0f} GB/s", flush=True)


ballast = []
for i in range(4):
    ballast.append(make(i)); report(f"ballast {i}", ballast[-1])
for pad in (0, 16, 64, 512, 2048):
    A = make(10 + pad, pad); report(f"behind 128 GiB of ballast, ld pad {pad:4d}", A, (2, 34, 10)); A.close(); time.sleep(2.5)
for i in (0, 1):
    ballast[i].close()
time.sleep(4.0)
A = make(20); report("after releasing ballast 0 and 1 (64 GiB free below the rest)", A)
B = make(21); report("a second one next to it", B)
C = make(22); report("a third one (has to go behind the ballast again)", C)
