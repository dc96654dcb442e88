"""How much does the speed of the row-strided kernels depend on WHICH device memory a 32 GiB matrix got -- with nothing being cleared in the background?
Seven 65536^2 matrices are allocated one after the other and all kept (224 GiB), each timed: one-pass step, K-fwd, K-adj, the read-only stream.
Then all are released, the device settles, and the same again (the allocator now hands out memory that has been used before).
Usage: python scripts/probes/placement_spread.py [rounds]    -> profiles/r06_placement.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

N = 65536
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hip.alloc_cache(False)
rng = np.random.RandomState(0)
b = rng.randn(N); x0 = rng.randn(N) * 0.01


def timed(ctx, fn, kid, reps):
    fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


for r in range(rounds):
    held = []
    for i in range(7):
        t0 = time.perf_counter()
        A = fa.DenseMatrixMap.synthetic(N, N, i, synthetic.lasso_scale(N, N))
        ctx = A.ctx
        ctx.set_loss_lsq(b); ctx.set_prox(hip.PROX_SHRINK, 0.02); ctx.set_vector(hip.VEC_X0, x0); ctx.init()
        ctx.sync(); t_alloc = time.perf_counter() - t0
        one = timed(ctx, lambda: ctx.step(0.2), hip.K_FUSED, 8)
        fwd = timed(ctx, lambda: ctx.fwd(0.2), hip.K_FWD, 3)
        adj = timed(ctx, lambda: ctx.adj(0.2), hip.K_ADJ, 3)
        ms, nbytes = ctx.stream_read_ms(3)
        print(f"[round {r}] matrix {i} ({(i + 1) * 32} GiB resident; allocation + generation {t_alloc:.2f} s): one-pass {one:.3f} ms   K-fwd {fwd:.3f}   K-adj {adj:.3f}   "
              f"stream {nbytes / ms / 1e6:.0f} GB/s", flush=True)
        held.append(A)
    # the first one again, now that the device is full
    ctx = held[0].ctx
    print(f"[round {r}] matrix 0 again: one-pass {timed(ctx, lambda: ctx.step(0.2), hip.K_FUSED, 8):.3f} ms", flush=True)
    for A in held:
        A.close()
    time.sleep(9.0)
