"""profiles/r06_alloc_settle.txt once more, now that the one-pass kernel deals its rows cyclically: is a 32 GiB matrix allocated RIGHT BEHIND a 128 GiB free still slow for
its lifetime?  Kept-block re-use OFF (every allocation is a fresh mapping), the library's wait before the allocation on / off, interleaved; and both dealings
on the un-waited allocation.   Usage: python scripts/probes/alloc_settle_cyclic.py [cycles]   -> profiles/r06_placement.txt"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

N = 65536
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hip.alloc_cache(False)


def steps_ms(A, b, samples=2, per=100, variant=None):
    np.random.seed(3)
    if variant is not None:
        A.ctx.set_tuning(hip.TUNE_FUSED_VARIANT, variant)
    solver = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(N), verbose=False, max_iters=10 + samples * per, tolerance=0.0)
    out = []
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        solver.setup(); solver.advance(10)
        for _ in range(samples):
            A.ctx.sync(); t0 = time.perf_counter()
            solver.advance(per)
            A.ctx.sync(); out.append((time.perf_counter() - t0) / per * 1e3)
    return out


def big_alloc_and_free():
    A5 = fa.ShardedDenseMatrixMap.synthetic(262144, N, seed=0, scale=synthetic.lasso_scale(262144, N), devices=[0] * 8)
    A5.ctx.sync()
    A5.close()


def matrix():
    return fa.DenseMatrixMap.synthetic(N, N, seed=0, scale=synthetic.lasso_scale(N, N))


fmt = lambda v: " ".join(f"{x:.3f}" for x in v)
A = matrix()
x_true = synthetic.sparse_signal(N, seed=1)
b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
print(f"fresh process, 32 GiB, ms per step (2 x 100 steps): cyclic {fmt(steps_ms(A, b, variant=34))}   blocked {fmt(steps_ms(A, b, variant=2))}", flush=True)
A.close()
time.sleep(3.0)
for c in range(cycles):
    for mode in ("off", "on") if c % 2 == 0 else ("on", "off"):
        hip.alloc_settle(mode == "on")
        big_alloc_and_free()
        w0 = hip.alloc_settle_waited()
        A = matrix()
        waited = hip.alloc_settle_waited() - w0
        cyc = steps_ms(A, b, variant=34); blk = steps_ms(A, b, variant=2); cyc2 = steps_ms(A, b, samples=1, variant=34)
        print(f"[{c}] wait {mode:3s} (the library waited {waited:.2f} s), 32 GiB right behind the 128 GiB free: cyclic {fmt(cyc)}   blocked {fmt(blk)}   cyclic again {fmt(cyc2)}", flush=True)
        A.close()
        time.sleep(6.0)
hip.alloc_settle(True); hip.alloc_cache(True)
