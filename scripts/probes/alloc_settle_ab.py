"""Does a 32 GiB matrix allocated RIGHT BEHIND a 128 GiB free run slow for its lifetime -- and does the library's wait before the allocation
(fh_alloc_settle: 35 ms per GiB freed) cure it?  Round 5 saw it once (profiles/r05_free_aftermath.txt, cycle [0]: 5.31-5.35 ms per step for 1200 steps
against 4.86-4.91; cycle [1] of the same process: nothing).  A throughput A/B, nothing has to "happen again":
  `cycles` x { wait ON:  128 GiB (config 5's matrix as 8 blocks on this GPU) generated and freed -> 32 GiB allocated at once (the library sleeps first) -> 300 steps
               wait OFF: the same, the library does not sleep
               control:  the 32 GiB matrix allocated BEFORE the 128 GiB one is generated and freed -> 300 steps right after the free, then 300 more 5 s later }
interleaved, one process.  -> profiles/r06_alloc_settle.txt.   Usage: python scripts/probes/alloc_settle_ab.py [cycles]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

N = 65536
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def steps_ms(A, b, samples=3, per=100):
    np.random.seed(3)
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
print(f"fresh process, 32 GiB, ms per step (3 x 100 steps): {fmt(steps_ms(A, b))}", flush=True)
A.close()
time.sleep(3.0)
table = {"on": [], "off": [], "control": []}
for c in range(cycles):
    for mode in ("off", "on", "control") if c % 2 == 0 else ("on", "off", "control"):
        hip.alloc_settle(mode == "on")
        if mode == "control":
            A = matrix()
            big_alloc_and_free()
            first = steps_ms(A, b)
            time.sleep(5.0)
            later = steps_ms(A, b, samples=1)
            print(f"[{c}] control: 32 GiB allocated BEFORE the 128 GiB alloc + free: right after the free {fmt(first)} | 5 s later {fmt(later)}", flush=True)
            table["control"].append(first + later)
        else:
            big_alloc_and_free()
            w0 = hip.alloc_settle_waited()
            t0 = time.perf_counter()
            A = matrix()
            alloc_s = time.perf_counter() - t0
            ms = steps_ms(A, b)
            print(f"[{c}] wait {mode:3s}: 32 GiB allocated right behind the 128 GiB free (allocation + generation {alloc_s:.2f} s, of which the library waited "
                  f"{hip.alloc_settle_waited() - w0:.2f} s): {fmt(ms)}", flush=True)
            table[mode].append(ms)
        A.close()
        time.sleep(6.0)          # (every case starts from a device whose earlier frees have been cleared)
# remedy 1 (kept-block re-use): a 32 GiB matrix given up and a 32 GiB matrix allocated AT ONCE -- the kept block (no wait) against a fresh
# allocation with neither remedy
for c in range(3):
    for mode in ("reuse", "neither"):
        hip.alloc_cache(mode == "reuse"); hip.alloc_settle(False)
        A = matrix(); A.ctx.sync(); A.close()
        h0 = hip.alloc_cache_hits()
        A = matrix()
        ms = steps_ms(A, b)
        print(f"[{c}] 32 GiB right behind the release of a 32 GiB matrix, {mode:7s} (kept-block hits {hip.alloc_cache_hits() - h0}): {fmt(ms)}", flush=True)
        table.setdefault(mode, []).append(ms)
        A.close(); hip.release_cached()
        time.sleep(3.0)
hip.alloc_settle(True); hip.alloc_cache(True)
print("\nsummary, ms per step (mean of the 3 x 100-step samples of each cycle):")
for mode in ("off", "on", "control", "reuse", "neither"):
    means = [float(np.mean(v[:3])) for v in table[mode]]
    print(f"  {mode:8s}: " + " ".join(f"{m:.3f}" for m in means) + f"   | worst {max(means):.3f}, best {min(means):.3f}")
slow = sum(1 for v in table["off"] if np.mean(v) > 1.03 * min(np.mean(w) for w in table["on"] + table["off"]))
print(f"un-waited cycles more than 3 % slower than the best cycle: {slow} of {len(table['off'])}")
