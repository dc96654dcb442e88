"""Who drives the loop: iterations/s of the bench's LASSO problem with driver="python" (FBSolver.step between all launches, rounds 1-5), the
library's host-side loop (fh_iterate, the default) and the loop on the device (fh_run, where it has a kernel; FH_TUNE_RUN_MAX_N = 7168 so that
every width it CAN take is measured).  Usage: python scripts/probes/driver_cost.py [m n]...   -> profiles/r06_device_loop.txt"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

sizes = [(512, 1024), (2048, 2048), (4096, 4096), (5000, 5000), (5632, 5632), (6144, 6144), (6656, 6656), (7168, 7168), (8192, 8192), (16384, 16384)]
if len(sys.argv) > 2:
    sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
iters = 1024
tuning = {hip.TUNE_RUN_MAX_N: 7168}
if os.environ.get("FH_RUN_CHAIN") == "1":          # the chained form of fh_run outside the persistent launch's window (opt-in)
    tuning[hip.TUNE_RUN_CHAIN] = 1
    tuning.pop(hip.TUNE_RUN_MAX_N)
    print("# FH_TUNE_RUN_CHAIN = 1, default window of the persistent launch: the device column is the chain of one-pass launches outside it")
if os.environ.get("FH_NT_LOADS") in ("0", "1"):        # A/B: the device loop (and K-fwd / K-adj) with plain / non-temporal loads of A (the one-pass kernel has no plain form)
    tuning[hip.TUNE_NT_LOADS] = int(os.environ["FH_NT_LOADS"])
    print(f"# FH_TUNE_NT_LOADS = {tuning[hip.TUNE_NT_LOADS]}")
if os.environ.get("FH_SEQ_POLL") == "0":          # A/B: wait for every launch with hipStreamSynchronize (rounds 1-5) instead of its sequence number
    tuning[hip.TUNE_SEQ_POLL] = 0
    print("# FH_TUNE_SEQ_POLL = 0: hipStreamSynchronize after every launch")
for m, n in sizes:
    A = fa.DenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n), tuning=tuning)
    try:
        x_true = synthetic.sparse_signal(n, seed=1)
        b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
        loss, reg = fa.LeastSquares(b), fa.Shrink(0.02)
        cols = []
        for name, kw in (("python", dict(driver="python")), ("library", dict(driver="library")), ("device K=64", dict(driver="device", device_iters=64))):
            best, dev, lib = 0.0, 0, 0
            for _ in range(3):
                np.random.seed(3)
                solver = fa.FBSolver(A, loss, reg, np.zeros(n), verbose=False, max_iters=iters, tolerance=0.0, **kw)
                with warnings.catch_warnings(), np.errstate(all="ignore"):
                    warnings.simplefilter("ignore")
                    solver.setup()
                    A.ctx.sync()
                    t0 = time.perf_counter()
                    solver.run()
                    A.ctx.sync()
                    best = max(best, solver.i / (time.perf_counter() - t0))
                dev, lib = solver.device_steps, solver.library_steps
            cols.append(f"{name:12s} {best:8.0f} it/s ({1e6 / best:6.1f} us; device {dev:4d}, library {lib:4d})")
        print(f"{m:6d} x {n:6d}: " + " | ".join(cols), flush=True)
    finally:
        A.close()
