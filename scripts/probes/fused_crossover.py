"""Where does the one-pass kernel start to beat K-fwd + K-adj under one sync?  Wall clock per call on small / mid shapes
(GPU box).  Round 2: the launch no longer refills its hand-off slots from the host (two self-re-arming arrays) nor zeroes
its counters, so its fixed cost dropped; this sweep re-measures the `fused_pays()` crossover."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

shapes = ((512, 1024), (2048, 2048), (4096, 4096), (8192, 4096), (16384, 4096), (1024, 8192), (4096, 8192), (8192, 8192),
          (512, 16384), (2048, 16384), (64, 65536), (512, 65536), (32, 8192), (32, 32768), (32, 65536))
for m, n in shapes:
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    res = []
    for name, fn in (("one-pass", lambda: ctx.step(0.2)), ("pair", lambda: ctx.fwd_adj(0.2))):
        best = 1e9
        for rep in range(3):
            for _ in range(5):
                fn()
            t0 = time.perf_counter()
            for _ in range(100):
                fn()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
        res.append((name, best))
    print(f"m={m:6d} n={n:6d} ({m * n / 2**20:7.1f} Mi)  " + "  ".join(f"{k} {v:7.1f} us" for k, v in res)
          + f"   fused_supported={ctx.fused_supported()}", flush=True)
    A.close()
