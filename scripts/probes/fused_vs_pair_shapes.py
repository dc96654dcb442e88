"""One-pass kernel vs K-fwd + K-adj under one sync for short/fat and small shapes (wall clock per call, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

shapes_tall = ((1048576, 1024), (262144, 2048), (262144, 4096), (65536, 4096), (16384, 4096), (4096, 4096), (131072, 8192), (16384, 8192), (65536, 12000))
for m, n in (shapes_tall if len(sys.argv) > 1 and sys.argv[1] == "tall" else ((64, 65536), (256, 65536), (1024, 65536), (4096, 65536), (256, 32768), (1024, 32768), (4096, 32768),
             (512, 16384), (2048, 16384), (8192, 16384), (16384, 16384))):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    res = []
    for name, fn in (("one-pass", lambda: ctx.step(0.2)), ("pair", lambda: ctx.fwd_adj(0.2))):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        res.append((name, (time.perf_counter() - t0) / 50 * 1e6))
    print(f"m={m:6d} n={n:6d}  " + "  ".join(f"{k} {v:8.1f} us" for k, v in res), flush=True)
    A.close()
