"""Fixed cost and per-row cost of the one-pass kernel: HIP-event kernel time vs m at fixed n (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

for n in (4096, 8192, 16384, 32768, 65536):
    pts = []
    for m in (32, 256, 1024, 4096, 8192, 16384):
        if m * n > 2 ** 31:
            continue
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
        c = A.ctx
        rng = np.random.RandomState(0)
        c.set_loss_lsq(rng.randn(m)); c.set_prox(hip.PROX_SHRINK, 0.02)
        c.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
        c.init()
        for _ in range(5):
            c.step(0.2)
        c.timing_reset(); c.timing_enable(True)
        for _ in range(30):
            c.step(0.2)
        c.timing_enable(False)
        ms, cnt = c.timing_get(hip.K_FUSED)
        pts.append((m, ms / cnt * 1e3))
        A.close()
    ms_ = np.array([p[0] for p in pts], float); us = np.array([p[1] for p in pts])
    slope, icpt = np.polyfit(ms_[-3:], us[-3:], 1)
    print(f"n={n:6d}: " + "  ".join(f"m={m}: {t:7.1f} us" for m, t in pts) + f"   | fit on the 3 largest: {icpt:6.1f} us + {slope * 1e3:7.2f} ns/row"
          f"  (stream rate {n * 8 / slope / 1e3:6.0f} GB/s)", flush=True)
