"""Per-trip cost of the one-pass kernel: time vs rows per team at fixed n (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

variants = [int(v) for v in sys.argv[1:]] or [2]
for n in (8192, 32768):
    for m in (32, 2048, 8192, 32768):
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
        ctx = A.ctx
        rng = np.random.RandomState(0)
        ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
        ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
        ctx.init()
        out = []
        for v in variants:
            ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
            ctx.step(0.2)
            ctx.timing_reset(); ctx.timing_enable(True)
            for _ in range(8):
                ctx.step(0.2)
            ctx.timing_enable(False)
            ms, cnt = ctx.timing_get(hip.K_FUSED)
            out.append(f"v{v}: {ms / cnt * 1e3:8.1f} us")
        print(f"n={n:6d} m={m:6d} rows/team={(m + 31) // 32:5d}  " + "  ".join(out), flush=True)
        A.close()
