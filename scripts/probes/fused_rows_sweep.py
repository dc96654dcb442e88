"""One-pass kernel time vs number of rows at n = 65536 (GPU box)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
n = 65536
for m in (64, 4096, 16384, 32768, 65536):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    out = []
    for v in (2, 34, 10):
        ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
        ctx.step(0.2)
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(6):
            ctx.step(0.2)
        ctx.timing_enable(False)
        ms, cnt = ctx.timing_get(hip.K_FUSED)
        out.append(f"v{v}: {ms / cnt * 1e3:8.1f} us ({m * n * 8 / (ms / cnt) / 1e6:5.0f} GB/s)")
    print(f"n={n} m={m:6d}  " + "  ".join(out), flush=True)
    A.close()
