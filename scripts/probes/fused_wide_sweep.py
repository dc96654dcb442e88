"""One-pass variants on wide matrices, n > 65536 (GPU box)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
for n, m in ((262144, 16384), (200000, 16384), (150000, 16384), (131072, 32768), (120000, 32768), (110000, 32768), (100000, 32768), (90000, 32768), (80000, 32768), (70000, 32768), (66000, 32768), (65536, 32768)):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    out = []
    for v in (2, 18):       # 2 = default (n > 65536: x slice in LDS, posting one row ahead), 18 = round-1 shape (16 pieces in registers, 3 buffers, exchange in line)
        ctx.set_tuning(hip.TUNE_FUSED_VARIANT, v)
        ctx.step(0.2)
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(5):
            ctx.step(0.2)
        ctx.timing_enable(False)
        ms, cnt = ctx.timing_get(hip.K_FUSED)
        out.append(f"v{v}: {ms / cnt:6.3f} ms ({m * n * 8 / (ms / cnt) / 1e6:5.0f} GB/s)")
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(3):
        ctx.fwd(0.2); ctx.adj(0.2)
    ctx.timing_enable(False)
    tf = ctx.timing_get(hip.K_FWD); ta = ctx.timing_get(hip.K_ADJ)
    print(f"n={n} m={m}: " + "  ".join(out) + f"  fwd {tf[0] / tf[1]:.3f} adj {ta[0] / ta[1]:.3f}", flush=True)
    A.close()
