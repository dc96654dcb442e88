"""Per-launch time of the one-pass kernel (and of K-fwd / K-adj at 65536^2) across the shapes a build serves: best of three solves,
HIP events.  Run it once per library on ONE box for an A/B: FASTA_HIP_LIB=fasta_python_amd/libfasta_hip_narrow.so python scripts/probes/onepass_ab.py
(`make -C fasta_python_amd/csrc narrow` builds that library: 8-byte hand-off stores / loads instead of the 16-byte ones)."""
import os, sys, time, warnings
sys.path.insert(0, os.getcwd())
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
print("library:", os.path.basename(hip.LIB_PATH), flush=True)
for m, n, fused in ((32768, 131072, "auto"), (65536, 65536, "auto"), (65536, 65536, False), (8192, 8192, "auto"), (16384, 16384, "auto"), (4096, 4096, "auto")):
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, seed=1), seed_noise=2, sigma=0.01)
    steps = 30 if m * n > (1 << 30) else 300
    best = {}
    for rep in range(3):
        np.random.seed(3)
        s = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=steps + 5, tolerance=0.0, fused=fused)
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            s.setup()
            for _ in range(5): s.step()
            A.ctx.timing_reset(); A.ctx.timing_enable(True); A.ctx.sync()
            for _ in range(steps): s.step()
            A.ctx.sync(); A.ctx.timing_enable(False)
        for name, kid in (("fused", hip.K_FUSED), ("fwd", hip.K_FWD), ("adj", hip.K_ADJ)):
            ms, cnt = A.ctx.timing_get(kid)
            if cnt: best[name] = min(best.get(name, 1e9), ms / cnt)
    print(f"{m:6d} x {n:6d} fused={fused!s:5}: " + "  ".join(f"{k} {v * 1e3:9.2f} us" for k, v in best.items()), flush=True)
    A.close()
