"""A/B of one-pass kernel shapes for float32 storage of A (GPU box): per-launch time of the one-pass kernel under FH_TUNE_FUSED_VARIANT bits,
the candidates measured in turns on matrices that stay resident (box clocks drift by a few per cent within a call).
Usage: python scripts/probes/f32_shapes.py [variant ...]   (default: 2 10 18)"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

variants = [int(v) for v in sys.argv[1:]] or [2, 10, 18]      # bit 2 (members of a team on one XCD) is the default
for m, n in ((8192, 8192), (16384, 16384), (32768, 32768), (65536, 65536)):
    maps = {}
    for slot, var in enumerate(variants):          # (a variant may be listed twice: two allocations of the same matrix)
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), storage="f32", tuning={hip.TUNE_FUSED_VARIANT: var})
        maps[slot] = (A, synthetic.lasso_observation(A, synthetic.sparse_signal(n, seed=1), seed_noise=2, sigma=0.01))
    steps = 40 if m * n > (1 << 30) else 200
    runs = {slot: [] for slot in maps}
    for rep in range(4):
        for slot in maps:
            A, b = maps[slot]
            np.random.seed(3)
            s = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=steps + 5, tolerance=0.0)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                s.setup()
                for _ in range(5): s.step()
                A.ctx.timing_reset(); A.ctx.timing_enable(True); A.ctx.sync()
                for _ in range(steps): s.step()
                A.ctx.sync(); A.ctx.timing_enable(False)
            ms, cnt = A.ctx.timing_get(hip.K_FUSED)
            runs[slot].append(ms / cnt * 1e3 if cnt else float("nan"))
    by = m * n * 4 + (3 * m + 7 * n) * 8
    for slot, var in enumerate(variants):
        best = min(runs[slot])
        print(f"{m:6d} x {n:6d} f32 storage, variant {var:3d}: "
              f"one-pass launch us {' '.join('%8.1f' % r for r in runs[slot])}   best {by / best / 1e3:7.0f} GB/s  frac {by / best / 1e3 / 8000:.3f}", flush=True)
    for A, _ in maps.values(): A.close()
