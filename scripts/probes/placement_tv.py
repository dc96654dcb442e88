"""Does the speed of the TV sweep (k_tv_onepass, 8192^2) depend on where its buffers lie in device memory, as the one-pass dense kernel's did with the blocked
dealing (profiles/r06_placement.txt)?  The TV problem is set up afresh behind 0, 1, ... 7 ballast blocks of 32 GiB (all kept) and timed each time: sweep time
by HIP events over 60 iterations, adaptive and FISTA.   Usage: python scripts/probes/placement_tv.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
from fasta_python_amd.examples.tv_denoising import checkerboard

hip.alloc_cache(False)
side = 8192
np.random.seed(7)
M = checkerboard(side, side, side // 32)
M += 0.1 * np.random.standard_normal(M.shape)
mu = 0.1


def tv_ms(accelerate, tune=None):
    A = fa.GradDivMap(M.shape)
    for k, v in (tune or {}).items():
        A.ctx.set_tuning(k, v)
    try:
        np.random.seed(3)
        s = fa.FBSolver(A, fa.LeastSquares(M / mu), fa.TVDualBall(), np.zeros(M.shape + (2,)), adaptive=not accelerate, accelerate=accelerate, verbose=False, max_iters=80, tolerance=0.0)
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            s.setup(); s.advance(20)
            A.ctx.sync(); A.ctx.timing_reset(); A.ctx.timing_enable(True)
            t0 = time.perf_counter(); s.advance(60); A.ctx.sync(); wall = (time.perf_counter() - t0) / 60 * 1e3
            A.ctx.timing_enable(False)
        ms, cnt = A.ctx.timing_get(hip.K_FUSED)
        return ms / cnt, wall
    finally:
        A.close()


tunes = [None] + [{int(a.split("=")[0]): int(a.split("=")[1]) for a in arg.split(",")} for arg in sys.argv[1:]]
ballast = []
for k in range(8):
    out = []
    for tune in tunes:
        a = tv_ms(False, tune); f = tv_ms(True, tune)
        out.append(f"{'default' if not tune else tune}: adaptive sweep {a[0]:.4f} ms ({a[1]:.3f} per iteration)  FISTA sweep {f[0]:.4f} ms")
    print(f"{k * 32:3d} GiB of ballast resident: " + " | ".join(out), flush=True)
    if k < 7:
        ballast.append(fa.DenseMatrixMap.synthetic(65536, 65536, k, synthetic.lasso_scale(65536, 65536)))
        ballast[-1].ctx.sync()
