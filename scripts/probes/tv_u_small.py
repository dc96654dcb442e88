"""Rows per TRIP (FH_TUNE_TV_U) and trip buffers (FH_TUNE_TV_PIPE) of the z-free stencil sweep on small images, at the chunk height of round 6
(8 rows up to 512^2 / 1024^2 plain): a chunk of 8 + 4 rows is 6 dependent two-row trips by default.  Usage: python scripts/probes/tv_u_small.py [side...]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip
from fasta_python_amd.examples.tv_denoising import checkerboard

for side in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    np.random.seed(7)
    M = checkerboard(side, side, max(1, side // 32)) + 0.1 * np.random.standard_normal((side, side))
    A = fa.GradDivMap(M.shape)
    loss, reg, x0 = fa.LeastSquares(M / 0.1), fa.TVDualBall(), np.zeros(M.shape + (2,))
    try:
        for acc in (False, True):
            for rows in (0, 12, 16):
                for u, pipe in ((0, 0), (2, 1), (4, 1), (8, 1), (2, 3), (4, 3)):
                    A.ctx.set_tuning(hip.TUNE_TV_ROWS, rows); A.ctx.set_tuning(hip.TUNE_TV_U, u); A.ctx.set_tuning(hip.TUNE_TV_PIPE, pipe)
                    best, sweep = 0.0, None
                    for rep in range(3):
                        timed = rep == 2
                        np.random.seed(3)
                        solver = fa.FBSolver(A, loss, reg, x0, adaptive=not acc, accelerate=acc, verbose=False, max_iters=460, tolerance=0.0)
                        with warnings.catch_warnings(), np.errstate(all="ignore"):
                            warnings.simplefilter("ignore")
                            solver.setup(); solver.advance(60)
                            A.ctx.timing_reset(); A.ctx.timing_enable(timed)
                            A.ctx.sync(); t0 = time.perf_counter()
                            solver.advance(400)
                            A.ctx.sync(); el = time.perf_counter() - t0
                            A.ctx.timing_enable(False)
                        if timed:
                            ms, cnt = A.ctx.timing_get(hip.K_FUSED); sweep = ms / cnt * 1e3
                        else:
                            best = max(best, 400 / el)
                    print(f"{side:5d}^2 {'FISTA   ' if acc else 'adaptive'} rows/wg {rows:3d} U {u} pipe {pipe}: {best:8.0f} it/s  {1e6 / best:7.2f} us/it   sweep {sweep:6.2f} us", flush=True)
    finally:
        A.close()
