"""What the clipping-level search (csrc/fh_prox.h) costs next to the step it precedes, for the two sort-free prox kinds
(fasta/proximal.py:12-41; examples/lasso.py:45), across matrix sizes (GPU box).  Wall clock per iteration of a full FBSolver.step(),
HIP-event time of the level search and of the step launch, and the same solve with the soft-threshold prox (no level search) for reference.
Usage: python scripts/probes/level_search_cost.py [sizes...]   e.g. 4096 8192 16384 65536"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536]
for n in sizes:
    m = n
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
    ctx = A.ctx
    x_true = synthetic.sparse_signal(n, seed=1)
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01)
    loss = fa.LeastSquares(b)
    for name, reg in (("shrink", fa.Shrink(0.02)), ("l1ball", fa.L1Ball(0.8 * float(np.abs(x_true).sum()))), ("linf", fa.LinfProx(0.02))):
        steps, warm = (200, 20) if n <= 16384 else (40, 5)
        np.random.seed(3)
        solver = fa.FBSolver(A, loss, reg, np.zeros(n), verbose=False, max_iters=steps + warm, tolerance=0.0)
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            solver.setup()
            for _ in range(warm):
                solver.step()
            ctx.timing_reset(); ctx.timing_enable(True); ctx.sync()
            bt0 = solver.total_backtracks
            t0 = time.perf_counter()
            for _ in range(steps):
                solver.step()
            ctx.sync()
            dt = time.perf_counter() - t0
            ctx.timing_enable(False)
        lv_ms, lv_n = ctx.timing_get(hip.K_LEVEL)
        st_ms, st_n = ctx.timing_get(hip.K_FUSED)
        fw_ms, fw_n = ctx.timing_get(hip.K_FWD)
        ad_ms, ad_n = ctx.timing_get(hip.K_ADJ)
        kern = st_ms + fw_ms + ad_ms
        print(f"n={n:6d} {name:7s} {steps / dt:9.1f} it/s  {dt / steps * 1e3:8.4f} ms/step  backtracks {solver.total_backtracks - bt0:3d} | "
              f"step launches {st_n + fw_n:4d} avg {kern / max(1, st_n + max(fw_n, ad_n)):8.4f} ms | level search {lv_n:4d} avg {lv_ms / max(1, lv_n) * 1e3:7.2f} us "
              f"= {100 * lv_ms / max(1e-9, kern + lv_ms):5.2f} % of kernel time", flush=True)
    A.close()
