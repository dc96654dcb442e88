"""Round 4: what ONE host thread pays per iteration to drive the in-process 8-row-block form, on one GPU.

    python scripts/probes/inproc_issue.py [blocks=8] [rows_per_block=8192] [n=65536] [steps=40]

For `ShardedDenseMatrixMap.synthetic(blocks * rows, n, devices=[0] * blocks)` and, next to it, the single-launch context on the same
matrix: wall time per iteration; (a) HOST issue time per iteration = entry of fh_step to the start of its one synchronisation
(FH_K_HOST_ISSUE); (b) DEVICE time of the blocks' one-pass launches, the sum over the blocks (k_sum_shards) and the n-side epilogues
(HIP events; FH_K_FUSED / FH_K_COMM / FH_K_AUX summed over the blocks); with the HIP-event timers off, the wall time alone.
On 8 GPUs the 8 launches and the 8 epilogues run side by side and only the host issue time stays serial."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
m = blocks * rows


def run(A, label, timers):
    ctx = A.ctx
    x_true = synthetic.sparse_signal(n, seed=1)
    b = synthetic.lasso_observation(A, x_true, seed_noise=2, sigma=0.01, row0=A.rows[0], m_total=m)
    np.random.seed(3)
    solver = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=steps + 5, tolerance=0.0)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        solver.setup()
        for _ in range(5):
            solver.step()
        ctx.timing_reset()
        ctx.timing_enable(timers)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.step()
        ctx.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
        ctx.timing_enable(False)
    out = f"{label:62s} wall {wall:7.4f} ms/iteration"
    if timers:
        t = {k: ctx.timing_get(v) for k, v in (("fused", hip.K_FUSED), ("comm", hip.K_COMM), ("aux", hip.K_AUX), ("host", hip.K_HOST_ISSUE))}
        per = {k: (ms / steps) for k, (ms, cnt) in t.items()}
        out += (f" | host issue {per['host']:.4f} ms | device: one-pass launches {per['fused']:.4f} ms ({t['fused'][1] // steps} per iteration), "
                f"sum over blocks {per['comm']:.4f} ms, epilogues {per['aux']:.4f} ms ({t['aux'][1] // steps} per iteration)"
                f" | wall - device = {wall - per['fused'] - per['comm'] - per['aux']:.4f} ms")
    print(out, flush=True)
    return wall


print(f"# {m} x {n} float64, {steps} timed iterations after 5 warm-up ones")
A1 = fa.DenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n), device=0)
try:
    run(A1, "one context, one launch per iteration, timers on", True)
    run(A1, "one context, one launch per iteration, timers off", False)
finally:
    A1.close()
A8 = fa.ShardedDenseMatrixMap.synthetic(m, n, seed=0, scale=synthetic.lasso_scale(m, n), devices=[0] * blocks)
try:
    run(A8, f"{blocks} row blocks of {rows} rows in one process, timers on", True)
    run(A8, f"{blocks} row blocks of {rows} rows in one process, timers off", False)
    run(A8, f"{blocks} row blocks of {rows} rows in one process, timers on (again)", True)
finally:
    A8.close()
