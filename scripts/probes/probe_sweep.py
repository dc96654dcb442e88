"""Sweep the launch shape of the streaming-read probe (fh_stream_read_ms) next to the product kernels' own rates.
Usage: python scripts/probes/probe_sweep.py [rows] [cols]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
c = A.ctx
for grid in (128, 256, 384, 512, 768, 1024, 2048):
    c.set_tuning(hip.TUNE_FWD_GRID_CAP, grid)
    best = min(c.stream_read_ms(5)[0] for _ in range(3))
    ms, nbytes = c.stream_read_ms(5)
    print(f"probe grid {grid:5d}: {nbytes / min(best, ms) / 1e6:8.1f} GB/s  ({min(best, ms):.3f} ms)")
c.set_tuning(hip.TUNE_FWD_GRID_CAP, 0)
# the product kernels on the same matrix
b = np.zeros(m)
c.set_loss_lsq(b); c.set_prox(hip.PROX_SHRINK, 0.02); c.set_vector(hip.VEC_X0, np.zeros(n)); c.init()
c.timing_enable(True)
for name, fn, kid, by in (("K-fwd", lambda: c.fwd(0.1), hip.K_FWD, (m * n + 4 * n + 2 * m) * 8),
                          ("K-adj", lambda: c.adj(0.1), hip.K_ADJ, (m * n + 2 * m + 5 * n) * 8),
                          ("K-fused", lambda: c.step(0.1), hip.K_FUSED, (m * n + 3 * m + 7 * n) * 8)):
    fn(); c.timing_reset()
    for _ in range(5):
        fn()
    ms, cnt = c.timing_get(kid)
    print(f"{name:8s}: {by / (ms / cnt) / 1e6:8.1f} GB/s  ({ms / cnt:.3f} ms)")
A.close()
