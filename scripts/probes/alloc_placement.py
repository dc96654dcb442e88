"""Does WHERE an allocation lands change what a kernel gets out of it?  (GPU box)  The bench's wide-row sub-result and the 128 GiB config-5 run come out
5-8 % slow on some leases while the headline on the same lease does not (profiles/r04_bench_boxes.txt, r05_bench_default.json).  This script allocates
the same matrices in different orders / next to different neighbours inside ONE process and times a read-only stream and the one-pass step on each.
Usage: python scripts/probes/alloc_placement.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic


def make(m, n):
    return fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))


def measure(A, label):
    m, n = A.shape
    c = A.ctx
    ms, nbytes = c.stream_read_ms(5)
    b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, seed=1), seed_noise=2, sigma=0.01)
    np.random.seed(3)
    s = fa.FBSolver(A, fa.LeastSquares(b), fa.Shrink(0.02), np.zeros(n), verbose=False, max_iters=40, tolerance=0.0)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        s.setup()
        for _ in range(4): s.step()
        c.timing_reset(); c.timing_enable(True); c.sync()
        for _ in range(20): s.step()
        c.sync(); c.timing_enable(False)
    k_ms, cnt = c.timing_get(hip.K_FUSED)
    print(f"{label:58s} {m:6d} x {n:6d}: stream {nbytes / ms / 1e6:7.0f} GB/s | one-pass step {k_ms / cnt:7.4f} ms", flush=True)


sq, wide = (65536, 65536), (32768, 131072)
for cycle in range(2):
    a1 = make(*sq);    measure(a1, f"[{cycle}] square, first allocation of the cycle")
    a2 = make(*wide);  measure(a2, f"[{cycle}] wide, allocated next to the square one")
    measure(a1, f"[{cycle}] square again (wide one resident)")
    a1.close()
    a3 = make(*wide);  measure(a3, f"[{cycle}] second wide, into the hole the square one left")
    a2.close()
    measure(a3, f"[{cycle}] second wide again (first wide freed)")
    a4 = make(*sq);    measure(a4, f"[{cycle}] square, into the hole the first wide left")
    a3.close(); a4.close()
    big = [make(32768, 65536) for _ in range(6)]          # 6 x 16 GiB
    for i in (0, 5): measure(big[i], f"[{cycle}] block {i} of six 16-GiB blocks")
    for b_ in big: b_.close()
