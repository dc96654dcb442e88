"""After a large hipFree: does the device clear the freed memory in the background, at the expense of running kernels?  (GPU box)
A 32 GiB matrix stays resident; 96 GiB more are allocated, touched and freed; the read-only stream rate over the RESIDENT matrix is sampled before and for
a while after the free.  (profiles/r05_soak.txt: a 1500-iteration run right after config 5's 128 GiB were freed ran 12 % slow from start to end.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

def rate(A, reps=3):
    ms, nbytes = A.ctx.stream_read_ms(reps)
    return nbytes / ms / 1e6

A = fa.DenseMatrixMap.synthetic(65536, 65536, 0, synthetic.lasso_scale(65536, 65536))
print("resident matrix, before anything else: " + " ".join(f"{rate(A):6.0f}" for _ in range(5)) + " GB/s", flush=True)
for cycle in range(2):
    big = [fa.DenseMatrixMap.synthetic(32768, 65536, 0, 1.0) for _ in range(6)]     # 6 x 16 GiB, generated (= touched) in HBM
    print(f"[{cycle}] with 96 GiB more resident:        " + " ".join(f"{rate(A):6.0f}" for _ in range(5)) + " GB/s", flush=True)
    t0 = time.perf_counter()
    for b in big: b.close()
    print(f"[{cycle}] the six frees took {time.perf_counter() - t0:.3f} s", flush=True)
    t0 = time.perf_counter()
    samples = []
    while time.perf_counter() - t0 < 12.0:
        samples.append((time.perf_counter() - t0, rate(A, 2)))
        time.sleep(0.25)
    print(f"[{cycle}] after the frees (s: GB/s): " + " ".join(f"{t:4.1f}:{r:5.0f}" for t, r in samples), flush=True)
    B = fa.DenseMatrixMap.synthetic(65536, 65536, 0, synthetic.lasso_scale(65536, 65536))     # a fresh 32 GiB into the hole
    print(f"[{cycle}] fresh 32 GiB allocated into the hole: " + " ".join(f"{rate(B):6.0f}" for _ in range(5)) + " GB/s ; the resident one: " + " ".join(f"{rate(A):6.0f}" for _ in range(3)), flush=True)
    B.close()
A.close()
