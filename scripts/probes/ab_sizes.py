"""Same-box A/B of two builds of libfasta_hip.so over matrix sizes: one-pass kernel time (HIP events, plain and accelerated steps), each
library in its own process, interleaved:   python scripts/probes/ab_sizes.py <old.so> <new.so> [sizes "m,n m,n ..."]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import os, sys, ctypes
sys.path.insert(0, %r)
import numpy as np
from fasta_python_amd import hip
lib, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
probe = ctypes.CDLL(lib)
for name in list(hip.SIGNATURES):
    if not hasattr(probe, name):
        del hip.SIGNATURES[name]
hip.load_library(lib)
import fasta_python_amd as fa
from fasta_python_amd import synthetic
A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n))
ctx = A.ctx
rng = np.random.RandomState(0)
ctx.set_loss_lsq(rng.randn(m)); ctx.set_prox(hip.PROX_SHRINK, 0.02)
ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
ctx.init()
def timed(fn, reps=30):
    for _ in range(5): fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps): fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(hip.K_FUSED)
    return ms / cnt
best = min(timed(lambda: ctx.step(0.2)) for _ in range(3))
s = ctx.step(0.2)
ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01); ctx.init()
def acc():
    ctx.step_accel(0.2, 0.3, True); ctx.commit(False)
besta = min(timed(acc) for _ in range(3))
print("%%.5f %%.5f %%.17g" %% (best, besta, s[hip.S_FSQ]))
''' % ROOT
old, new = sys.argv[1], sys.argv[2]
sizes = sys.argv[3].split() if len(sys.argv) > 3 else ["4096,4096", "8192,8192", "16384,16384", "32768,32768", "8192,65536", "65536,65536"]
print("# one-pass kernel, HIP events, best of 3 x 30 launches, ms (plain | accelerated); algorithmic GB/s of the plain step")
for sz in sizes:
    m, n = (int(v) for v in sz.split(","))
    rows = {}
    for tag, lib in (("old", old), ("new", new), ("old", old), ("new", new)):
        out = subprocess.run([sys.executable, "-c", WORKER, lib, str(m), str(n)], capture_output=True, text=True)
        if out.returncode:
            print(out.stderr[-1500:]); sys.exit(1)
        a, b, f = out.stdout.split()[-3:]
        rows.setdefault(tag, []).append((float(a), float(b), float(f)))
    by = m * n * 8 + (3 * m + 7 * n) * 8
    o = min(r[0] for r in rows["old"]); nw = min(r[0] for r in rows["new"])
    oa = min(r[1] for r in rows["old"]); na = min(r[1] for r in rows["new"])
    print(f"{m:6d} x {n:6d}  old {o:.4f} | {oa:.4f}   new {nw:.4f} | {na:.4f}   {by / o / 1e6:6.0f} -> {by / nw / 1e6:6.0f} GB/s  ({(nw / o - 1) * 100:+.1f} %, accel {(na / oa - 1) * 100:+.1f} %)"
          f"   frac {by / nw / 1e6 / 8000:.3f}   FSQ rel diff {abs(rows['new'][0][2] - rows['old'][0][2]) / abs(rows['old'][0][2]):.1e}", flush=True)
