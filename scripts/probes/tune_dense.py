"""Sweep the dense-kernel tunables on the GPU box and print avg launch ms / achieved GB/s per variant.
    python scripts/probes/tune_dense.py [--m 65536 --n 65536]"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=65536)
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--pads", default="0,32")
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
m, n = args.m, args.n
mat_bytes = m * n * 8


def time_kernel(ctx, kid, fn, reps):
    fn()                     # warm
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


for pad in [int(p) for p in args.pads.split(",")]:
    A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), tuning={hip.TUNE_LD_PAD: pad})
    ctx = A.ctx
    ms, nbytes = ctx.stream_read_ms(3)
    print(f"# ld_pad={pad}: stream-read ceiling {nbytes / ms / 1e6:.0f} GB/s ({ms:.3f} ms per pass)", flush=True)
    rng = np.random.RandomState(0)
    ctx.set_loss_lsq(rng.randn(m))
    ctx.set_prox(hip.PROX_SHRINK, 0.02)
    ctx.set_vector(hip.VEC_X0, rng.randn(n) * 0.01)
    ctx.init()
    tau = 0.2
    fwd_grid = [(r, nt, cap) for r in (4, 8, 16) for nt in (1, 0) for cap in ((0, 2048, 1024) if not args.quick else (0,))]
    for r, nt, cap in fwd_grid:
        ctx.set_tuning(hip.TUNE_FWD_ROWS, r); ctx.set_tuning(hip.TUNE_NT_LOADS, nt); ctx.set_tuning(hip.TUNE_FWD_GRID_CAP, cap)
        t = time_kernel(ctx, hip.K_FWD, lambda: ctx.fwd(tau), args.reps)
        print(f"fwd pad={pad:3d} rows={r:2d} nt={nt} cap={cap:5d}: {t:8.3f} ms  {mat_bytes / t / 1e6:7.0f} GB/s", flush=True)
    ctx.set_tuning(hip.TUNE_FWD_ROWS, 0); ctx.set_tuning(hip.TUNE_FWD_GRID_CAP, 0)
    slabs = (128, 256, 512, 1024, 2048) if not args.quick else (512, 1024)
    for cpt, slab, nt in itertools.product((1, 2, 4), slabs, (1, 0)):
        ctx.set_tuning(hip.TUNE_ADJ_CPT, cpt); ctx.set_tuning(hip.TUNE_ADJ_SLAB_ROWS, slab); ctx.set_tuning(hip.TUNE_NT_LOADS, nt)
        t = time_kernel(ctx, hip.K_ADJ, lambda: ctx.adj(tau), args.reps)
        print(f"adj pad={pad:3d} cpt={cpt} slab={slab:4d} nt={nt}: {t:8.3f} ms  {mat_bytes / t / 1e6:7.0f} GB/s", flush=True)
    A.close()
