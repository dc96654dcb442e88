"""Sweep the stencil-kernel tunables (rows per workgroup, rows in flight) and separate streaming from prox cost.
    python scripts/probes/tune_tv.py [side]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import fasta_python_amd as fa
from fasta_python_amd import hip

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = side * side
rng = np.random.RandomState(0)
A = fa.GradDivMap((side, side))
ctx = A.ctx
ctx.set_loss_lsq(rng.standard_normal(P))
ctx.set_vector(hip.VEC_X0, rng.standard_normal(2 * P) * 0.7)


def timed(kid, fn, reps=5):
    fn()
    ctx.timing_reset()
    ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(kid)
    return ms / cnt


import itertools
for nt, u, rows in itertools.product((1, 0), (2, 4, 8), (16, 32, 64, 128)):
    if True:
        ctx.set_tuning(hip.TUNE_TV_NT, nt)
        ctx.set_tuning(hip.TUNE_TV_U, u)
        ctx.set_tuning(hip.TUNE_TV_ROWS, rows)
        out = []
        for kind, name in ((hip.PROX_TVBALL, "ball"), (hip.PROX_IDENTITY, "ident")):
            ctx.set_prox(kind)
            ctx.init()
            tf = timed(hip.K_FWD, lambda: ctx.fwd(0.1))
            ta = timed(hip.K_ADJ, lambda: ctx.adj(0.1))
            out.append(f"{name}: fwd {tf:6.3f} ms {64 * P / tf / 1e6:6.0f} GB/s | adj {ta:6.3f} ms "
                       f"{80 * P / ta / 1e6:6.0f} GB/s(80P) {72 * P / ta / 1e6:6.0f}(72P)")
        print(f"nt={nt} U={u} rows={rows:3d}  " + "  ||  ".join(out), flush=True)
A.close()
