"""Sweep the z-free one-pass stencil sweep (k_tv_onepass) over its tunables on the GPU box and check every variant against the
round-2 form bit for bit:
    python scripts/probes/tune_tvz.py [side] [nt list] [pipe list] [u list] [rows list] [xcd list] [pad list] [ring list] [slots list]
(HIP-event time of fh_step / fh_step_accel; 40*P resp. 56*P algorithmic bytes)"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from fasta_python_amd import hip
if os.environ.get("FASTA_LIB"):                    # an experimental build of the library (A/B on one box)
    hip.load_library(os.environ["FASTA_LIB"])
import fasta_python_amd as fa


def ints(i, default):
    return tuple(int(v) for v in sys.argv[i].split(",")) if len(sys.argv) > i else default


side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nts, pipes, us, rowss = ints(2, (3, 2)), ints(3, (1, 3)), ints(4, (2, 4, 8)), ints(5, (128, 0))       # TV_NT 3 = plain stores, 2 = nt stores; rows 0 = auto
pads = ints(7, (0,))                                                                                # TV_LDS_PAD bytes (occupancy limiter)
slotss = ints(9, (0,))                                                                              # TV_SLOTS 0 = one workgroup per chunk, n = persistent, n workgroups per CU
rings = ints(8, (1,))                                                                               # TV_RING 1 = register trips, 2 / 3 = LDS-DMA ring slots
xcds = ints(6, (2,))                                                                                # TV_XCD 2 = plain blockIdx order, 1 = XCD by XCD
P = side * side
rng = np.random.RandomState(0)
A = fa.GradDivMap((side, side))
ctx = A.ctx
ctx.set_loss_lsq(rng.standard_normal(P))
ctx.set_prox(hip.PROX_TVBALL)
x0 = rng.standard_normal(2 * P) * 0.7


def timed(fn, reps=8):
    fn()
    ctx.timing_reset()
    ctx.timing_enable(True)
    for _ in range(reps):
        fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(hip.K_FUSED)
    return ms / cnt


def signature():
    """scalars + prox output of one plain step and two accelerated steps from the same start"""
    ctx.set_vector(hip.VEC_X0, x0)
    ctx.init()
    s = ctx.step(0.1)
    xp = ctx.get_vector(hip.VEC_XPROX, 2 * P)
    ctx.set_vector(hip.VEC_X0, x0)
    ctx.init()
    a1 = ctx.step_accel(0.1, 0.0, True)
    ctx.commit(False)
    a2 = ctx.step_accel(0.1, 0.3, False)
    xa = ctx.get_vector(hip.VEC_XPROX, 2 * P)
    return s, xp, a1, a2, xa


for key, v in ((hip.TUNE_TV_NT, 3), (hip.TUNE_TV_PIPE, 1), (hip.TUNE_TV_U, 2), (hip.TUNE_TV_ROWS, 128), (hip.TUNE_TV_XCD, 2), (hip.TUNE_TV_RING, 1), (hip.TUNE_TV_SLOTS, 0)):
    ctx.set_tuning(key, v)
ref = signature()
NT = {0: "default", 1: "nt st (1)", 2: "nt st", 3: "plain st"}
for nt, pipe, u, rows, xcd, pad, ring, slots in itertools.product(nts, pipes, us, rowss, xcds, pads, rings, slotss):
    ctx.set_tuning(hip.TUNE_TV_RING, ring)
    ctx.set_tuning(hip.TUNE_TV_SLOTS, slots)
    ctx.set_tuning(hip.TUNE_TV_XCD, xcd)
    ctx.set_tuning(hip.TUNE_TV_LDS_PAD, pad)
    ctx.set_tuning(hip.TUNE_TV_NT, nt)
    ctx.set_tuning(hip.TUNE_TV_PIPE, pipe)
    ctx.set_tuning(hip.TUNE_TV_U, u)
    ctx.set_tuning(hip.TUNE_TV_ROWS, rows)
    got = signature()
    # the prox outputs are elementwise: bit-identical; the sums depend on the chunking (rows per workgroup) only
    same_x = np.array_equal(got[1], ref[1]) and np.array_equal(got[4], ref[4])
    close = all(np.allclose(g, r, rtol=1e-12, atol=0) for g, r in ((got[0], ref[0]), (got[2], ref[2]), (got[3], ref[3])))
    exact = all(np.array_equal(g, r) for g, r in ((got[0], ref[0]), (got[2], ref[2]), (got[3], ref[3])))
    ctx.set_vector(hip.VEC_X0, x0)
    ctx.init()
    t = timed(lambda: ctx.step(0.1))
    ctx.set_vector(hip.VEC_X0, x0)
    ctx.init()

    def acc():
        ctx.step_accel(0.1, 0.3, True)
        ctx.commit(False)
    ta = timed(acc)
    print(f"{NT[nt]:9s} pipe={pipe} U={u} rows={rows:3d} xcd={xcd} pad={pad:5d} ring={ring} slots={slots}  plain {t:6.4f} ms {40 * P / t / 1e6:6.0f} GB/s | FISTA {ta:6.4f} ms {56 * P / ta / 1e6:6.0f} GB/s"
          f" | xprox {'same bits' if same_x else 'DIFFERS'}, scalars {'same bits' if exact else ('rtol 1e-12' if close else 'DIFFER')}", flush=True)
A.close()
