"""How much of the stencil one-pass sweep's time is arithmetic?  Same sweep with the identity prox (no divisions / square
root in the projection) against the unit-ball prox, 8192^2, HIP-event kernel times (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip

side = 8192
rng = np.random.RandomState(0)
M = rng.standard_normal((side, side))
op = fa.GradDivMap((side, side))
c = op.ctx
c.set_loss_lsq(M)
for name, kind in (("unit-ball prox", hip.PROX_TVBALL), ("identity prox ", hip.PROX_IDENTITY)):
    c.set_prox(kind)
    c.set_vector(hip.VEC_X0, np.zeros((side, side, 2)))
    c.init()
    for accel in (False, True):
        c.set_vector(hip.VEC_X0, rng.standard_normal((side, side, 2)) * 0.5)
        c.init()
        fn = (lambda: c.step_accel(0.1, 0.3, False)) if accel else (lambda: c.step(0.1))
        for _ in range(3):
            fn(); c.commit()
        c.timing_reset(); c.timing_enable(True)
        for _ in range(10):
            fn(); c.commit()
        c.timing_enable(False)
        ms, cnt = c.timing_get(hip.K_FUSED)
        P = side * side
        print(f"{name} {'FISTA' if accel else 'plain'}: {ms / cnt:.4f} ms  ({(56 if accel else 40) * P / (ms / cnt) / 1e6:.0f} GB/s algorithmic)", flush=True)
op.close()
