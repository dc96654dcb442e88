"""fh_setup (one read of A for the solver's set-up) with the rows dealt in blocks against cyclically (bit 32 of FH_TUNE_FUSED_VARIANT), by where the matrix lies.
Usage: python scripts/probes/setup_cyclic.py   -> profiles/r06_placement.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

hip.alloc_cache(False)
rng = np.random.RandomState(0)


def make(m, n, seed, f32=False):
    A = fa.DenseMatrixMap.synthetic(m, n, seed, synthetic.lasso_scale(m, n), storage="f32" if f32 else "f64")
    c = A.ctx
    c.set_loss_lsq(rng.randn(m)); c.set_prox(hip.PROX_SHRINK, 0.02)
    for which in (hip.VEC_T0, hip.VEC_T1):
        c.set_vector(which, rng.randn(n))
    c.set_vector(hip.VEC_X0, np.zeros(n))
    return A


def setup_ms(c, variant):
    c.set_tuning(hip.TUNE_FUSED_VARIANT, variant)
    c.setup(); c.sync(); c.timing_reset(); c.timing_enable(True)
    for _ in range(3):
        c.setup()
    c.timing_enable(False)
    ms, cnt = c.timing_get(hip.K_FUSED)
    return ms / cnt


held = []
for i in range(7):
    A = make(65536, 65536, i); held.append(A)
    r = [(setup_ms(A.ctx, 2), setup_ms(A.ctx, 34)) for _ in range(2)]
    print(f"65536^2 matrix {i} ({(i + 1) * 32:3d} GiB resident): fh_setup blocked {r[0][0]:.3f} {r[1][0]:.3f}   cyclic {r[0][1]:.3f} {r[1][1]:.3f} ms", flush=True)
for A in held:
    A.close()
time.sleep(9.0)
for (m, n, f32) in ((65536, 65536, True), (32768, 131072, False), (65536, 32768, False), (32768, 32768, False), (16384, 16384, False), (8192, 8192, False), (20000, 30000, False), (100000, 10000, False)):
    A = make(m, n, 1, f32)
    r = [(setup_ms(A.ctx, 2), setup_ms(A.ctx, 34)) for _ in range(2)]
    print(f"{m} x {n}{' float32 storage' if f32 else ''}: fh_setup blocked {r[0][0]:.4f} {r[1][0]:.4f}   cyclic {r[0][1]:.4f} {r[1][1]:.4f} ms", flush=True)
    A.close(); time.sleep(1.5)
