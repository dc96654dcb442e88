import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic
"""One read of A for the whole set-up against the three passes and a step launch, float32 storage (k_setup_dense<..., F32 = 1>).  Usage: python scripts/probes/setup_cost_f32.py"""
for storage, variant, shapes in (("f32", 2, [(65536, 65536), (32768, 32768), (65536, 32768), (16384, 16384), (8192, 8192)]), ("f64", 2, [(65536, 65536)])):
    print(f"# {storage}, FH_TUNE_FUSED_VARIANT = {variant}")
    for m, n in shapes:
        A = fa.DenseMatrixMap.synthetic(m, n, 0, synthetic.lasso_scale(m, n), storage=storage, tuning={hip.TUNE_FUSED_VARIANT: variant})
        c = A.ctx
        b = synthetic.lasso_observation(A, synthetic.sparse_signal(n, 1), 2, 0.01)
        c.set_loss_lsq(b); c.set_prox(hip.PROX_SHRINK, 0.02)
        rng = np.random.RandomState(0)
        c.set_vector(hip.VEC_T0, rng.randn(n)); c.set_vector(hip.VEC_T1, rng.randn(n)); c.set_vector(hip.VEC_X0, np.zeros(n))
        c.setup()
        c.timing_reset(); c.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(10): c.setup()
        one = (time.perf_counter() - t0) / 10 * 1e3
        ms1, cnt1 = c.timing_get(hip.K_FUSED)
        c.timing_reset()
        t0 = time.perf_counter()
        for _ in range(10):
            c.gradient_at(hip.VEC_T0, hip.VEC_T2); c.gradient_at(hip.VEC_T1, hip.VEC_T3); c.init()
        three = (time.perf_counter() - t0) / 10 * 1e3
        ms3, cnt3 = c.timing_get(hip.K_FUSED)
        c.timing_reset()
        c.set_vector(hip.VEC_X0, np.zeros(n)); c.init()
        for _ in range(10): c.step(0.05); c.commit()
        mss, cnts = c.timing_get(hip.K_FUSED)
        print(f"{storage} {m:6d} x {n:6d}: fh_setup {one:7.3f} ms wall ({cnt1 // 10} launch: {ms1 / cnt1:6.3f} ms) | three passes {three:7.3f} ms wall ({cnt3 // 10} launches) | a step launch {mss / cnts:6.3f} ms", flush=True)
        A.close()
