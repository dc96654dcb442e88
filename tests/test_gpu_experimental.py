"""The experimental forms of the stencil sweep (csrc/fh_experimental.h), compiled only into libfasta_hip_experimental.so
(`make -C fasta_python_amd/csrc experimental`).  Its own job, in its own process:

    FASTA_HIP_LIB=$PWD/fasta_python_amd/libfasta_hip_experimental.so python -m pytest tests/test_gpu_experimental.py -m gpu -q

tests/conftest.py leaves this file out of every collection that runs against the shipped library.  Every form must produce the
bits of the shipped sweep: FH_TUNE_TV_ZFREE = 0 (the round-1 one-pass kernels that stream z), FH_TUNE_TV_RING (trips prefetched by
LDS-DMA into a per-wave ring), FH_TUNE_TV_SLOTS (persistent workgroups walking short chunks band-major)."""
import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from tests import test_gpu_prox_tv as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H_,W_", T.ONE_PASS_SHAPES)
def test_z_streaming_one_pass_step_equals_two_launch_step(H_, W_):
    T.one_pass_tv_step_equals_two_launch_step(H_, W_, 0)


@pytest.mark.parametrize("prox", ["tvball", "identity"])
@pytest.mark.parametrize("restart", [True, False])
@pytest.mark.parametrize("H_,W_", T.ONE_PASS_ACCEL_SHAPES)
def test_z_streaming_accelerated_steps_equal_two_launch_steps(H_, W_, restart, prox):
    T.one_pass_accelerated_tv_steps_equal_two_launch_steps(H_, W_, restart, prox, 0)


def test_zfree_switch_is_refused_while_the_accelerated_iterate_is_kept_lazily():
    """in one-pass FISTA mode the z-free sweep rotates its image buffers without writing them, so switching
    FH_TUNE_TV_ZFREE to 0 mid-solve would make the z-streaming kernel read stale images: the library refuses (FH_E_STATE) until
    the next fh_init / fh_set_vector(X0); setting the value it already has stays allowed."""
    rng = np.random.RandomState(6)
    H_, W_ = 40, 90
    op = fa.GradDivMap((H_, W_))
    try:
        c = op.ctx
        c.set_loss_lsq(rng.randn(H_, W_))
        c.set_prox(hip.PROX_TVBALL)
        c.set_vector(hip.VEC_X0, rng.randn(H_, W_, 2) * 0.5)
        c.init()
        c.step_accel(0.2, 0.0, True)
        c.commit(False)
        c.set_tuning(hip.TUNE_TV_ZFREE, 1)                      # no change: fine
        with pytest.raises(hip.HipError, match="TV_ZFREE"):
            c.set_tuning(hip.TUNE_TV_ZFREE, 0)
        c.step_accel(0.2, 0.3, True)                            # the solve goes on undisturbed
        c.set_vector(hip.VEC_X0, np.zeros((H_, W_, 2)))         # a new start lifts the restriction
        c.set_tuning(hip.TUNE_TV_ZFREE, 0)
        c.init()
        c.step_accel(0.2, 0.0, True)
    finally:
        op.close()



@pytest.mark.parametrize("ring,slots,rows", [(2, 0, 0), (3, 0, 0), (1, 2, 16), (2, 3, 8), (1, 1, 4), (2, 5, 32)])
@pytest.mark.parametrize("H_,W_", [(2, 4), (5, 64), (33, 62), (40, 258), (97, 130), (64, 1000), (130, 122), (300, 4000), (37, 61)])
def test_ring_and_persistent_forms_of_the_one_pass_sweep_equal_the_default(H_, W_, ring, slots, rows):
    """Round 4: FH_TUNE_TV_RING (2-row trips prefetched by LDS-DMA into a per-wave ring; needs an even width -- an odd one runs the
    register form) and FH_TUNE_TV_SLOTS (persistent workgroups walking short chunks) change how the sweep's bytes arrive and which
    workgroup sums what, never a stored value: xprox must be BIT-identical to the default sweep's, the sums equal to rounding -- plain
    and accelerated steps, with and without a lagging extrapolation coefficient."""
    rng = np.random.RandomState(H_ * 11 + W_)
    M = rng.randn(H_, W_)
    Y0 = rng.randn(H_, W_, 2) * 0.8
    tau = 0.11
    op = fa.GradDivMap((H_, W_))
    try:
        c = op.ctx

        def signature():
            c.set_loss_lsq(M)
            c.set_prox(hip.PROX_TVBALL)
            c.set_vector(hip.VEC_X0, Y0)
            c.init()
            s = c.step(tau)
            xp = c.get_vector(hip.VEC_XPROX, Y0.size)
            c.set_vector(hip.VEC_X0, Y0)
            c.init()
            a1 = c.step_accel(tau, 0.0, True)
            c.commit(False)
            a2 = c.step_accel(tau, 0.3, False)          # reads both prox outputs (lagging coefficient of the previous step)
            c.commit(False)
            a3 = c.step_accel(tau, 0.45, True)
            xa = c.get_vector(hip.VEC_XPROX, Y0.size)
            return s, xp, a1, a2, a3, xa
        for key, v in ((hip.TUNE_TV_RING, 1), (hip.TUNE_TV_SLOTS, 0), (hip.TUNE_TV_ROWS, 0)):
            c.set_tuning(key, v)
        ref = signature()
        for key, v in ((hip.TUNE_TV_RING, ring), (hip.TUNE_TV_SLOTS, slots), (hip.TUNE_TV_ROWS, rows)):
            c.set_tuning(key, v)
        got = signature()
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[5], ref[5])
        for g, r in ((got[0], ref[0]), (got[2], ref[2]), (got[3], ref[3]), (got[4], ref[4])):
            np.testing.assert_allclose(g, r, rtol=1e-11, atol=1e-300)
        assert np.array_equal(signature()[0], got[0])                 # and the chosen form is bitwise repeatable
    finally:
        op.close()

