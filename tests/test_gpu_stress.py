"""Hand-off stress: the cross-workgroup protocols (sc1 partials + tickets, team slots of the one-pass kernel) must
give BITWISE identical results on every repetition -- a stale or torn partial would show up as a differing bit."""
import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip, synthetic

pytestmark = pytest.mark.gpu


def _prepare(m, n, prox=hip.PROX_SHRINK):
    op = fa.DenseMatrixMap.synthetic(m, n, 7, synthetic.lasso_scale(m, n))
    rng = np.random.RandomState(m + n)
    c = op.ctx
    c.set_loss_lsq(rng.randn(m))
    c.set_prox(prox, 0.02)
    c.set_vector(hip.VEC_X0, rng.randn(n) * 0.05)
    c.init()
    return op, c


@pytest.mark.parametrize("m,n,reps", [(1000, 3000, 60), (8192, 8192, 40), (4096, 32768, 40), (20000, 65536, 15)])
def test_two_launch_kernels_are_bitwise_repeatable_under_repetition(m, n, reps):
    op, c = _prepare(m, n)
    try:
        s0 = c.fwd(0.3)
        a0 = c.adj(0.3)
        g0 = c.get_vector(hip.VEC_G1, n)
        z0 = c.get_vector(hip.VEC_Z, m)
        for _ in range(reps):
            s = c.fwd(0.3)
            a = c.adj(0.3)
            assert np.array_equal(s[:8], s0[:8]) and np.array_equal(a[8:14], a0[8:14])   # each launch owns its half of the block
        assert np.array_equal(c.get_vector(hip.VEC_G1, n), g0)
        assert np.array_equal(c.get_vector(hip.VEC_Z, m), z0)
    finally:
        op.close()


@pytest.mark.parametrize("m,n,reps", [(777, 4096, 60), (5000, 16384, 40), (30000, 32768, 20), (20000, 65536, 20)])
def test_one_pass_kernel_is_bitwise_repeatable_under_repetition(m, n, reps):
    op, c = _prepare(m, n)
    try:
        f0 = c.step(0.3)
        g0 = c.get_vector(hip.VEC_G1, n)
        z0 = c.get_vector(hip.VEC_Z, m)
        for k in range(reps):
            f = c.step(0.3)
            assert np.array_equal(f, f0), k
            if k % 5 == 0:
                assert np.array_equal(c.get_vector(hip.VEC_G1, n), g0), k
        assert np.array_equal(c.get_vector(hip.VEC_Z, m), z0)
        # and it agrees with the two-launch path on the same state
        s = c.fwd(0.3)
        a = c.adj(0.3)
        np.testing.assert_allclose(f0[hip.S_FSQ], s[hip.S_FSQ], rtol=1e-12)
        np.testing.assert_allclose(f0[hip.S_DXDG], a[hip.S_DXDG], rtol=1e-9)
        np.testing.assert_allclose(g0, c.get_vector(hip.VEC_G1, n), rtol=1e-9, atol=1e-13)
    finally:
        op.close()


def test_tv_one_pass_kernel_is_bitwise_repeatable():
    rng = np.random.RandomState(3)
    H_, W_ = 1500, 2100
    op = fa.GradDivMap((H_, W_))
    try:
        c = op.ctx
        c.set_loss_lsq(rng.standard_normal(H_ * W_))
        c.set_prox(hip.PROX_TVBALL)
        c.set_vector(hip.VEC_X0, rng.standard_normal(2 * H_ * W_) * 0.7)
        c.init()
        f0 = c.step(0.1)
        xp0 = c.get_vector(hip.VEC_XPROX, 2 * H_ * W_)
        for _ in range(30):
            assert np.array_equal(c.step(0.1), f0)
        assert np.array_equal(c.get_vector(hip.VEC_XPROX, 2 * H_ * W_), xp0)
    finally:
        op.close()
