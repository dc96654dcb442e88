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


@pytest.mark.parametrize("m,n,reps", [(777, 4096, 60), (5000, 16384, 40), (30000, 32768, 20), (20000, 65536, 20), (6000, 100000, 12), (5000, 131072, 12), (2500, 200000, 10)])
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


@pytest.mark.parametrize("accel", [False, True])
@pytest.mark.parametrize("m,n,reps", [(70, 65536, 25), (300, 4096, 40), (90, 20000, 30), (4000, 32768, 12), (3000, 65536, 10), (700, 90000, 12), (900, 131072, 10), (400, 262144, 8)])
def test_one_pass_kernel_never_consumes_a_previous_launch_s_partials(m, n, reps, accel):
    """Consecutive launches reuse the same slot lines.  Alternate between two different iterates and check every launch
    against its own two-launch reference: a slot value left over from the previous launch (a read that bypasses the
    sentinel protocol, e.g. through a non-coherent cache) would be a plausible number for the WRONG x."""
    rng = np.random.RandomState(m ^ n)
    if m * n <= 2 ** 23:       # small cases from a host matrix (fh_set_matrix), the large ones generated on the device
        op = fa.DenseMatrixMap(rng.randn(m, n) / (np.sqrt(m) + np.sqrt(n)))
    else:
        op = fa.DenseMatrixMap.synthetic(m, n, 11, synthetic.lasso_scale(m, n))
    c = op.ctx
    b = rng.randn(m)
    xs = [rng.randn(n) * 0.05, rng.randn(n) * 0.2]
    tau = 0.3

    def state(x):
        c.set_loss_lsq(b); c.set_prox(hip.PROX_SHRINK, 0.02); c.set_vector(hip.VEC_X0, x); c.init()
        if accel:      # one committed iteration so that x_accel0 / z_accel0 are not x0 / z0
            c.fwd(tau); c.adj(tau, True, 0.0); c.commit()

    try:
        refs = []
        for x in xs:
            state(x)
            s = c.fwd(tau)
            a = c.adj(tau, accel, 0.3 if accel else 0.0)
            refs.append((s[hip.S_FSQ], a[hip.S_DXDG], c.get_vector(hip.VEC_Z, m), c.get_vector(hip.VEC_G1, n)))
        for k in range(reps):
            which = k % 2
            state(xs[which])
            f = c.step_accel(tau, 0.3, False) if accel else c.step(tau)
            fsq, dxdg, z, g = refs[which]
            np.testing.assert_allclose(f[hip.S_FSQ], fsq, rtol=1e-12, err_msg=f"launch {k}")
            np.testing.assert_allclose(f[hip.S_DXDG], dxdg, rtol=1e-9, err_msg=f"launch {k}")
            np.testing.assert_allclose(c.get_vector(hip.VEC_Z, m), z, rtol=1e-12, atol=1e-14, err_msg=f"launch {k}")
            np.testing.assert_allclose(c.get_vector(hip.VEC_G1, n), g, rtol=1e-9, atol=1e-13, err_msg=f"launch {k}")
    finally:
        op.close()
