"""Row-sharded arithmetic of the product on ONE GPU (a lease has one): two contexts in one process hold the two row
blocks of the same synthetic matrix (non-zero row0), each with its own 1-rank RCCL communicator so that every launch takes
the sharded code path (local launch in mode 2 -> all-reduce -> separate n-side epilogue).  The test plays the all-reduce:
it sums the two ranks' A_k^T r_k partials (FH_VEC_G1) and loss sums (FH_S_FSQ) on the host and compares them with the
unsharded launch on the whole matrix.  The operator being sharded is `A @ x` / `A.T @ r` of fasta/linalg.py:41."""
import numpy as np
import pytest

import fasta_python_amd as fa
from fasta_python_amd import hip
from oracle import problems as pr

pytestmark = pytest.mark.gpu


def _prepare(op, b, x0, g0, mu=0.02):
    c = op.ctx
    c.set_loss_lsq(b)
    c.set_prox(hip.PROX_SHRINK, mu)
    c.set_vector(hip.VEC_X0, x0)
    c.init()                                  # arms the buffers (z history, prox history); g0 is then overwritten
    c.set_vector(hip.VEC_G0, g0)
    return c


@pytest.mark.parametrize("m,n,one_pass,storage", [(300, 4096, True, "f64"), (300, 4096, False, "f64"), (2 * 32768, 65536, True, "f64"),
                                                  (2 * 32768, 65536, False, "f64"), (300, 20000, True, "f32"), (2 * 8192, 65536, True, "f32"),
                                                  (200, 100000, True, "f64")])
def test_two_row_blocks_sum_to_the_unsharded_launch(m, n, one_pass, storage):
    """(m/2 x n) per emulated rank; 32768 x 65536 is BASELINE config 5's per-GPU shard shape.  Also in float32 storage and on
    a wide-row shape (x slice in LDS)."""
    scale = 1.0 / (np.sqrt(m) + np.sqrt(n))
    rng = np.random.RandomState(4)
    x0 = rng.randn(n) * 0.1
    g0 = rng.randn(n) * 0.1
    b = rng.randn(m)
    tau = 0.7
    half = m // 2
    whole = fa.DenseMatrixMap.synthetic(m, n, 0, scale, storage=storage)
    shards = [fa.DenseMatrixMap.synthetic(half, n, 0, scale, row0=k * half, m_total=m, storage=storage) for k in range(2)]
    try:
        # the shards really are the two row blocks of the same matrix
        for k, r in ((0, 0), (0, half - 1), (1, 0), (1, half - 1)):
            assert np.array_equal(shards[k].host_rows(r, 1), whole.host_rows(k * half + r, 1))
        cw = _prepare(whole, b, x0, g0)
        cs = [_prepare(s, b[k * half:(k + 1) * half], x0, g0) for k, s in enumerate(shards)]
        for c in cs:
            c.comm_init(1, 0, hip.comm_unique_id())
            assert c.comm_count() == 1 and c.sharded
        if one_pass:
            assert cw.fused_supported() in (1, 3)
            sw = cw.step(tau)
            ss = [c.step(tau) for c in cs]
        else:
            sw = cw.fwd(tau)
            ss = [c.fwd(tau) for c in cs]
            sw_adj = cw.adj(tau)
            for c in cs:
                c.adj(tau)
        g_ref = cw.get_vector(hip.VEC_G1, n)
        g_sum = cs[0].get_vector(hip.VEC_G1, n) + cs[1].get_vector(hip.VEC_G1, n)
        z_ref = cw.get_vector(hip.VEC_Z, m)
        z_cat = np.concatenate([c.get_vector(hip.VEC_Z, half) for c in cs])
        xp = [c.get_vector(hip.VEC_XPROX, n) for c in [cw] + cs]
    finally:
        whole.close()
        for s in shards:
            s.close()
    # forward half: rows are local, so z is the same up to the summation order along a row (team shapes can differ)
    np.testing.assert_allclose(z_cat, z_ref, rtol=1e-12, atol=1e-13 * np.abs(z_ref).max())
    assert np.array_equal(xp[0], xp[1]) and np.array_equal(xp[0], xp[2])          # the prox is replicated work
    # the loss sum the all-reduce would produce
    fsq = ss[0][hip.S_FSQ] + ss[1][hip.S_FSQ]
    np.testing.assert_allclose(fsq, sw[hip.S_FSQ], rtol=1e-12)
    # the A^T partial sums the all-reduce would produce
    np.testing.assert_allclose(g_sum, g_ref, rtol=1e-12, atol=1e-12 * np.abs(g_ref).max())
    # n-side reductions computed before the exchange are replicated: identical on every rank and equal to the unsharded ones
    for k in (hip.S_DXG0, hip.S_DX2, hip.S_XH2, hip.S_G02, hip.S_GSUM, hip.S_GMAX):
        assert ss[0][k] == ss[1][k]
        np.testing.assert_allclose(ss[0][k], sw[k], rtol=1e-12)


def test_shard_of_the_synthetic_matrix_matches_the_host_twin():
    """Row block with a non-zero row0 = the same rows of the host twin of the generator, bit for bit."""
    m_total, n, half = 96, 160, 48
    scale = 1.0 / (np.sqrt(m_total) + np.sqrt(n))
    full = pr.synth_matrix(m_total, n, 0, scale)
    for k in range(2):
        op = fa.DenseMatrixMap.synthetic(half, n, 0, scale, row0=k * half, m_total=m_total)
        try:
            assert np.array_equal(op.host_rows(0, half), full[k * half:(k + 1) * half])
        finally:
            op.close()
