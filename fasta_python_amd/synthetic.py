"""Constants of the counter-based synthetic matrix generator (device kernel k_gen_matrix).

element(idx) = (sum of the eight 16-bit fields of two splitmix64 words keyed by (seed, idx) - 262140) * coef,
an Irwin-Hall(8) variate scaled to unit variance times `scale` -- integer arithmetic plus one IEEE
multiply, so the host twin in oracle/problems.py reproduces it bit for bit (tests/test_gpu_dense.py).
"""
import math

IH_STD = math.sqrt(8.0 * (65536.0 ** 2 - 1.0) / 12.0)


def synth_coef(scale):
    return float(scale) / IH_STD


def lasso_scale(m_total, n):
    """A = G / (sqrt(m) + sqrt(n)) stands in for `A /= la.norm(A, 2)` (sparse_least_squares.py:67-68)."""
    return 1.0 / (math.sqrt(m_total) + math.sqrt(n))


def sparse_signal(n, seed):
    """K = ceil(n/100) ones at seeded positions (scaled-up sparse_least_squares.py:62-64; K=10 of N=1000 there)."""
    import numpy as np
    x = np.zeros(n)
    x[np.random.RandomState(seed).permutation(n)[:int(math.ceil(n / 100))]] = 1
    return x


def lasso_observation(A_map, x_true, seed_noise, sigma, row0=0, m_total=None):
    """b = A x_true + sigma*N(0,1) for the rows this rank holds (sparse_least_squares.py:71); A x_true runs on the device."""
    import numpy as np
    m = A_map.Wshape[0]
    m_total = m if m_total is None else m_total
    noise = np.random.RandomState(seed_noise).randn(m_total)[row0:row0 + m]
    return A_map(x_true) + sigma * noise
