"""Restated example recipes + the synthetic-matrix generator twin.  TEST INFRASTRUCTURE ONLY.

The reference's example modules cannot be imported, so their
`construct()` / `solve()` recipes are restated here from the source text; each function cites
the lines it follows.  RNG draws are made in the reference's order so that a given
`np.random.seed` produces the same problem instance the reference example would build.
"""

import numpy as np
from numpy import linalg as la

from . import fasta_np as fo


class Problem:
    """Bundle of (A, At, f, gradf, g, proxg, x0) closures in the examples' calling convention."""

    def __init__(self, kind, A, At, f, gradf, g, proxg, x0, data):
        self.kind = kind
        self.A, self.At = A, At
        self.f, self.gradf, self.g, self.proxg = f, gradf, g, proxg
        self.x0 = x0
        self.data = data            # dict of the arrays/scalars that define the instance

    def args7(self):
        return (self.A, self.At, self.f, self.gradf, self.g, self.proxg, self.x0)


def _least_squares(b):
    # examples/sparse_least_squares.py:41-42 (same closures in lasso.py:42-43, nn_least_squares.py:39-40)
    f = lambda z: .5 * la.norm((z - b).ravel()) ** 2
    gradf = lambda z: z - b
    return f, gradf


def _sparse_signal_and_matrix(M, N, K, sigma, normalise=True):
    # examples/sparse_least_squares.py:62-74: permutation, randn(M,N), spectral normalisation, noise
    x = np.zeros(N)
    x[np.random.permutation(N)[:K]] = 1
    A = np.random.randn(M, N)
    if normalise:
        A /= la.norm(A, 2)
    b = A @ x + sigma * np.random.randn(M)
    return A, b, x


def sparse_least_squares(M=200, N=1000, K=10, sigma=0.01, mu=0.02, normalise=True):
    """examples/sparse_least_squares.py:41-44 (closures), :50-76 (construct)."""
    A, b, x = _sparse_signal_and_matrix(M, N, K, sigma, normalise)
    return sparse_least_squares_from(A, b, mu, x_true=x)


def sparse_least_squares_from(A, b, mu, x_true=None):
    f, gradf = _least_squares(b)
    g = lambda x: mu * la.norm(x.ravel(), 1)
    proxg = lambda x, t: fo.shrink(x, t * mu)
    return Problem("sparse_ls", A, A.T, f, gradf, g, proxg, np.zeros(A.shape[1]),
                   dict(A=A, b=b, mu=mu, x_true=x_true))


def nn_least_squares(M=200, N=1000, K=10, sigma=0.005, normalise=True):
    """examples/nn_least_squares.py:39-42 (closures), :46-72 (construct)."""
    A, b, x = _sparse_signal_and_matrix(M, N, K, sigma, normalise)
    return nn_least_squares_from(A, b, x_true=x)


def nn_least_squares_from(A, b, x_true=None):
    f, gradf = _least_squares(b)
    g = lambda x: 0
    proxg = lambda x, t: np.maximum(x, 0)
    return Problem("nnls", A, A.T, f, gradf, g, proxg, np.zeros(A.shape[1]),
                   dict(A=A, b=b, x_true=x_true))


def l1_ball_lasso(M=200, N=1000, K=10, sigma=0.01, mu=0.8):
    """examples/lasso.py:42-45 (closures), :51-79 (construct; mu scaled by ||x||_1 BEFORE A is drawn)."""
    x = np.zeros(N)
    x[np.random.permutation(N)[:K]] = 1
    mu = mu * la.norm(x, 1)
    A = np.random.randn(M, N)
    A /= la.norm(A, 2)
    b = A @ x + sigma * np.random.randn(M)
    return l1_ball_lasso_from(A, b, mu, x_true=x)


def l1_ball_lasso_from(A, b, mu, x_true=None):
    f, gradf = _least_squares(b)
    g = lambda x: 0
    proxg = lambda x, t: fo.project_l1(x, mu)
    return Problem("l1ball", A, A.T, f, gradf, g, proxg, np.zeros(A.shape[1]),
                   dict(A=A, b=b, mu=mu, x_true=x_true))


def linf_regularised(M=96, N=96, mu=0.05):
    """Dense-matrix variant of examples/democratic_representation.py:39-42 (closures): the
    reference's operator there is a masked DCT (out of scope); the prox and objective are the same."""
    A = np.random.randn(M, N)
    A /= la.norm(A, 2)
    b = np.random.randn(M)
    return linf_regularised_from(A, b, mu)


def linf_regularised_from(A, b, mu):
    f, gradf = _least_squares(b)
    g = lambda x: mu * la.norm(x, np.inf)
    proxg = lambda x, t: fo.prox_linf(x, t * mu)
    return Problem("linf", A, A.T, f, gradf, g, proxg, np.zeros(A.shape[1]),
                   dict(A=A, b=b, mu=mu))


def sparse_logistic(M=1000, N=2000, K=5, mu=40):
    """examples/sparse_logistic.py:47-50 (closures), :54-80 (construct): labels b in {-1,+1}, A unnormalised."""
    x = np.zeros(N)
    x[np.random.permutation(N)[:K]] = 1
    A = np.random.randn(M, N)
    p = 1 / (1 + np.exp(-A @ x))
    b = 2.0 * (np.random.rand(M) < p) - 1
    return sparse_logistic_from(A, b, mu, x_true=x)


def sparse_logistic_from(A, b, mu, x_true=None):
    f = lambda z: np.sum(np.log(1 + np.exp(z)) - (b == 1) * z)
    gradf = lambda z: -b / (1 + np.exp(b * z))
    g = lambda x: mu * la.norm(x.ravel(), 1)
    proxg = lambda x, t: fo.shrink(x, t * mu)
    return Problem("logistic", A, A.T, f, gradf, g, proxg, np.zeros(A.shape[1]),
                   dict(A=A, b=b, mu=mu, x_true=x_true))


# ---- total variation (examples/tv_denoising.py) ---------------------------------------------
def grad(X):
    """examples/tv_denoising.py:26-40: out[..., d] = roll(X, +1, axis=d) - X (periodic)."""
    out = np.zeros(X.shape + (X.ndim,))
    for d in range(X.ndim):
        out[..., d] = np.roll(X, 1, axis=d) - X
    return out


def div(Y):
    """examples/tv_denoising.py:43-63: sum_d roll(Y[..., d], -1, axis=d) - Y[..., d]."""
    nd = Y.shape[-1]
    assert nd == Y.ndim - 1
    out = np.zeros(Y.shape[:-1])
    for d in range(nd):
        comp = Y[..., d]
        out += np.roll(comp, -1, axis=d) - comp
    return out


def checkerboard(H, W, square):
    ii, jj = np.indices((H, W))
    return (((ii // square) + (jj // square)) % 2).astype(float)


def tv_denoising(H=32, W=32, square=8, sigma=0.1, mu=0.1):
    """examples/tv_denoising.py:85-96 (closures), :105-125 (construct).  `scipy.misc.ascent` is
    gone and needs a download in modern SciPy, so the clean image is a {0,1} checkerboard
    (already max-normalised); noise and Y0 follow the reference."""
    M = checkerboard(H, W, square)
    M /= np.max(M)
    M += sigma * np.random.randn(*M.shape)
    return tv_denoising_from(M, mu)


def tv_denoising_from(M, mu):
    target = M / mu
    f = lambda Z: .5 * la.norm((Z - M / mu).ravel()) ** 2
    gradf = lambda Z: Z - M / mu
    g = lambda Y: 0
    Y0 = np.zeros(M.shape + (2,))
    return Problem("tv", div, grad, f, gradf, g, fo.tv_dual_ball, Y0, dict(M=M, mu=mu, target=target))


def tv_primal(M, mu, Ysol):
    return M - mu * div(Ysol)                       # tv_denoising.py:101


FROM_DATA = {
    "sparse_ls": lambda d: sparse_least_squares_from(d["A"], d["b"], float(d["mu"])),
    "nnls": lambda d: nn_least_squares_from(d["A"], d["b"]),
    "l1ball": lambda d: l1_ball_lasso_from(d["A"], d["b"], float(d["mu"])),
    "linf": lambda d: linf_regularised_from(d["A"], d["b"], float(d["mu"])),
    "tv": lambda d: tv_denoising_from(d["M"], float(d["mu"])),
    "logistic": lambda d: sparse_logistic_from(d["A"], d["b"], float(d["mu"])),
}


# ---- synthetic generator twin (bit-identical to csrc gen_matrix kernel) -----------------------
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
IH_MEAN = 8 * 32767.5                                  # mean of the sum of eight u16 fields
IH_STD = float(np.sqrt(8.0 * (65536.0 ** 2 - 1.0) / 12.0))


def _mix(z):
    with np.errstate(over="ignore"):
        z = z + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _fields_sum(h):
    s = np.zeros(h.shape, dtype=np.int64)
    for sh in (0, 16, 32, 48):
        s += ((h >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    return s


def synth_values(seed, idx, coef):
    """value(idx) = (sum of the eight 16-bit fields of two splitmix64 words - 262140) * coef.
    Pure integer arithmetic + one IEEE multiply => identical on host and device."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed))
        h1 = _mix(key + np.uint64(2) * idx)
        h2 = _mix(key + np.uint64(2) * idx + np.uint64(1))
    s = _fields_sum(h1) + _fields_sum(h2)
    return (s - 262140).astype(np.float64) * np.float64(coef)


def synth_coef(scale):
    return np.float64(scale) / np.float64(IH_STD)


def synth_matrix(m, n, seed, scale, row0=0, n_total=None):
    """Rows row0..row0+m of the (.., n_total) synthetic matrix; element (i,j) uses counter i*n_total+j."""
    n_total = n if n_total is None else n_total
    rows = (np.arange(row0, row0 + m, dtype=np.uint64)[:, None] * np.uint64(n_total))
    idx = rows + np.arange(n, dtype=np.uint64)[None, :]
    return synth_values(seed, idx, synth_coef(scale))


def synth_lasso(m, n, seed_A=0, seed_x=1, seed_noise=2, sigma=0.01, mu=0.02, row0=0, m_total=None):
    """BASELINE.md section 4 recipe: A = G/(sqrt(m)+sqrt(n)), K=ceil(n/100)-sparse x_true of ones,
    b = A x_true + sigma N(0,1), x0 = 0 (scaled-up sparse_least_squares.py:62-76)."""
    m_total = m if m_total is None else m_total
    scale = 1.0 / (np.sqrt(m_total) + np.sqrt(n))
    A = synth_matrix(m, n, seed_A, scale, row0=row0)
    x_true = synth_sparse_signal(n, seed_x)
    noise = np.random.RandomState(seed_noise).randn(m_total)[row0:row0 + m]
    b = A @ x_true + sigma * noise
    return A, b, x_true, mu


def synth_sparse_signal(n, seed):
    K = int(np.ceil(n / 100))
    x = np.zeros(n)
    x[np.random.RandomState(seed).permutation(n)[:K]] = 1
    return x
