"""Linear maps behind the reference's `fasta.linalg` names (fasta/linalg.py:13-160).

`LinearMap` keeps the reference's callable-pair contract (shape asserts, `.H`, algebra).  Two
subclasses are *recognised* by `fasta()` and run on the device:

  DenseMatrixMap  -- row-major float64 matrix, resident in HBM (from a host ndarray, or generated
                     on the device by the counter-based synthetic generator, optionally one row
                     block of a matrix sharded over GPUs);
  GradDivMap      -- the periodic div/grad stencil pair of examples/tv_denoising.py:26-63.

Calling a recognised map on a host array applies the operator ON THE DEVICE (fh_apply); there is
no NumPy matvec in this module.
"""

from functools import reduce
from operator import mul

import numpy as np

from . import hip

Matrix = np.ndarray
Vector = np.ndarray

__all__ = ["LinearMap", "LinearOperator", "DenseMatrixMap", "GradDivMap", "Matrix", "Vector"]


class LinearMap:
    """Callable pair (map, adjoint) between array spaces V and W (fasta/linalg.py:13-69)."""

    def __init__(self, map_func, adj_func, Vshape, Wshape=None):
        # 3-argument form (map, adj, shape) = the old `LinearOperator` call used by
        # examples/democratic_representation.py:80-82: an endomorphism on `shape`.
        if Wshape is None:
            Wshape = Vshape
        self.map_func, self.adj_func = map_func, adj_func
        self.Vshape, self.Wshape = tuple(Vshape), tuple(Wshape)

    @staticmethod
    def from_matrix(A, device=0, storage="f64"):
        """fasta/linalg.py:37-41.  Returns a device-resident DenseMatrixMap (storage="f32": opt-in float32 storage of A)."""
        assert A.ndim == 2
        return DenseMatrixMap(A, device=device, storage=storage)

    @staticmethod
    def identity(shape):
        return LinearMap(lambda x: x, lambda x: x, shape, shape)

    def __call__(self, v):
        assert v.shape == self.Vshape            # AssertionError like fasta/linalg.py:58
        w = self.map_func(v)
        assert w.shape == self.Wshape            # :60
        return w

    @property
    def H(self):
        return LinearMap(self.adj_func, self.map_func, self.Wshape, self.Vshape)

    # ---- algebra (host-side composition of callables; fasta/linalg.py:71-160) -------------------
    def __matmul__(self, B):
        assert isinstance(B, LinearMap) and self.Wshape == B.Vshape     # same check as :77
        return LinearMap(lambda x: self(B(x)), lambda y: B.H(self.H(y)), self.Vshape, B.Wshape)

    def __rmul__(self, k):
        assert np.isscalar(k)
        return LinearMap(lambda x: k * self(x), lambda y: k * self.H(y), self.Vshape, self.Wshape)

    __mul__ = __rmul__

    def __neg__(self):
        return -1 * self

    def __add__(self, B):
        assert isinstance(B, LinearMap) and (self.Vshape, self.Wshape) == (B.Vshape, B.Wshape)
        return LinearMap(lambda x: self(x) + B(x), lambda y: self.H(y) + B.H(y), self.Vshape, self.Wshape)

    def __sub__(self, B):
        return self + (-B)

    @property
    def is_operator(self):
        return self.Vshape == self.Wshape

    def __pow__(self, n, modulo=None):
        assert self.is_operator
        out = LinearMap.identity(self.Vshape)
        for _ in range(n):
            out = out @ self
        return out

    @property
    def _scipy(self):
        from scipy.sparse import linalg as sla
        M = reduce(mul, self.Vshape, 1)
        N = reduce(mul, self.Wshape, 1)
        return sla.LinearOperator((M, N), matvec=lambda x: np.ravel(self(x.reshape(self.Vshape))),
                                  rmatvec=lambda y: np.ravel(self.H(y.reshape(self.Wshape))))

    def eigs(self, k=1):
        from scipy.sparse import linalg as sla
        assert self.is_operator
        values, vectors = sla.eigs(self._scipy, k)
        return values, np.reshape(vectors.T, (k,) + self.Wshape)


LinearOperator = LinearMap      # name the reference's examples import (sparse_least_squares.py:12)


class _DeviceMap(LinearMap):
    """A LinearMap whose operator lives in a HipContext; `fasta()` runs the fused device loop on it."""

    def __init__(self, Vshape, Wshape, device, storage="f64"):
        self.ctx = hip.HipContext(device, storage)
        self.device = device
        LinearMap.__init__(self, self._apply_fwd, self._apply_adj, Vshape, Wshape)

    def _apply_fwd(self, v):
        return self.ctx.apply(np.asarray(v, dtype=np.float64), adjoint=False).reshape(self.Wshape)

    def _apply_adj(self, w):
        return self.ctx.apply(np.asarray(w, dtype=np.float64), adjoint=True).reshape(self.Vshape)

    def close(self):
        self.ctx.close()

    def spectral_norm_squared(self, iters=50, rtol=1e-6, seed=None):
        """||A||_2^2 by power iteration on A^H A, the matvecs running on the device: each iteration is one
        `fh_gradient_at` with a zero target (the one-pass kernel reads a dense A once for both directions); the host only
        normalises the n-vector in between.  OPT-IN replacement for `LinearMap.eigs` (linalg.py:149-160) and for the
        four setup passes of fasta()'s random-probe estimate (fasta/__init__.py:100-113): for f = .5||Ax - b||^2 pass
        `L = op.spectral_norm_squared()` and `tau0 = (2 / L) / 10` to fasta().  A different L changes every iterate with
        respect to the reference's RNG-based estimate, so fasta() never calls this by itself.
        Overwrites the context's loss with a zero least-squares target; call it before fasta() (which sets its own)."""
        c = self.ctx
        n = int(np.prod(self.Vshape))
        m = int(np.prod(self.Wshape))
        rng = np.random.RandomState(seed)
        x = rng.randn(n)
        x /= np.linalg.norm(x)
        c.set_loss_lsq(np.zeros(m))
        lam = 0.0
        for _ in range(int(iters)):
            c.set_vector(hip.VEC_T0, x)
            c.gradient_at(hip.VEC_T0, hip.VEC_T1)                  # y = A^H (A x - 0)
            y = c.get_vector(hip.VEC_T1, n)
            new = float(np.dot(x, y))                              # Rayleigh quotient (||x|| = 1)
            ny = float(np.linalg.norm(y))
            if ny == 0.0:
                return 0.0
            x = y / ny
            if abs(new - lam) <= rtol * abs(new):
                lam = new
                break
            lam = new
        return lam


class DenseMatrixMap(_DeviceMap):
    """Dense float64 matrix held row-major in HBM (the operator of fasta/linalg.py:41).

    `rows` = (row0, m_total) describes a row block of a larger matrix when A is sharded across
    ranks (one process per GPU); the default is the whole matrix on one GPU.
    """

    def __init__(self, A=None, device=0, tuning=None, _defer=False, storage="f64"):
        """storage="f32" (opt-in): keep the device copy of A in float32 -- half the bytes per pass, ~2x the iterations/s on
        large matrices.  The solve is then the reference's solve on the ROUNDED matrix A.astype(float32) (all vectors and
        arithmetic stay float64), so iterates differ from the float64-matrix run by the effect of that rounding."""
        self.rows = None
        self.shape = None
        self.storage = storage
        if _defer:
            _DeviceMap.__init__(self, (0,), (0,), device, storage)
        else:
            assert A is not None and A.ndim == 2
            _DeviceMap.__init__(self, (A.shape[1],), (A.shape[0],), device, storage)
            self._tune(tuning)
            self.ctx.set_matrix(A)
            self.shape = tuple(A.shape)

    def _tune(self, tuning):
        for key, value in (tuning or {}).items():
            self.ctx.set_tuning(key, value)

    @classmethod
    def synthetic(cls, m, n, seed, scale, row0=0, m_total=None, device=0, tuning=None, storage="f64"):
        """Rows [row0, row0+m) of the counter-based synthetic matrix, generated in HBM."""
        from .synthetic import synth_coef
        self = cls(_defer=True, device=device, storage=storage)
        self._tune(tuning)
        self.ctx.generate_matrix(m, n, row0, seed, synth_coef(scale))
        self.Vshape, self.Wshape = (n,), (m,)
        self.shape = (m, n)
        self.rows = (row0, m if m_total is None else m_total)
        return self

    def host_rows(self, row0, nrows):
        return self.ctx.get_matrix_rows(row0, nrows)

    @property
    def T(self):
        return self.H


class GradDivMap(_DeviceMap):
    """A = div : (H, W, 2) -> (H, W) and A^H = grad, periodic (examples/tv_denoising.py:26-63)."""

    def __init__(self, image_shape, device=0):
        H, W = image_shape
        _DeviceMap.__init__(self, (H, W, 2), (H, W), device)
        self.ctx.set_stencil(H, W)
        self.image_shape = (H, W)
