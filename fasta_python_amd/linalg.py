"""Linear maps behind the reference's `fasta.linalg` names (fasta/linalg.py:13-160).

`LinearMap` keeps the reference's callable-pair contract (shape asserts, `.H`, algebra).  Two
subclasses are *recognised* by `fasta()` and run on the device:

  DenseMatrixMap  -- row-major float64 matrix, resident in HBM (from a host ndarray, or generated
                     on the device by the counter-based synthetic generator, optionally one row
                     block of a matrix sharded over GPUs, one process per GPU);
  ShardedDenseMatrixMap -- the same matrix split into contiguous row blocks over several devices
                     of THIS process (single call): `fasta(ShardedDenseMatrixMap(A,
                     devices=[0, 1, ...]), ls.f, ls.gradf, reg.g, reg.prox, x0)`;
  GradDivMap      -- the periodic div/grad stencil pair of examples/tv_denoising.py:26-63.

A DenseMatrixMap built from a host ndarray uploads LAZILY: it keeps a reference to the array (as the
closures of fasta/linalg.py:41 do) and copies it into HBM when the device loop first asks for its
context.  Called on a host array -- which is what the generic host loop, the reference's own loop or
user code do -- it applies `A @ x` / `A.T @ y` on the host, exactly the reference's expressions, so
`LinearMap.from_matrix(A)` together with Python closures (examples/lasso.py:42-47, :79) runs
bit-identically to the reference with or without a GPU.  Maps without a host copy (the synthetic
generator, GradDivMap) apply on the device (fh_apply); `device_apply` always does.
"""

from functools import reduce
from operator import mul

import numpy as np

from . import hip

Matrix = np.ndarray
Vector = np.ndarray

__all__ = ["LinearMap", "LinearOperator", "DenseMatrixMap", "ShardedDenseMatrixMap", "GradDivMap", "Matrix", "Vector"]


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
    def from_matrix(A, device=0, storage="f64", devices=None):
        """fasta/linalg.py:37-41.  Returns a DenseMatrixMap: `A @ x` / `A.T @ y` on host arrays (the reference's closures), the
        device-resident operator once the device loop adopts it (storage="f32": opt-in float32 storage of the device copy;
        devices=[...]: row blocks over several devices of this process, i.e. a ShardedDenseMatrixMap)."""
        assert A.ndim == 2
        if devices is not None:
            return ShardedDenseMatrixMap(A, devices=devices, storage=storage)
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

    def __init__(self, Vshape, Wshape, device, storage="f64", devices=None, lazy=False, rccl_shell=False):
        self._ctx = None
        self._ctx_args = (device, storage, devices, rccl_shell)
        self.device = device
        LinearMap.__init__(self, self._apply_fwd, self._apply_adj, Vshape, Wshape)
        if not lazy:
            self.ctx                                   # create the device context now (raises without a GPU)

    @property
    def ctx(self):
        """The device context; created -- and, for a map built from a host matrix, filled -- on first use."""
        if self._ctx is None:
            device, storage, devices, rccl_shell = self._ctx_args
            self._ctx = hip.HipContext(device, storage, devices=devices, rccl_shell=rccl_shell)
            try:
                self._on_context(self._ctx)
            except Exception:
                self._ctx.close()
                self._ctx = None
                raise
        return self._ctx

    @ctx.setter
    def ctx(self, value):                              # (test stand-ins install their own context object)
        self._ctx = value

    def _on_context(self, ctx):
        pass

    def device_apply(self, v, adjoint=False):
        """A v (or A^H v) computed by the device kernels on a host array (fh_apply)."""
        shape = self.Vshape if adjoint else self.Wshape
        return self.ctx.apply(np.asarray(v, dtype=np.float64), adjoint=adjoint).reshape(shape)

    def _apply_fwd(self, v):
        return self.device_apply(v, adjoint=False)

    def _apply_adj(self, w):
        return self.device_apply(w, adjoint=True)

    def close(self):
        if self._ctx is not None:
            self._ctx.close()

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

    def __init__(self, A=None, device=0, tuning=None, _defer=False, storage="f64", _devices=None, _rccl_shell=False):
        """storage="f32" (opt-in): keep the device copy of A in float32 -- half the bytes per pass, ~2x the iterations/s on
        large matrices.  The solve is then the reference's solve on the ROUNDED matrix A.astype(float32) (all vectors and
        arithmetic stay float64), so iterates differ from the float64-matrix run by the effect of that rounding."""
        self.rows = None
        self.shape = None
        self.storage = storage
        self.matrix = None                 # host copy (a reference to the caller's array, like the closures of linalg.py:41)
        self._tuning = dict(tuning or {})
        if _defer:
            _DeviceMap.__init__(self, (0,), (0,), device, storage, _devices, rccl_shell=_rccl_shell)
        else:
            assert A is not None and A.ndim == 2
            self.matrix = A
            self.shape = tuple(A.shape)
            _DeviceMap.__init__(self, (A.shape[1],), (A.shape[0],), device, storage, _devices, lazy=True, rccl_shell=_rccl_shell)

    def _on_context(self, ctx):
        """First use of the device context: tuning, then the one H2D copy of the host matrix."""
        for key, value in self._tuning.items():
            ctx.set_tuning(key, value)
        if self.matrix is not None:
            ctx.set_matrix(self.matrix)

    def _tune(self, tuning):
        self._tuning.update(tuning or {})
        for key, value in (tuning or {}).items():
            self.ctx.set_tuning(key, value)

    # host arrays in, host arrays out: with a host copy of A these ARE the reference's closures (linalg.py:41); a storage="f32"
    # map applies the rounded matrix, i.e. the operator its device copy holds
    def _host_matrix(self):
        if self.storage == "f32" and self.matrix.dtype != np.float32:
            if getattr(self, "_rounded", None) is None:
                self._rounded = self.matrix.astype(np.float32).astype(np.float64)
            return self._rounded
        return self.matrix

    def _apply_fwd(self, v):
        if self.matrix is None:
            return self.device_apply(v, adjoint=False)
        return self._host_matrix() @ v

    def _apply_adj(self, w):
        if self.matrix is None:
            return self.device_apply(w, adjoint=True)
        return self._host_matrix().T @ w

    @classmethod
    def synthetic(cls, m, n, seed, scale, row0=0, m_total=None, device=0, tuning=None, storage="f64", _devices=None):
        """Rows [row0, row0+m) of the counter-based synthetic matrix, generated in HBM."""
        from .synthetic import synth_coef
        self = cls(_defer=True, device=device, storage=storage, _devices=_devices)
        self._tune(tuning)
        self.ctx.generate_matrix(m, n, row0, seed, synth_coef(scale))
        self.Vshape, self.Wshape = (n,), (m,)
        self.shape = (m, n)
        self.rows = (row0, m if m_total is None else m_total)
        return self

    def host_rows(self, row0, nrows):
        """Rows of the DEVICE copy of A pulled back to the host."""
        return self.ctx.get_matrix_rows(row0, nrows)

    @property
    def T(self):
        return self.H


class ShardedDenseMatrixMap(DenseMatrixMap):
    """Dense matrix split into contiguous ROW BLOCKS over several devices driven from this one process -- the single-call
    multi-GPU form of the operator of fasta/linalg.py:41:

        op = ShardedDenseMatrixMap(A, devices=[0, 1, 2, 3])            # or .synthetic(m, n, seed, scale, devices=[...])
        fasta(op, ls.f, ls.gradf, reg.g, reg.prox, x0)                 # b and z are split like the rows, x / g are replicated

    Each iteration is one local launch per device, ONE sum of the A_k^T r_k partials (with the loss sums riding along) and one
    host synchronisation (csrc/fasta_hip.hip:dense_step).  `devices` all different: one GPU per block, the sum is a grouped
    RCCL all-reduce over xGMI; all equal (e.g. [0] * 8): every block on that GPU, summed in block order by an in-library
    kernel -- the same arithmetic on a one-GPU box.  Results differ from the unsharded operator only by the order of the
    float64 sums (~1e-15 relative)."""

    def __init__(self, A=None, devices=(0, 0), tuning=None, storage="f64", _defer=False, _rccl_shell=False):
        """`_rccl_shell` (tests): with a single device id, still build the multi-device form -- one row block whose exchange is the
        grouped ncclAllReduce on a communicator from ncclCommInitAll (FH_CREATE_RCCL_SHELL)."""
        devices = [int(d) for d in devices]
        self.devices = devices
        DenseMatrixMap.__init__(self, A, device=devices[0], tuning=tuning, storage=storage, _defer=_defer, _devices=devices,
                                _rccl_shell=_rccl_shell)

    @classmethod
    def synthetic(cls, m, n, seed, scale, devices=(0, 0), tuning=None, storage="f64"):
        """The whole (m, n) counter-based synthetic matrix, every device generating its own row block in HBM."""
        from .synthetic import synth_coef
        self = cls(devices=devices, storage=storage, _defer=True)
        self._tune(tuning)
        self.ctx.generate_matrix(m, n, 0, seed, synth_coef(scale))
        self.Vshape, self.Wshape = (n,), (m,)
        self.shape = (m, n)
        self.rows = (0, m)
        return self

    def row_blocks(self):
        """[(row0, rows)] of every block, in device-list order."""
        return [self.ctx.shard(k)[1:] for k in range(self.ctx.shard_count())]


class GradDivMap(_DeviceMap):
    """A = div : (H, W, 2) -> (H, W) and A^H = grad, periodic (examples/tv_denoising.py:26-63)."""

    def __init__(self, image_shape, device=0):
        H, W = image_shape
        _DeviceMap.__init__(self, (H, W, 2), (H, W), device)
        self.ctx.set_stencil(H, W)
        self.image_shape = (H, W)
