"""Proximal operators: device-tagged objects + the reference's function names (fasta/proximal.py).

A tagged object bundles `g` and `proxg` of one regulariser and tells `fasta()` which fused device
prox to run (`kind`, `mu`).  Use it for both arguments:

    reg = Shrink(mu);  fasta(A, At, ls.f, ls.gradf, reg.g, reg.prox, x0)

With device-recognised operands `fasta()` never CALLS these objects: it maps the tag to the prox fused
into K-fwd / the one-pass kernel.  Called on a host array -- by a user's closure, by the generic host loop
(generic.py), by the reference itself -- a tag and the module-level functions (`shrink`, `project_Linf_ball`,
`project_L1_ball`, `project_Lnuc_ball`) are ordinary callables with the reference's NumPy semantics
(fasta/proximal.py:12-67), so the reference's `solve()` bodies run unmodified against this module.
`device_prox(tag, x, t)` evaluates the same prox with the DEVICE kernel on a host array, through a scratch
context cached per (device, shape); the -m gpu tests use it to pin the kernels to the reference's known
answers.
"""

import numpy as np

from . import hip

__all__ = ["Shrink", "NonNeg", "LinfProx", "L1Ball", "Box", "TVDualBall", "NoProx", "device_prox", "release_scratch",
           "shrink", "project_Linf_ball", "project_L1_ball", "project_Lnuc_ball"]


class ProxTag:
    kind = hip.PROX_IDENTITY
    mu = 0.0
    lo = hi = 0.0
    step_scaled = True          # prox parameter is t*mu (True) or mu alone (False, constraint sets)

    def g_from_sums(self, gsum, gmax):
        """g(x) from the device reductions sum|x_i| and max|x_i|."""
        return 0

    # ordinary callables on host arrays (reference semantics); the device loop never calls them ----
    def prox(self, x, t):
        return x

    def __call__(self, x, t):
        return self.prox(x, t)

    def g(self, x):
        return 0

    def prox_on_device(self, x, t, device=0):
        """The fused device prox applied to a host array (scratch context cached per shape)."""
        return device_prox(self, x, t, device)


class NoProx(ProxTag):
    """g = None branch of fasta/__init__.py:88-90 (plain gradient descent)."""


class Shrink(ProxTag):
    """g(x) = mu*||x||_1, proxg(x, t) = shrink(x, t*mu)  (examples/sparse_least_squares.py:43-44)."""
    kind = hip.PROX_SHRINK

    def __init__(self, mu):
        self.mu = float(mu)

    def g_from_sums(self, gsum, gmax):
        return self.mu * gsum

    def prox(self, x, t):
        return shrink(x, t * self.mu)

    def g(self, x):
        return self.mu * np.linalg.norm(np.ravel(x), 1)


class NonNeg(ProxTag):
    """g = 0 on x >= 0, proxg = max(x, 0)  (examples/nn_least_squares.py:41-42)."""
    kind = hip.PROX_NONNEG

    def prox(self, x, t):
        return np.maximum(x, 0)


class LinfProx(ProxTag):
    """g(x) = mu*||x||_inf, proxg(x, t) = project_Linf_ball(x, t*mu)
    (examples/democratic_representation.py:41-42)."""
    kind = hip.PROX_LINF

    def __init__(self, mu):
        self.mu = float(mu)

    def g_from_sums(self, gsum, gmax):
        return self.mu * gmax

    def prox(self, x, t):
        return project_Linf_ball(x, t * self.mu)

    def g(self, x):
        return self.mu * np.linalg.norm(np.ravel(x), np.inf)


class L1Ball(ProxTag):
    """g = 0, proxg(x, t) = project_L1_ball(x, mu) -- radius mu, independent of t (examples/lasso.py:44-45)."""
    kind = hip.PROX_L1BALL
    step_scaled = False

    def __init__(self, mu):
        self.mu = float(mu)

    def prox(self, x, t):
        return project_L1_ball(x, self.mu)


class Box(ProxTag):
    """Clip to [lo, hi] (examples/svm.py:71)."""
    kind = hip.PROX_BOX

    def __init__(self, lo, hi):
        self.lo, self.hi = float(lo), float(hi)

    def prox(self, x, t):
        return np.minimum(np.maximum(x, self.lo), self.hi)


class TVDualBall(ProxTag):
    """Per-pixel projection of 2-vectors onto the unit ball (examples/tv_denoising.py:89-96)."""
    kind = hip.PROX_TVBALL

    def prox(self, Y, t):
        lengths = np.linalg.norm(Y, axis=-1)
        return Y / np.maximum(lengths, 1)[..., None]


# ---- the device prox on host arrays -----------------------------------------------------------------
_scratch = {}            # (device, "dense", n) or (device, "tv", H, W) -> operator holding a scratch HipContext


def _scratch_op(device, x, tv):
    """One scratch operator per (device, shape), kept for the life of the process (`release_scratch()` frees them):
    a prox inside a caller's loop costs two small copies and one launch, not a context create/destroy."""
    from .linalg import DenseMatrixMap, GradDivMap
    key = (device, "tv") + tuple(x.shape[:2]) if tv else (device, "dense", x.size)
    op = _scratch.get(key)
    if op is None:
        if len(_scratch) >= 8:                                   # bounded: drop the oldest shape
            _scratch.pop(next(iter(_scratch))).close()
        op = GradDivMap(x.shape[:2], device=device) if tv else DenseMatrixMap(np.zeros((1, x.size)), device=device)
        if not tv:
            op.ctx.set_loss_lsq(np.zeros(1))
            op.ctx.set_vector(hip.VEC_G0, np.zeros(x.size))
        _scratch[key] = op
    return op


def release_scratch():
    """Free the cached scratch contexts of `device_prox`."""
    while _scratch:
        _scratch.popitem()[1].close()


def device_prox(tag, x, t, device=0):
    """prox_{t g}(x) for a host array through the DEVICE kernels: one K-fwd with x0 := x and a zero gradient
    gives xprox = prox(x, t)."""
    x = np.asarray(x, dtype=np.float64)
    flat = x.ravel()
    tv = tag.kind == hip.PROX_TVBALL
    if tv:
        assert x.ndim == 3 and x.shape[-1] == 2
    ctx = _scratch_op(device, x, tv).ctx
    ctx.set_prox(tag.kind, tag.mu, tag.lo, tag.hi)
    ctx.set_vector(hip.VEC_X0, flat)
    if tv:
        # the stencil path recomputes g0 = grad(A x0 - b); with b := A x it is exactly zero, so xhat = x
        ctx.set_loss_lsq(ctx.apply(flat))
        ctx.init()
    ctx.fwd(float(t))
    return ctx.get_vector(hip.VEC_XPROX, flat.size).reshape(x.shape)


# ---- the reference's function names (host arrays, NumPy) ------------------------------------------------
def shrink(x, t):
    """Soft-threshold by t (fasta/proximal.py:58-67); small negatives come back as -0.0, as in the reference."""
    return np.sign(x) * np.maximum(np.abs(x) - t, 0)


def project_Linf_ball(x, t):
    """The prox of t*||.||_inf, as the reference implements it (fasta/proximal.py:12-31): clip |x| at the level
    alpha = max_k (sum of the k largest |x_i| - t)/k, or return float64 zeros when that level is not positive."""
    mags = np.abs(x)
    ranked = mags.copy()
    ranked[::-1].sort()                                          # descending, in place through the reversed view
    alpha = np.max((np.cumsum(ranked) - t) / np.arange(1, len(x) + 1))
    if alpha > 0:
        return np.minimum(mags, alpha) * np.sign(x)
    return np.zeros(len(x))


def project_L1_ball(x, t):
    """Euclidean projection onto {||x||_1 <= t} by Moreau's identity (fasta/proximal.py:34-41)."""
    return x - project_Linf_ball(x, t)


def project_Lnuc_ball(X, t):
    """Soft-threshold the singular values of X by t (fasta/proximal.py:44-55).  Host only: a dense SVD is not on the
    MI355X hot path."""
    U, s, Vh = np.linalg.svd(X, full_matrices=False)
    return (U * shrink(s, t)) @ Vh
