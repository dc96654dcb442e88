"""Proximal operators: device-tagged objects + the reference's function names (fasta/proximal.py).

A tagged object bundles `g` and `proxg` of one regulariser and tells `fasta()` which fused device
prox to run (`kind`, `mu`).  Use it for both arguments:

    reg = Shrink(mu);  fasta(A, At, ls.f, ls.gradf, reg.g, reg.prox, x0)

Calling a tagged prox on a host array evaluates it on the device through a scratch context; the
module-level functions (`shrink`, `project_Linf_ball`, `project_L1_ball`) do the same, keeping the
reference's names and argument meaning.  `project_Lnuc_ball` (dense SVD, fasta/proximal.py:44-55) is
out of scope for this build and raises.
"""

import numpy as np

from . import hip

__all__ = ["Shrink", "NonNeg", "LinfProx", "L1Ball", "Box", "TVDualBall", "NoProx",
           "shrink", "project_Linf_ball", "project_L1_ball", "project_Lnuc_ball"]


class ProxTag:
    kind = hip.PROX_IDENTITY
    mu = 0.0
    lo = hi = 0.0
    step_scaled = True          # prox parameter is t*mu (True) or mu alone (False, constraint sets)

    def g_from_sums(self, gsum, gmax):
        """g(x) from the device reductions sum|x_i| and max|x_i|."""
        return 0

    # host-array conveniences: run the device prox on a scratch identity problem -----------------
    def prox(self, x, t):
        return _device_prox(self, np.asarray(x, dtype=np.float64), float(t))

    __call__ = prox

    def g(self, x):
        x = np.asarray(x, dtype=np.float64).ravel()
        return self.g_from_sums(float(np.abs(x).sum()), float(np.abs(x).max(initial=0.0)))


class NoProx(ProxTag):
    """g = None branch of fasta/__init__.py:88-90 (plain gradient descent)."""


class Shrink(ProxTag):
    """g(x) = mu*||x||_1, proxg(x, t) = shrink(x, t*mu)  (examples/sparse_least_squares.py:43-44)."""
    kind = hip.PROX_SHRINK

    def __init__(self, mu):
        self.mu = float(mu)

    def g_from_sums(self, gsum, gmax):
        return self.mu * gsum


class NonNeg(ProxTag):
    """g = 0 on x >= 0, proxg = max(x, 0)  (examples/nn_least_squares.py:41-42)."""
    kind = hip.PROX_NONNEG


class LinfProx(ProxTag):
    """g(x) = mu*||x||_inf, proxg(x, t) = project_Linf_ball(x, t*mu)
    (examples/democratic_representation.py:41-42)."""
    kind = hip.PROX_LINF

    def __init__(self, mu):
        self.mu = float(mu)

    def g_from_sums(self, gsum, gmax):
        return self.mu * gmax


class L1Ball(ProxTag):
    """g = 0, proxg(x, t) = project_L1_ball(x, mu) -- radius mu, independent of t (examples/lasso.py:44-45)."""
    kind = hip.PROX_L1BALL
    step_scaled = False

    def __init__(self, mu):
        self.mu = float(mu)


class Box(ProxTag):
    """Clip to [lo, hi] (examples/svm.py:71)."""
    kind = hip.PROX_BOX

    def __init__(self, lo, hi):
        self.lo, self.hi = float(lo), float(hi)


class TVDualBall(ProxTag):
    """Per-pixel projection of 2-vectors onto the unit ball (examples/tv_denoising.py:89-96)."""
    kind = hip.PROX_TVBALL


def _device_prox(tag, x, t):
    """prox on the device: one K-fwd with x0 := x, g0 := 0 gives xprox = prox(x, t)."""
    from .linalg import GradDivMap
    flat = x.ravel()
    if tag.kind == hip.PROX_TVBALL:
        assert x.ndim == 3 and x.shape[-1] == 2
        with _Scratch(GradDivMap(x.shape[:2])) as op:
            # the stencil path recomputes g0 = grad(A x0 - b); with b := A x it is exactly zero, so xhat = x
            ctx = op.ctx
            ctx.set_loss_lsq(ctx.apply(flat))
            ctx.set_prox(tag.kind, tag.mu, tag.lo, tag.hi)
            ctx.set_vector(hip.VEC_X0, flat)
            ctx.init()
            ctx.fwd(t)
            return ctx.get_vector(hip.VEC_XPROX, flat.size).reshape(x.shape)
    ctx = hip.HipContext(0)
    try:
        ctx.set_matrix(np.zeros((1, flat.size)))
        return _one_prox(ctx, tag, flat, t).reshape(x.shape)
    finally:
        ctx.close()


class _Scratch:
    def __init__(self, op):
        self.op = op

    def __enter__(self):
        return self.op

    def __exit__(self, *exc):
        self.op.close()


def _one_prox(ctx, tag, flat, t):
    m, n = ctx.shape()
    ctx.set_loss_lsq(np.zeros(m))
    ctx.set_prox(tag.kind, tag.mu, tag.lo, tag.hi)
    ctx.set_vector(hip.VEC_X0, flat)
    ctx.set_vector(hip.VEC_G0, np.zeros(n))
    ctx.fwd(t)
    return ctx.get_vector(hip.VEC_XPROX, n)


# ---- the reference's function names ------------------------------------------------------------
def shrink(x, t):
    """Soft-threshold by t (fasta/proximal.py:58-67)."""
    return Shrink(1.0).prox(x, t)


def project_Linf_ball(x, t):
    """The prox of t*||.||_inf, as the reference implements it (fasta/proximal.py:12-31)."""
    return LinfProx(1.0).prox(x, t)


def project_L1_ball(x, t):
    """Euclidean projection onto {||x||_1 <= t} (fasta/proximal.py:34-41)."""
    return L1Ball(t).prox(x, 1.0)


def project_Lnuc_ball(X, t):
    raise NotImplementedError("nuclear-norm prox (dense SVD, fasta/proximal.py:44-55) is out of scope "
                              "for the MI355X FBS hot path (SURVEY.md section 8(a))")
