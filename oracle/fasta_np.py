"""NumPy oracle: a CPU restatement of the reference FBS solver and its operator library.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity: PINNED bit-for-bit against fixtures
captured from the reference core by oracle/make_golden.py (tests/test_oracle_golden.py).

Every routine cites the reference lines it restates (paths relative to /root/reference).  The
floating-point expressions are kept in the reference's evaluation order so that, with the same
NumPy/BLAS underneath, histories come out identical to the last bit; the control structure is
re-organised (one helper per phase of the iteration) rather than transcribed.
"""

from time import time

import numpy as np
from numpy import linalg as la

EPS = 1e-12            # fasta/__init__.py:32
RESTART_EPS = 1e-30    # fasta/__init__.py:231


# --------------------------------------------------------------------------------------------
# linear maps (fasta/linalg.py:13-69)
# --------------------------------------------------------------------------------------------
class LinearMap:
    """Pair (forward, adjoint) with fixed domain/codomain shapes; fasta/linalg.py:23-69."""

    def __init__(self, fwd, adj, Vshape, Wshape):
        self.fwd, self.adj = fwd, adj
        self.Vshape, self.Wshape = tuple(Vshape), tuple(Wshape)

    def __call__(self, v):                       # linalg.py:52-61 (asserts on both sides)
        assert v.shape == self.Vshape
        w = self.fwd(v)
        assert w.shape == self.Wshape
        return w

    @property
    def H(self):                                 # linalg.py:63-69
        return LinearMap(self.adj, self.fwd, self.Wshape, self.Vshape)

    @staticmethod
    def from_matrix(M):                          # linalg.py:37-41: A @ x and A.T @ x
        assert M.ndim == 2
        return LinearMap(lambda x: M @ x, lambda y: M.T @ y, (M.shape[1],), (M.shape[0],))

    @staticmethod
    def identity(shape):                         # linalg.py:43-50
        return LinearMap(lambda x: x, lambda x: x, shape, shape)


def coerce_map(A, At, x0):
    """Accept the operator forms the reference's examples pass."""
    if isinstance(A, LinearMap):
        return A
    if A is None:                                # svm.py:74 style: identity
        return LinearMap.identity(x0.shape)
    if isinstance(A, np.ndarray):                # sparse_least_squares.py:46: raw matrices
        return LinearMap.from_matrix(A)
    if callable(A) and callable(At):             # tv_denoising.py:99: bare functions
        Wshape = np.shape(A(np.zeros(x0.shape)))
        return LinearMap(A, At, x0.shape, Wshape)
    raise TypeError("unsupported operator form")


# --------------------------------------------------------------------------------------------
# proximal operators (fasta/proximal.py:12-67, examples)
# --------------------------------------------------------------------------------------------
def shrink(x, t):                                # proximal.py:58-67
    return np.sign(x) * np.maximum(np.abs(x) - t, 0)


def prox_linf(x, t):
    """proximal.py:12-31 (`project_Linf_ball`, which is the prox of t*||.||_inf)."""
    mag = np.abs(x)
    desc = mag.copy()
    desc[::-1].sort()                            # descending, in place through the reversed view
    level = np.max((np.cumsum(desc) - t) / np.arange(1, len(x) + 1))
    if level > 0:
        return np.minimum(mag, level) * np.sign(x)
    return np.zeros(len(x))


def project_l1(x, t):                            # proximal.py:34-41 (Moreau complement)
    return x - prox_linf(x, t)


def nonneg(x, t=None):                           # examples/nn_least_squares.py:42
    return np.maximum(x, 0)


def tv_dual_ball(Y, t=None):                     # examples/tv_denoising.py:89-96
    mags = la.norm(Y, axis=Y.ndim - 1)
    mags = np.maximum(mags, 1)
    return Y / mags[..., np.newaxis]


# --------------------------------------------------------------------------------------------
# stop rules (fasta/stopping.py:6-51) -- signature (i, resid, norm_resid, max_resid, tol)
# --------------------------------------------------------------------------------------------
def residual(i, resid, norm_resid, max_resid, tol):          # stopping.py:6-15
    return resid < tol


def norm_residual(i, resid, norm_resid, max_resid, tol):     # stopping.py:18-27
    return norm_resid < tol


def ratio_residual(i, resid, norm_resid, max_resid, tol):    # stopping.py:30-39
    return resid / max_resid < tol


def hybrid_residual(i, resid, norm_resid, max_resid, tol):   # stopping.py:42-51
    return resid / max_resid < tol or norm_resid < tol


# --------------------------------------------------------------------------------------------
# result record (fasta/__init__.py:323-351)
# --------------------------------------------------------------------------------------------
class Convergence:
    FIELDS = ("residuals", "norm_residuals", "stepsizes", "backtracks", "times",
              "iteration_count", "solution", "objectives", "iterates", "function_hist")

    def __init__(self, **kw):
        for name in self.FIELDS:
            setattr(self, name, kw.get(name))
        # oracle-only extras (not part of the reference surface): operator pass counters
        self.passes = kw.get("passes")


# --------------------------------------------------------------------------------------------
# the solver (fasta/__init__.py:38-320)
# --------------------------------------------------------------------------------------------
def _sqnorm_like_ref(v):
    """`la.norm(v.ravel())**2`, the form used at fasta/__init__.py:200 and :258."""
    return la.norm(v.ravel()) ** 2


def estimate_lipschitz(A, gradf, shape):
    """fasta/__init__.py:100-113: two global-RNG probes (x1 then x2), L and tau0 = (2/L)/10."""
    p1 = np.random.randn(*shape)
    p2 = np.random.randn(*shape)
    d1 = A.H(gradf(A(p1)))
    d2 = A.H(gradf(A(p2)))
    L = la.norm((d1 - d2).ravel()) / la.norm((p1 - p2).ravel())
    return L, (2 / L) / 10


def fasta(A, *rest, adaptive=True, accelerate=False, verbose=False, max_iters=1000, tolerance=1e-5,
          stop_rule=hybrid_residual, L=None, tau0=None, backtrack=True, stepsize_shrink=None,
          window=10, max_backtracks=20, restart=True, evaluate_objective=False,
          record_iterates=False, func=None):
    """Oracle FBS solve.  Positional forms: (A, f, gradf, g, proxg, x0) -- fasta/__init__.py:38-40 --
    or (A, At, f, gradf, g, proxg, x0) -- the examples' form, e.g. sparse_least_squares.py:46.

    NOTE: `verbose` defaults to False here (reference default True, :42); the print format is
    restated from :118-120 and :302-306 when enabled.
    """
    if len(rest) == 6:
        At, f, gradf, g, proxg, x0 = rest
    elif len(rest) == 5:
        At = None
        f, gradf, g, proxg, x0 = rest
    else:
        raise TypeError("fasta() takes 6 or 7 positional arguments")
    A = coerce_map(A, At, x0)
    count = {"A": 0, "AH": 0}
    fwd_raw, adj_raw = A.fwd, A.adj

    def _cf(v):
        count["A"] += 1
        return fwd_raw(v)

    def _ca(v):
        count["AH"] += 1
        return adj_raw(v)
    A = LinearMap(_cf, _ca, A.Vshape, A.Wshape)

    if g is None:                                 # :88-90 gradient descent option
        g = lambda x: 0
        proxg = lambda x, t: x
    if stepsize_shrink is None and backtrack:     # :92-97
        stepsize_shrink = 0.2 if adaptive else 0.5

    if not L or not tau0:                         # :100 -- either missing => both recomputed
        L, tau0 = estimate_lipschitz(A, gradf, x0.shape)
    if not tau0:                                  # :115-116 (unreachable, kept for fidelity)
        tau0 = 1 / L

    if verbose:                                   # :118-120
        print("Initializing FASTA...\n")
        print("Iteration #\tResidual\tStepsize\tAccel. param\tBacktracks\tObjective")

    resid_h = np.zeros(max_iters)                 # :123-127
    nresid_h = np.zeros(max_iters)
    tau_h = np.zeros(max_iters)
    f_h = np.zeros(max_iters + 1)
    stamps = np.zeros(max_iters + 1)

    x_new = x0                                    # :132-137
    tau_next = tau0
    z_new = A(x_new)
    f_new = f(z_new)
    grad_new = A.H(gradf(z_new))
    f_h[0] = f_new

    obj_h = it_h = fn_h = None
    if evaluate_objective:                        # :141-143
        obj_h = np.zeros(max_iters + 1)
        obj_h[0] = f_new + g(x_new)
    if record_iterates:                           # :145-147
        it_h = np.zeros((max_iters + 1,) + x0.shape)
        it_h[0] = x_new
    if func:                                      # :149-151
        fn_h = np.zeros(max_iters + 1)
        fn_h[0] = func(x_new)

    if accelerate:                                # :154-157
        xa_new, za_new, alpha_new = x_new, z_new, 1.0

    n_backtracks = 0
    peak_resid = -np.inf                          # :165-167
    best_q, best_x = np.inf, x0

    k = 0
    while k < max_iters:                          # :171
        stamps[k] = time()                        # :173
        x_old, grad_old, tau = x_new, grad_new, tau_next          # :176-178

        # forward-backward step (:181-188)
        x_hat = x_old - tau * grad_new
        x_new = proxg(x_hat, tau)
        step = x_new - x_old
        z_new = A(x_new)
        f_new = f(z_new)

        # non-monotone backtracking (:195-217)
        bt = 0
        if backtrack:
            ceiling = np.max(f_h[max(k - window + 1, 0):(k + 1)])
            while (f_new - (ceiling + np.real(step.ravel().T @ grad_old.ravel())
                            + _sqnorm_like_ref(step) / (2 * tau)) > EPS) and bt < max_backtracks:
                tau *= stepsize_shrink
                x_hat = x_old - tau * grad_old
                x_new = proxg(x_hat, tau)
                step = x_new - x_old
                z_new = A(x_new)
                f_new = f(z_new)
                bt += 1
            n_backtracks += bt

        # FISTA extrapolation (:220-245)
        alpha_old = None
        if accelerate:
            xa_old, za_old = xa_new, za_new
            xa_new, za_new = x_new, z_new
            alpha_old = alpha_new
            if restart and (x_old - x_new).ravel().T @ (x_new - xa_old).ravel() > RESTART_EPS:
                alpha_old = 1.0
                if verbose:
                    print("Restarted acceleration.")
            alpha_new = (1 + np.sqrt(1 + 4 * alpha_old ** 2)) / 2
            x_new = x_new + (alpha_old - 1) / alpha_new * (xa_new - xa_old)
            z_new = z_new + (alpha_old - 1) / alpha_new * (za_new - za_old)
            f_new = f(z_new)

        grad_new = A.H(gradf(z_new))              # :248
        tau_next = tau                            # :249

        if adaptive:                              # Barzilai-Borwein, :253-270
            dgrad = grad_new + (x_hat - x_old) / tau
            inner = np.real(step.ravel().T @ dgrad.ravel())
            tau_s = _sqnorm_like_ref(step) / inner
            tau_m = max(inner / _sqnorm_like_ref(dgrad), 0)
            tau_next = tau_m if 2 * tau_m > tau_s else tau_s - .5 * tau_m
            if tau_next <= 0 or np.isinf(tau_next) or np.isnan(tau_next):
                tau_next = tau * 1.5

        # residuals and bookkeeping (:272-300)
        resid_h[k] = la.norm(step.ravel()) / tau
        scale = max(la.norm(grad_old.ravel()), la.norm((x_new - x_hat).ravel()) / tau) + EPS
        tau_h[k] = tau
        nresid_h[k] = resid_h[k] / scale
        f_h[k + 1] = f_new
        peak_resid = max(peak_resid, resid_h[k])
        if evaluate_objective:
            obj_h[k + 1] = f_new + g(x_new)
            q = obj_h[k + 1]
        else:
            q = resid_h[k]
        if record_iterates:
            it_h[k + 1, ...] = x_new
        if func:
            fn_h[k + 1] = func(x_new)
        if q < best_q:
            best_x, best_q = x_new, q

        if verbose:                               # :302-306 (prints obj_h[k], the previous one)
            print("[{:<6}]\t{:e}\t{:e}\t{:e}\t{:6}\t{:e}".format(
                k, resid_h[k], tau_h[k], alpha_old if accelerate else 0.0,
                bt if backtrack else 0, obj_h[k] if evaluate_objective else 0))

        k += 1                                    # :308-312 (increment happens on both paths)
        if stop_rule(k - 1, resid_h[k - 1], nresid_h[k - 1], peak_resid, tolerance):
            break

    stamps[k] = time()                            # :315
    return Convergence(residuals=resid_h, norm_residuals=nresid_h, stepsizes=tau_h,
                       backtracks=n_backtracks, times=stamps, iteration_count=k, solution=best_x,
                       objectives=obj_h, iterates=it_h, function_hist=fn_h, passes=dict(count))
