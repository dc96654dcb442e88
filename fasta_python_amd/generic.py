"""The generic (host) FBS loop: `fasta()` for operands that cannot run inside a kernel.

Arbitrary Python closures -- the form every reference example passes (`f = lambda z: ...`,
examples/sparse_least_squares.py:41-44), a callable pair `A, At` (examples/tv_denoising.py:99), `A = None` for
the identity (examples/svm.py:74) or a plain host `LinearMap` -- are evaluated where they live: on the host, on
NumPy arrays, with the reference's semantics (fasta/__init__.py:38-53, 88-320).  This is NOT a fallback for the
device path: `fasta()` picks it from the operand TYPES alone (solver.py:_recognise); device-recognisable operands
never come here unless the caller says `backend="numpy"`, and they raise when the GPU path is unavailable.

Same floating-point expressions as the reference, so a run is bit-identical to it on the same NumPy/BLAS
(tests/test_generic_cpu.py drives every fixture captured from the reference through this loop and compares
bitwise).  Organised as a small state machine -- trial point, line search, extrapolation, step-size rule,
bookkeeping -- that shares `Convergence`, option defaults and the verbose format with the device driver.
"""

from time import time

import numpy as np
from numpy import linalg as la

from . import stopping
from .linalg import LinearMap

__all__ = ["HostFBS", "host_map"]

EPSILON = 1E-12          # fasta/__init__.py:32
RESTART_EPS = 1E-30      # fasta/__init__.py:231


def host_map(A, At, x0):
    """The operator forms the reference's callers use -> a host LinearMap (shape asserts of linalg.py:58,60 kept)."""
    if isinstance(A, LinearMap):
        return A
    if A is None:                                    # svm.py:74, nn_factorization.py:63: identity
        return LinearMap.identity(np.shape(x0))
    if isinstance(A, np.ndarray):                    # sparse_least_squares.py:46 passes the raw matrix and its transpose
        assert A.ndim == 2                           # linalg.py:40
        if isinstance(At, np.ndarray):
            assert At.shape == A.shape[::-1]
            return LinearMap(lambda x: A @ x, lambda y: At @ y, (A.shape[1],), (A.shape[0],))
        return LinearMap(lambda x: A @ x, lambda y: A.T @ y, (A.shape[1],), (A.shape[0],))      # linalg.py:41
    if callable(A) and callable(At):                 # tv_denoising.py:99: bare functions, codomain found by probing
        return LinearMap(A, At, np.shape(x0), np.shape(A(np.zeros(np.shape(x0)))))
    raise TypeError("fasta(): operator A must be a LinearMap, a 2-D ndarray, a callable pair (A, At) or None")


def _flat_dot(u, v):
    return np.real(u.ravel().T @ v.ravel())          # the form of fasta/__init__.py:200, :255


def _norm(v):
    return la.norm(v.ravel())


class _Point:
    """A trial point of one iteration: forward point, prox output, its image and the smooth value there."""
    __slots__ = ("xhat", "x", "dx", "z", "f")


class HostFBS:
    """FBS on the host.  `setup()` then `step()` until it returns True, or `run()`."""

    def __init__(self, A, f, gradf, g, proxg, x0, adaptive=True, accelerate=False, verbose=True, max_iters=1000,
                 tolerance=1e-5, stop_rule=stopping.hybrid_residual, L=None, tau0=None, backtrack=True,
                 stepsize_shrink=None, window=10, max_backtracks=20, restart=True, evaluate_objective=False,
                 record_iterates=False, func=None):
        if g is None:                                                   # :88-90 plain gradient descent
            g, proxg = (lambda x: 0), (lambda x, t: x)
        if stepsize_shrink is None and backtrack:                       # :92-97
            stepsize_shrink = 0.2 if adaptive else 0.5
        self.A, self.f, self.gradf, self.g, self.proxg, self.x_start = A, f, gradf, g, proxg, x0
        self.adaptive, self.accelerate, self.verbose = adaptive, accelerate, verbose
        self.max_iters, self.tolerance, self.stop_rule = max_iters, tolerance, stop_rule
        self.L, self.tau0 = L, tau0
        self.backtrack, self.shrink_by = backtrack, stepsize_shrink
        self.window, self.max_backtracks, self.restart = window, max_backtracks, restart
        self.evaluate_objective, self.record_iterates, self.func = evaluate_objective, record_iterates, func

    # ---- phases -----------------------------------------------------------------------------------
    def _gradient(self, z):
        return self.A.H(self.gradf(z))

    def _estimate_step(self):
        """:100-116: L from two global-RNG probes whenever L or tau0 is missing; tau0 = (2/L)/10."""
        L, tau0 = self.L, self.tau0
        if not L or not tau0:
            shape = self.x_start.shape
            p, q = np.random.randn(*shape), np.random.randn(*shape)
            gp, gq = self._gradient(self.A(p)), self._gradient(self.A(q))
            L = _norm(gp - gq) / _norm(p - q)
            tau0 = (2 / L) / 10
        if not tau0:
            tau0 = 1 / L
        self.L, self.tau0 = L, tau0

    def _trial(self, tau):
        """:181-188 / :207-213: forward step, prox, image, smooth value."""
        t = _Point()
        t.xhat = self.x - tau * self.grad
        t.x = self.proxg(t.xhat, tau)
        t.dx = t.x - self.x
        t.z = self.A(t.x)
        t.f = self.f(t.z)
        return t

    def _accepts(self, t, tau, ceiling):
        """:200: the non-monotone sufficient-decrease test."""
        return not (t.f - (ceiling + _flat_dot(t.dx, self.grad) + _norm(t.dx) ** 2 / (2 * tau)) > EPSILON)

    def _extrapolate(self, t):
        """:220-245 FISTA: returns (x1, z1, f1, alpha0); rotates the prox-output history."""
        xa_prev, za_prev = self.xa, self.za
        self.xa, self.za = t.x, t.z
        alpha0 = self.alpha
        if self.restart and (self.x - t.x).ravel().T @ (t.x - xa_prev).ravel() > RESTART_EPS:
            alpha0 = 1.0
            if self.verbose:
                print("Restarted acceleration.")
        self.alpha = (1 + np.sqrt(1 + 4 * alpha0 ** 2)) / 2
        x1 = t.x + (alpha0 - 1) / self.alpha * (self.xa - xa_prev)
        z1 = t.z + (alpha0 - 1) / self.alpha * (self.za - za_prev)
        return x1, z1, self.f(z1), alpha0

    def _next_stepsize(self, t, tau, grad1):
        """:253-270 Barzilai-Borwein with the reference's safeguards."""
        dg = grad1 + (t.xhat - self.x) / tau
        d = _flat_dot(t.dx, dg)
        tau_s = _norm(t.dx) ** 2 / d
        tau_m = max(d / _norm(dg) ** 2, 0)
        nxt = tau_m if 2 * tau_m > tau_s else tau_s - .5 * tau_m
        if nxt <= 0 or np.isinf(nxt) or np.isnan(nxt):
            nxt = tau * 1.5
        return nxt

    # ---- driver -----------------------------------------------------------------------------------
    def setup(self):
        self._estimate_step()
        if self.verbose:                                                # :118-120
            print("Initializing FASTA...\n")
            print("Iteration #\tResidual\tStepsize\tAccel. param\tBacktracks\tObjective")
        K = self.max_iters
        self.residuals, self.norm_residuals, self.stepsizes = np.zeros(K), np.zeros(K), np.zeros(K)
        self.f_hist, self.times = np.zeros(K + 1), np.zeros(K + 1)
        self.x = self.x_start                                           # :132-137
        self.tau_next = self.tau0
        z = self.A(self.x)
        self.f_hist[0] = fx = self.f(z)
        self.grad = self._gradient(z)
        self.objectives = self.iterates = self.function_hist = None
        if self.evaluate_objective:
            self.objectives = np.zeros(K + 1)
            self.objectives[0] = fx + self.g(self.x)
        if self.record_iterates:
            self.iterates = np.zeros((K + 1,) + self.x_start.shape)
            self.iterates[0] = self.x
        if self.func:
            self.function_hist = np.zeros(K + 1)
            self.function_hist[0] = self.func(self.x)
        if self.accelerate:                                             # :154-157
            self.xa, self.za, self.alpha = self.x, z, 1.0
        self.total_backtracks = 0
        self.max_residual, self.best_quality, self.best = -np.inf, np.inf, self.x_start
        self.i = 0
        return self

    def step(self):
        i = self.i
        self.times[i] = time()
        tau = self.tau_next
        t = self._trial(tau)
        bt = 0
        if self.backtrack:                                              # :195-217
            ceiling = np.max(self.f_hist[max(i - self.window + 1, 0):(i + 1)])
            while not self._accepts(t, tau, ceiling) and bt < self.max_backtracks:
                tau *= self.shrink_by
                t = self._trial(tau)
                bt += 1
            self.total_backtracks += bt
        x1, z1, f1, alpha0 = t.x, t.z, t.f, 0.0
        if self.accelerate:
            x1, z1, f1, alpha0 = self._extrapolate(t)
        grad1 = self._gradient(z1)                                      # :248
        self.tau_next = self._next_stepsize(t, tau, grad1) if self.adaptive else tau

        self.residuals[i] = _norm(t.dx) / tau                           # :272-281
        normalizer = max(_norm(self.grad), _norm(x1 - t.xhat) / tau) + EPSILON
        self.stepsizes[i] = tau
        self.norm_residuals[i] = self.residuals[i] / normalizer
        self.f_hist[i + 1] = f1
        self.max_residual = max(self.max_residual, self.residuals[i])
        if self.evaluate_objective:                                     # :284-289
            self.objectives[i + 1] = f1 + self.g(x1)
            quality = self.objectives[i + 1]
        else:
            quality = self.residuals[i]
        if self.record_iterates:
            self.iterates[i + 1, ...] = x1
        if self.func:
            self.function_hist[i + 1] = self.func(x1)
        if quality < self.best_quality:                                 # :298-300
            self.best, self.best_quality = x1, quality
        if self.verbose:                                                # :302-306
            print("[{:<6}]\t{:e}\t{:e}\t{:e}\t{:6}\t{:e}".format(
                i, self.residuals[i], self.stepsizes[i], alpha0 if self.accelerate else 0.0,
                bt if self.backtrack else 0, self.objectives[i] if self.evaluate_objective else 0))
        self.x, self.grad = x1, grad1                                   # :176-177 of the next round
        self.i = i + 1
        return bool(self.stop_rule(i, self.residuals[i], self.norm_residuals[i], self.max_residual, self.tolerance))

    def run(self):
        from .solver import Convergence
        while self.i < self.max_iters:
            if self.step():
                break
        self.times[self.i] = time()                                     # :315
        return Convergence(self.residuals, self.norm_residuals, self.stepsizes, self.total_backtracks, self.times,
                           self.i, self.best, self.objectives, self.iterates, self.function_hist)
