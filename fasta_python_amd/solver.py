"""`fasta()` -- the FBS driver, host control in Python, all vector work in fused HIP kernels.

Drop-in for the reference call (fasta/__init__.py:38-53):

    fasta(A, f, gradf, g, proxg, x0, **options)            # 6-positional core form
    fasta(A, At, f, gradf, g, proxg, x0, **options)        # 7-positional form used by the examples

with the reference's options, defaults and `Convergence` result.  Per iteration ONE kernel is launched -- the one-pass step
(`fh_step` / `fh_step_accel`: both directions from a single read of A, or one stencil sweep) -- wherever the operator shape has one
and it pays, otherwise K-fwd and K-adj, plus one more launch per backtrack; the 16 float64 scalars of the FH_S_* block that the launch
returns drive the reference's branch decisions (backtracking test :200, restart test :231, Barzilai-Borwein rule :253-270, stop rule :308).
Those decisions are taken, by default, by the LIBRARY's own host-side loop (`HipContext.iterate`, csrc/fh_host_iterate.h: the arithmetic
of `FBSolver.step` below, bit for bit, without the interpreter between two launches); `driver="python"` keeps them here,
`driver="device"` moves the whole loop into a persistent launch where a kernel for it exists (csrc/fh_run.h).
Iterates, gradients and residual vectors never leave HBM unless `record_iterates` / `func` ask.

Which loop runs is decided by the operand TYPES alone (`_recognise`), never by whether a GPU happens to be there:
  * device-recognisable operands (matrix / DenseMatrixMap / GradDivMap + tagged loss + tagged prox) run the HIP loop
    below and RAISE when libfasta_hip.so or the GPU is missing -- there is no CPU fallback for them;
  * anything else -- Python closures, a callable pair, `A=None`, a host LinearMap: the forms the reference's own
    examples pass -- cannot execute inside a kernel and runs the generic host loop (generic.py, reference semantics,
    bit-identical to the reference on the same NumPy);
  * `backend="hip"` insists on the device loop (TypeError for unrecognisable operands), `backend="numpy"` on the host one.
"""

import math
import warnings
from time import time

import numpy as np

from . import hip, stopping
from .linalg import DenseMatrixMap, GradDivMap, LinearMap, _DeviceMap
from .losses import LeastSquares, LogisticLoss
from .proximal import NoProx, ProxTag

__all__ = ["fasta", "Convergence", "FBSolver", "EPSILON"]

PAIR_MAX_ELEMENTS = 1 << 26     # dense operators up to 64 Mi elements (<= ~0.1 ms per launch) take K-fwd + K-adj under one sync
EPSILON = 1E-12      # fasta/__init__.py:32


def _sqrt(x):
    """sqrt of a device sum of squares as np.float64: same IEEE result as np.sqrt on the scalar, a fraction of its call cost
    (the arguments are sums of squares or NaN, never negative)."""
    return np.float64(math.sqrt(x))


class Convergence:
    """Result record with the reference's ten public attributes (fasta/__init__.py:323-351)."""

    def __init__(self, residuals, norm_residuals, stepsizes, backtracks, times, iteration_count, solution,
                 objectives=None, iterates=None, function_hist=None):
        self.residuals = residuals
        self.norm_residuals = norm_residuals
        self.stepsizes = stepsizes
        self.backtracks = backtracks
        self.times = times
        self.iteration_count = iteration_count
        self.solution = solution
        self.objectives = objectives
        self.iterates = iterates
        self.function_hist = function_hist


def _tag_of(obj, cls):
    """A tagged object passed directly, or as a bound method (`ls.f`, `reg.prox`, ...)."""
    if isinstance(obj, cls):
        return obj
    owner = getattr(obj, "__self__", None)
    return owner if isinstance(owner, cls) else None


def _unrecognised(A, At, f, gradf, g, proxg):
    """None when the seven operands can run on the device, else the reason they cannot (a sentence)."""
    if not isinstance(A, (np.ndarray, _DeviceMap)):
        return ("operator A is not device-resident (pass a 2-D float64 ndarray, a linalg.DenseMatrixMap / "
                "LinearMap.from_matrix(A), or a linalg.GradDivMap); arbitrary Python callables cannot run inside the fused HIP kernels")
    loss_f, loss_g = _tag_of(f, (LeastSquares, LogisticLoss)), _tag_of(gradf, (LeastSquares, LogisticLoss))
    if loss_f is None or loss_f is not loss_g:
        return "f and gradf must be the `.f` / `.gradf` of one losses.LeastSquares(b) or losses.LogisticLoss(b) object"
    if g is None and proxg is None:
        return None
    prox, owner_g = _tag_of(proxg, ProxTag), _tag_of(g, ProxTag)
    if prox is None or (owner_g is not None and owner_g is not prox) or (owner_g is None and g is not None):
        return ("g and proxg must be the `.g` / `.prox` of one proximal.* tag object "
                "(Shrink, NonNeg, LinfProx, L1Ball, Box, TVDualBall)")
    return None


def _recognise(A, At, f, gradf, g, proxg, x0):
    """Map the reference's seven operands onto device objects (call only when `_unrecognised` returned None)."""
    if isinstance(A, np.ndarray):
        if A.ndim != 2:
            raise AssertionError("matrix operator must be 2-D")            # linalg.py:40
        if isinstance(At, np.ndarray) and At.shape != A.shape[::-1]:
            raise AssertionError("At must have the transposed shape of A")
        A = DenseMatrixMap(A)
    loss_f = _tag_of(f, (LeastSquares, LogisticLoss))
    prox = NoProx() if (g is None and proxg is None) else _tag_of(proxg, ProxTag)      # :88-90
    if tuple(x0.shape) != A.Vshape:
        raise AssertionError(f"x0 has shape {x0.shape}, operator expects {A.Vshape}")   # linalg.py:58
    if loss_f.b.shape != A.Wshape:
        raise AssertionError(f"b has shape {loss_f.b.shape}, operator produces {A.Wshape}")
    return A, loss_f, prox


class FBSolver:
    """The FBS loop as an object: `setup()` then `step()` until it returns True.  `fasta()` wraps it;
    bench.py drives it directly so that warm-up and timed steps are the same code path."""

    def __init__(self, A, loss, prox, x0, adaptive=True, accelerate=False, verbose=True, max_iters=1000,
                 tolerance=1e-5, stop_rule=stopping.hybrid_residual, L=None, tau0=None, backtrack=True,
                 stepsize_shrink=None, window=10, max_backtracks=20, restart=True, evaluate_objective=False,
                 record_iterates=False, func=None, *, fused="auto", device_iters="auto", driver=None):
        """Reference options (fasta/__init__.py:42-53) plus two build-only, keyword-only switches:
        fused = "auto" | True | False -- use the one-pass kernel (`HipContext.step` / `step_accel`, csrc/fh_fused.h)
        when the operator shape supports it (with acceleration: dense operator only).  It is speculative: the
        launch assumes the step is accepted; when the backtracking test fails the iteration falls back to
        K-fwd/K-adj (same results).
        driver -- who takes the reference's decisions between two launches:
          "library" (default)  the library's own host-side loop (`HipContext.iterate`, csrc/fh_host_iterate.h): the launches, scalars and
                     float64 decisions of `step()` below, bit for bit, without the interpreter's 17-19 us per iteration -- for every
                     operator, prox and sharding form;
          "device"   the loop itself on the device (`HipContext.run`, csrc/fh_run.h: one persistent launch per call; backtracking, FISTA,
                     the Barzilai-Borwein rule and the stop rule decided there) where the context has a kernel for it (dense float64,
                     n <= 6144, separable prox), the library's host-side loop elsewhere and after a launch of it timed out;
          "python"   this class's `step()` per iteration (Python between all launches: rounds 1-5).
        device_iters -- iterations per call of the library / device loop: "auto" (calls sized to ~50 ms, at most 256 iterations) or K > 0.
        Shorthands kept from round 5: `device_iters=K` alone means driver="device", `device_iters=0` means driver="python".
        The library and device loops are taken only when nothing needs Python between two iterations -- `stop_rule` is one of the four of
        fasta/stopping.py, no `func`, no `record_iterates`, window <= 64; otherwise `step()` runs (same results).  `verbose` lines are
        printed after each call from its history records: same text and order as the reference's (:302-306), in bursts.  `times[i]`
        within one call are interpolated between its start and its end."""
        self.A, self.loss, self.prox = A, loss, prox
        self.fused_opt = fused
        self.ctx = A.ctx
        self.x0 = np.asarray(x0, dtype=np.float64)
        self.shape = self.x0.shape
        self.n = int(self.x0.size)
        if stepsize_shrink is None and backtrack:                       # :92-97
            stepsize_shrink = 0.2 if adaptive else 0.5
        self.adaptive, self.accelerate, self.verbose = adaptive, accelerate, verbose
        self.max_iters, self.tolerance, self.stop_rule = max_iters, tolerance, stop_rule
        self.L, self.tau0 = L, tau0
        self.backtrack, self.stepsize_shrink = backtrack, stepsize_shrink
        self.window, self.max_backtracks, self.restart = window, max_backtracks, restart
        self.evaluate_objective, self.record_iterates, self.func = evaluate_objective, record_iterates, func
        if device_iters is None or device_iters == "auto":
            device_iters = -1
        device_iters = int(device_iters)
        if driver is None:
            driver = "python" if device_iters == 0 else ("device" if device_iters > 0 else "library")
        if driver not in ("library", "device", "python"):
            raise ValueError('driver must be "library", "device" or "python"')
        self.driver = driver
        self.device_iters = device_iters if device_iters > 0 else -1        # -1: calls sized by time
        self.device_steps = 0                  # iterations that ran inside persistent launches (fh_run)
        self.library_steps = 0                 # iterations driven by the library's host-side loop (fh_iterate)

    # ------------------------------------------------------------------------------------------
    def setup(self):
        c = self.ctx
        self.loss.bind(c)
        fval = self.loss.f_from_device
        self._fval = fval
        c.set_prox(self.prox.kind, self.prox.mu, self.prox.lo, self.prox.hi)
        L, tau0 = self.L, self.tau0
        probes = not L or not tau0
        if probes:                                                      # :100-113
            p1 = np.random.randn(*self.shape)                           # same two global-RNG draws
            p2 = np.random.randn(*self.shape)
            c.set_vector(hip.VEC_T0, p1)
            c.set_vector(hip.VEC_T1, p2)
        c.set_vector(hip.VEC_X0, self.x0)
        if probes and hasattr(c, "setup"):
            # probes and the initial pass (:135-137) in ONE call: one read of a dense A where the set-up kernel (csrc/fh_setup.h) has a shape
            s = c.setup()
            L = _sqrt(s[hip.S_DG2]) / _sqrt(s[hip.S_DX2])
            tau0 = (2 / L) / 10
        else:
            if probes:
                c.gradient_at(hip.VEC_T0, hip.VEC_T2)
                c.gradient_at(hip.VEC_T1, hip.VEC_T3)
                L = np.float64(c.diff_norm(hip.VEC_T2, hip.VEC_T3)) / np.float64(c.diff_norm(hip.VEC_T0, hip.VEC_T1))
                tau0 = (2 / L) / 10
            s = c.init()                                                # :135-137
        if not tau0:                                                    # :115-116
            tau0 = 1 / L
        self.L, self.tau0 = L, tau0

        if self.verbose:                                                # :118-120
            print("Initializing FASTA...\n")
            print("Iteration #\tResidual\tStepsize\tAccel. param\tBacktracks\tObjective")

        K = self.max_iters
        self.residuals = np.zeros(K)                                    # :123-127
        self.norm_residuals = np.zeros(K)
        self.stepsizes = np.zeros(K)
        self.f_hist = np.zeros(K + 1)
        self.times = np.zeros(K + 1)
        self.total_backtracks = 0

        f1 = fval(s[hip.S_FSQ])
        self.f_hist[0] = f1
        self.objectives = self.iterates = self.function_hist = None
        if self.evaluate_objective:                                     # :141-143
            self.objectives = np.zeros(K + 1)
            self.objectives[0] = f1 + self.prox.g_from_sums(s[hip.S_GSUM], s[hip.S_GMAX])
        if self.record_iterates:                                        # :145-147
            self.iterates = np.zeros((K + 1,) + self.shape)
            self.iterates[0] = self.x0
        if self.func:                                                   # :149-151
            self.function_hist = np.zeros(K + 1)
            self.function_hist[0] = self.func(self.x0)
        kind = c.fused_agree() if self.fused_opt is not False else 0
        # How the two halves of an iteration reach the device:
        #   "always"      one-pass kernel for every launch of the loop, backtracking retries included: where it costs no more
        #                 than K-fwd alone -- the stencil, and the dense operator from n = 16384 (kind 1: 65536^2 5.0 vs 4.9 ms)
        #   "speculative" one-pass kernel on a dense operator below that size (kind 3): a rejected step wastes the A^T half, so
        #                 back off for a few iterations after a backtrack.  Since round 5 (two-level grid barrier and final arrival:
        #                 a one-pass launch of a small matrix costs 18-27 us) this is what fused="auto" takes at EVERY size down to
        #                 64 x 128 -- 2048^2: 21 300 it/s against 10 400 for the pair below (profiles/r05_crossover.txt)
        #   "pair"        K-fwd and K-adj enqueued back to back under one synchronisation (operators without a one-pass kernel, or whose
        #                 co-residency probe said no; no acceleration): the host round trip is what costs there; speculative in the same way
        #   None          fh_fwd, decide, fh_adj
        self.mode = None
        if kind in (1, 2):
            self.mode = "always"
        elif kind == 3 and self.fused_opt in (True, "auto"):
            self.mode = "speculative"
        elif (self.fused_opt == "auto" and kind in (0, 3) and not self.accelerate and hasattr(c, "fwd_adj")
              and getattr(self.A, "shape", None) is not None and len(self.A.shape) == 2
              and self.A.shape[0] * self.A.shape[1] <= PAIR_MAX_ELEMENTS):
            self.mode = "pair"
        self.use_fused = self.mode in ("always", "speculative")
        self.fused_always = self.mode == "always"
        if self.fused_opt is True and not self.use_fused:
            raise ValueError("fused=True needs a dense operator with n <= 262144, or a stencil operator")
        self._spec_cooldown = 0            # iterations to wait after a backtrack before speculating again
        self._fused_backoff = 64           # iterations to stay on K-fwd / K-adj after a one-pass launch timed out (doubles per failure)
        self._fused_retry_at = None
        self.fused_steps = 0
        self.pair_steps = 0
        self.alpha1 = 1.0                                               # :157
        self.max_residual = -np.inf                                     # :165-167
        self.best_quality = np.inf
        self.tau_next = tau0
        self.i = 0
        self.done = False
        self._run_opts = self._library_loop_options() if self.driver != "python" else None
        # fh_run (the loop on the device) only on request and where the context has a kernel for it
        self._use_run = bool(self._run_opts is not None and self.driver == "device" and self.fused_opt is not False
                             and hasattr(self.ctx, "run_supported") and self.ctx.run_supported())
        self._run_backoff, self._run_retry_at = 64, None      # after a grid-barrier timeout of fh_run: iterations on the host-side loop before the next try
        self._chunk = 8                                       # device_iters = "auto": iterations of the next library call
        return self

    # ------------------------------------------------------------------------------------------
    def _library_loop_options(self):
        """hip.RunOpts when the library can drive the loop (fh_run / fh_iterate: see __init__), else None."""
        rules = {getattr(stopping, name): k for k, name in enumerate(hip.STOP_RULES)}
        c = self.ctx
        if (self.stop_rule not in rules or self.func or self.record_iterates or not hasattr(c, "iterate")
                or not 1 <= int(self.window) <= hip.RUN_WINDOW_MAX):
            return None
        o = hip.RunOpts()
        o.adaptive, o.accelerate, o.backtrack, o.restart = int(bool(self.adaptive)), int(bool(self.accelerate)), int(bool(self.backtrack)), int(bool(self.restart))
        o.evaluate_objective, o.stop_rule, o.window, o.max_backtracks = int(bool(self.evaluate_objective)), rules[self.stop_rule], int(self.window), int(self.max_backtracks)
        o.stepsize_shrink = float(self.stepsize_shrink) if self.backtrack else 1.0
        o.tolerance = float(self.tolerance)
        o.launch_mode = hip.LAUNCH_MODES[self.mode]
        return o

    def _library_call(self, upto=None):
        """One call of the library's loop: fh_run (persistent launch) where enabled, else fh_iterate; adopts its histories and state.
        upto: do not go past this iteration index."""
        c, st, o = self.ctx, hip.RunState(), self._run_opts
        i = self.i
        st.tau_next, st.alpha1, st.max_residual, st.best_quality = self.tau_next, self.alpha1, self.max_residual, self.best_quality
        st.iteration, st.backtracks, st.stopped = i, self.total_backtracks, 0
        lo = max(i - self.window + 1, 0)
        for j in range(lo, i + 1):
            st.f_window[j % hip.RUN_WINDOW_MAX] = self.f_hist[j]
        if self._run_retry_at is not None and i >= self._run_retry_at:
            self._use_run, self._run_retry_at = True, None
        K = self.device_iters if self.device_iters > 0 else self._chunk
        K = min(K, (self.max_iters if upto is None else min(upto, self.max_iters)) - i)
        on_device = self._use_run
        if not on_device:       # the launch policy's memory travels with the state (csrc/fh_host_iterate.h)
            fused_mode = self.mode in ("always", "speculative")
            o.launch_mode = hip.LAUNCH_MODES[self.mode] if (self.use_fused or self._fused_retry_at is not None or not fused_mode) else hip.LAUNCH_SEPARATE
            st.spec_cooldown, st.onepass_backoff = self._spec_cooldown, self._fused_backoff
            st.onepass_off_until = -1 if (self.use_fused or self._fused_retry_at is None) else self._fused_retry_at
        t0 = time()
        h = c.run(K, o, st) if on_device else c.iterate(K, o, st)
        t1 = time()
        k = len(h)
        self.residuals[i:i + k], self.norm_residuals[i:i + k], self.stepsizes[i:i + k] = h[:, 0], h[:, 1], h[:, 2]
        self.f_hist[i + 1:i + k + 1] = h[:, 3]
        if self.evaluate_objective:
            self.objectives[i + 1:i + k + 1] = h[:, 4]
        self.times[i:i + k] = t0 + (t1 - t0) * np.arange(k) / max(k, 1)         # (one call: the iterations' stamps are interpolated)
        self.tau_next, self.alpha1, self.max_residual, self.best_quality = st.tau_next, st.alpha1, st.max_residual, st.best_quality
        self.total_backtracks = int(st.backtracks)
        self.i = int(st.iteration)
        if on_device:
            self.device_steps += k
            if st.stopped == 3:        # a grid barrier of the persistent launch timed out: state and context are those of the last completed iteration
                warnings.warn(f"device-side loop (fh_run) timed out after {k} iterations of its launch (workgroups not co-resident?): "
                              f"continuing on the library's host-side loop for the next {self._run_backoff} iterations")
                self._use_run, self._run_retry_at = False, self.i + self._run_backoff
                self._run_backoff *= 2
        else:
            self.library_steps += k
            self.fused_steps += int(st.onepass_launches)
            self.pair_steps += int(st.pair_launches)
            self._spec_cooldown, self._fused_backoff = int(st.spec_cooldown), int(st.onepass_backoff)
            if st.onepass_timeouts:
                warnings.warn(f"fused one-pass kernel disabled until iteration {int(st.onepass_off_until)}: its team hand-off timed out "
                              f"{int(st.onepass_timeouts)} time(s) in this call (workgroups not co-resident?); K-fwd / K-adj meanwhile")
            if fused_mode:
                self.use_fused = st.onepass_off_until < 0 and o.launch_mode != hip.LAUNCH_SEPARATE
                self._fused_retry_at = None if st.onepass_off_until < 0 else int(st.onepass_off_until)
            if self.device_iters < 0 and k == K and t1 > t0:        # "auto": size the next call to ~50 ms, 1..256 iterations
                self._chunk = int(min(256, max(1, 0.05 * k / (t1 - t0))))
        if self.verbose:                                            # :302-306, from the records: same text, same order, per call
            for q in range(k):
                if int(h[q, 7]) & 2:
                    print("Restarted acceleration.")
                print("[{:<6}]\t{:e}\t{:e}\t{:e}\t{:6}\t{:e}".format(
                    i + q, self.residuals[i + q], self.stepsizes[i + q], h[q, 6] if self.accelerate else 0.0,
                    int(h[q, 5]) if self.backtrack else 0, self.objectives[i + q] if self.evaluate_objective else 0))
        self.done = st.stopped == 1 or self.i >= self.max_iters

    def _forward(self, tau, one_pass):
        """(fwd scalars, adj scalars or None): the one-pass kernel when enabled and asked for, else K-fwd alone."""
        c = self.ctx
        if self.mode == "pair" and one_pass:
            self.pair_steps += 1
            s = c.fwd_adj(tau)
            return s, s
        if self.use_fused and one_pass:
            try:
                if self.accelerate:     # the launch decides the restart itself (:231); step() mirrors it afterwards
                    a1 = (1 + np.sqrt(1 + 4 * self.alpha1 ** 2)) / 2
                    s = c.step_accel(tau, (self.alpha1 - 1) / a1, self.restart)
                else:
                    s = c.step(tau)
                self.fused_steps += 1
                return s, s
            except hip.HipTimeout as exc:                               # bounded-spin timeout ONLY: any other status propagates
                # (row-sharded runs all-reduce the timeout word with g1, so every rank gets here in the same iteration
                # and the ranks' collective sequences stay aligned.)  The usual cause is a co-tenant on the GPU -- the launch
                # needs every CU at once -- which may be gone later: fall back to K-fwd / K-adj now and try the one-pass
                # kernel again after `_fused_backoff` iterations, doubling the wait each time it fails (a failed try costs
                # the ~0.4 s of the bounded spins).
                warnings.warn(f"fused one-pass kernel disabled for the next {self._fused_backoff} iterations: {exc}")
                self.use_fused = False
                self._fused_retry_at = self.i + self._fused_backoff
                self._fused_backoff *= 2
        return c.fwd(tau), None

    # ------------------------------------------------------------------------------------------
    def step(self):
        """One FBS iteration (fasta/__init__.py:171-312).  Returns True when the stop rule fires."""
        c, i = self.ctx, self.i
        self.times[i] = time()                                          # :173
        tau = self.tau_next                                             # :178

        fval = self._fval
        if self._fused_retry_at is not None and i >= self._fused_retry_at:      # (every rank holds the same counters: same decision)
            self.use_fused, self._fused_retry_at = True, None
        speculate = self.fused_always or self._spec_cooldown == 0
        s, a = self._forward(tau, speculate)                            # :181-188  (K-fwd, or K-fwd + K-adj in one pass)
        if not speculate:
            self._spec_cooldown -= 1
        f1 = fval(s[hip.S_FSQ])
        bt = 0
        if self.backtrack:                                              # :195-217
            M = self.f_hist[max(i - self.window + 1, 0):(i + 1)].max()     # (ndarray.max: np.max's wrapper costs 3 us per call)
            while (f1 - (M + s[hip.S_DXG0] + _sqrt(s[hip.S_DX2]) ** 2 / (2 * tau)) > EPSILON
                   and bt < self.max_backtracks):
                tau *= self.stepsize_shrink
                # :207-213: K-fwd again (a speculative K-adj, if any, is void) -- or the one-pass kernel again where it
                # costs what K-fwd costs
                s, a = self._forward(tau, self.fused_always)
                f1 = fval(s[hip.S_FSQ])
                bt += 1
            self.total_backtracks += bt
            if bt:
                self._spec_cooldown = 8

        alpha0, coef = 0.0, 0.0
        if self.accelerate:                                             # :220-238
            alpha0 = self.alpha1
            if self.restart and s[hip.S_RDOT] > 1E-30:
                alpha0 = 1.0
                if self.verbose:
                    print("Restarted acceleration.")
            self.alpha1 = (1 + np.sqrt(1 + 4 * alpha0 ** 2)) / 2
            coef = (alpha0 - 1) / self.alpha1

        if a is None:
            a = c.adj(tau, self.accelerate, coef)                       # :242-248  (K-adj)
        if self.accelerate:
            f1 = fval(a[hip.S_FSQ_ADJ])                                 # :245
            xh2, gsum, gmax = a[hip.S_XH2_ADJ], a[hip.S_GSUM_ADJ], a[hip.S_GMAX_ADJ]
        else:
            xh2, gsum, gmax = s[hip.S_XH2], s[hip.S_GSUM], s[hip.S_GMAX]

        tau_next = tau                                                  # :249
        dx_norm = _sqrt(s[hip.S_DX2])
        if self.adaptive:                                               # :253-270
            dot = a[hip.S_DXDG]
            tau_s = dx_norm ** 2 / dot
            tau_m = max(dot / _sqrt(a[hip.S_DG2]) ** 2, 0)
            tau_next = tau_m if 2 * tau_m > tau_s else tau_s - .5 * tau_m
            if tau_next <= 0 or np.isinf(tau_next) or np.isnan(tau_next):
                tau_next = tau * 1.5
        self.tau_next = tau_next

        self.residuals[i] = dx_norm / tau                               # :272
        normalizer = max(_sqrt(s[hip.S_G02]), _sqrt(xh2) / tau) + EPSILON          # :274
        self.stepsizes[i] = tau
        self.norm_residuals[i] = self.residuals[i] / normalizer
        self.f_hist[i + 1] = f1
        self.max_residual = max(self.max_residual, self.residuals[i])   # :281

        if self.evaluate_objective:                                     # :284-289
            self.objectives[i + 1] = f1 + self.prox.g_from_sums(gsum, gmax)
            quality = self.objectives[i + 1]
        else:
            quality = self.residuals[i]
        better = bool(quality < self.best_quality)                      # :298-300
        if better:
            self.best_quality = quality
        c.commit(save_best=better)                                      # x0 <- x1, g0 <- g1 (:176-177)

        if self.record_iterates or self.func:                           # :291-296 (D2H of x1, off the fast path)
            x1 = c.get_vector(hip.VEC_X0, self.n).reshape(self.shape)
            if self.record_iterates:
                self.iterates[i + 1, ...] = x1
            if self.func:
                self.function_hist[i + 1] = self.func(x1)

        if self.verbose:                                                # :302-306
            print("[{:<6}]\t{:e}\t{:e}\t{:e}\t{:6}\t{:e}".format(
                i, self.residuals[i], self.stepsizes[i], alpha0 if self.accelerate else 0.0,
                bt if self.backtrack else 0, self.objectives[i] if self.evaluate_objective else 0))

        self.i = i + 1                                                  # :308-312
        self.done = bool(self.stop_rule(i, self.residuals[i], self.norm_residuals[i], self.max_residual,
                                        self.tolerance)) or self.i >= self.max_iters
        return self.done

    def advance(self, k):
        """Up to k more iterations, by whoever drives this solve (the library's loop in calls that end at iteration i + k, else `step()`):
        what `run()` does, in instalments -- bench.py times exactly K iterations with it.  Returns True once the solve has ended."""
        target = min(self.i + int(k), self.max_iters)
        with np.errstate(all="ignore"):
            while self.i < target and not self.done:
                if self._run_opts is not None:
                    self._library_call(upto=target)
                else:
                    self.step()
        return self.done

    def run(self):
        with warnings.catch_warnings():
            # the reference relies on float64 inf/nan semantics (e.g. 0/0 in the BB rule once converged)
            warnings.simplefilter("ignore", RuntimeWarning)
            with np.errstate(all="ignore"):
                while self.i < self.max_iters and not self.done:
                    if self._run_opts is not None:
                        self._library_call()
                    elif self.step():
                        break
        return self.result()

    def result(self):
        self.times[self.i] = time()                                     # :315
        solution = self.ctx.get_vector(hip.VEC_BEST, self.n).reshape(self.shape)
        conv = Convergence(self.residuals, self.norm_residuals, self.stepsizes, self.total_backtracks, self.times,
                           self.i, solution, self.objectives, self.iterates, self.function_hist)
        conv.device_steps = self.device_steps         # (build-only diagnostics: iterations that ran inside persistent launches -- fh_run --
        conv.library_steps = self.library_steps       #  and iterations driven by the library's host-side loop -- fh_iterate)
        return conv


def fasta(A, *operands, backend="auto", **options):
    """Run FASTA.  Same positional forms, keyword options and defaults as the reference (fasta/__init__.py:38-53);
    returns `Convergence`.  Build-only keywords:
      backend = "auto"  -- device loop for device-recognisable operands (raises if the GPU path is unavailable), generic
                           host loop for operands that cannot run in a kernel (closures, callable pair, None, host LinearMap);
                "hip"   -- device loop or TypeError;   "numpy" -- generic host loop (operands are called as given);
      fused   = "auto" | True | False -- one-pass kernel policy of the device loop (see FBSolver);
      driver  = "library" | "device" | "python", device_iters = "auto" | K -- who takes the decisions between two launches: the library's
                host-side loop (default), the loop itself on the device where there is a kernel for it, or Python per iteration; and how
                many iterations one call of the first two runs (see FBSolver)."""
    if len(operands) == 6:
        At, f, gradf, g, proxg, x0 = operands
    elif len(operands) == 5:
        At = None
        f, gradf, g, proxg, x0 = operands
    else:
        raise TypeError("fasta() takes (A, f, gradf, g, proxg, x0) or (A, At, f, gradf, g, proxg, x0)")
    if backend not in ("auto", "hip", "numpy"):
        raise ValueError('backend must be "auto", "hip" or "numpy"')
    why_not = "backend='numpy' was requested" if backend == "numpy" else _unrecognised(A, At, f, gradf, g, proxg)
    if why_not is not None:
        if backend == "hip":
            raise TypeError("fasta(backend='hip'): " + why_not)
        from .generic import HostFBS, host_map
        options.pop("fused", None)                                      # device-loop policies, meaningless on the host
        options.pop("device_iters", None)
        options.pop("driver", None)
        x0 = np.asarray(x0)
        return HostFBS(host_map(A, At, x0), f, gradf, g, proxg, x0, **options).setup().run()
    x0 = np.asarray(x0, dtype=np.float64)
    owns = isinstance(A, np.ndarray)
    A, loss, prox = _recognise(A, At, f, gradf, g, proxg, x0)
    try:
        return FBSolver(A, loss, prox, x0, **options).setup().run()
    finally:
        if owns:
            A.close()
