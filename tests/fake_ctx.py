"""NumPy stand-in for `fasta_python_amd.hip.HipContext` -- TEST INFRASTRUCTURE.

It implements the C-ABI contract of include/fasta_hip.h (the scalar block of fh_init / fh_fwd / fh_adj /
fh_step, the vector ids, fh_commit's rotation and best-iterate rule) with the oracle's NumPy operators, so the
product's HOST logic (fasta_python_amd/solver.py: backtracking, restart, Barzilai-Borwein, the speculative
one-pass path, histories, best iterate) can be exercised by the CPU test tier against the reference-captured
golden fixtures.  It is never imported by the product."""
import numpy as np

from fasta_python_amd import hip
from fasta_python_amd.linalg import LinearMap, _DeviceMap
from oracle import fasta_np as fo
from oracle import problems as pr


class FakeContext:
    def __init__(self, fwd, adj, n_shape, m_shape, fused_kind=0):
        self.fwd_op, self.adj_op = fwd, adj
        self.n_shape, self.m_shape = tuple(n_shape), tuple(m_shape)
        self.fused_kind = fused_kind
        self.calls = {"fwd": 0, "adj": 0, "step": 0, "pair": 0}
        self.vec = {}
        self.loss = "lsq"
        self.prox_kind, self.mu, self.lo, self.hi = hip.PROX_IDENTITY, 0.0, 0.0, 0.0

    # ---- problem data ----
    def shape(self):
        return int(np.prod(self.m_shape)), int(np.prod(self.n_shape))

    def set_loss_lsq(self, b):
        self.b, self.loss = np.asarray(b, dtype=float).reshape(self.m_shape), "lsq"

    def set_loss_logistic(self, b):
        self.b, self.loss = np.asarray(b, dtype=float).reshape(self.m_shape), "logistic"

    def set_prox(self, kind, mu=0.0, lo=0.0, hi=0.0):
        self.prox_kind, self.mu, self.lo, self.hi = kind, mu, lo, hi

    def set_vector(self, which, v):
        self.vec[which] = np.array(v, dtype=float).reshape(self.n_shape)

    def get_vector(self, which, length):
        if which == hip.VEC_BEST:
            return self.best.ravel().copy()
        if which == hip.VEC_X0:
            return self.x0.ravel().copy()
        return self.vec[which].ravel().copy()

    # ---- maths ----
    def _f_sum(self, z):
        if self.loss == "logistic":
            return float(np.sum(np.log(1 + np.exp(z)) - (self.b == 1) * z))
        return float(np.sum((z - self.b) ** 2))

    def _gradf(self, z):
        if self.loss == "logistic":
            return -self.b / (1 + np.exp(self.b * z))
        return z - self.b

    def _prox(self, x, tau):
        k = self.prox_kind
        if k == hip.PROX_SHRINK:
            return fo.shrink(x, tau * self.mu)
        if k == hip.PROX_NONNEG:
            return np.maximum(x, 0)
        if k == hip.PROX_LINF:
            return fo.prox_linf(x, tau * self.mu)
        if k == hip.PROX_L1BALL:
            return fo.project_l1(x, self.mu)
        if k == hip.PROX_TVBALL:
            return fo.tv_dual_ball(x)
        if k == hip.PROX_BOX:
            return np.clip(x, self.lo, self.hi)
        return x

    @staticmethod
    def _gterms(x, s):
        s[hip.S_GSUM] = np.abs(x).sum()
        s[hip.S_GMAX] = np.abs(x).max(initial=0.0)

    # ---- solver steps ----
    def init(self):
        self.x0 = self.vec[hip.VEC_X0]
        self.z_acc = self.fwd_op(self.x0)
        self.zcur = self.z_acc
        self.g0 = self.adj_op(self._gradf(self.z_acc))
        self.xacc = self.x0
        self.best = self.x0
        s = np.zeros(hip.NSCALARS)
        s[hip.S_FSQ] = self._f_sum(self.z_acc)
        self._gterms(self.x0, s)
        return s

    def gradient_at(self, src, dst):
        self.vec[dst] = self.adj_op(self._gradf(self.fwd_op(self.vec[src])))

    def diff_norm(self, a, b):
        return float(np.linalg.norm((self.vec[a] - self.vec[b]).ravel()))

    def fwd(self, tau):
        self.calls["fwd"] += 1
        x0, g0 = self.x0, self.g0
        self.xhat = x0 - tau * g0
        self.xp = self._prox(self.xhat, tau)
        self.z1 = self.fwd_op(self.xp)
        dx = (self.xp - x0).ravel()
        s = np.zeros(hip.NSCALARS)
        s[hip.S_FSQ] = self._f_sum(self.z1)
        s[hip.S_DXG0] = dx @ g0.ravel()
        s[hip.S_DX2] = dx @ dx
        s[hip.S_XH2] = np.sum((self.xp - self.xhat) ** 2)
        s[hip.S_G02] = np.sum(g0 ** 2)
        s[hip.S_RDOT] = (x0 - self.xp).ravel() @ (self.xp - self.xacc).ravel()
        self._gterms(self.xp, s)
        self._fwd_scalars = s
        return s.copy()

    def adj(self, tau, accel=False, coef=0.0):
        self.calls["adj"] += 1
        z, x1 = self.z1, self.xp
        if accel:
            z = z + coef * (z - self.z_acc)
            x1 = x1 + coef * (x1 - self.xacc)
        self.accel, self.x1, self.z_ext = bool(accel), x1, z
        self.g1 = self.adj_op(self._gradf(z))
        dg = self.g1 + (self.xhat - self.x0) / tau
        dx = (self.xp - self.x0).ravel()
        s = self._fwd_scalars.copy()
        s[hip.S_DXDG] = dx @ dg.ravel()
        s[hip.S_DG2] = np.sum(dg ** 2)
        s[hip.S_FSQ_ADJ] = self._f_sum(z)
        s[hip.S_XH2_ADJ] = np.sum((x1 - self.xhat) ** 2)
        s[hip.S_GSUM_ADJ] = np.abs(x1).sum()
        s[hip.S_GMAX_ADJ] = np.abs(x1).max(initial=0.0)
        return s

    def fused_supported(self):
        return self.fused_kind

    def fused_agree(self):          # (no communicator: the agreed verdict is the local one)
        return self.fused_kind

    def step(self, tau):
        self.calls["step"] += 1
        self.fwd(tau)
        self.calls["fwd"] -= 1
        s = self.adj(tau)
        self.calls["adj"] -= 1
        return s

    def fwd_adj(self, tau):
        self.calls["pair"] += 1
        self.fwd(tau)
        self.calls["fwd"] -= 1
        s = self.adj(tau)
        self.calls["adj"] -= 1
        return s

    def step_accel(self, tau, coef, restart):
        self.calls["step"] += 1
        s = self.fwd(tau)
        self.calls["fwd"] -= 1
        applied = 0.0 if (restart and s[hip.S_RDOT] > 1E-30) else coef      # the launch's own restart rule (:231)
        a = self.adj(tau, True, applied)
        self.calls["adj"] -= 1
        return a

    # ---- the library's host-side loop (csrc/fh_host_iterate.h: fh_iterate), restated on this stand-in -----------------------------------
    def iterate(self, max_steps, o, st):
        """NumPy twin of fh_iterate: the same launches by the same policy, the expressions of fasta_python_amd/solver.py:FBSolver.step for the
        decisions, the same history records and state updates -- so that the CPU tier can drive `FBSolver._library_call` (history slices,
        state carried from call to call, verbose lines, launch counters) without a GPU."""
        import math
        sq = lambda v: np.float64(math.sqrt(v))
        fval = (lambda s_: np.float64(s_)) if self.loss == "logistic" else (lambda s_: .5 * sq(s_) ** 2)
        gval = {hip.PROX_SHRINK: lambda gs, gm: self.mu * gs, hip.PROX_LINF: lambda gs, gm: self.mu * gm}.get(self.prox_kind, lambda gs, gm: 0)
        hist = np.zeros((int(max_steps), hip.RUN_HIST))
        always = o.launch_mode == hip.LAUNCH_ONEPASS_ALWAYS
        if st.onepass_backoff <= 0:
            st.onepass_backoff = 64
        st.stopped = 0

        def forward(tau, one_pass, alpha1):
            if o.launch_mode == hip.LAUNCH_PAIR and one_pass:
                st.pair_launches += 1
                return self.fwd_adj(tau), True
            fused_on = o.launch_mode in (hip.LAUNCH_ONEPASS_ALWAYS, hip.LAUNCH_ONEPASS_SPECULATIVE) and st.onepass_off_until < 0
            if fused_on and one_pass:
                try:
                    if o.accelerate:
                        a1 = (1 + np.sqrt(1 + 4 * alpha1 ** 2)) / 2
                        s_ = self.step_accel(tau, (alpha1 - 1) / a1, bool(o.restart))
                    else:
                        s_ = self.step(tau)
                    st.onepass_launches += 1
                    return s_, True
                except hip.HipTimeout:
                    st.onepass_off_until = int(st.iteration) + st.onepass_backoff
                    st.onepass_backoff *= 2
                    st.onepass_timeouts += 1
            return self.fwd(tau), False

        done = 0
        for step in range(int(max_steps)):
            i = int(st.iteration)
            tau = np.float64(st.tau_next)
            if st.onepass_off_until >= 0 and i >= st.onepass_off_until:
                st.onepass_off_until = -1
            speculate = always or st.spec_cooldown == 0
            s, have_adj = forward(tau, speculate, np.float64(st.alpha1))
            if not speculate:
                st.spec_cooldown -= 1
            f1 = fval(s[hip.S_FSQ])
            bt = 0
            if o.backtrack:
                lo = max(i - o.window + 1, 0)
                M = np.array([st.f_window[j % hip.RUN_WINDOW_MAX] for j in range(lo, i + 1)]).max()
                while f1 - (M + s[hip.S_DXG0] + sq(s[hip.S_DX2]) ** 2 / (2 * tau)) > 1E-12 and bt < o.max_backtracks:
                    tau = tau * o.stepsize_shrink
                    s, have_adj = forward(tau, always, np.float64(st.alpha1))
                    f1 = fval(s[hip.S_FSQ])
                    bt += 1
                if bt:
                    st.spec_cooldown = 8
            alpha0, coef, alpha1, restarted = 0.0, 0.0, np.float64(st.alpha1), False
            if o.accelerate:
                alpha0 = alpha1
                if o.restart and s[hip.S_RDOT] > 1E-30:
                    alpha0, restarted = 1.0, True
                alpha1 = (1 + np.sqrt(1 + 4 * alpha0 ** 2)) / 2
                coef = (alpha0 - 1) / alpha1
            a = s if have_adj else self.adj(tau, bool(o.accelerate), coef)
            if o.accelerate:
                f1 = fval(a[hip.S_FSQ_ADJ])
                xh2, gsum, gmax = a[hip.S_XH2_ADJ], a[hip.S_GSUM_ADJ], a[hip.S_GMAX_ADJ]
            else:
                xh2, gsum, gmax = s[hip.S_XH2], s[hip.S_GSUM], s[hip.S_GMAX]
            tau_next = tau
            dx_norm = sq(s[hip.S_DX2])
            if o.adaptive:
                dot = a[hip.S_DXDG]
                tau_s = dx_norm ** 2 / dot
                tau_m = max(dot / sq(a[hip.S_DG2]) ** 2, 0)
                tau_next = tau_m if 2 * tau_m > tau_s else tau_s - .5 * tau_m
                if tau_next <= 0 or np.isinf(tau_next) or np.isnan(tau_next):
                    tau_next = tau * 1.5
            resid = dx_norm / tau
            norm_resid = resid / (max(sq(s[hip.S_G02]), sq(xh2) / tau) + 1E-12)
            st.max_residual = max(st.max_residual, resid)
            objective, quality = 0.0, resid
            if o.evaluate_objective:
                objective = f1 + gval(gsum, gmax)
                quality = objective
            better = bool(quality < st.best_quality)
            if better:
                st.best_quality = quality
            self.commit(save_best=better)
            ratio, normed = resid / st.max_residual < o.tolerance, norm_resid < o.tolerance
            stop = [resid < o.tolerance, normed, ratio, ratio or normed][o.stop_rule]
            hist[step] = (resid, norm_resid, tau, f1, objective, bt, alpha0, (1.0 if better else 0.0) + (2.0 if restarted else 0.0))
            st.f_window[(i + 1) % hip.RUN_WINDOW_MAX] = f1
            st.tau_next, st.alpha1 = tau_next, alpha1
            st.backtracks += bt
            st.iteration = i + 1
            done = step + 1
            if stop:
                st.stopped = 1
                break
        return hist[:done]

    def commit(self, save_best=False):
        self.xacc, self.z_acc = self.xp, self.z1          # FISTA history (pre-extrapolation values)
        self.x0, self.g0 = self.x1, self.g1
        if save_best:
            self.best = self.x1

    def close(self):
        pass


class FakeDenseMap(_DeviceMap):
    """Device-map look-alike over a host matrix (bypasses HipContext creation)."""

    def __init__(self, A, fused_kind=0):
        A = np.asarray(A, dtype=float)
        self.matrix = A
        self.shape = A.shape
        self.ctx = FakeContext(lambda x: A @ x, lambda y: A.T @ y, (A.shape[1],), (A.shape[0],), fused_kind)
        LinearMap.__init__(self, self.ctx.fwd_op, self.ctx.adj_op, (A.shape[1],), (A.shape[0],))


class FakeStencilMap(_DeviceMap):
    def __init__(self, image_shape, fused_kind=2):
        H, W = image_shape
        self.ctx = FakeContext(pr.div, pr.grad, (H, W, 2), (H, W), fused_kind)
        LinearMap.__init__(self, pr.div, pr.grad, (H, W, 2), (H, W))
