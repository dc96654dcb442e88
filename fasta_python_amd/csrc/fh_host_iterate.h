// fh_host_iterate.h -- fh_iterate: the FBS loop (fasta/__init__.py:171-312) driven from the HOST SIDE OF THE LIBRARY (round 6).
//
// fh_run (csrc/fh_run.h) keeps the loop on the device, for the shapes where a workgroup can own whole rows.  Every other operator -- a dense
// matrix of any width, the l-infinity / l1-ball prox kinds with their level search, the stencil, float32 storage, row blocks in this process
// or one rank of a row-sharded run -- still advances one launch (or two) per iteration, and between two launches somebody has to take the
// reference's branch decisions from the 16 scalars the launch returns.  Until round 5 that somebody was Python (fasta_python_amd/solver.py:
// FBSolver.step): 17-19 us per iteration of interpreter time, 14 % of an 8192^2 step and most of a 512 x 512 stencil sweep.  fh_iterate
// is that same driver in C++: it issues the same launches through the same entry points (fh_step / fh_step_accel / fh_fwd / fh_adj /
// fh_fwd_adj / fh_commit), reads the same scalar block after the same single synchronisation and takes the same decisions with the
// same float64 arithmetic -- the backtracking test (:195-217), FISTA restart and alpha recursion (:220-238), the Barzilai-Borwein rule
// (:253-270), residuals, best iterate (:272-300), the four built-in stop rules (fasta/stopping.py:6-51) -- and returns the histories of
// the iterations in fh_run's record format.  Same signature, same options / state / history structs as fh_run, so a caller picks per
// context: fh_run where it exists and wins, fh_iterate everywhere else.
//
// Bit for bit the Python driver: every expression below is the expression of solver.py in the same order, and where Python squares a
// NumPy float64 scalar (`x ** 2`) it calls the C library's pow(x, 2.0) exactly as NumPy's scalar power does -- which is NOT always x * x
// (glibc's pow is within 0.52 ulp, not correctly rounded: ~0.08 % of squares differ in the last bit); the call goes through a volatile
// function pointer because clang would otherwise fold pow(x, 2.0) into x * x.  tests/test_gpu_iterate.py compares the two drivers
// with `==` on every history of every fixture.
//
// The launch POLICY (which kernel serves a forward launch: the one-pass kernel always / speculatively / a K-fwd + K-adj pair under one
// synchronisation / separate launches) arrives in opts->launch_mode -- the caller has settled it, over the ranks if need be
// (fh_fused_agree) -- and the policy's memory (cool-down after a backtrack, back-off after a hand-off timeout) travels in the state, so
// the solve does not depend on where the calls are cut.  Policy never changes results, only which launches produce them.
#pragma once
#include <math.h>

static double (*volatile fh_libm_pow)(double, double) = pow;
static inline double it_sq(double x) { return fh_libm_pow(x, 2.0); }          // NumPy's float64_scalar ** 2

struct IterCtl {
  fh_ctx* c; const fh_run_opts* o; fh_run_state* st;
  double s[FH_NSCALARS];       // the scalar block of the latest forward launch (both halves after a one-pass / pair launch)
  bool have_adj;               // ... and it already carries this tau's K-adj half
};

// f1 from the device sum (losses.py: LeastSquares.f_from_device / LogisticLoss.f_from_device)
static inline double it_fval(const fh_ctx* c, double sum) {
#pragma clang fp contract(off)
  if (shard_of(const_cast<fh_ctx*>(c), 0)->loss_kind != LOSS_LSQ) return sum;
  return .5 * it_sq(sqrt(sum));
}
// g(x) from the device reductions (proximal.py: g_from_sums): mu * sum|x_i| (Shrink), mu * max|x_i| (LinfProx), else 0
static inline double it_gval(const fh_ctx* c, double gsum, double gmax) {
#pragma clang fp contract(off)
  const fh_ctx* s0 = shard_of(const_cast<fh_ctx*>(c), 0);
  if (s0->prox_kind == FH_PROX_SHRINK) return s0->mu * gsum;
  if (s0->prox_kind == FH_PROX_LINF) return s0->mu * gmax;
  return 0.0;
}

// solver.py:_forward -- the one-pass kernel when the policy has it and `one_pass` asks for it, else K-fwd alone
static int it_forward(IterCtl& k, double tau, bool one_pass, double alpha1) {
#pragma clang fp contract(off)
  fh_run_state* st = k.st;
  const int mode = k.o->launch_mode;
  k.have_adj = false;
  if (mode == FH_LAUNCH_PAIR && one_pass) {
    FH_TRY(fh_fwd_adj(k.c, tau, k.s));
    st->pair_launches += 1;
    k.have_adj = true;
    return 0;
  }
  const bool fused_on = (mode == FH_LAUNCH_ONEPASS_ALWAYS || mode == FH_LAUNCH_ONEPASS_SPECULATIVE) && st->onepass_off_until < 0;
  if (fused_on && one_pass) {
    int rc;
    if (k.o->accelerate) {      // the launch decides the restart itself (:231); the loop mirrors it afterwards
      const double a1 = (1.0 + sqrt(1.0 + 4.0 * it_sq(alpha1))) / 2.0;
      rc = fh_step_accel(k.c, tau, (alpha1 - 1.0) / a1, k.o->restart ? 1 : 0, k.s);
    } else rc = fh_step(k.c, tau, k.s);
    if (rc == 0 && k.s[15] == 0.0) { st->onepass_launches += 1; k.have_adj = true; return 0; }
    if (rc != 0 && rc != FH_E_TIMEOUT) return rc;      // bounded-spin timeouts ONLY: any other status propagates
    // the hand-off between the launch's workgroups timed out (a co-tenant on the GPU?): K-fwd / K-adj now, the one-pass kernel again after
    // `onepass_backoff` iterations, doubling the wait per failure.  (Row-sharded runs all-reduce the timeout word with g1: every rank gets
    // here in the same iteration, the ranks' collective sequences stay aligned.)
    st->onepass_off_until = (long long)st->iteration + st->onepass_backoff;
    st->onepass_backoff *= 2;
    st->onepass_timeouts += 1;
  }
  return fh_fwd(k.c, tau, k.s);
}

extern "C" int fh_iterate(fh_ctx* c, int max_steps, const fh_run_opts* o, fh_run_state* st, double* history, int* steps_done) {
#pragma clang fp contract(off)
  FH_TRY(check_ready(c, true));
  if (!o || !st || !history || !steps_done) return fail(FH_E_ARG, "fh_iterate: null argument");
  *steps_done = 0;
  if (max_steps < 0) return fail(FH_E_ARG, "fh_iterate: max_steps must be >= 0");
  if (o->window < 1 || o->window > FH_RUN_WINDOW_MAX) return fail(FH_E_ARG, "fh_iterate: window must be in [1, %d]", FH_RUN_WINDOW_MAX);
  if (o->stop_rule < 0 || o->stop_rule > 3) return fail(FH_E_ARG, "fh_iterate: stop_rule must be 0..3 (the four rules of fasta/stopping.py)");
  if (o->launch_mode < FH_LAUNCH_SEPARATE || o->launch_mode > FH_LAUNCH_PAIR) return fail(FH_E_ARG, "fh_iterate: unknown launch_mode %d", o->launch_mode);
  if (o->launch_mode == FH_LAUNCH_PAIR && o->accelerate) return fail(FH_E_ARG, "fh_iterate: the K-fwd + K-adj pair has no accelerated form");
  if (st->onepass_backoff <= 0) st->onepass_backoff = 64;
  IterCtl k;
  k.c = c; k.o = o; k.st = st; k.have_adj = false;
  const bool always = o->launch_mode == FH_LAUNCH_ONEPASS_ALWAYS;
  st->stopped = 0;
  for (int step = 0; step < max_steps; ++step) {
    const unsigned long long i = st->iteration;
    double tau = st->tau_next;                                                    // :178
    if (st->onepass_off_until >= 0 && (long long)i >= st->onepass_off_until) st->onepass_off_until = -1;      // (every rank holds the same counters: same decision)
    const bool speculate = always || st->spec_cooldown == 0;
    FH_TRY(it_forward(k, tau, speculate, st->alpha1));                            // :181-188  (K-fwd, or K-fwd + K-adj in one pass)
    if (!speculate) st->spec_cooldown -= 1;
    double f1 = it_fval(c, k.s[FH_S_FSQ]);
    int bt = 0;
    if (o->backtrack) {                                                           // :195-217
      const unsigned long long lo = i + 1ull > (unsigned long long)o->window ? i + 1ull - (unsigned long long)o->window : 0ull;
      double M = st->f_window[lo % FH_RUN_WINDOW_MAX];                            // f_hist[lo : i + 1].max() -- ndarray.max: a NaN anywhere wins
      for (unsigned long long j = lo + 1ull; j <= i; ++j) {
        const double v = st->f_window[j % FH_RUN_WINDOW_MAX];
        if (M != M) break;
        if (v != v || v > M) M = v;
      }
      while (f1 - (M + k.s[FH_S_DXG0] + it_sq(sqrt(k.s[FH_S_DX2])) / (2.0 * tau)) > 1E-12 && bt < o->max_backtracks) {
        tau *= o->stepsize_shrink;
        // :207-213: K-fwd again (a speculative K-adj, if any, is void) -- or the one-pass kernel again where it costs what K-fwd costs
        FH_TRY(it_forward(k, tau, always, st->alpha1));
        f1 = it_fval(c, k.s[FH_S_FSQ]);
        bt += 1;
      }
      if (bt) st->spec_cooldown = 8;
    }
    double alpha0 = 0.0, coef = 0.0, alpha1 = st->alpha1;
    bool restarted = false;
    if (o->accelerate) {                                                          // :220-238
      alpha0 = alpha1;
      if (o->restart && k.s[FH_S_RDOT] > 1E-30) { alpha0 = 1.0; restarted = true; }
      alpha1 = (1.0 + sqrt(1.0 + 4.0 * it_sq(alpha0))) / 2.0;
      coef = (alpha0 - 1.0) / alpha1;
    }
    // the forward half of the block is final now; K-adj (:242-248) overwrites only its own half
    const double dxg2 = k.s[FH_S_DX2], g02 = k.s[FH_S_G02];
    double xh2 = k.s[FH_S_XH2], gsum = k.s[FH_S_GSUM], gmax = k.s[FH_S_GMAX];
    if (!k.have_adj) {
      double a[FH_NSCALARS];
      FH_TRY(fh_adj(c, tau, o->accelerate ? 1 : 0, coef, a));
      for (int q = FH_S_DXDG; q < FH_NSCALARS; ++q) k.s[q] = a[q];
    }
    if (o->accelerate) {
      f1 = it_fval(c, k.s[FH_S_FSQ_ADJ]);                                         // :245
      xh2 = k.s[FH_S_XH2_ADJ]; gsum = k.s[FH_S_GSUM_ADJ]; gmax = k.s[FH_S_GMAX_ADJ];
    }
    double tau_next = tau;                                                        // :249
    const double dx_norm = sqrt(dxg2);
    if (o->adaptive) {                                                            // :253-270
      const double dot = k.s[FH_S_DXDG];
      const double tau_s = it_sq(dx_norm) / dot;
      const double q = dot / it_sq(sqrt(k.s[FH_S_DG2]));
      const double tau_m = 0.0 > q ? 0.0 : q;                                     // Python's max(q, 0)
      tau_next = (2.0 * tau_m > tau_s) ? tau_m : tau_s - .5 * tau_m;
      if (tau_next <= 0.0 || isinf(tau_next) || isnan(tau_next)) tau_next = tau * 1.5;
    }
    const double resid = dx_norm / tau;                                           // :272
    const double na = sqrt(g02), nb = sqrt(xh2) / tau;
    const double normalizer = (nb > na ? nb : na) + 1E-12;                        // max(a, b) + EPSILON  (:274)
    const double norm_resid = resid / normalizer;
    if (resid > st->max_residual) st->max_residual = resid;                       // :281  (Python's max(a, b): a unless b > a)
    double objective = 0.0, quality = resid;
    if (o->evaluate_objective) {                                                  // :284-289
      objective = f1 + it_gval(c, gsum, gmax);
      quality = objective;
    }
    const bool better = quality < st->best_quality;                               // :298-300
    if (better) st->best_quality = quality;
    FH_TRY(fh_commit(c, better ? 1 : 0));                                         // x0 <- x1, g0 <- g1 (:176-177)
    bool stop = false;                                                            // stopping.py:6-51
    if (o->stop_rule == 0) stop = resid < o->tolerance;
    else if (o->stop_rule == 1) stop = norm_resid < o->tolerance;
    else if (o->stop_rule == 2) stop = resid / st->max_residual < o->tolerance;
    else stop = (resid / st->max_residual < o->tolerance) || (norm_resid < o->tolerance);
    double* h = history + (size_t)step * FH_RUN_HIST;
    h[0] = resid; h[1] = norm_resid; h[2] = tau; h[3] = f1; h[4] = objective; h[5] = (double)bt; h[6] = alpha0;
    h[7] = (better ? 1.0 : 0.0) + (restarted ? 2.0 : 0.0);
    st->f_window[(i + 1ull) % FH_RUN_WINDOW_MAX] = f1;
    st->tau_next = tau_next; st->alpha1 = alpha1;
    st->backtracks += (uint64_t)bt;
    st->iteration = i + 1ull;
    *steps_done = step + 1;
    if (stop) { st->stopped = 1; break; }
  }
  return 0;
}
