// fh_fused.h -- ONE-PASS FBS iteration for the dense operator: z = A xprox AND g1 = A^T grad f(z) from a
// single read of A (the two-launch path reads A twice: 2*m*n*8 bytes per iteration; this reads m*n*8).
//
// Why it is possible: g1 = sum_i a_i * r_i with r_i = grad f(a_i . xprox) -- row i is needed twice, first
// whole (the dot product), then again for the rank-1 update.  A row (up to 1 MiB) does not fit one CU beyond n = 4096, so
// a TEAM of 2, 4, 8 or 16 co-resident workgroups splits the columns (one workgroup owns whole rows up to n = 4096): each member keeps its piece of the row (PPT 16-byte
// pieces per lane) IN REGISTERS, publishes its partial dot product as one write-through (`sc1`) 8-byte store into
// the row's slot line(s), reads the whole line back with scalar loads until no slot holds the sentinel (bounded),
// sums the partials in member order, and applies r_i * (its row pieces) to its register-resident slice of g1.
// The prox'd x slice also lives in registers, computed once per launch.
//
// Two schedules of the row loop (template parameter PIPE; the host picks per shape, fh_host_launch.h:fused_shape_for):
//   PIPE = D >= 1  "exchange D trips ahead": in trip t the team posts row t+D and waits for row t, which was posted D
//             whole trips earlier, so the ~0.5 us hand-off is off the critical path; NB = 5-6 row buffers rotate
//             (rows t..t+D held until their updates, the rest prefetching).  PPT <= 8.  D = 1 for teams of 8,
//             D = 2 for teams of 16 (more members, more skew: 65536^2 5.25 -> 4.99 ms).
//   PIPE = 0  exchange in line (post row t, wait for row t, update): three buffers of PPT = 16 pieces are all the
//             512 registers hold; two rows of loads stay in flight during the exchange.
// What made the loop fast (each step measured, profiles/r01b_tune_fused.txt and r01d_fused_tuning.txt):
//   * raw `s_barrier` + lgkmcnt-only waits, branch-free unconditional (clamped) prefetches and b[r] via s_load, so that
//     hipcc's waitcnt pass keeps counted `vmcnt(N)` waits instead of draining every prefetched row each trip;
//   * the slot poll uses SCALAR loads: a vector poll's `vmcnt(0)` also waits for the wave's own prefetch (one HBM
//     latency per trip) and for its write-through store's acknowledgement;
//   * the post is issued by wave 1, the poll by wave 0; cross-lane sums by DPP/readlane instead of ds_bpermute.
//
//   grid  = (#CUs / TEAM) teams x TEAM members, 256 threads, 1 workgroup per CU (all 512 registers => 1 wave per
//           SIMD), so the whole grid is co-resident by construction; every spin is bounded and raises p.err
//           instead of hanging if that assumption is ever violated.
//   team t owns rows [t*rows_per_team, ...); member j owns 16-byte pieces [j*256*PPT, (j+1)*256*PPT).
//   After the rows: slice partials -> workspace, bounded grid barrier, then all workgroups sum the team
//   partials for their share of the columns in team order and run the n-side epilogue (same arithmetic as
//   K-adj's finaliser), last arriver sums the scalars.  No float atomics: bitwise repeatable.
//
// Host driver (solver.py): from n = 16384 the launch costs what K-fwd alone costs, so it serves every launch of the loop,
// backtracking retries included; below that it is used speculatively (the launch assumes the step will be accepted; if
// the backtracking test fails the driver re-runs K-fwd with the smaller step and K-adj: identical results either way).
// With acceleration (p.accel, fh_step_accel) the FISTA coefficient depends on this launch's own restart dot: every team
// exchanges it through one extra slot line before its first row.
// Requires n <= 262144: a row must fit TEAM*256*PPT 16-byte pieces; lanes past the row's last piece load a clamped
// address, carry x = 0 and are masked out of every store (fh_host_launch.h:fused_shape_for picks the next shape up).
#pragma once
#include "fh_dense.h"
#include "fh_loop.h"

#define FT_TEAM_MAX 32                              // members per team: 1, 2, 4, 8, 16 or 32 (template parameter TEAM)
#define FT_SENTINEL_HI 0x7FF8DEADu                  // slot filler: the NaN 0x7FF8DEAD7FF8DEAD (hipMemsetD32)
#define FT_SPIN_TICKS 50000000ull                   // 0.5 s of the 100 MHz s_memrealtime clock (grid barrier)
#define FT_SPIN_POLLS 1000000u                      // slot-poll budget: ~0.3-0.5 us per poll (s_load glc + s_sleep) => ~0.4 s

// Workgroup barrier for LDS hand-offs inside the row loop.  `__syncthreads()` makes hipcc drain `vmcnt(0)` first,
// which would land every prefetched row before each of the two per-row barriers (pipeline depth 0); this waits
// for the LDS traffic only and leaves the row loads in flight.
__device__ __forceinline__ void ft_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Cross-lane sums without LDS round trips (the row loop's serial chain is dot -> wave sum -> barrier -> exchange -> barrier ->
// update): the DPP reduction of fh_device.h, every lane gets the same sum (fixed order).
__device__ __forceinline__ double ft_wave_sum(double v) { return wave_sum(v); }

// -DFT_PROFILE: wave 0 of block 0 accumulates s_memtime ticks per phase of the row loop and prints them (debug builds only)
#ifdef FT_PROFILE
#define FT_T(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); prof[i] += _t - tprev; tprev = _t; } while (0)
#define FT_PHASE(i) do { phase[i] = __builtin_amdgcn_s_memrealtime(); } while (0)     /* 100 MHz wall clock of the whole launch's phases */
#else
#define FT_T(i) do { } while (0)
#define FT_PHASE(i) do { } while (0)
#endif

struct FusedP {
  const double* A;
  uint64_t ld;
  uint32_t ld2, n, m, mp;   // ld2: 16-byte pieces of a row that hold data (<= TEAM*256*PPT); ld: row stride of A in elements
  uint32_t ldp, nv2;        // ldp: row stride of A in 16-byte pieces; nv2: double pairs per n-side vector (ld2 x 1 or x 2, see PieceOf)
  uint32_t nteams, rows_per_team;
  const double* x0; const double* g0;
  double* xhat; double* xp;
  const double* b; double* z;
  double tau;
  int loss;
  int mode;              // 0 = full epilogue, 2 = row-sharded (g1 partial + local loss only)
  ProxP px;
  // FISTA (fasta/__init__.py:220-243): accel != 0 => x1 = xp + c*(xp - xacc0), the gradient is taken at z1 + c*(z1 - zacc0);
  // c = coef unless restart != 0 and this step's restart dot <(x0 - xp), (xp - xacc0)> exceeds 1e-30 (:231), then 0.
  int accel, restart;
  double coef;
  const double* xacc0; const double* zacc0; double* x1;
  double* coef_out;      // optional: the coefficient actually applied (for the separate n-side epilogue of row-sharded runs)
  double* pack;          // optional (row-sharded runs): 3 doubles behind g1 -- local loss sum, timeout word, loss sum at the extrapolated
                         // point -- so that ONE all-reduce of g1 carries them along
  double* slots;         // [mp + nteams][max(TEAM, 8)] partial dot products (last nteams lines: restart dot), holding the sentinel on entry
  double* slots_next;    // the same array of the NEXT launch: every slot this launch posts is re-armed there with the sentinel, so
                         // no launch needs a host-side refill (two arrays alternate; the host refills both only when the shape changes)
  double* gpart;         // [nteams][ld]
  double* g1;
  double* red;           // [grid][16] reduction partials
  unsigned* bar;         // [1] final arrivals   (zero on entry; the finaliser zeroes it again)
  unsigned* gbar;        // 2 x GB_WORDS words: the two-level grid barrier and the two-level final arrival (fh_device.h:grid_barrier2 / arrive_last2); zero on entry, zeroed again by the finaliser
  unsigned* err;         // set to 1 on a spin timeout
  int variant;           // bits: 2 = team members 32 blocks apart (one XCD), 4 = no s_sleep between polls, 8 = n=65536 as 8 members x 16 pieces, 32 = rows dealt cyclically to the teams, 64 = fault injection (tests)
  double* out;
};

__device__ __forceinline__ double ft_sq(double r) {
#pragma clang fp contract(off)
  return r * r;
}
__device__ __forceinline__ double ft_sentinel() { return __hiloint2double((int)FT_SENTINEL_HI, (int)FT_SENTINEL_HI); }

// XLDS = 1 (wide rows, PPT 9..16): the member's prox'd x slice lives in LDS (PPT x 4 KiB) instead of PPT x 4 registers per lane.
// That is what lets these widths POST AHEAD (PIPE = 1) like the narrower ones instead of exchanging in line: with the slice in
// registers, two hot row buffers + the g1 slice + the x slice exceed what hipcc can keep out of scratch (300-600 spilled
// registers, 2x slower); with it in LDS and NBO = 3-4 row buffers the loop compiles without spills and streams at the rate of
// the narrow shapes (n = 131072: 5.36 -> 7.16 TB/s, profiles/r02_fused_wide.txt).
// Workgroups per CU: one everywhere, except the float32-storage shapes of 8 and 16 members x 4 pieces with NBO = 4 (x and g1 slices of 32 registers
// each, four 16-register row buffers: inside the 256-register budget of two waves per SIMD), where two co-resident workgroups per CU hide
// each other's conversion, LDS and hand-off stalls.  All 2 x CUs workgroups are resident at once (two of these fit a CU and nothing else
// runs), which is what the slot exchange and the grid barrier need; fh_host_launch.h:fused_wpc_of is the host's copy of this rule.
template <int PPT, int TEAM, int XLDS, int F32> __host__ __device__ constexpr int fused_wpc() { return (F32 && TEAM >= 8 && PPT == 4 && !XLDS) ? 2 : 1; }
struct ChainP;
__device__ void chain_controller(const ChainP& ch, const FusedP& p, const double (&a)[8], const double (&bq)[5], double rdot, double timed_out);
template <int PPT, int NT, int PIPE, int TEAM, int XLDS = 0, int NBO = 0, int F32 = 0>   // NBO: number of row buffers (0 = the schedule's default); F32: float32-storage A
__global__ __launch_bounds__(FH_WG, (fused_wpc<PPT, TEAM, XLDS, F32>())) void k_fused_dense(const FusedP p) {
#define FUSED_BODY_CHAIN 0
#include "fh_fused_body.inc"
#undef FUSED_BODY_CHAIN
}

// =====================================================================================================================================
// CHAINED one-pass launches (round 6): the loop on the device for the shapes the persistent launch of fh_run.h does not serve.
//
// Between two one-pass launches somebody takes the reference's decisions from the launch's sums.  With the host in that place an iteration
// costs the kernel + ~7-10 us (launch, dispatch, the scalars' way back over PCIe); two DEPENDENT launches enqueued back to back on one
// stream start 4.3 us apart (profiles/r06_launchgap.txt).  To enqueue launch k + 1 before launch k has ended, nothing of launch k + 1 may
// depend on the host having seen launch k: its step size and its buffer roles come from a STATE BLOCK in device memory, and the finaliser
// of launch k -- the last workgroup to arrive, which holds all of the launch's sums -- runs the loop's controller (csrc/fh_run.h phase C:
// backtracking test fasta/__init__.py:195-217, FISTA restart and alpha recursion :220-238, Barzilai-Borwein :253-270, residuals, best
// iterate :272-300, the four stop rules of stopping.py:6-51) and rewrites that block: a retry keeps the roles and shrinks the step, an
// accepted iteration rotates the roles exactly as fh_commit does and appends its record to the host-mapped history.  One launch = one
// ATTEMPT; a launch that finds the solve stopped (stop rule, step budget, a hand-off timeout) returns at once.  The host enqueues K
// launches, copies the state back once and adopts it as after fh_run -- same entry point (fh_run), same options / state / history.
// Arithmetic: K-fused's sums, fh_run's controller (x * x where NumPy squares: histories rtol 1e-6 against the host-driven loop, counts equal).
// Served: float64, teams of 1 / 2 / 4 members (n <= 16384), separable prox kinds -- where a launch is short enough for 5 us to matter.
// =====================================================================================================================================
struct ChainState {          // device memory; read by every workgroup at the start of a launch, rewritten by its finaliser
  RunState rs;               // rs.tau_next: the step of the NEXT attempt
  double tau_iter;           // the step the current ITERATION started with (what a caller that redoes it after a timeout must use)
  int bt;                    // retries of the current iteration so far
  int steps_done;            // iterations completed since the host uploaded the block
  int attempts;              // launches that did work
  int reserved;
};
struct ChainP {
  double* nbuf[5];           // physical n-side buffers (X pool of three, P pair) as the context holds them
  double* G[2]; double* Z[2];
  double mu;
  RunOpts o;
  int g_kind;                // g(x) for the objective: 0 = none, 1 = mu * sum|x|
  int max_steps;             // iterations this chain may complete
  ChainState* st;
  double* hist;              // [max_steps][FR_HIST], host-mapped
};

template <typename T> __device__ __forceinline__ T* chain_sel5(T* const (&b)[5], int i) {      // (no dynamic index into a kernel-argument array: that would put it into scratch)
  return i == 0 ? b[0] : (i == 1 ? b[1] : (i == 2 ? b[2] : (i == 3 ? b[3] : b[4])));
}

// the controller of csrc/fh_run.h (phase C), run by ONE thread: the finaliser of a chained launch
__device__ inline void chain_controller(const ChainP& ch, const FusedP& p, const double (&a)[8], const double (&bq)[5], double rdot, double timed_out) {
#pragma clang fp contract(off)
  ChainState* st = ch.st;
  RunState& rs = st->rs;
  const RunOpts& o = ch.o;
  const double tau = p.tau;
  if (timed_out != 0.0) {            // a team hand-off ran out: this attempt's sums are void; everything the iteration read is intact
    rs.stopped = 3;
    rs.tau_next = st->tau_iter;
    return;
  }
  st->attempts += 1;
  const double fsq = a[0], dxg0 = a[1], dx2 = a[2], xh2 = a[3], g02 = a[4], gsum = a[5], gmax = a[6], fsq_adj = p.accel ? a[7] : a[0];
  auto fval = [&](double s) -> double { if (p.loss != LOSS_LSQ) return s; const double q = sqrt(s); return .5 * (q * q); };
  double f1 = fval(fsq);
  const unsigned long long ita = rs.iteration;
  if (o.backtrack) {                                                          // :195-217
    const unsigned long long lo_ = ita + 1ull > (unsigned long long)o.window ? ita + 1ull - (unsigned long long)o.window : 0ull;
    double M = rs.f_window[lo_ % FR_WINDOW_MAX];
    for (unsigned long long j = lo_ + 1ull; j <= ita; ++j) { const double v = rs.f_window[j % FR_WINDOW_MAX]; M = v > M ? v : M; }
    const double dxn = sqrt(dx2);
    if (f1 - (M + dxg0 + (dxn * dxn) / (2.0 * tau)) > 1E-12 && st->bt < o.max_backtracks) {
      rs.tau_next = tau * o.stepsize_shrink;                                  // same x0 / g0, smaller step (:204-215): the roles stay
      st->bt += 1;
      return;
    }
  }
  double alpha0 = 0.0, alpha1_new = rs.alpha1;
  bool restarted = false;
  if (o.accelerate) {                                                         // :220-238 (the launch applied the same restart rule to its coefficient)
    alpha0 = rs.alpha1;
    if (o.restart && rdot > 1E-30) { alpha0 = 1.0; restarted = true; }
    alpha1_new = (1.0 + sqrt(1.0 + 4.0 * (alpha0 * alpha0))) / 2.0;
    f1 = fval(fsq_adj);                                                       // :245
  }
  const double xh2u = o.accelerate ? bq[2] : xh2, gsu = o.accelerate ? bq[3] : gsum, gmu = o.accelerate ? bq[4] : gmax;
  double tau_nx = tau;                                                        // :249
  const double dx_norm = sqrt(dx2);
  if (o.adaptive) {                                                           // :253-270
    const double dot = bq[0];
    const double tau_s = (dx_norm * dx_norm) / dot;
    const double sg = sqrt(bq[1]);
    const double q = dot / (sg * sg);
    const double tau_m = 0.0 > q ? 0.0 : q;                                   // Python's max(q, 0)
    tau_nx = (2.0 * tau_m > tau_s) ? tau_m : tau_s - .5 * tau_m;
    if (tau_nx <= 0.0 || isinf(tau_nx) || isnan(tau_nx)) tau_nx = tau * 1.5;
  }
  const double resid = dx_norm / tau;                                         // :272
  const double a_ = sqrt(g02), b_ = sqrt(xh2u) / tau;
  const double normalizer = (b_ > a_ ? b_ : a_) + 1E-12;                      // max(a, b) + EPSILON  (:274)
  const double norm_resid = resid / normalizer;
  if (resid > rs.max_residual) rs.max_residual = resid;                       // :281
  double objective = 0.0, quality = resid;
  if (o.evaluate_objective) {                                                 // :284-289
    (void)gmu;
    objective = f1 + (ch.g_kind == 1 ? ch.mu * gsu : 0.0);
    quality = objective;
  }
  const bool better = quality < rs.best_quality;                              // :298-300
  if (better) rs.best_quality = quality;
  bool stop = false;                                                          // stopping.py:6-51
  const bool ratio = resid / rs.max_residual < o.tolerance, normed = norm_resid < o.tolerance;
  if (o.stop_rule == 0) stop = resid < o.tolerance;
  else if (o.stop_rule == 1) stop = normed;
  else if (o.stop_rule == 2) stop = ratio;
  else stop = ratio || normed;
  double* h = ch.hist + (uint64_t)st->steps_done * FR_HIST;
  h[0] = resid; h[1] = norm_resid; h[2] = tau; h[3] = f1; h[4] = objective; h[5] = (double)st->bt; h[6] = alpha0;
  h[7] = (better ? 1.0 : 0.0) + (restarted ? 2.0 : 0.0);
  // commit: fh_commit's pointer bookkeeping on the roles
  if (o.accelerate) { rs.alpha1 = alpha1_new; rs.pc ^= 1; rs.last_accel = 1; }
  else { const int x_ = rs.perm[rs.ti], y_ = rs.perm[3 + (rs.pc ^ 1)]; rs.perm[rs.ti] = y_; rs.perm[3 + (rs.pc ^ 1)] = x_; rs.last_accel = 0; }      // std::swap(X[ti], P[pc ^ 1])
  rs.xi = rs.ti;
  if (better) rs.bi = rs.xi;
  for (int k = 0; k < 3; ++k) if (k != rs.xi && k != rs.bi) { rs.ti = k; break; }
  rs.zc ^= 1; rs.gc ^= 1;
  rs.f_window[(ita + 1ull) % FR_WINDOW_MAX] = f1;
  rs.tau_next = tau_nx;
  st->tau_iter = tau_nx;
  rs.iteration = ita + 1ull;
  rs.backtracks += (unsigned long long)st->bt;
  st->bt = 0;
  st->steps_done += 1;
  if (stop) rs.stopped = 1;
}

// the arguments of this attempt from the state block the previous launch's finaliser left (uniform: every workgroup reads the same block and
// takes the same way); false = the solve has stopped or the chain's step budget is used up: the launch returns at once
__device__ __forceinline__ bool chain_params(const FusedP& base, const ChainP& ch, FusedP& p) {
#pragma clang fp contract(off)
  const ChainState* st = ch.st;
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  if (uni(st->rs.stopped) != 0 || uni(st->steps_done) >= ch.max_steps) return false;
  const int xi = uni(st->rs.xi), ti = uni(st->rs.ti), pcx = uni(st->rs.pc), gc = uni(st->rs.gc), zc = uni(st->rs.zc);
  int perm[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) perm[q] = uni(st->rs.perm[q]);
  const double tau = st->rs.tau_next, alpha1 = st->rs.alpha1;
  p = base;
  p.tau = tau;
  p.px.thr = tau * ch.mu;
  p.x0 = chain_sel5(ch.nbuf, xi == 0 ? perm[0] : (xi == 1 ? perm[1] : perm[2]));
  p.g0 = gc ? ch.G[1] : ch.G[0];
  p.xp = chain_sel5(ch.nbuf, pcx ? perm[3] : perm[4]);                        // P[pc ^ 1]
  p.z = zc ? ch.Z[0] : ch.Z[1];                                               // Z[zc ^ 1]
  p.g1 = gc ? ch.G[0] : ch.G[1];                                              // G[gc ^ 1]
  if (ch.o.accelerate) {
    const double a1 = (1.0 + sqrt(1.0 + 4.0 * (alpha1 * alpha1))) / 2.0;
    p.accel = 1; p.restart = ch.o.restart; p.coef = (alpha1 - 1.0) / a1;
    p.xacc0 = chain_sel5(ch.nbuf, pcx ? perm[4] : perm[3]);                   // P[pc]
    p.zacc0 = zc ? ch.Z[1] : ch.Z[0];                                         // Z[zc]
    p.x1 = chain_sel5(ch.nbuf, ti == 0 ? perm[0] : (ti == 1 ? perm[1] : perm[2]));
  }
  return true;
}

template <int PPT, int NT, int PIPE, int TEAM, int XLDS = 0, int NBO = 0, int F32 = 0>
__global__ __launch_bounds__(FH_WG, (fused_wpc<PPT, TEAM, XLDS, F32>())) void k_fused_chain(const FusedP base, const ChainP ch) {
  FusedP p;
  if (!chain_params(base, ch, p)) return;
#define FUSED_BODY_CHAIN 1
#include "fh_fused_body.inc"
#undef FUSED_BODY_CHAIN
}
