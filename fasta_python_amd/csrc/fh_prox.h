// fh_prox.h -- sort-free clipping level for the l-infinity prox / l1-ball projection.
//
// The reference sorts: alpha = max_k (cumsum(sort_desc|x|)[k] - t) / k   (fasta/proximal.py:22-26).
// That alpha is the root of  sum_i max(|x_i| - alpha, 0) = t  when ||x||_1 > t, and <= 0 otherwise.
// On the device it is found by Michelot's fixed-point iteration over reductions (no sort):
//     alpha_0 = (sum_i |x_i| - t) / n ;   A_k = { i : |x_i| > alpha_k } ;
//     alpha_{k+1} = (sum_{A_k} |x_i| - t) / |A_k| ,   stop when |A_k| stops shrinking.
// alpha_k increases monotonically to the root; the result differs from the sort-based value only by
// the rounding of the sums (~1e-16 relative).  One 1024-thread workgroup; the |xhat_i| = |x0_i - tau*g0_i|
// live in registers (EPT per thread) for n <= 65536 and are re-read from L2 beyond that.
#pragma once
#include "fh_device.h"

#define LVL_WG 1024

template <int WG = LVL_WG>
__device__ __forceinline__ void lvl_reduce2(double& s, double& c, double* sa, double* sb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s = wave_sum(s); c = wave_sum(c);
  __syncthreads();                        // previous round's readers are done with sa/sb
  if (lane == 0) { sa[wave] = s; sb[wave] = c; }
  __syncthreads();
  double ts = 0.0, tc = 0.0;
#pragma unroll
  for (int w = 0; w < WG / 64; ++w) { ts += sa[w]; tc += sb[w]; }   // same order in every thread
  s = ts; c = tc;
}

// WARM START (round 5).  For ANY alpha the update (sum_{|x_i| > alpha} |x_i| - t) / #{|x_i| > alpha} is a lower bound of the root (every
// subset S of the entries satisfies sum_S (|x_i| - root) <= t), so the search may start from any guess: one pass from the guess gives
// a lower bound, from which the monotone iteration runs as before.  The guess is the level the PREVIOUS launch found (it still sits in
// `level_out`): between two FBS iterations the level moves by a few per cent, and the search then takes 2-4 passes instead of 8-12
// (profiles/r05_level_search.txt).  The result is the same as from a cold start: the iteration ends on the same final active set,
// and the level is that set's (sum - t) / count, summed in the same fixed order.  A guess that is not a positive finite number, or
// lies above every entry, is ignored (cold start).  x0 / g0 are read as 16-byte pairs.
// WG threads: 1024, or 256 for n <= 8192 (four waves: a pass is one short scan and a four-term sum -- a quarter of the barrier and
// reduction work of sixteen waves, which is what a pass costs at these sizes).
template <int EPT, int WG = LVL_WG>
__global__ __launch_bounds__(WG) void k_level_search(const double* x0, const double* g0, uint32_t n, double tau,
                                                     double radius, double* level_out) {
  constexpr uint32_t LVL_WG_ = WG;
  __shared__ __attribute__((aligned(16))) double sa[WG / 64];
  __shared__ __attribute__((aligned(16))) double sb[WG / 64];
  const uint32_t tid = threadIdx.x;
  constexpr int PAIRS = EPT >= 2 ? EPT / 2 : 0;           // EPT = 1: one element per thread; EPT = 0: re-read every pass (n > 65536)
  double ax[EPT > 0 ? EPT : 1];
  if (EPT == 1) ax[0] = tid < n ? fabs(fwd_point(x0[tid], g0[tid], tau)) : -INFINITY;   // -inf is never "> alpha"
  if (PAIRS > 0) {
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
      const uint32_t i = 2u * (tid + (uint32_t)k * LVL_WG_);           // (n-side vectors are padded to a multiple of 16 doubles: in bounds)
      d2 xv = {0.0, 0.0}, gv = {0.0, 0.0};
      if (i < n) { xv = reinterpret_cast<const d2*>(x0)[i / 2]; gv = reinterpret_cast<const d2*>(g0)[i / 2]; }
      ax[2 * k] = i < n ? fabs(fwd_point(xv.x, gv.x, tau)) : -INFINITY;
      ax[2 * k + 1] = i + 1u < n ? fabs(fwd_point(xv.y, gv.y, tau)) : -INFINITY;
    }
  }
  const double guess = level_out[0];                      // the previous launch's level (any value is safe: see above)
  double alpha = (guess > 0.0 && guess < INFINITY) ? guess : -INFINITY;
  double prev_cnt = -1.0;
  bool warm = alpha > 0.0;
  for (int pass = 0; pass < 100000; ++pass) {
    double s = 0.0, cnt = 0.0;
    if (EPT > 0) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) if (ax[k] > alpha) { s += ax[k]; cnt += 1.0; }
    } else {
      for (uint32_t i = 2u * tid; i < n; i += 2u * LVL_WG_) {
        const d2 xv = reinterpret_cast<const d2*>(x0)[i / 2], gv = reinterpret_cast<const d2*>(g0)[i / 2];
        const double a0 = fabs(fwd_point(xv.x, gv.x, tau)), a1 = fabs(fwd_point(xv.y, gv.y, tau));
        if (a0 > alpha) { s += a0; cnt += 1.0; }
        if (i + 1u < n && a1 > alpha) { s += a1; cnt += 1.0; }
      }
    }
    lvl_reduce2<WG>(s, cnt, sa, sb);          // every thread now holds the same totals -> uniform control flow
    if (warm) {                           // the pass from the guess: its update is a lower bound of the root, whatever the guess was
      warm = false;
      if (cnt == 0.0) { alpha = -INFINITY; continue; }      // the guess lies above every entry: cold start
      {
#pragma clang fp contract(off)
        alpha = (s - radius) / cnt;
      }
      continue;                           // (prev_cnt stays -1: the monotone iteration starts here)
    }
    if (cnt == 0.0 || cnt == prev_cnt) break;
    prev_cnt = cnt;
    {
#pragma clang fp contract(off)
      alpha = (s - radius) / cnt;
    }
  }
  if (tid == 0) level_out[0] = alpha;
}

// ---- the same search on SEVERAL workgroups (round 5; n > 16384) --------------------------------------------------------------------
// One workgroup keeps n <= 16384 values in registers (16 per thread); beyond that they spilled (64 per thread at n = 65536: 79 us per
// search, 1.6 % of the step it precedes -- profiles/r05_level_search.txt).  Here G workgroups hold LVL_MEPT values per thread each
// (G = ceil(n / 8192) <= 32, all resident at once: the launch runs alone on its stream), and a pass is: local (sum, count) ->
// one write-through 16-byte record per workgroup + one ticket on the pass's counter -> every workgroup waits (bounded) for the G
// tickets, reads the G records with sc1 loads and adds them IN WORKGROUP ORDER -> the same alpha, the same decision, everywhere.
// Hand-off form: CDNA4 guide, Guideline 16 (one lane stores and drains, then signals; the poller reads after its poll matched, the
// other waves after a workgroup barrier).  The pass counters are left zero by the last workgroup to leave (a final ticket).
//
// FAILURE PATH (round 6).  The G workgroups must be co-resident, which nobody guarantees beside a CU-capped co-tenant, and LVL_MAXPASS
// bounds the record array while Michelot's iteration may need O(n) passes on adversarial magnitudes.  A workgroup whose wait for the
// tickets of a pass runs out (0.2 s), or that reaches pass LVL_MAXPASS, therefore does the WHOLE search again BY ITSELF
// (lvl_search_alone below: no hand-off, no co-residency, 100000 passes), emulating the G workgroups one after the other so that every
// sum is formed in exactly the order of the shared search -- the level it finds is bit-identical to the one an undisturbed launch finds.
// Every workgroup that gave up does that (they all find the same level); workgroup 0 writes it.  `diag[0]` counts launches that took the
// fall-back, `diag[1]` launches that could not produce a level at all (only the test hook LVL_HOOK_NO_FALLBACK gets there): those
// write NaN, which prox_scalar PROPAGATES into every prox output and sum of the step, and which the host turns into FH_E_TIMEOUT
// (csrc/fh_host_ctx.h:collect_scalars) -- never a finite wrong prox.
#define LVL_MEPT 8
#define LVL_MAXPASS 96
#define LVL_MAXG 32
#define LVL_HOOK_WITHHOLD 1        // test hooks (FH_TUNE_TEST_HOOKS bits 4 / 8): the last workgroup never posts its pass-0 record ...
#define LVL_HOOK_NO_FALLBACK 2     // ... and a workgroup that gave up writes NaN instead of searching alone
struct LevelWs { double* rec; unsigned* cnt; unsigned* diag; };      // rec[LVL_MAXPASS][LVL_MAXG][2], cnt[LVL_MAXPASS + 1] (the last one: leavers), diag[2]

// the decisions of one search (warm start, monotone iteration, stop), shared by the G-workgroup search and its single-workgroup twin
struct LvlCtl {
  double alpha, prev_cnt; bool warm;
  __device__ __forceinline__ void start(double guess) {
    alpha = (guess > 0.0 && guess < INFINITY) ? guess : -INFINITY;
    prev_cnt = -1.0;
    warm = alpha > 0.0;
  }
  // totals (s, cnt) of the entries above `alpha` -> true when the search has ended (alpha is the level)
  __device__ __forceinline__ bool next(double s, double cnt, double radius) {
#pragma clang fp contract(off)
    if (warm) {                           // the pass from the guess: its update is a lower bound of the root, whatever the guess was
      warm = false;
      if (cnt == 0.0) { alpha = -INFINITY; return false; }      // the guess lies above every entry: cold start
      alpha = (s - radius) / cnt;
      return false;                       // (prev_cnt stays -1: the monotone iteration starts here)
    }
    if (cnt == 0.0 || cnt == prev_cnt) return true;
    prev_cnt = cnt;
    alpha = (s - radius) / cnt;
    return false;
  }
};

// The search of k_level_search_multi done by ONE workgroup of LVL_WG threads: per pass it plays workgroup 0, 1, ..., G - 1 in turn -- the
// same LVL_MEPT entries per thread, the same wave and workgroup reductions, the totals added in workgroup order -- so every pass yields
// the bits the shared search yields.  The entries are re-read from L2 (1 MiB at n = 65536) in every pass.
__device__ __forceinline__ double lvl_search_alone(const double* x0, const double* g0, uint32_t n, double tau, double radius, double guess,
                                                   uint32_t G, double* sa, double* sb) {
  const uint32_t tid = threadIdx.x;
  LvlCtl ctl;
  ctl.start(guess);
  for (int pass = 0; pass < 100000; ++pass) {
    double S = 0.0, C = 0.0;
    for (uint32_t w = 0; w < G; ++w) {
      double s = 0.0, cnt = 0.0;
#pragma unroll
      for (int k = 0; k < LVL_MEPT / 2; ++k) {
        const uint32_t i = 2u * ((w * (LVL_MEPT / 2) + (uint32_t)k) * LVL_WG + tid);
        d2 xv = {0.0, 0.0}, gv = {0.0, 0.0};
        if (i < n) { xv = reinterpret_cast<const d2*>(x0)[i / 2]; gv = reinterpret_cast<const d2*>(g0)[i / 2]; }
        const double a0 = i < n ? fabs(fwd_point(xv.x, gv.x, tau)) : -INFINITY;
        const double a1 = i + 1u < n ? fabs(fwd_point(xv.y, gv.y, tau)) : -INFINITY;
        if (a0 > ctl.alpha) { s += a0; cnt += 1.0; }
        if (a1 > ctl.alpha) { s += a1; cnt += 1.0; }
      }
      lvl_reduce2(s, cnt, sa, sb);
      S += s; C += cnt;
    }
    if (ctl.next(S, C, radius)) break;
  }
  return ctl.alpha;
}

__global__ __launch_bounds__(LVL_WG) void k_level_search_multi(const double* x0, const double* g0, uint32_t n, double tau, double radius,
                                                               double* level_out, const LevelWs ws, int hooks) {
  __shared__ __attribute__((aligned(16))) double sa[LVL_WG / 64];
  __shared__ __attribute__((aligned(16))) double sb[LVL_WG / 64];
  __shared__ __attribute__((aligned(16))) double s_rec[LVL_MAXG][2];
  __shared__ unsigned s_ok;
  const uint32_t tid = threadIdx.x, G = gridDim.x, wg = blockIdx.x;
  double ax[LVL_MEPT];
#pragma unroll
  for (int k = 0; k < LVL_MEPT / 2; ++k) {
    const uint32_t i = 2u * ((wg * (LVL_MEPT / 2) + (uint32_t)k) * LVL_WG + tid);
    d2 xv = {0.0, 0.0}, gv = {0.0, 0.0};
    if (i < n) { xv = reinterpret_cast<const d2*>(x0)[i / 2]; gv = reinterpret_cast<const d2*>(g0)[i / 2]; }
    ax[2 * k] = i < n ? fabs(fwd_point(xv.x, gv.x, tau)) : -INFINITY;
    ax[2 * k + 1] = i + 1u < n ? fabs(fwd_point(xv.y, gv.y, tau)) : -INFINITY;
  }
  const double guess = level_out[0];
  LvlCtl ctl;
  ctl.start(guess);
  bool gave_up = (hooks & LVL_HOOK_WITHHOLD) && G > 1u && wg == G - 1u;      // (test hook: this workgroup's pass-0 record never arrives)
  bool found = false;
  for (int pass = 0; pass < LVL_MAXPASS && !gave_up; ++pass) {
    double s = 0.0, cnt = 0.0;
#pragma unroll
    for (int k = 0; k < LVL_MEPT; ++k) if (ax[k] > ctl.alpha) { s += ax[k]; cnt += 1.0; }
    lvl_reduce2(s, cnt, sa, sb);
    if (tid == 0) {
      double* rec = ws.rec + ((size_t)pass * LVL_MAXG + wg) * 2;
      store_partial(rec, s); store_partial(rec + 1, cnt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(ws.cnt + pass, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned ok = 1u;
      while (__hip_atomic_load(ws.cnt + pass, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { ok = 0u; break; }       // 0.2 s of the 100 MHz clock
        __builtin_amdgcn_s_sleep(1);
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) { gave_up = true; break; }                    // (uniform: every thread reads the same LDS word)
    if (tid < G) {
      const double* rec = ws.rec + ((size_t)pass * LVL_MAXG + tid) * 2;
      s_rec[tid][0] = load_partial(rec); s_rec[tid][1] = load_partial(rec + 1);
    }
    __syncthreads();
    s = 0.0; cnt = 0.0;
    for (uint32_t w = 0; w < G; ++w) { s += s_rec[w][0]; cnt += s_rec[w][1]; }     // workgroup order: the same totals in every thread of every workgroup
    if (ctl.next(s, cnt, radius)) { found = true; break; }
  }
  double alpha = ctl.alpha;
  if (!found) {
    // a hand-off timed out, or LVL_MAXPASS passes were not enough: this workgroup searches alone (same sums, same order, same level)
    if (hooks & LVL_HOOK_NO_FALLBACK) alpha = __builtin_nan("");      // (test hook) no level at all: NaN, which the step propagates and the host reports
    else alpha = lvl_search_alone(x0, g0, n, tau, radius, guess, G, sa, sb);
    if (tid == 0 && wg == 0) __hip_atomic_fetch_add(ws.diag + ((hooks & LVL_HOOK_NO_FALLBACK) ? 1 : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) {
    if (wg == 0) level_out[0] = alpha;
    // every poll of this launch lies behind this workgroup; the last one to get here zeroes the counters for the next launch
    const unsigned t = __hip_atomic_fetch_add(ws.cnt + LVL_MAXPASS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == G - 1u) {
      for (int q = 0; q <= LVL_MAXPASS; ++q) __hip_atomic_store(ws.cnt + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
