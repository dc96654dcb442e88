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
#define LVL_MEPT 8
#define LVL_MAXPASS 96
#define LVL_MAXG 32
struct LevelWs { double* rec; unsigned* cnt; };      // rec[LVL_MAXPASS][LVL_MAXG][2], cnt[LVL_MAXPASS + 1] (the last one: leavers)
__global__ __launch_bounds__(LVL_WG) void k_level_search_multi(const double* x0, const double* g0, uint32_t n, double tau, double radius,
                                                               double* level_out, const LevelWs ws) {
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
  double alpha = (guess > 0.0 && guess < INFINITY) ? guess : -INFINITY;
  double prev_cnt = -1.0;
  bool warm = alpha > 0.0, failed = false;
  int pass = 0;
  for (; pass < LVL_MAXPASS; ++pass) {
    double s = 0.0, cnt = 0.0;
#pragma unroll
    for (int k = 0; k < LVL_MEPT; ++k) if (ax[k] > alpha) { s += ax[k]; cnt += 1.0; }
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
    if (!s_ok) { failed = true; break; }                     // (uniform: every thread reads the same LDS word)
    if (tid < G) {
      const double* rec = ws.rec + ((size_t)pass * LVL_MAXG + tid) * 2;
      s_rec[tid][0] = load_partial(rec); s_rec[tid][1] = load_partial(rec + 1);
    }
    __syncthreads();
    s = 0.0; cnt = 0.0;
    for (uint32_t w = 0; w < G; ++w) { s += s_rec[w][0]; cnt += s_rec[w][1]; }     // workgroup order: the same totals in every thread of every workgroup
    if (warm) {
      warm = false;
      if (cnt == 0.0) { alpha = -INFINITY; continue; }
      {
#pragma clang fp contract(off)
        alpha = (s - radius) / cnt;
      }
      continue;
    }
    if (cnt == 0.0 || cnt == prev_cnt) break;
    prev_cnt = cnt;
    {
#pragma clang fp contract(off)
      alpha = (s - radius) / cnt;
    }
  }
  if (failed || pass >= LVL_MAXPASS) alpha = __builtin_nan("");     // never a silently wrong level: NaN poisons the step's sums, the solver stops
  if (tid == 0) {
    if (wg == 0) level_out[0] = alpha;
    // every poll of this launch lies behind this workgroup; the last one to get here zeroes the counters for the next launch
    const unsigned t = __hip_atomic_fetch_add(ws.cnt + LVL_MAXPASS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == G - 1u) {
      for (int q = 0; q <= LVL_MAXPASS; ++q) __hip_atomic_store(ws.cnt + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
