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

__device__ __forceinline__ void lvl_reduce2(double& s, double& c, double* sa, double* sb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s = wave_sum(s); c = wave_sum(c);
  __syncthreads();                        // previous round's readers are done with sa/sb
  if (lane == 0) { sa[wave] = s; sb[wave] = c; }
  __syncthreads();
  double ts = 0.0, tc = 0.0;
#pragma unroll
  for (int w = 0; w < LVL_WG / 64; ++w) { ts += sa[w]; tc += sb[w]; }   // same order in every thread
  s = ts; c = tc;
}

template <int EPT>
__global__ __launch_bounds__(LVL_WG) void k_level_search(const double* x0, const double* g0, uint32_t n, double tau,
                                                         double radius, double* level_out) {
  __shared__ __attribute__((aligned(16))) double sa[LVL_WG / 64];
  __shared__ __attribute__((aligned(16))) double sb[LVL_WG / 64];
  const uint32_t tid = threadIdx.x;
  double ax[EPT > 0 ? EPT : 1];
  if (EPT > 0) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const uint32_t i = tid + (uint32_t)k * LVL_WG;
      ax[k] = i < n ? fabs(fwd_point(x0[i], g0[i], tau)) : -INFINITY;   // -inf is never "> alpha"
    }
  }
  double alpha = -INFINITY, prev_cnt = -1.0;
  for (int pass = 0; pass < 100000; ++pass) {
    double s = 0.0, cnt = 0.0;
    if (EPT > 0) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) if (ax[k] > alpha) { s += ax[k]; cnt += 1.0; }
    } else {
      for (uint32_t i = tid; i < n; i += LVL_WG) {
        const double a = fabs(fwd_point(x0[i], g0[i], tau));
        if (a > alpha) { s += a; cnt += 1.0; }
      }
    }
    lvl_reduce2(s, cnt, sa, sb);          // every thread now holds the same totals -> uniform control flow
    if (cnt == 0.0 || cnt == prev_cnt) break;
    prev_cnt = cnt;
    {
#pragma clang fp contract(off)
      alpha = (s - radius) / cnt;
    }
  }
  if (tid == 0) level_out[0] = alpha;
}
