// fh_tv.h -- fused FBS kernels for the periodic difference-stencil operator pair of
// examples/tv_denoising.py:26-63 (A = div : (H,W,2) -> (H,W),  A^H = grad : (H,W) -> (H,W,2)).
//
//   div(Y)[i,j]    = (Y[i+1,j,0] - Y[i,j,0]) + (Y[i,j+1,1] - Y[i,j,1])        (np.roll(.., -1), :61)
//   grad(X)[i,j,0] =  X[i-1,j] - X[i,j] ;  grad(X)[i,j,1] = X[i,j-1] - X[i,j]  (np.roll(.., +1), :38)
//
// Layout: Y-space vectors are pixel-interleaved float64 pairs (C order of (H,W,2)): one pixel = one
// aligned 16-byte access; X-space vectors are H*W float64.
//
// Wave-strip streaming, no LDS tiles and no barriers in the hot loop: each 64-lane wave owns a strip of
// 64 image columns (aligned 1 KiB row pieces) and walks `rows_wg` rows top to bottom.  The vertical
// neighbour is the previous row's value kept in a register; the horizontal neighbour comes from the
// adjacent lane by __shfl, and for the strip's edge lane from a halo pixel that lanes 0..TV_U-1 fetch
// (one lane per row of the batch) and broadcast.  Loads are issued TV_U rows ahead.
// The first pair below (k_fwd_tv / k_adj_tv) is the PLAIN pair: z = div(x0) and g = grad(z - b) with
// materialised outputs -- used by fh_init, fh_apply and the Lipschitz probes.  The FBS-step pair
// (k_fwd_tv_step / k_adj_tv_step, end of this file) never materialises the gradient.
#pragma once
#include "fh_device.h"

#define TV_SW 64       // owned columns per wave
// rows per workgroup (p.rows_wg) and rows of loads in flight (template TV_U) are tunables, see fh_set_tuning

// unit-ball projection of one pixel's 2-vector: Y / max(||Y||_2, 1)   (examples/tv_denoising.py:89-96)
__device__ __forceinline__ d2 tv_ball(d2 y) {
#pragma clang fp contract(off)
  const double a = y.x * y.x;
  const double b = y.y * y.y;
  const double nr = sqrt(a + b);
  const double d = fmax(nr, 1.0);
  d2 r;
  r.x = y.x / d;
  r.y = y.y / d;
  return r;
}

// (base + off) mod H for the few rows a sweep steps outside its chunk (base < H, off in [-2, rows + 1], rows <= H - base): two
// conditional corrections per side cover every H >= 1.  NOT a 64-bit `%`: hipcc expands that into ~140 scalar instructions, two
// of them per image row in the round-2 sweeps (~560 of the ~920 instructions of a two-row trip).  Removing them did not change the
// sweep's time -- it is not issue-bound (profiles/r03_tune_tv.txt, part b) -- but there is no reason to keep them.
__device__ __forceinline__ uint32_t tv_wrap_row(uint32_t base, int off, uint32_t H) {
  int r = (int)base + off;
  const int h = (int)H;
  if (r < 0) r += h;
  if (r < 0) r += h;
  if (r >= h) r -= h;
  if (r >= h) r -= h;
  return (uint32_t)r;
}

// Workgroups are handed to the 8 XCDs round-robin by workgroup id, and every XCD has its own L2: with the plain order, horizontally
// adjacent strip groups -- which share their halo columns' cache lines -- always sit on DIFFERENT XCDs, so each shared line is fetched
// from HBM twice (PMC: reads 1.15x algorithmic).  This bijection gives every XCD a contiguous range of logical ids instead
// (workgroup b -> XCD b % 8 -> its (b / 8)-th logical id), so that neighbours in the image are neighbours in one L2.  A speed
// heuristic only: nothing depends on where a workgroup really runs.
__device__ __forceinline__ uint32_t tv_xcd_order(uint32_t b, uint32_t grid, int on) {
  const uint32_t per = grid / 8u;
  if (!on || b >= per * 8u) return b;            // the grid's last grid % 8 workgroups keep their ids
  return (b % 8u) * per + b / 8u;
}

template <int NT>
__device__ __forceinline__ void store_d2(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <int NT>
__device__ __forceinline__ void store_f64(double* p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <int NT>
__device__ __forceinline__ double load_f64(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// ---- PLAIN pair -----------------------------------------------------------------------------------
// z = div(x) (+ ||z - b||^2): aligned 64-column strips; lane 63's right neighbour is a halo pixel that lanes
// 0..TV_U-1 fetch (one per row of the batch) and broadcast.
struct TvFwdP {
  uint32_t H, W;
  uint32_t strip_groups;      // ceil(ceil(W/64)/4): 4 wave strips per workgroup
  uint32_t rows_wg;           // rows per workgroup
  const double* x0;           // (H,W,2)
  const double* b; double* z; // (H,W)
  int sub_b;
  double* red;                // [grid][8]
  unsigned* counter;
  double* out;
};

template <int TV_U, int NT>
__global__ __launch_bounds__(FH_WG) void k_fwd_tv(const TvFwdP p) {
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sg = blockIdx.x % p.strip_groups, rc = blockIdx.x / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TV_SW;
  const uint32_t c = first + lane;                        // image column of this lane
  const bool own = c < p.W;
  const uint32_t cl = own ? c : c % p.W;                  // columns past W wrap: they act as right neighbours
  const uint32_t hc = (first + TV_SW) % p.W;              // halo column right of the strip (for lane 63)
  double fs = 0.0;

  d2 cur = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + (uint64_t)i0 * p.W + cl);
  for (uint32_t r0 = 0; r0 < rows; r0 += TV_U) {
    d2 xv[TV_U];
    double bv[TV_U];
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      const uint32_t rr = min(r0 + u + 1u, rows);                 // row below output row r0+u (clamped past the chunk)
      uint32_t nrow = i0 + rr; if (nrow >= p.H) nrow -= p.H;      // periodic
      xv[u] = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + (uint64_t)nrow * p.W + cl);
      const uint32_t orow = min(i0 + r0 + u, p.H - 1u);
      bv[u] = (p.sub_b && own) ? load_f64<NT>(p.b + (uint64_t)orow * p.W + c) : 0.0;
    }
    double halo_y = 0.0;                                          // lane u: pixel right of the strip in output row r0+u
    if (lane < (uint32_t)TV_U) {
      const uint32_t orow = min(i0 + r0 + lane, p.H - 1u);
      halo_y = reinterpret_cast<const d2*>(p.x0)[(uint64_t)orow * p.W + hc].y;
    }
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      if (r0 + u < rows) {                                        // wave-uniform
        const d2 nxt = xv[u];
        double right_y = __shfl_down(cur.y, 1, 64);               // pixel (row, col+1), component 1
        const double edge_y = __shfl(halo_y, u, 64);
        if (lane == 63u) right_y = edge_y;
        double zv;
        {
#pragma clang fp contract(off)
          const double t0 = nxt.x - cur.x;                        // roll(Y0, -1, axis 0) - Y0
          const double t1 = right_y - cur.y;                      // roll(Y1, -1, axis 1) - Y1
          zv = t0 + t1;
        }
        if (own) {
          store_f64<NT>(p.z + (uint64_t)(i0 + r0 + u) * p.W + c, zv);
          const double rv = p.sub_b ? sub_nofma(zv, bv[u]) : zv;
          fs = fma(rv, rv, fs);
        }
        cur = nxt;
      }
    }
  }
  double w[8] = {fs, 0, 0, 0, 0, 0, 0, 0};
  block_reduce<8>(w, s_scr, -1);
  if (!publish_partials<8>(p.red + (uint64_t)blockIdx.x * 8, w, p.counter, gridDim.x, s_flag)) return;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) t[0] += load_partial(p.red + (uint64_t)i * 8);
  block_reduce<8>(t, s_scr, -1);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.out[k] = t[k];
    __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// g = grad(z - b) materialised (Lipschitz probes, fh_apply adjoint); lane 0's left neighbour is the broadcast halo.
struct TvAdjP {
  uint32_t H, W;
  uint32_t strip_groups;
  uint32_t rows_wg;
  const double* z; const double* b;
  int sub_b;
  double* g1;           // (H,W,2)
  double* red;          // [grid][8]
  unsigned* counter;
  double* out;
};

template <int TV_U, int NT>
__global__ __launch_bounds__(FH_WG) void k_adj_tv(const TvAdjP p) {
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sg = blockIdx.x % p.strip_groups, rc = blockIdx.x / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TV_SW;
  const uint32_t c = first + lane;
  const bool own = c < p.W;
  const uint32_t cl = own ? c : c % p.W;
  const uint32_t hc = (first == 0u) ? p.W - 1u : (first - 1u) % p.W;     // halo column left of the strip (for lane 0)
  double fs = 0.0;

  auto resid = [&](uint64_t pix) -> double {
    const double zv = load_f64<NT>(p.z + pix);
    return p.sub_b ? sub_nofma(zv, load_f64<NT>(p.b + pix)) : zv;
  };

  double up;
  {
    const uint32_t prow = (i0 == 0u) ? p.H - 1u : i0 - 1u;
    up = resid((uint64_t)prow * p.W + cl);
  }
  for (uint32_t r0 = 0; r0 < rows; r0 += TV_U) {
    double me[TV_U];
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      const uint32_t row = min(i0 + r0 + u, p.H - 1u);
      me[u] = resid((uint64_t)row * p.W + cl);
    }
    double halo_r = 0.0;                                          // lane u: residual left of the strip in row r0+u
    if (lane < (uint32_t)TV_U) {
      const uint32_t row = min(i0 + r0 + lane, p.H - 1u);
      halo_r = resid((uint64_t)row * p.W + hc);
    }
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      if (r0 + u < rows) {                                       // wave-uniform
        double left = __shfl_up(me[u], 1, 64);                   // residual at (row, col-1)
        const double edge = __shfl(halo_r, u, 64);
        if (lane == 0u) left = edge;
        d2 g;
        g.x = sub_nofma(up, me[u]);                              // roll(X, +1, axis 0) - X
        g.y = sub_nofma(left, me[u]);                            // roll(X, +1, axis 1) - X
        if (own) {
          fs = fma(me[u], me[u], fs);
          store_d2<NT>(reinterpret_cast<d2*>(p.g1) + (uint64_t)(i0 + r0 + u) * p.W + c, g);
        }
        up = me[u];
      }
    }
  }
  double w[6] = {0, 0, 0, 0, 0, fs};
  block_reduce<6>(w, s_scr, 4);
  if (!publish_partials<6>(p.red + (uint64_t)blockIdx.x * 8, w, p.counter, gridDim.x, s_flag)) return;
  double t[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) t[5] += load_partial(p.red + (uint64_t)i * 8 + 5);
  block_reduce<6>(t, s_scr, 4);
  if (tid == 0) {
    p.out[S_DXDG] = 0.0; p.out[S_DG2] = 0.0; p.out[S_XH2_ADJ] = 0.0; p.out[S_GSUM_ADJ] = 0.0;
    p.out[S_GMAX_ADJ] = 0.0; p.out[S_FSQ_ADJ] = t[5];
    __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// =================================================================================================
// FBS-step kernels that never materialise the gradient.
//
// For the stencil pair, g = grad(r) is a 2-point difference of the residual r = z - b, so storing g
// (16 B/pixel written by K-adj, 16 B/pixel read by K-fwd) costs more than recomputing it from z and b
// (8 + 8 B/pixel, and b is needed anyway).  With z_cur = A x0 (the previous launch's output):
//   K-fwd step : g0 = grad(z_cur - b) on the fly; xhat, xprox, z_new = div(xprox), reductions.
//                reads x0 16 + z_cur 8 + b 8, writes xprox 16 + z_new 8            = 56 B/pixel
//   K-adj step : g1 = grad(z_new' - b), g0 again (for xhat), BB reductions only.
//                reads z_new 8 + z_cur 8 + b 8 + x0 16 + xprox 16, writes nothing   = 56 B/pixel
//                (+ FISTA: reads z_acc0 8 + x_acc0 16, writes z' 8 + x1 16)
// i.e. 112*P per iteration against the materialised-vector model's 136*P; the
// roofline line still prices the launches at the model's 64*P / 72*P.  Every value is produced by the same
// IEEE operations as the reference (r = z - b, g = r_neighbour - r, xhat = x0 - tau*g), so parity is
// unchanged.  Waves own overlapping strips (one halo lane per needed side) so all lanes run one code path.
// =================================================================================================
#define TVS_FWD_OWN 62     // lanes 1..62 own; lane 0 / 63 are the left / right halo columns
#define TVS_ADJ_OWN 63     // lanes 1..63 own; lane 0 is the left halo column

struct TvStepFwdP {
  uint32_t H, W, strip_groups, rows_wg;
  const double* x0; const double* xacc0;   // (H,W,2)
  double* xp;                               // (H,W,2)
  const double* zc; const double* b;        // (H,W): z at x0, target
  double* zn;                               // (H,W): div(xprox)
  double tau;
  double* red; unsigned* counter; double* out;
};

template <int IDENT, int TV_U, int NT>
__global__ __launch_bounds__(FH_WG) void k_fwd_tv_step(const TvStepFwdP p) {
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sg = blockIdx.x % p.strip_groups, rc = blockIdx.x / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TVS_FWD_OWN;
  const uint32_t c = first + lane - 1u;                      // 0xFFFFFFFF for the very first halo lane
  const bool own = lane >= 1u && lane <= (uint32_t)TVS_FWD_OWN && c < p.W;
  const uint32_t cl = (lane == 0u && first == 0u) ? p.W - 1u : c % p.W;
  double v[5] = {0, 0, 0, 0, 0};                             // dxg0, dx2, xh2, g02, rdot
  double fs = 0.0;

  // forward point + prox of one pixel given its residual neighbourhood; owner side effects when `mine`
  auto step_pixel = [&](d2 x0v, double r_me, double r_up, double r_left, uint64_t pix, bool mine) -> d2 {
    d2 g0v, xh;
    g0v.x = sub_nofma(r_up, r_me);                           // grad(r)[..,0] = roll(r,+1,axis 0) - r
    g0v.y = sub_nofma(r_left, r_me);                         // grad(r)[..,1] = roll(r,+1,axis 1) - r
    xh.x = fwd_point(x0v.x, g0v.x, p.tau);
    xh.y = fwd_point(x0v.y, g0v.y, p.tau);
    const d2 xp = IDENT ? xh : tv_ball(xh);
    if (mine) {
      store_d2<NT>(reinterpret_cast<d2*>(p.xp) + pix, xp);
      d2 xav = {0.0, 0.0};
      if (p.xacc0) xav = reinterpret_cast<const d2*>(p.xacc0)[pix];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double dx = sub_nofma(xp[e], x0v[e]);
        const double dh = sub_nofma(xp[e], xh[e]);
        v[0] = fma(dx, g0v[e], v[0]);
        v[1] = fma(dx, dx, v[1]);
        v[2] = fma(dh, dh, v[2]);
        v[3] = fma(g0v[e], g0v[e], v[3]);
        v[4] = fma(sub_nofma(x0v[e], xp[e]), sub_nofma(xp[e], xav[e]), v[4]);
      }
    }
    return xp;
  };

  double r_up, b_cur;
  d2 cur;
  {
    const uint32_t prow = (i0 == 0u) ? p.H - 1u : i0 - 1u;
    const uint64_t ppix = (uint64_t)prow * p.W + cl;
    r_up = sub_nofma(load_f64<NT>(p.zc + ppix), load_f64<NT>(p.b + ppix));
    const uint64_t pix = (uint64_t)i0 * p.W + cl;
    const d2 x0v = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + pix);
    b_cur = load_f64<NT>(p.b + pix);
    const double r_me = sub_nofma(load_f64<NT>(p.zc + pix), b_cur);
    const double r_left = __shfl_up(r_me, 1, 64);
    cur = step_pixel(x0v, r_me, r_up, r_left, pix, own);
    r_up = r_me;
  }
  for (uint32_t r0 = 0; r0 < rows; r0 += TV_U) {
    d2 xv[TV_U];
    double zv[TV_U], bv[TV_U];
    uint64_t npix[TV_U];
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      const uint32_t rr = min(r0 + u + 1u, rows);                 // row below output row r0+u (clamped past the chunk)
      uint32_t nrow = i0 + rr; if (nrow >= p.H) nrow -= p.H;      // periodic
      npix[u] = (uint64_t)nrow * p.W + cl;
      xv[u] = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + npix[u]);
      zv[u] = load_f64<NT>(p.zc + npix[u]);
      bv[u] = load_f64<NT>(p.b + npix[u]);
    }
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      if (r0 + u < rows) {                                        // wave-uniform
        const double r_n = sub_nofma(zv[u], bv[u]);
        const double r_left = __shfl_up(r_n, 1, 64);
        const d2 nxt = step_pixel(xv[u], r_n, r_up, r_left, npix[u], own && (r0 + u + 1u < rows));
        const double right_y = __shfl_down(cur.y, 1, 64);         // xprox(row, col+1)[1]
        double zo;
        {
#pragma clang fp contract(off)
          const double t0 = nxt.x - cur.x;                        // roll(Y0, -1, axis 0) - Y0
          const double t1 = right_y - cur.y;                      // roll(Y1, -1, axis 1) - Y1
          zo = t0 + t1;
        }
        if (own) {
          store_f64<NT>(p.zn + (uint64_t)(i0 + r0 + u) * p.W + c, zo);
          const double rv = sub_nofma(zo, b_cur);
          fs = fma(rv, rv, fs);
        }
        cur = nxt; b_cur = bv[u]; r_up = r_n;
      }
    }
  }
  double w[8] = {fs, v[0], v[1], v[2], v[3], 0.0, 0.0, v[4]};   // S_FSQ, S_DXG0, S_DX2, S_XH2, S_G02, -, -, S_RDOT
  block_reduce<8>(w, s_scr, -1);
  if (!publish_partials<8>(p.red + (uint64_t)blockIdx.x * 8, w, p.counter, gridDim.x, s_flag)) return;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] += load_partial(p.red + (uint64_t)i * 8 + k);
  }
  block_reduce<8>(t, s_scr, -1);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.out[k] = t[k];
    __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct TvStepAdjP {
  uint32_t H, W, strip_groups, rows_wg;
  const double* zn; const double* zacc0; const double* zc; const double* b;   // (H,W)
  int accel; double coef; double tau;
  const double* x0; const double* xp; const double* xacc0;                    // (H,W,2)
  double* x1; double* zx;                                                      // FISTA outputs: x1 (H,W,2), z' (H,W)
  double* red; unsigned* counter; double* out;
};

template <int TV_U, int NT>
__global__ __launch_bounds__(FH_WG) void k_adj_tv_step(const TvStepAdjP p) {
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sg = blockIdx.x % p.strip_groups, rc = blockIdx.x / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TVS_ADJ_OWN;
  const uint32_t c = first + lane - 1u;
  const bool own = lane >= 1u && c < p.W;
  const uint32_t cl = (lane == 0u && first == 0u) ? p.W - 1u : c % p.W;
  double v[5] = {0, 0, 0, 0, 0};   // dxdg, dg2, xh2, gsum, gmax
  double fs = 0.0;

  auto z_new_at = [&](uint64_t pix) -> double {          // z1' of fasta/__init__.py:243 (z1 itself without FISTA)
    double zv = load_f64<NT>(p.zn + pix);
    if (p.accel) zv = extrapolate(zv, p.zacc0[pix], p.coef);
    return zv;
  };

  double rn_up, rc_up;
  {
    const uint32_t prow = (i0 == 0u) ? p.H - 1u : i0 - 1u;
    const uint64_t ppix = (uint64_t)prow * p.W + cl;
    const double bp = load_f64<NT>(p.b + ppix);
    rn_up = sub_nofma(z_new_at(ppix), bp);
    rc_up = sub_nofma(load_f64<NT>(p.zc + ppix), bp);
  }
  for (uint32_t r0 = 0; r0 < rows; r0 += TV_U) {
    double zn[TV_U], zc[TV_U], bv[TV_U];
    d2 x0v[TV_U], xpv[TV_U], xav[TV_U];
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      const uint32_t row = min(i0 + r0 + u, p.H - 1u);
      const uint64_t pix = (uint64_t)row * p.W + cl;
      zn[u] = z_new_at(pix);
      zc[u] = load_f64<NT>(p.zc + pix);
      bv[u] = load_f64<NT>(p.b + pix);
      x0v[u] = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + pix);
      xpv[u] = load_stream<NT>(reinterpret_cast<const d2*>(p.xp) + pix);
      xav[u] = (d2){0.0, 0.0};
      if (p.accel) xav[u] = reinterpret_cast<const d2*>(p.xacc0)[pix];
    }
#pragma unroll
    for (int u = 0; u < TV_U; ++u) {
      if (r0 + u < rows) {                                       // wave-uniform
        const double rn = sub_nofma(zn[u], bv[u]);
        const double rcur = sub_nofma(zc[u], bv[u]);
        const double rn_left = __shfl_up(rn, 1, 64);
        const double rc_left = __shfl_up(rcur, 1, 64);
        if (own) {
          const uint64_t pix = (uint64_t)(i0 + r0 + u) * p.W + c;
          d2 g1, g0;
          g1.x = sub_nofma(rn_up, rn);    g1.y = sub_nofma(rn_left, rn);
          g0.x = sub_nofma(rc_up, rcur);  g0.y = sub_nofma(rc_left, rcur);
          fs = fma(rn, rn, fs);
          if (p.accel) p.zx[pix] = zn[u];
          d2 x1v;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double xh = fwd_point(x0v[u][e], g0[e], p.tau);       // same bits as K-fwd's xhat
            double x1 = xpv[u][e];
            if (p.accel) x1 = extrapolate(xpv[u][e], xav[u][e], p.coef);
            const double dx = sub_nofma(xpv[u][e], x0v[u][e]);
            const double dg = bb_dgrad(g1[e], xh, x0v[u][e], p.tau);
            const double dh = sub_nofma(x1, xh);
            v[0] = fma(dx, dg, v[0]);
            v[1] = fma(dg, dg, v[1]);
            v[2] = fma(dh, dh, v[2]);
            v[3] += fabs(x1);
            v[4] = fmax(v[4], fabs(x1));
            x1v[e] = x1;
          }
          if (p.accel) reinterpret_cast<d2*>(p.x1)[pix] = x1v;
        }
        rn_up = rn; rc_up = rcur;
      }
    }
  }
  double w[6] = {v[0], v[1], v[2], v[3], v[4], fs};
  block_reduce<6>(w, s_scr, 4);
  if (!publish_partials<6>(p.red + (uint64_t)blockIdx.x * 8, w, p.counter, gridDim.x, s_flag)) return;
  double t[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double q = load_partial(p.red + (uint64_t)i * 8 + k);
      if (k == 4) t[k] = fmax(t[k], q); else t[k] += q;
    }
  }
  block_reduce<6>(t, s_scr, 4);
  if (tid == 0) {
    p.out[S_DXDG] = t[0]; p.out[S_DG2] = t[1]; p.out[S_XH2_ADJ] = t[2]; p.out[S_GSUM_ADJ] = t[3];
    p.out[S_GMAX_ADJ] = t[4]; p.out[S_FSQ_ADJ] = t[5];
    __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// =================================================================================================
// ONE-PASS FBS iteration for the stencil pair (no acceleration): everything k_fwd_tv_step does PLUS the
// K-adj reductions, because g1 = grad(z_new - b) is a 2-point difference of values this sweep has just
// produced (previous row: a register; left column: __shfl_up).  Per pixel it reads x0 16 + z_cur 8 + b 8 and
// writes xprox 16 + z_new 8 -- the traffic of K-fwd alone; K-adj's 56*P disappear.
//   lanes 2..62 own (61 columns per wave): lane 0 supplies r_cur to lane 1, lanes 1 and 63 supply xprox / z_new
//   / r_new halos; each workgroup also rolls through one halo row above its chunk (r_new of row i0-1 feeds
//   g1 of row i0) and looks one row ahead (xprox of row i0+rows feeds z_new of the last row).
// =================================================================================================
#ifdef FH_EXPERIMENTAL   // round-1 one-pass stencil kernels that stream z: superseded by k_tv_onepass, kept for A/B runs (csrc/fh_experimental.h)
#define TVF_OWN 61

template <int IDENT, int TV_U, int NT>
__global__ __launch_bounds__(FH_WG) void k_fused_tv_step(const TvStepFwdP p) {
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sg = blockIdx.x % p.strip_groups, rc = blockIdx.x / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TVF_OWN;
  // lane L <-> image column first + L - 2 (periodic); lanes 2..62 own
  const uint32_t cw = (first + lane + 2u * p.W - 2u) % p.W;
  const uint32_t c = first + lane - 2u;                           // valid as an index only for owning lanes
  const bool own = lane >= 2u && lane <= 62u && c < p.W;
  double v[4] = {0, 0, 0, 0};                                     // dxg0, dx2, xh2, g02
  double u[5] = {0, 0, 0, 0, 0};                                  // dxdg, dg2, xh2(adj), gsum, gmax
  double fs = 0.0;

  auto row_of = [&](uint32_t base, int off) -> uint32_t { return tv_wrap_row(base, off, p.H); };   // (base + off) mod H, periodic
  auto resid_at = [&](uint32_t row) -> double {
    const uint64_t pix = (uint64_t)row * p.W + cw;
    return sub_nofma(load_f64<NT>(p.zc + pix), load_f64<NT>(p.b + pix));
  };
  // forward point + prox of one pixel from its r_cur neighbourhood
  auto prox_pixel = [&](d2 x0v, double r_me, double r_up, double r_left, d2& g0v, d2& xh) -> d2 {
    g0v.x = sub_nofma(r_up, r_me);
    g0v.y = sub_nofma(r_left, r_me);
    xh.x = fwd_point(x0v.x, g0v.x, p.tau);
    xh.y = fwd_point(x0v.y, g0v.y, p.tau);
    return IDENT ? xh : tv_ball(xh);
  };

  // ---- prologue: halo row i0-1 (needs r_cur of rows i0-2 and i0-1), then row i0 --------------------------------
  double rc_up = resid_at(row_of(i0, -2));
  d2 x0_prev, xh_prev, xp_prev;        // pixel (row-1) state kept for its z_new / epilogue one step later
  double b_prev, rn_prev = 0.0;        // r_new of the row above the one being finished
  {
    const uint32_t hrow = row_of(i0, -1);
    const uint64_t pix = (uint64_t)hrow * p.W + cw;
    const d2 x0v = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + pix);
    b_prev = load_f64<NT>(p.b + pix);
    const double r_me = sub_nofma(load_f64<NT>(p.zc + pix), b_prev);
    const double r_left = __shfl_up(r_me, 1, 64);
    d2 g0v;
    xp_prev = prox_pixel(x0v, r_me, rc_up, r_left, g0v, xh_prev);
    x0_prev = x0v;
    rc_up = r_me;
  }
  // process rows t = 0 .. rows: step t loads row i0+t, finishes z_new / epilogue of row i0+t-1
  // (t = 0 finishes the halo row i0-1: only its r_new is kept)
  for (uint32_t t0 = 0; t0 <= rows; t0 += TV_U) {
    d2 xv[TV_U];
    double zv[TV_U], bv[TV_U];
    uint64_t npix[TV_U];
#pragma unroll
    for (int q = 0; q < TV_U; ++q) {
      const uint32_t t = min(t0 + q, rows);                        // clamp past the chunk (loads stay in bounds)
      const uint32_t nrow = row_of(i0, (int)t);
      npix[q] = (uint64_t)nrow * p.W + cw;
      xv[q] = load_stream<NT>(reinterpret_cast<const d2*>(p.x0) + npix[q]);
      zv[q] = load_f64<NT>(p.zc + npix[q]);
      bv[q] = load_f64<NT>(p.b + npix[q]);
    }
#pragma unroll
    for (int q = 0; q < TV_U; ++q) {
      const uint32_t t = t0 + q;
      if (t <= rows) {                                             // wave-uniform
        // ---- new row i0+t: forward point + prox (owned when t < rows) ----
        const double r_n = sub_nofma(zv[q], bv[q]);
        const double r_left = __shfl_up(r_n, 1, 64);
        d2 g0v, xh;
        const d2 xp = prox_pixel(xv[q], r_n, rc_up, r_left, g0v, xh);
        const bool mine = own && t < rows;
        if (mine) {
          store_d2<NT>(reinterpret_cast<d2*>(p.xp) + npix[q], xp);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double dx = sub_nofma(xp[e], xv[q][e]);
            const double dh = sub_nofma(xp[e], xh[e]);
            v[0] = fma(dx, g0v[e], v[0]);
            v[1] = fma(dx, dx, v[1]);
            v[2] = fma(dh, dh, v[2]);
            v[3] = fma(g0v[e], g0v[e], v[3]);
          }
        }
        // ---- finish row i0+t-1: z_new, r_new, g1, BB terms ----
        const double right_y = __shfl_down(xp_prev.y, 1, 64);
        double zo;
        {
#pragma clang fp contract(off)
          const double a0 = xp.x - xp_prev.x;
          const double a1 = right_y - xp_prev.y;
          zo = a0 + a1;
        }
        const double rn = sub_nofma(zo, b_prev);
        const double rn_left = __shfl_up(rn, 1, 64);
        if (own && t >= 1u) {                                      // rows i0 .. i0+rows-1
          const uint32_t orow = i0 + t - 1u;
          store_f64<NT>(p.zn + (uint64_t)orow * p.W + c, zo);
          fs = fma(rn, rn, fs);
          d2 g1;
          g1.x = sub_nofma(rn_prev, rn);
          g1.y = sub_nofma(rn_left, rn);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double dx = sub_nofma(xp_prev[e], x0_prev[e]);
            const double dg = bb_dgrad(g1[e], xh_prev[e], x0_prev[e], p.tau);
            const double dh = sub_nofma(xp_prev[e], xh_prev[e]);
            u[0] = fma(dx, dg, u[0]);
            u[1] = fma(dg, dg, u[1]);
            u[2] = fma(dh, dh, u[2]);
            u[3] += fabs(xp_prev[e]);
            u[4] = fmax(u[4], fabs(xp_prev[e]));
          }
        }
        rn_prev = rn;
        x0_prev = xv[q]; xh_prev = xh; xp_prev = xp; b_prev = bv[q]; rc_up = r_n;
      }
    }
  }
  double w[8] = {fs, v[0], v[1], v[2], v[3], u[0], u[1], u[2]};
  block_reduce<8>(w, s_scr, -1);
  double w2[2] = {u[3], u[4]};
  block_reduce<2>(w2, s_scr, 1);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) store_partial(p.red + (uint64_t)blockIdx.x * 16 + k, w[k]);
    store_partial(p.red + (uint64_t)blockIdx.x * 16 + 8, w2[0]);
    store_partial(p.red + (uint64_t)blockIdx.x * 16 + 9, w2[1]);
  }
  if (!arrive_last(p.counter, gridDim.x, s_flag)) return;
  double t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const double qv = load_partial(p.red + (uint64_t)i * 16 + k);
      if (k == 9) t[k] = fmax(t[k], qv); else t[k] += qv;
    }
  }
  {
    double a[8] = {t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]};
    block_reduce<8>(a, s_scr, -1);
    double bq[2] = {t[8], t[9]};
    block_reduce<2>(bq, s_scr, 1);
    if (tid == 0) {
      p.out[S_FSQ] = a[0]; p.out[S_DXG0] = a[1]; p.out[S_DX2] = a[2]; p.out[S_XH2] = a[3]; p.out[S_G02] = a[4];
      p.out[S_GSUM] = bq[0]; p.out[S_GMAX] = bq[1]; p.out[S_RDOT] = 0.0;
      p.out[S_DXDG] = a[5]; p.out[S_DG2] = a[6]; p.out[S_XH2_ADJ] = a[7];
      p.out[S_GSUM_ADJ] = bq[0]; p.out[S_GMAX_ADJ] = bq[1]; p.out[S_FSQ_ADJ] = a[0];
      p.out[15] = 0.0;
      __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// =================================================================================================
// ONE-PASS FBS iteration for the stencil pair WITH acceleration (FISTA, fasta/__init__.py:220-248).
//
// The extrapolation coefficient of an accelerated step depends on this step's own restart dot
// <x0 - xprox, xprox - x_accel0> (:231) -- a sum over the whole image that is only complete when the sweep ends,
// while g1 = grad(z1' - b) and the BB sums need the coefficient pixel by pixel.  Two things make one sweep enough:
//   * the sweep carries BOTH candidates (c = coef and c = 0, the restart case) through the g1 / BB arithmetic -- a
//     second residual stream in registers and five more accumulators, no extra memory traffic -- and the finalising
//     workgroup, which knows the dot, publishes the set the reference's branch would have computed;
//   * the extrapolated iterate and its image are never written: the state is (last two prox outputs P1, P0, their
//     images Z1 = div P1, Z0 = div P0, the coefficient c_prev that was applied), and the next sweep forms
//     x0 = P1 + c_prev (P1 - P0) and z(x0) = Z1 + c_prev (Z1 - Z0) on the fly -- the same IEEE expressions the
//     reference evaluates at :242-243, so the bits are the ones it would have stored.
// Per pixel: reads P1 16 + P0 16 + Z1 8 + Z0 8 + b 8 (P0 / Z0 skipped while c_prev = 0), writes xprox 16 + z_new 8
// = 80 B against 160 B for the two-launch pair (k_fwd_tv_step 56 + k_adj_tv_step 104 with its x1 / z' outputs).
// Lane / halo layout as k_fused_tv_step.
// =================================================================================================
struct TvAccelP {
  uint32_t H, W, strip_groups, rows_wg;
  const double* p1; const double* p0;   // (H,W,2): the last two prox outputs; x0 = p1 + cprev*(p1 - p0); x_accel0 of this step = p1
  double* pn;                            // (H,W,2): this step's prox output
  const double* z1; const double* z0;   // (H,W): div of p1 / p0; z at x0 = z1 + cprev*(z1 - z0); z_accel0 of this step = z1
  double* zn;                            // (H,W): div of this step's prox output
  const double* b;
  double tau, cprev, coef;
  int restart;
  double* red; unsigned* counter; double* out;
};

template <int IDENT, int TV_U, int NT>
__global__ __launch_bounds__(FH_WG) void k_fused_tv_accel(const TvAccelP p) {
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t sg = blockIdx.x % p.strip_groups, rc = blockIdx.x / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TVF_OWN;
  const uint32_t cw = (first + lane + 2u * p.W - 2u) % p.W;       // lane L <-> image column first + L - 2 (periodic); lanes 2..62 own
  const uint32_t c = first + lane - 2u;
  const bool own = lane >= 2u && lane <= 62u && c < p.W;
  const bool lag = p.cprev != 0.0;                                 // uniform: the previous step extrapolated
  double v[5] = {0, 0, 0, 0, 0};                                  // dxg0, dx2, xh2, g02, restart dot
  double u0[4] = {0, 0, 0, 0};                                    // candidate c = 0   : dxdg, dg2, gsum, gmax  (xh2 = v[2], f = fs)
  double u1[6] = {0, 0, 0, 0, 0, 0};                              // candidate c = coef: dxdg, dg2, xh2, gsum, gmax, f
  double fs = 0.0;

  auto row_of = [&](uint32_t base, int off) -> uint32_t { return tv_wrap_row(base, off, p.H); };
  auto x0_of = [&](d2 p1v, d2 p0v) -> d2 {                        // :242 of the previous iteration, evaluated now
    if (!lag) return p1v;
    d2 x;
    x.x = extrapolate(p1v.x, p0v.x, p.cprev);
    x.y = extrapolate(p1v.y, p0v.y, p.cprev);
    return x;
  };
  auto zc_of = [&](double z1v, double z0v) -> double { return lag ? extrapolate(z1v, z0v, p.cprev) : z1v; };   // :243
  auto resid_at = [&](uint32_t row) -> double {
    const uint64_t pix = (uint64_t)row * p.W + cw;
    const double z1v = load_f64<NT>(p.z1 + pix);
    return sub_nofma(zc_of(z1v, lag ? load_f64<NT>(p.z0 + pix) : 0.0), load_f64<NT>(p.b + pix));
  };
  auto prox_pixel = [&](d2 x0v, double r_me, double r_up, double r_left, d2& g0v, d2& xh) -> d2 {
    g0v.x = sub_nofma(r_up, r_me);
    g0v.y = sub_nofma(r_left, r_me);
    xh.x = fwd_point(x0v.x, g0v.x, p.tau);
    xh.y = fwd_point(x0v.y, g0v.y, p.tau);
    return IDENT ? xh : tv_ball(xh);
  };

  // ---- prologue: halo row i0-1 (needs the residual of rows i0-2 and i0-1) ----------------------------------------
  double rc_up = resid_at(row_of(i0, -2));
  d2 x0_prev, xh_prev, xp_prev, p1_prev;
  double b_prev, z1_prev, rn0_prev = 0.0, rn1_prev = 0.0;
  {
    const uint64_t pix = (uint64_t)row_of(i0, -1) * p.W + cw;
    p1_prev = load_stream<NT>(reinterpret_cast<const d2*>(p.p1) + pix);
    d2 p0v = {0.0, 0.0};
    if (lag) p0v = load_stream<NT>(reinterpret_cast<const d2*>(p.p0) + pix);
    x0_prev = x0_of(p1_prev, p0v);
    b_prev = load_f64<NT>(p.b + pix);
    z1_prev = load_f64<NT>(p.z1 + pix);
    const double r_me = sub_nofma(zc_of(z1_prev, lag ? load_f64<NT>(p.z0 + pix) : 0.0), b_prev);
    const double r_left = __shfl_up(r_me, 1, 64);
    d2 g0v;
    xp_prev = prox_pixel(x0_prev, r_me, rc_up, r_left, g0v, xh_prev);
    rc_up = r_me;
  }
  // step t loads row i0+t and finishes z_new / the BB terms of row i0+t-1 (t = 0 finishes the halo row: only its r_new is kept)
  for (uint32_t t0 = 0; t0 <= rows; t0 += TV_U) {
    d2 xv1[TV_U], xv0[TV_U];
    double zv1[TV_U], zv0[TV_U], bv[TV_U];
    uint64_t npix[TV_U];
#pragma unroll
    for (int q = 0; q < TV_U; ++q) {
      const uint32_t t = min(t0 + q, rows);
      npix[q] = (uint64_t)row_of(i0, (int)t) * p.W + cw;
      xv1[q] = load_stream<NT>(reinterpret_cast<const d2*>(p.p1) + npix[q]);
      zv1[q] = load_f64<NT>(p.z1 + npix[q]);
      bv[q] = load_f64<NT>(p.b + npix[q]);
      xv0[q] = (d2){0.0, 0.0};
      zv0[q] = 0.0;
      if (lag) {
        xv0[q] = load_stream<NT>(reinterpret_cast<const d2*>(p.p0) + npix[q]);
        zv0[q] = load_f64<NT>(p.z0 + npix[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < TV_U; ++q) {
      const uint32_t t = t0 + q;
      if (t <= rows) {                                             // wave-uniform
        // ---- new row i0+t: x0, forward point, prox (owned when t < rows) ----
        const d2 x0v = x0_of(xv1[q], xv0[q]);
        const double r_n = sub_nofma(zc_of(zv1[q], zv0[q]), bv[q]);
        const double r_left = __shfl_up(r_n, 1, 64);
        d2 g0v, xh;
        const d2 xp = prox_pixel(x0v, r_n, rc_up, r_left, g0v, xh);
        if (own && t < rows) {
          store_d2<NT>(reinterpret_cast<d2*>(p.pn) + npix[q], xp);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double dx = sub_nofma(xp[e], x0v[e]);
            const double dh = sub_nofma(xp[e], xh[e]);
            v[0] = fma(dx, g0v[e], v[0]);
            v[1] = fma(dx, dx, v[1]);
            v[2] = fma(dh, dh, v[2]);
            v[3] = fma(g0v[e], g0v[e], v[3]);
            v[4] = fma(sub_nofma(x0v[e], xp[e]), sub_nofma(xp[e], xv1[q][e]), v[4]);      // x_accel0 = P1 (:222, :231)
          }
        }
        // ---- finish row i0+t-1: z_new, both candidates of z1' / r_new / g1 / the BB terms ----
        const double right_y = __shfl_down(xp_prev.y, 1, 64);
        double zo;
        {
#pragma clang fp contract(off)
          const double a0 = xp.x - xp_prev.x;
          const double a1 = right_y - xp_prev.y;
          zo = a0 + a1;
        }
        const double rn0 = sub_nofma(zo, b_prev);
        const double rn1 = sub_nofma(extrapolate(zo, z1_prev, p.coef), b_prev);          // z_accel0 = Z1 (:224, :243)
        const double rn0_left = __shfl_up(rn0, 1, 64);
        const double rn1_left = __shfl_up(rn1, 1, 64);
        if (own && t >= 1u) {
          store_f64<NT>(p.zn + (uint64_t)(i0 + t - 1u) * p.W + c, zo);
          fs = fma(rn0, rn0, fs);
          u1[5] = fma(rn1, rn1, u1[5]);
          d2 ga, gb;
          ga.x = sub_nofma(rn0_prev, rn0);  ga.y = sub_nofma(rn0_left, rn0);
          gb.x = sub_nofma(rn1_prev, rn1);  gb.y = sub_nofma(rn1_left, rn1);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double dx = sub_nofma(xp_prev[e], x0_prev[e]);
            const double dga = bb_dgrad(ga[e], xh_prev[e], x0_prev[e], p.tau);
            const double dgb = bb_dgrad(gb[e], xh_prev[e], x0_prev[e], p.tau);
            const double x1 = extrapolate(xp_prev[e], p1_prev[e], p.coef);              // :242
            const double dh = sub_nofma(x1, xh_prev[e]);
            u0[0] = fma(dx, dga, u0[0]);
            u0[1] = fma(dga, dga, u0[1]);
            u0[2] += fabs(xp_prev[e]);
            u0[3] = fmax(u0[3], fabs(xp_prev[e]));
            u1[0] = fma(dx, dgb, u1[0]);
            u1[1] = fma(dgb, dgb, u1[1]);
            u1[2] = fma(dh, dh, u1[2]);
            u1[3] += fabs(x1);
            u1[4] = fmax(u1[4], fabs(x1));
          }
        }
        rn0_prev = rn0; rn1_prev = rn1;
        x0_prev = x0v; xh_prev = xh; xp_prev = xp; b_prev = bv[q]; p1_prev = xv1[q]; z1_prev = zv1[q]; rc_up = r_n;
      }
    }
  }
  // partials per workgroup: [0] fs [1..4] v0..v3 [5] rdot [6..8] u0 dxdg, dg2, gsum [9..13] u1 dxdg, dg2, xh2, gsum, f  [14] u0 gmax [15] u1 gmax
  {
    double w[8] = {fs, v[0], v[1], v[2], v[3], v[4], u0[0], u0[1]};
    block_reduce<8>(w, s_scr, -1);
    double w2[6] = {u0[2], u1[0], u1[1], u1[2], u1[3], u1[5]};
    block_reduce<6>(w2, s_scr, -1);
    double w3[2] = {u0[3], u1[4]};
    { const double m0 = wave_max(w3[0]), m1 = wave_max(w3[1]); if (lane == 0) { s_scr[wave * 2] = m0; s_scr[wave * 2 + 1] = m1; } }
    __syncthreads();
    if (tid == 0) {
      double* slot = p.red + (uint64_t)blockIdx.x * 16;
#pragma unroll
      for (int k = 0; k < 8; ++k) store_partial(slot + k, w[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) store_partial(slot + 8 + k, w2[k]);
      store_partial(slot + 14, fmax(fmax(s_scr[0], s_scr[2]), fmax(s_scr[4], s_scr[6])));
      store_partial(slot + 15, fmax(fmax(s_scr[1], s_scr[3]), fmax(s_scr[5], s_scr[7])));
    }
  }
  if (!arrive_last(p.counter, gridDim.x, s_flag)) return;
  double t[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = 0.0;
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double qv = load_partial(p.red + (uint64_t)i * 16 + k);
      if (k >= 14) t[k] = fmax(t[k], qv); else t[k] += qv;
    }
  }
  {
    double a[8] = {t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]};
    block_reduce<8>(a, s_scr, -1);
    double bq[6] = {t[8], t[9], t[10], t[11], t[12], t[13]};
    block_reduce<6>(bq, s_scr, -1);
    const double m0 = wave_max(t[14]), m1 = wave_max(t[15]);
    if (lane == 0) { s_scr[wave * 2] = m0; s_scr[wave * 2 + 1] = m1; }
    __syncthreads();
    if (tid == 0) {
      const double gmax0 = fmax(fmax(s_scr[0], s_scr[2]), fmax(s_scr[4], s_scr[6]));
      const double gmax1 = fmax(fmax(s_scr[1], s_scr[3]), fmax(s_scr[5], s_scr[7]));
      const double rdot = a[5];
      const bool plain = (p.restart && rdot > 1E-30) || p.coef == 0.0;        // :231 -- the branch the reference takes
      p.out[S_FSQ] = a[0]; p.out[S_DXG0] = a[1]; p.out[S_DX2] = a[2]; p.out[S_XH2] = a[3]; p.out[S_G02] = a[4];
      p.out[S_GSUM] = bq[0]; p.out[S_GMAX] = gmax0; p.out[S_RDOT] = rdot;
      p.out[S_DXDG] = plain ? a[6] : bq[1];
      p.out[S_DG2] = plain ? a[7] : bq[2];
      p.out[S_XH2_ADJ] = plain ? a[3] : bq[3];
      p.out[S_GSUM_ADJ] = plain ? bq[0] : bq[4];
      p.out[S_GMAX_ADJ] = plain ? gmax0 : gmax1;
      p.out[S_FSQ_ADJ] = plain ? a[0] : bq[5];
      p.out[S_ALPHA] = 0.0;
      p.out[15] = 0.0;
      __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif   // FH_EXPERIMENTAL

// out = a + coef*(a - b) elementwise (:242): materialises a lazily-kept iterate for fh_get_vector

__global__ __launch_bounds__(FH_WG) void k_extrapolate_vec(double* out, const double* a, const double* b, double coef, uint64_t len) {
  for (uint64_t i = (uint64_t)blockIdx.x * FH_WG + threadIdx.x; i < len; i += (uint64_t)gridDim.x * FH_WG)
    out[i] = coef != 0.0 ? extrapolate(a[i], b[i], coef) : a[i];
}

// =================================================================================================
// ONE-PASS stencil iteration that does not touch z at all (round 2; replaces k_fused_tv_step / k_fused_tv_accel as the
// default `fh_step` / `fh_step_accel` path of the stencil operator).
//
// z = div(x) is a 3-point combination of values the sweep is reading anyway, so the residual source z_cur = div(x0) is
// RECOMPUTED from the x0 rows in flight instead of being read (8 B/pixel), and z_new = div(xprox) -- needed only for this
// iteration's f, g1 and BB sums -- is never WRITTEN (8 B/pixel): the next sweep recomputes it from the xprox it reads.
// Every recomputed value is the same IEEE expression on the same inputs as the reference's z1 = A(x1) (fasta/__init__.py:187),
// so the bits do not change.  With FISTA (ACCEL) the image of the extrapolated point is z1 + c (z1 - z_accel0) with
// z1 = div(P1), z_accel0 = div(P0) (:243): both are recomputed, separately, from the two prox outputs the sweep reads.
//   per pixel: reads x0 16 + b 8, writes xprox 16 = 40 B   (k_fused_tv_step: 56 B; two launches: 112 B; model: 136 B)
//   FISTA    : reads P1 16 + P0 16 + b 8, writes 16   = 56 B   (k_fused_tv_accel: 80 B; two launches: 160 B)
// Software pipeline over the rows s = i0-2 .. i0+rows+1 of a chunk (row s "arrives" = its loads are consumed):
//   A  z_cur, r_cur of row s-1      from x rows s-1, s and the right neighbour          (lanes 0..62 valid)
//   B  g0, xhat, xprox of row s-1   from r_cur rows s-2, s-1 and the left neighbour      (lanes 1..62)   -> store xprox, forward sums
//   C  z_new, r_new of row s-2      from xprox rows s-2, s-1 and the right neighbour     (lanes 1..61)
//   D  g1 and the BB sums of row s-2 from r_new rows s-3, s-2 and the left neighbour     (lanes 2..61 = the 60 owned columns)
// so a wave strip owns 60 columns (two halo lanes per side) and a chunk reads rows+4 rows for `rows` owned ones.
// =================================================================================================
#define TVZ_OWN 60

// neighbour exchange by DPP wavefront shifts (VALU, no LDS crossbar round trip): lane i <- lane i-1 / lane i+1.  The end lane has
// no source and reads 0 (bound_ctrl): it is a halo lane whose result is never used -- and with no "old" value to preserve, the shift
// is ONE v_mov_b32_dpp per half instead of a copy plus the shift (round 4: 8 fewer vector instructions per pixel row).
__device__ __forceinline__ double tvz_from_left(double v) {       // == __shfl_up(v, 1) on lanes 1..63
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xF, 0xF, true);   // wave_shr:1
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tvz_from_right(double v) {      // == __shfl_down(v, 1) on lanes 0..62
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xF, 0xF, true);   // wave_shl:1
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// running maximum of |x| in ONE v_max_f64 (fmax(m, fabs(x)) compiles to three: hipcc canonicalises both operands first)
__device__ __forceinline__ double tvz_max_abs(double m, double x) {
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(x));
  return r;
}

// ---- LDS-DMA helpers for the RING form of k_tv_onepass (round 4) ----------------------------------------------------------------
// One `global_load_lds_dwordx4`: every lane names its own 16-byte SOURCE, the destination is wave-uniform: M0 base + lane * 16.
// M0 is compiler-reserved, so it is written and restored inside the statement that reads it.  Counted in vmcnt like any load.
__device__ __forceinline__ void tv_glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void tv_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((address_space(3))) const d2* tv_lds_d2;
typedef __attribute__((address_space(3))) const double* tv_lds_f64;

struct TvZP {
  uint32_t H, W, strip_groups, rows_wg;
  const double* p1; const double* p0;   // (H,W,2): x0 = p1 (ACCEL: p1 + cprev*(p1 - p0), the last two prox outputs)
  double* pn;                            // (H,W,2): this step's prox output
  const double* b;                       // (H,W)
  double tau, cprev, coef;
  int restart;
  int xcd_order;                         // 1 = logical workgroup ids dealt out XCD by XCD (tv_xcd_order)
  uint32_t nchunks;                      // strip_groups * row bands; a smaller grid walks the chunk ids with stride gridDim.x
  unsigned seq;                          // (in the struct's padding) != 0: published behind the scalar block by the finaliser (fh_device.h:publish_seq)
  double* red; unsigned* counter; double* out;
  unsigned* arrive;                      // GB_WORDS words of the two-level final arrival (fh_device.h:arrive_last2): ~1260 workgroups finish together here
};

// NT: bit 0 = non-temporal loads, bit 1 = non-temporal stores.  The host only instantiates NT = 2 (non-temporal stores, the default)
// and NT = 0 (plain accesses, FH_TUNE_TV_NT = 3): non-temporal LOADS lose 12 % here (halo columns and rows come back through L2).
// RING > 0 (round 4): the trips are not loaded into registers but prefetched by LDS-DMA into a ring of RING trip slots PER WAVE
// (x1 rows 1 KiB each | x0 rows likewise with ACCEL | b rows 512 B each, two rows per DMA instruction).  The wave that issues a
// DMA is the wave that reads the slot, so its own counted vmcnt is the only ordering needed -- no barrier, no flags; the loads in
// flight cost no registers, so RING - 1 trips stay in flight behind the one being consumed at the register budget of the burst
// form (which has none in flight while it computes).  Needs an even W (16-byte aligned b pieces) and TV_U = 2.
template <int IDENT, int ACCEL, int TV_U, int NT, int NB = 1, int RING = 0>
__global__ __launch_bounds__(FH_WG) void k_tv_onepass(const TvZP p) {
  constexpr int NTL = NT & 1, NTS = (NT >> 1) & 1;
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t w0 = tv_xcd_order(blockIdx.x, gridDim.x, p.xcd_order);      // logical workgroup id
  const bool lag = ACCEL && p.cprev != 0.0;                       // uniform: the previous step extrapolated
  const double rtau = 1.0 / p.tau;
  double v[5] = {0, 0, 0, 0, 0};                                  // dxg0, dx2, xh2, g02, restart dot
  double u0[4] = {0, 0, 0, 0};                                    // c = 0     : dxdg, dg2, gsum, gmax   (xh2 = v[2], f = fs)
  double u1[6] = {0, 0, 0, 0, 0, 0};                              // c = coef  : dxdg, dg2, xh2, gsum, gmax, f   (ACCEL only)
  double fs = 0.0;
  // PERSISTENT form (round 4): a grid smaller than the number of chunks walks the chunk ids w0, w0 + grid, w0 + 2 grid, ... (ids are
  // band-major: all strip groups of a row band, then the next band).  With short chunks the resident workgroups then sweep the image
  // as ONE compact window of a few hundred rows that moves top to bottom -- the order in which the strip walk's traffic runs fastest
  // (profiles/r04_tvshape.txt) -- while the sums stay in registers across chunks and are reduced once.  The assignment is static, so the
  // summation order, hence every bit of the result, is fixed by (grid, rows per chunk).  grid = chunks: one chunk per workgroup (round 3).
  for (uint32_t wg = w0; wg < p.nchunks; wg += gridDim.x) {
  const uint32_t sg = wg % p.strip_groups, rc = wg / p.strip_groups;
  const uint32_t i0 = rc * p.rows_wg;
  const uint32_t rows = min(p.rows_wg, p.H - i0);
  const uint32_t first = (sg * 4u + wave) * TVZ_OWN;
  const uint32_t cw = (first + lane + 2u * p.W - 2u) % p.W;       // lane L <-> image column first + L - 2 (periodic)
  const uint32_t c = first + lane - 2u;                           // valid as an index only for owning lanes
  const bool own = lane >= 2u && lane <= 61u && c < p.W;

  auto row_of = [&](int off) -> uint32_t { return tv_wrap_row(i0, off, p.H); };   // (i0 + off) mod H, periodic
  auto div_at = [&](d2 me, d2 below) -> double {                  // div of one field at (row, col) given the row below; right neighbour by shuffle
    const double right_y = tvz_from_right(me.y);
    const double a0 = sub_nofma(below.x, me.x);                   // roll(Y0, -1, axis 0) - Y0
    const double a1 = sub_nofma(right_y, me.y);                   // roll(Y1, -1, axis 1) - Y1
    return add_nofma(a0, a1);
  };

  // rolling state: index 1 = row s-1, index 2 = row s-2
  d2 P1a = {0, 0}, P0a = {0, 0};               // prox outputs of row s-1 (loaded)
  double b1 = 0.0, b2 = 0.0;                   // targets of rows s-1, s-2
  double rc2 = 0.0;                            // r_cur of row s-2
  d2 x0_2 = {0, 0}, xh_2 = {0, 0}, xp_2 = {0, 0}, q1_2 = {0, 0};   // row s-2: x0, xhat, xprox, P1 (x_accel0 of this step)
  double z1_2 = 0.0;                           // row s-2: div(P1) (z_accel0 of this step)
  double rn0_3 = 0.0, rn1_3 = 0.0;             // r_new of row s-3 (both candidates)

  const int total = (int)rows + 4;             // rows i0-2 .. i0+rows+1
  // One TRIP = TV_U consecutive rows.  load_trip issues a trip's loads (rows past the chunk are clamped: they re-read its last
  // row and are never consumed); eat_trip runs stages A-D on a landed trip.  The empty asm statements pin the issue order
  // (hipcc otherwise sinks every load to just ahead of its first use, i.e. nothing stays in flight across trips).
  struct Trip { d2 x1[TV_U]; d2 x0[TV_U]; double b[TV_U]; };
  auto load_trip = [&](Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < TV_U; ++q) {
      const int s = min(t0 + q, total - 1) - 2;                   // clamp past the chunk (loads stay in bounds, rows not consumed)
      const uint64_t npix = (uint64_t)row_of(s) * p.W + cw;
      T.x1[q] = load_stream<NTL>(reinterpret_cast<const d2*>(p.p1) + npix);
      T.b[q] = load_f64<NTL>(p.b + npix);
      T.x0[q] = (d2){0.0, 0.0};
      if (lag) T.x0[q] = load_stream<NTL>(reinterpret_cast<const d2*>(p.p0) + npix);
    }
    asm volatile("" ::: "memory");
  };
  auto eat_trip = [&](const Trip& T, int t0) {
    const d2 (&xv1)[TV_U] = T.x1;
    const d2 (&xv0)[TV_U] = T.x0;
    const double (&bv)[TV_U] = T.b;
#pragma unroll
    for (int q = 0; q < TV_U; ++q) {
      const int s = t0 + q - 2;                                    // relative row that arrives now
      if (t0 + q < total) {                                        // wave-uniform
        if (s >= -1) {
          // ---- A: image of x0 at row s-1 and its residual ----
          const double z1v = div_at(P1a, xv1[q]);                  // div(P1) at row s-1
          double zc = z1v;
          if (lag) zc = extrapolate(z1v, div_at(P0a, xv0[q]), p.cprev);       // (:243 of the previous iteration)
          const double rc1 = sub_nofma(zc, b1);
          if (s >= 0) {
            // ---- B: forward point and prox at row s-1 ----
            const double r_left = tvz_from_left(rc1);
            d2 x0v = P1a;
            if (lag) { x0v.x = extrapolate(P1a.x, P0a.x, p.cprev); x0v.y = extrapolate(P1a.y, P0a.y, p.cprev); }   // (:242)
            d2 g0v, xh;
            g0v.x = sub_nofma(rc2, rc1);
            g0v.y = sub_nofma(r_left, rc1);
            xh.x = fwd_point(x0v.x, g0v.x, p.tau);
            xh.y = fwd_point(x0v.y, g0v.y, p.tau);
            const d2 xp = IDENT ? xh : tv_ball(xh);
            if (own && s >= 1 && s <= (int)rows) {                 // row s-1 in [0, rows)
              store_d2<NTS>(reinterpret_cast<d2*>(p.pn) + (uint64_t)(i0 + s - 1) * p.W + c, xp);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const double dx = sub_nofma(xp[e], x0v[e]);
                const double dh = sub_nofma(xp[e], xh[e]);
                v[0] = fma(dx, g0v[e], v[0]);
                v[1] = fma(dx, dx, v[1]);
                v[2] = fma(dh, dh, v[2]);
                v[3] = fma(g0v[e], g0v[e], v[3]);
                if (ACCEL) v[4] = fma(sub_nofma(x0v[e], xp[e]), sub_nofma(xp[e], P1a[e]), v[4]);      // x_accel0 = P1 (:222, :231)
              }
            }
            if (s >= 1) {
              // ---- C: z_new and r_new of row s-2 ----
              const double zo = div_at(xp_2, xp);
              const double rn0 = sub_nofma(zo, b2);
              double rn1 = rn0;
              if (ACCEL) rn1 = sub_nofma(extrapolate(zo, z1_2, p.coef), b2);                           // z_accel0 = div(P1) (:224, :243)
              const double rn0_left = tvz_from_left(rn0);
              double rn1_left = rn0_left;
              if (ACCEL) rn1_left = tvz_from_left(rn1);
              if (own && s >= 2) {                                 // ---- D: row s-2 in [0, rows) ----
                fs = fma(rn0, rn0, fs);
                if (ACCEL) u1[5] = fma(rn1, rn1, u1[5]);
                d2 ga, gb;
                ga.x = sub_nofma(rn0_3, rn0);  ga.y = sub_nofma(rn0_left, rn0);
                gb.x = sub_nofma(rn1_3, rn1);  gb.y = sub_nofma(rn1_left, rn1);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  const double dx = sub_nofma(xp_2[e], x0_2[e]);
                  const double dga = bb_dgrad_rcp(ga[e], xh_2[e], x0_2[e], p.tau, rtau);
                  u0[0] = fma(dx, dga, u0[0]);
                  u0[1] = fma(dga, dga, u0[1]);
                  u0[2] += fabs(xp_2[e]);
                  u0[3] = tvz_max_abs(u0[3], xp_2[e]);
                  if (ACCEL) {
                    const double dgb = bb_dgrad_rcp(gb[e], xh_2[e], x0_2[e], p.tau, rtau);
                    const double x1 = extrapolate(xp_2[e], q1_2[e], p.coef);                          // (:242)
                    const double dh = sub_nofma(x1, xh_2[e]);
                    u1[0] = fma(dx, dgb, u1[0]);
                    u1[1] = fma(dgb, dgb, u1[1]);
                    u1[2] = fma(dh, dh, u1[2]);
                    u1[3] += fabs(x1);
                    u1[4] = tvz_max_abs(u1[4], x1);
                  }
                }
              }
              rn0_3 = rn0; rn1_3 = rn1;
            }
            x0_2 = x0v; xh_2 = xh; xp_2 = xp; q1_2 = P1a; z1_2 = z1v;
          }
          rc2 = rc1;
        }
        b2 = b1;
        P1a = xv1[q]; P0a = xv0[q]; b1 = bv[q];
      }
    }
    asm volatile("" ::: "memory");
  };
#ifdef FH_EXPERIMENTAL
  if constexpr (RING > 0) {
    static_assert(TV_U % 2 == 0, "the RING form pairs the rows of a trip for its b pieces");
    constexpr uint32_t XB = (uint32_t)TV_U * 1024u;                       // the x1 rows of one trip
    constexpr uint32_t BOFF = XB * (ACCEL ? 2u : 1u);                     // the b pieces follow the x1 (and x0) rows
    constexpr uint32_t SLOT = BOFF + (uint32_t)TV_U * 512u;
    constexpr int G1 = TV_U + TV_U / 2, G2 = 2 * TV_U + TV_U / 2;         // DMA instructions per trip without / with the P0 stream
    __shared__ __attribute__((aligned(16))) unsigned char s_ring[4 * RING * SLOT];
    const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)s_ring;
    const uint32_t ring = __builtin_amdgcn_readfirstlane(ring0 + wave * ((uint32_t)RING * SLOT));   // this wave's ring (LDS byte address)
    const uint32_t cwb = (first + 2u * (lane & 31u) + 2u * p.W - 2u) % p.W;     // b piece: lane l of each half wave fetches pixels 2l, 2l+1 of the strip
    const bool stores = __builtin_amdgcn_ballot_w64(own) != 0;            // this wave issues store instructions at all
    const int ntrips = (total + TV_U - 1) / TV_U;
    auto issue_trip = [&](int it) {                                       // DMA trip `it` into slot it % RING (rows past the chunk clamped)
      const uint32_t slot = ring + (uint32_t)(it % RING) * SLOT;
      uint64_t ro[TV_U];
#pragma unroll
      for (int q = 0; q < TV_U; ++q) ro[q] = (uint64_t)row_of(min(it * TV_U + q, total - 1) - 2) * p.W;
#pragma unroll
      for (int q = 0; q < TV_U; ++q) {
        tv_glds16(reinterpret_cast<const d2*>(p.p1) + ro[q] + cw, slot + (uint32_t)q * 1024u);
        if (lag) tv_glds16(reinterpret_cast<const d2*>(p.p0) + ro[q] + cw, slot + XB + (uint32_t)q * 1024u);
      }
#pragma unroll
      for (int q = 0; q < TV_U; q += 2) tv_glds16(p.b + (lane < 32u ? ro[q] : ro[q + 1]) + cwb, slot + BOFF + (uint32_t)q * 512u);
    };
    auto read_trip = [&](Trip& T, int it) {
      const uint32_t slot = ring + (uint32_t)(it % RING) * SLOT;
#pragma unroll
      for (int q = 0; q < TV_U; ++q) {
        T.x1[q] = *(tv_lds_d2)(uintptr_t)(slot + (uint32_t)q * 1024u + lane * 16u);
        T.x0[q] = (d2){0.0, 0.0};
        if (lag) T.x0[q] = *(tv_lds_d2)(uintptr_t)(slot + XB + (uint32_t)q * 1024u + lane * 16u);
        T.b[q] = *(tv_lds_f64)(uintptr_t)(slot + BOFF + (uint32_t)q * 512u + lane * 8u);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the slot may be refilled from here on
    };
    for (int it = 0; it < RING - 1; ++it) issue_trip(it);
    for (int it = 0; it < ntrips; ++it) {
      issue_trip(it + RING - 1);                                          // into the slot consumed by the previous iteration
      // Younger than trip it's DMAs: the DMAs of the RING - 1 later trips and the stores of the RING - 1 trips consumed since (vmcnt
      // counts loads, stores and LDS-DMA together, in issue order).  Only when every one of those trips stored all of its TV_U rows
      // is the larger count exact; waiting for FEWER outstanding operations is always safe.
      // (round 4 took the larger count whenever TV_U stores per consumed trip could be assumed; nothing in the build pins the number of
      // store instructions hipcc emits per trip, and assuming too many would read a slot before its DMA has landed.  vmcnt retires
      // loads, stores and LDS-DMA in issue order, so waiting for the DMAs alone is always safe.)
      (void)stores;
      if (lag) tv_wait_vm<(RING - 1) * G2>(); else tv_wait_vm<(RING - 1) * G1>();
      Trip T0;
      read_trip(T0, it);
      eat_trip(T0, it * TV_U);
    }
    tv_wait_vm<0>();                                                      // the clamped DMAs past the chunk still target this wave's ring
  } else
#endif
  if constexpr (NB == 3) {
    // three rotating trips: two stay in flight behind the one being consumed -- the shape that sustains this read/write mix
    // best in scripts/probes/bench_mem/mixprobe.hip (in the burst form below the waves sit in s_waitcnt 65 % of their cycles)
    Trip T0, T1, T2;
    load_trip(T0, 0);
    load_trip(T1, TV_U);
    for (int t0 = 0; t0 < total; t0 += 3 * TV_U) {
      load_trip(T2, t0 + 2 * TV_U); eat_trip(T0, t0);
      load_trip(T0, t0 + 3 * TV_U); eat_trip(T1, t0 + TV_U);
      load_trip(T1, t0 + 4 * TV_U); eat_trip(T2, t0 + 2 * TV_U);
    }
  } else {
    for (int t0 = 0; t0 < total; t0 += TV_U) {                    // burst form (round 2): load a trip, consume it
      Trip T0;
      load_trip(T0, t0);
      eat_trip(T0, t0);
    }
  }
  }   // chunks of this workgroup
  // partials per workgroup: [0] fs [1..4] v0..v3 [5] rdot [6..8] u0 dxdg, dg2, gsum [9..13] u1 dxdg, dg2, xh2, gsum, f  [14] u0 gmax [15] u1 gmax
  {
    double w[8] = {fs, v[0], v[1], v[2], v[3], v[4], u0[0], u0[1]};
    block_reduce<8>(w, s_scr, -1);
    double w2[6] = {u0[2], u1[0], u1[1], u1[2], u1[3], u1[5]};
    block_reduce<6>(w2, s_scr, -1);
    { const double m0 = wave_max(u0[3]), m1 = wave_max(u1[4]); if (lane == 0) { s_scr[wave * 2] = m0; s_scr[wave * 2 + 1] = m1; } }
    __syncthreads();
    if (tid == 0) {
      double* slot = p.red + (uint64_t)w0 * 16;              // by logical id: the finaliser's summation order does not depend on the dealing
#pragma unroll
      for (int k = 0; k < 8; ++k) store_partial(slot + k, w[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) store_partial(slot + 8 + k, w2[k]);
      store_partial(slot + 14, fmax(fmax(s_scr[0], s_scr[2]), fmax(s_scr[4], s_scr[6])));
      store_partial(slot + 15, fmax(fmax(s_scr[1], s_scr[3]), fmax(s_scr[5], s_scr[7])));
    }
  }
  if (!arrive_last2(p.arrive, s_flag)) return;
  double t[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = 0.0;
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double qv = load_partial(p.red + (uint64_t)i * 16 + k);
      if (k >= 14) t[k] = fmax(t[k], qv); else t[k] += qv;
    }
  }
  {
    double a[8] = {t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]};
    block_reduce<8>(a, s_scr, -1);
    double bq[6] = {t[8], t[9], t[10], t[11], t[12], t[13]};
    block_reduce<6>(bq, s_scr, -1);
    const double m0 = wave_max(t[14]), m1 = wave_max(t[15]);
    if (lane == 0) { s_scr[wave * 2] = m0; s_scr[wave * 2 + 1] = m1; }
    __syncthreads();
    if (tid == 0) {
      const double gmax0 = fmax(fmax(s_scr[0], s_scr[2]), fmax(s_scr[4], s_scr[6]));
      const double gmax1 = fmax(fmax(s_scr[1], s_scr[3]), fmax(s_scr[5], s_scr[7]));
      const double rdot = ACCEL ? a[5] : 0.0;
      const bool plain = !ACCEL || (p.restart && rdot > 1E-30) || p.coef == 0.0;      // :231 -- the branch the reference takes
      // (system-scope stores: the block is host-mapped memory that the host reads as soon as the sequence number below arrives)
      scal_store(p.out + S_FSQ, a[0]); scal_store(p.out + S_DXG0, a[1]); scal_store(p.out + S_DX2, a[2]); scal_store(p.out + S_XH2, a[3]);
      scal_store(p.out + S_G02, a[4]); scal_store(p.out + S_GSUM, bq[0]); scal_store(p.out + S_GMAX, gmax0); scal_store(p.out + S_RDOT, rdot);
      scal_store(p.out + S_DXDG, plain ? a[6] : bq[1]);
      scal_store(p.out + S_DG2, plain ? a[7] : bq[2]);
      scal_store(p.out + S_XH2_ADJ, plain ? a[3] : bq[3]);
      scal_store(p.out + S_GSUM_ADJ, plain ? bq[0] : bq[4]);
      scal_store(p.out + S_GMAX_ADJ, plain ? gmax0 : gmax1);
      scal_store(p.out + S_FSQ_ADJ, plain ? a[0] : bq[5]);
      scal_store(p.out + S_ALPHA, 0.0);
      scal_store(p.out + 15, 0.0);
      publish_seq(p.out, p.seq);
    }
  }
  if (tid < GB_GROUPS + 1) __hip_atomic_store(p.arrive + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // leave the arrival counters zero for the next launch
}
