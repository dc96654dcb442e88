// fh_tv.h -- fused FBS kernels for the periodic difference-stencil operator pair of
// examples/tv_denoising.py:26-63 (A = div : (H,W,2) -> (H,W),  A^H = grad : (H,W) -> (H,W,2)).
//
//   div(Y)[i,j]    = (Y[i+1,j,0] - Y[i,j,0]) + (Y[i,j+1,1] - Y[i,j,1])        (np.roll(.., -1), :61)
//   grad(X)[i,j,0] =  X[i-1,j] - X[i,j] ;  grad(X)[i,j,1] = X[i,j-1] - X[i,j]  (np.roll(.., +1), :38)
//
// Layout: Y-space vectors are pixel-interleaved float64 pairs (C order of (H,W,2)): one pixel = one
// aligned 16-byte access; X-space vectors are H*W float64.  A workgroup owns a TH x TW pixel tile.
//   K-fwd: forward point + unit-ball prox (examples/tv_denoising.py:89-96) for the tile PLUS a one-pixel
//          halo below/right (recomputed, 7 % extra arithmetic, the loads hit L2), staged in LDS; then
//          z = div(xprox) from LDS, r = z - b, and the line-search reductions.  Algorithmic HBM bytes
//          64*P: read Y0,G0 (32P) + b (8P), write xhat? no -- see DESIGN.md.
//   K-adj: r = z' - b for the tile plus a halo above/left in LDS, g1 = grad(r), BB epilogue.
#pragma once
#include "fh_device.h"

#define TV_TH 16
#define TV_TW 128

struct TvFwdP {
  uint32_t H, W;
  uint32_t tiles_x, tiles_y;
  const double* x0; const double* g0; const double* xacc0;   // (H,W,2)
  double* xhat; double* xp;                                   // (H,W,2)
  const double* b; double* z;                                 // (H,W)
  double tau;
  int sub_b;
  double* red;          // [grid][8]
  unsigned* counter;
  double* out;
};

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

// PLAIN = 1: xprox := x0 (fh_init / fh_apply / Lipschitz probes); PLAIN = 0: FBS step with the TV-ball prox
// (IDENT = 1 swaps the prox for the identity: plain gradient descent on the dual).
template <int PLAIN, int IDENT>
__global__ __launch_bounds__(FH_WG) void k_fwd_tv(const TvFwdP p) {
  __shared__ __attribute__((aligned(16))) d2 s_y[(TV_TH + 1) * (TV_TW + 1)];
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x;
  const uint32_t ty = blockIdx.x / p.tiles_x, tx = blockIdx.x % p.tiles_x;
  const uint32_t i0 = ty * TV_TH, j0 = tx * TV_TW;
  const uint32_t rows = min((uint32_t)TV_TH, p.H - i0), cols = min((uint32_t)TV_TW, p.W - j0);
  const uint32_t span = cols + 1u;
  double v[7] = {0, 0, 0, 0, 0, 0, 0};   // dxg0, dx2, xh2, g02, gsum(unused), gmax(unused), rdot

  // ---- phase 1: prox'd tile + halo (row `rows` = next row, column `cols` = next column, periodic) ----
  for (uint32_t t = tid; t < (rows + 1u) * span; t += FH_WG) {
    const uint32_t r = t / span, c = t % span;
    uint32_t gi = i0 + r; if (gi >= p.H) gi -= p.H;
    uint32_t gj = j0 + c; if (gj >= p.W) gj -= p.W;
    const uint64_t pix = (uint64_t)gi * p.W + gj;
    const d2 x0v = reinterpret_cast<const d2*>(p.x0)[pix];
    d2 xp = x0v;
    if (!PLAIN) {
      const d2 g0v = reinterpret_cast<const d2*>(p.g0)[pix];
      d2 xh;
      xh.x = fwd_point(x0v.x, g0v.x, p.tau);
      xh.y = fwd_point(x0v.y, g0v.y, p.tau);
      xp = IDENT ? xh : tv_ball(xh);
      if (r < rows && c < cols) {        // interior pixel: this workgroup owns its outputs and reductions
        reinterpret_cast<d2*>(p.xhat)[pix] = xh;
        reinterpret_cast<d2*>(p.xp)[pix] = xp;
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
          v[6] = fma(sub_nofma(x0v[e], xp[e]), sub_nofma(xp[e], xav[e]), v[6]);
        }
      }
    }
    s_y[r * (TV_TW + 1) + c] = xp;
  }
  __syncthreads();

  // ---- phase 2: z = div(xprox) from LDS, residual, f partial ----------------------------------------
  double fs = 0.0;
  for (uint32_t t = tid; t < rows * cols; t += FH_WG) {
    const uint32_t r = t / cols, c = t % cols;
    const d2 me = s_y[r * (TV_TW + 1) + c];
    const double dn = s_y[(r + 1) * (TV_TW + 1) + c].x;
    const double rt = s_y[r * (TV_TW + 1) + c + 1].y;
    double zv;
    {
#pragma clang fp contract(off)
      const double t0 = dn - me.x;        // roll(Y0, -1, axis 0) - Y0
      const double t1 = rt - me.y;        // roll(Y1, -1, axis 1) - Y1
      zv = t0 + t1;
    }
    const uint64_t pix = (uint64_t)(i0 + r) * p.W + (j0 + c);
    p.z[pix] = zv;
    const double rv = p.sub_b ? sub_nofma(zv, p.b[pix]) : zv;
    fs = fma(rv, rv, fs);
  }
  double w[8] = {fs, v[0], v[1], v[2], v[3], v[4], v[5], v[6]};
  block_reduce<8>(w, s_scr, -1);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.red[(uint64_t)blockIdx.x * 8 + k] = w[k];
  }
  if (!arrive_last(p.counter, gridDim.x, s_flag)) return;
  double u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] += p.red[(uint64_t)i * 8 + k];
  }
  block_reduce<8>(u, s_scr, -1);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.out[k] = u[k];     // S_FSQ, S_DXG0, S_DX2, S_XH2, S_G02, S_GSUM, S_GMAX, S_RDOT
    __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct TvAdjP {
  uint32_t H, W;
  uint32_t tiles_x, tiles_y;
  const double* z; const double* zacc0; const double* b;
  int sub_b; int accel; double coef; int mode; double tau;
  const double* x0; const double* xp; const double* xacc0; const double* xhat;
  double* x1; double* g1;
  double* red;          // [grid][8]
  unsigned* counter;
  double* out;
};

__global__ __launch_bounds__(FH_WG) void k_adj_tv(const TvAdjP p) {
  __shared__ __attribute__((aligned(16))) double s_r[(TV_TH + 1) * (TV_TW + 1)];
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x;
  const uint32_t ty = blockIdx.x / p.tiles_x, tx = blockIdx.x % p.tiles_x;
  const uint32_t i0 = ty * TV_TH, j0 = tx * TV_TW;
  const uint32_t rows = min((uint32_t)TV_TH, p.H - i0), cols = min((uint32_t)TV_TW, p.W - j0);
  const uint32_t span = cols + 1u;

  // ---- phase 1: residual r = z' - b for the tile plus the halo row above / column left (periodic) -----
  // LDS row 0 / column 0 hold the halo; tile pixel (r,c) sits at (r+1, c+1).
  double fs = 0.0;
  for (uint32_t t = tid; t < (rows + 1u) * span; t += FH_WG) {
    const uint32_t r = t / span, c = t % span;
    const uint32_t gi = (r == 0) ? (i0 == 0 ? p.H - 1u : i0 - 1u) : i0 + r - 1u;
    const uint32_t gj = (c == 0) ? (j0 == 0 ? p.W - 1u : j0 - 1u) : j0 + c - 1u;
    const uint64_t pix = (uint64_t)gi * p.W + gj;
    double zv = p.z[pix];
    if (p.accel) zv = extrapolate(zv, p.zacc0[pix], p.coef);
    const double rv = p.sub_b ? sub_nofma(zv, p.b[pix]) : zv;
    s_r[r * (TV_TW + 1) + c] = rv;
    if (r > 0 && c > 0) fs = fma(rv, rv, fs);
  }
  __syncthreads();

  // ---- phase 2: g1 = grad(r), n-side epilogue ---------------------------------------------------------
  double v[5] = {0, 0, 0, 0, 0};   // dxdg, dg2, xh2, gsum, gmax
  for (uint32_t t = tid; t < rows * cols; t += FH_WG) {
    const uint32_t r = t / cols, c = t % cols;
    const double me = s_r[(r + 1) * (TV_TW + 1) + c + 1];
    d2 g;
    g.x = sub_nofma(s_r[r * (TV_TW + 1) + c + 1], me);         // roll(X, +1, axis 0) - X
    g.y = sub_nofma(s_r[(r + 1) * (TV_TW + 1) + c], me);       // roll(X, +1, axis 1) - X
    const uint64_t pix = (uint64_t)(i0 + r) * p.W + (j0 + c);
    reinterpret_cast<d2*>(p.g1)[pix] = g;
    if (p.mode == 0) {
      const d2 x0v = reinterpret_cast<const d2*>(p.x0)[pix];
      const d2 xpv = reinterpret_cast<const d2*>(p.xp)[pix];
      const d2 xhv = reinterpret_cast<const d2*>(p.xhat)[pix];
      d2 xav = {0.0, 0.0};
      if (p.accel) xav = reinterpret_cast<const d2*>(p.xacc0)[pix];
      d2 x1v;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        double x1 = xpv[e];
        if (p.accel) x1 = extrapolate(xpv[e], xav[e], p.coef);
        const double dx = sub_nofma(xpv[e], x0v[e]);
        const double dg = bb_dgrad(g[e], xhv[e], x0v[e], p.tau);
        const double dh = sub_nofma(x1, xhv[e]);
        v[0] = fma(dx, dg, v[0]);
        v[1] = fma(dg, dg, v[1]);
        v[2] = fma(dh, dh, v[2]);
        v[3] += fabs(x1);
        v[4] = fmax(v[4], fabs(x1));
        x1v[e] = x1;
      }
      if (p.accel) reinterpret_cast<d2*>(p.x1)[pix] = x1v;
    }
  }
  double w[6] = {v[0], v[1], v[2], v[3], v[4], fs};
  block_reduce<6>(w, s_scr, 4);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) p.red[(uint64_t)blockIdx.x * 8 + k] = w[k];
  }
  if (!arrive_last(p.counter, gridDim.x, s_flag)) return;
  double u[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double t = p.red[(uint64_t)i * 8 + k];
      if (k == 4) u[k] = fmax(u[k], t); else u[k] += t;
    }
  }
  block_reduce<6>(u, s_scr, 4);
  if (tid == 0) {
    p.out[S_DXDG] = u[0]; p.out[S_DG2] = u[1]; p.out[S_XH2_ADJ] = u[2]; p.out[S_GSUM_ADJ] = u[3];
    p.out[S_GMAX_ADJ] = u[4]; p.out[S_FSQ_ADJ] = u[5];
    __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
