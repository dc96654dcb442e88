// fh_run.h -- the FBS loop itself on the device: up to K iterations of fasta/__init__.py:171-312 in ONE persistent launch (round 5).
//
// Why: a launch of the one-pass kernel has ~45 us of fixed cost and the Python driver adds ~17 us per iteration (profiles/r04_sizes.txt);
// at the reference's own sizes (200 x 1000 ... a few thousand columns) that is most of an iteration.  Here the workgroups stay resident and
// run iteration after iteration; what the host driver decides between two launches -- the non-monotone backtracking test (:195-217), the
// FISTA restart and alpha recursion (:220-238), the Barzilai-Borwein step (:253-270), residuals, best iterate (:272-300) and the four built-in stop
// rules (stopping.py:6-51) -- is decided on the device, identically by EVERY workgroup from the same sums (no broadcast step), and the
// histories of the iterations go back to the host in one block at the end.
//
// Shapes: a workgroup owns whole rows, one workgroup per CU, all resident: n <= 4096 is the "team of one" shape of fh_fused.h (1-8 pieces per
// lane); n in (4096, 7168] runs 10-14 pieces per lane with three row buffers (the wide-row register budget of fh_fused.h) and keeps only x0 and
// the prox output in LDS -- the gradient and x_accel0 are re-read from L2 in every attempt.  Gains there: 5000^2 +40 %, 6000^2 +16 %, 7000^2 +3 %
// over the per-iteration path; 16 pieces (n up to 8192) lose 12 % at 8192^2 and are not instantiated (profiles/r05_device_loop.txt).
// One attempt (an iteration, or a backtracking retry of it):
//   phase A   every workgroup: xhat = x0 - tau g0, xprox = prox(xhat) for the WHOLE n side (so every workgroup holds the forward sums
//             <Dx,g0>, ||Dx||^2, ... and the restart dot itself: nothing to exchange), then its rows: z_i = a_i . xprox, the gradient
//             factor, the rank-1 update of its slice of g1, the loss terms;
//   barrier 1 (slice partials published write-through)
//   phase B   every workgroup sums the partials of its share of the columns in workgroup order -> g1, the BB terms, x1;
//   barrier 2
//   phase C   every workgroup reads all partial sums (13 doubles per workgroup) and runs the controller below: same inputs, same
//             arithmetic, same decision everywhere.  Workgroup 0 also writes the iteration's history record.
// Vectors that one workgroup writes and another reads in the NEXT attempt (g1, x1, xprox, xhat) are stored write-through (sc1) and
// loaded past L1 (sc1), as the CDNA4 guide's Guideline 16 prescribes for in-launch hand-offs; barriers are generation counters with
// bounded spins (a timeout ends the launch with `stopped = 3`; the host adopts the state of the last completed iteration and the caller carries
// on with the per-iteration path: csrc/fasta_hip.hip:fh_run).
// Measured and NOT shipped (round 5): rows of 4096 < n <= 8192 columns with 512-thread workgroups (two waves per SIMD, x0 / g0 in LDS):
// correct, but hipcc spills ~210 registers per lane at two waves per SIMD and an iteration at 8192^2 takes 209 us against 136 us on the
// per-iteration path (profiles/r05_device_loop.txt).
// Not here (the caller keeps the per-iteration path): Python hooks (stop_rule other than the four, func, record_iterates, verbose),
// the level-search prox kinds, float32 storage, row sharding, rows wider than 7168 columns.
#pragma once
#include "fh_fused.h"

// -DFR_PROFILE: workgroup 0 adds up the 100 MHz wall clock of each phase of an attempt and prints the averages at the end (debug builds)
#ifdef FR_PROFILE
#define FR_STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memrealtime(); fr_prof[i] += _t - fr_prev; fr_prev = _t; } while (0)
#else
#define FR_STAMP(i) do { } while (0)
#endif
#include "fh_loop.h"
struct RunP {
  const double* A;
  uint32_t ld2, n, m, mp, ldp, nv2, nteams, rows_per_team;
  double* nbuf[5];           // physical n-side buffers (X pool of three, P pair) as the context holds them at launch
  double* G[2]; double* Z[2]; double* xhat;
  const double* b;
  int loss, prox_kind;
  int nt;                    // 1 = stream A with non-temporal loads, 0 = default cache policy (a matrix that fits the Infinity Cache is re-read from it)
  double mu, lo, hi;
  int g_kind;                // g(x) for the objective: 0 = none, 1 = mu * sum|x|, 2 = mu * max|x|
  RunOpts o;
  int max_steps;
  RunState init;             // the state on entry, BY VALUE (kernel argument: no copy to order against the launch, no cache to go stale)
  double* hist;              // [max_steps][FR_HIST], host-mapped
  RunState* st_out;          // the state on exit, host-mapped
  double* gpart; double* red;
  unsigned* bar;             // GB_WORDS words of the two-level grid barrier (fh_device.h:grid_barrier2), zero on entry
  unsigned* err;
  int hook_attempt;          // test hook (FH_TUNE_TEST_HOOKS bits 8..15): the last workgroup stays away from barrier 1 of this attempt (1-based; 0 = off)
};

// (the grid barriers of the loop: grid_barrier2 of fh_device.h -- two-level, 1.5 us instead of 3.6 us at 256 workgroups)

// One double through the SCALAR unit, past the scalar cache (glc): for a value another wave of this launch has stored write-through
// (a plain scalar load may return the cached bytes of an earlier iteration).  Counts on lgkmcnt, so it does not wait for row prefetches.
__device__ __forceinline__ double fr_sload_glc(const double* ptr) {
  typedef unsigned fr_u2 __attribute__((ext_vector_type(2)));
  // (the address is the same in every lane, but derived from values loaded at run time: readfirstlane makes that provable to hipcc)
  const uintptr_t a = (uintptr_t)ptr;
  const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xFFFFFFFFu));
  fr_u2 r;
  asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(u) : "memory");
  return __hiloint2double((int)r.y, (int)r.x);
}

// Python's max(a, b) on floats: a unless b > a (so a NaN in `a` stays)
__device__ __forceinline__ double fr_pymax(double a, double b) { return b > a ? b : a; }

template <int PPT>
__global__ __launch_bounds__(FH_WG, 1) void k_run_dense(const RunP p) {
  typedef d2 PT;
  constexpr bool XLDS = PPT >= 7;          // 7+ pieces per lane: the dot products read the prox'd x slice from LDS, or the persistent loop spills
  constexpr bool BIG = PPT > 8;            // n in (4096, 7168]: only x0 and the prox output fit the LDS; g0 and x_accel0 come from L2 in every attempt
  // The n-side state of the solve lives in LDS, one copy per workgroup (every workgroup forms the whole forward point and prox itself,
  // so it can also keep x0, x_accel0 and the last prox output across attempts): between two iterations only the new gradient -- summed
  // over all workgroups -- has to be read back; a backtracking retry reads nothing.  PPT x 4 KiB each (PPT = 8: 128 KiB of the 160).
  __shared__ __attribute__((aligned(16))) d2 s_x0[PPT * FH_WG];     // x0
  __shared__ __attribute__((aligned(16))) d2 s_g0[BIG ? 1 : PPT * FH_WG];     // gradient at x0
  __shared__ __attribute__((aligned(16))) d2 s_xa[BIG ? 1 : PPT * FH_WG];     // x_accel0 (FISTA)
  __shared__ __attribute__((aligned(16))) d2 s_x[PPT * FH_WG];      // this attempt's prox output
  __shared__ __attribute__((aligned(16))) d2 s_fin[FH_WG];
  __shared__ __attribute__((aligned(16))) double s_part2[2][4];
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 16];
  __shared__ __attribute__((aligned(16))) double s_ctl[8];
  __shared__ __attribute__((aligned(16))) double s_win[FR_WINDOW_MAX];     // f_hist window: every workgroup keeps its own (identical) copy
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t team = blockIdx.x, G = gridDim.x;
  const uint32_t c0 = tid;
  const uint32_t row_base = min(team * p.rows_per_team, p.mp);
  const uint32_t r_end = min(row_base + p.rows_per_team, p.mp) - row_base;
  const uint32_t r_last = r_end - 1u;
  uint32_t pc[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) pc[k] = min(c0 + k * FH_WG, p.ld2 - 1u);
  auto load_row = [&](PT (&buf)[PPT], uint32_t r) {
    const PT* src = reinterpret_cast<const PT*>(p.A) + (uint64_t)(row_base + r) * p.ldp;
    if (p.nt) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) buf[k] = load_stream<1>(src + pc[k]);
    } else {
#pragma unroll
      for (int k = 0; k < PPT; ++k) buf[k] = load_stream<0>(src + pc[k]);
    }
  };
  constexpr int NB = BIG ? 3 : (PPT >= 7 ? 4 : (PPT >= 5 ? 5 : 6));       // row buffers: what stays out of scratch inside the persistent loop
  const auto* bq = (const __attribute__((address_space(4))) double*)(uintptr_t)p.b;
  const RunOpts o = p.o;

  // ---- solver state (every workgroup keeps its own, identical copy) -----------------------------------------------------------------
  const RunState* st = &p.init;
  double tau = st->tau_next, alpha1 = st->alpha1, max_residual = st->max_residual, best_quality = st->best_quality;
  // (iteration and backtrack counters: 32 bits inside the launch, uniform)
  const unsigned long long it0 = st->iteration, bt0 = st->backtracks;
  unsigned it = 0u, total_bt = 0u;      // relative to it0 / bt0
  // (readfirstlane: these are the same in every lane; telling hipcc so keeps the buffer selection and the loop control on the scalar unit)
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  int xi = uni(st->xi), ti = uni(st->ti), bi = uni(st->bi), pcx = uni(st->pc), gc = uni(st->gc), zc = uni(st->zc), last_accel = uni(st->last_accel);
  int perm[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) perm[q] = uni(st->perm[q]);
  auto Xb = [&](int q) -> double* { return p.nbuf[perm[q]]; };            // X[q]
  auto Pb = [&](int q) -> double* { return p.nbuf[perm[3 + q]]; };        // P[q]
  int stopped = 0, steps = 0, bt = 0;
  unsigned nbar = 0;                                                       // grid barriers passed so far in this launch
  unsigned attempt = 0;                                                    // parity selects one of two `red` blocks (see phase C)
  if (tid < FR_WINDOW_MAX) s_win[tid] = st->f_window[tid];
  __syncthreads();

#ifdef FR_PROFILE
  unsigned long long fr_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fr_prev = __builtin_amdgcn_s_memrealtime();
#endif
  // Row buffers live across attempts: a workgroup streams the SAME rows in every attempt, so the first NB - 1 of them are requested as
  // soon as the previous attempt's row loop has ended and land behind its barriers, column sums and controller (round 5: the row loop of
  // a 16-row block spent half its time filling its pipeline).
  PT B[NB][PPT];
  if (r_end > 0) {
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) load_row(B[k], min((uint32_t)k, r_last));
  }
  bool load_x = true, load_g = true;       // (uniform) the LDS copies of x0 / x_accel0 resp. g0 must be (re)read from memory
  double tau_iter = tau;                   // the step the current ITERATION started with (what a caller that redoes it after a timeout must use)
  while (steps < p.max_steps && !stopped) {
    // ================= one attempt with step `tau` =================
    FR_STAMP(7);
    const double* x0 = Xb(xi);
    const double* g0 = p.G[gc];
    const double* xacc0 = Pb(pcx);
    double* xp_out = Pb(pcx ^ 1);
    double* z_out = p.Z[zc ^ 1];
    const double* zacc0 = p.Z[zc];
    double* g1 = p.G[gc ^ 1];
    double* x1_out = Xb(ti);
    // `red` alternates between two blocks: a workgroup may still be reading the previous attempt's sums (phase C) when a faster one
    // already publishes this attempt's (end of phase A); a block is rewritten only two attempts later, behind a grid barrier
    double* red = p.red + (size_t)(attempt & 1u) * (size_t)G * 16;
    attempt += 1u;
    ProxP px;
    px.kind = p.prox_kind;
    {
#pragma clang fp contract(off)
      px.thr = tau * p.mu;
    }
    px.lo = p.lo; px.hi = p.hi; px.level = nullptr;

    // ---------------- phase A, n side: the whole forward point and prox in every workgroup
    if (load_x) {                          // first attempt of the launch
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        s_x0[k * FH_WG + tid] = load_partial16(reinterpret_cast<const d2*>(x0), pc[k]);
        if constexpr (!BIG) s_xa[k * FH_WG + tid] = o.accelerate ? load_partial16(reinterpret_cast<const d2*>(xacc0), pc[k]) : (d2){0.0, 0.0};
      }
      load_x = false;
    }
    if (load_g) {                          // after every accepted iteration: g1 was summed over all workgroups (phase B)
      if constexpr (!BIG) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) s_g0[k * FH_WG + tid] = load_partial16(reinterpret_cast<const d2*>(g0), pc[k]);
      }
      load_g = false;
    }
    d2 xq[XLDS ? 1 : PPT];
    double v[7] = {0, 0, 0, 0, 0, 0, 0};     // dxg0, dx2, xh2, g02, gsum, gmax, restart dot
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const uint32_t c = c0 + k * FH_WG;
      const d2 x0v = s_x0[k * FH_WG + tid];                                                                  // (each lane reads back only its own entries)
      d2 g0v, xav = {0.0, 0.0};
      if constexpr (BIG) {
        g0v = load_partial16(reinterpret_cast<const d2*>(g0), pc[k]);
        if (o.accelerate) xav = load_partial16(reinterpret_cast<const d2*>(xacc0), pc[k]);
      } else { g0v = s_g0[k * FH_WG + tid]; xav = s_xa[k * FH_WG + tid]; }
      d2 xh, xp;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool valid = (2u * c + e) < p.n;
        double xhe = fwd_point(x0v[e], g0v[e], tau);
        double xpe = prox_scalar_rt(p.prox_kind, xhe, px, 0.0);
        if (!valid) { xhe = 0.0; xpe = 0.0; }
        xh[e] = xhe; xp[e] = xpe;
        if (valid) {
          const double dx = sub_nofma(xpe, x0v[e]);
          const double dh = sub_nofma(xpe, xhe);
          v[0] = fma(dx, g0v[e], v[0]);
          v[1] = fma(dx, dx, v[1]);
          v[2] = fma(dh, dh, v[2]);
          v[3] = fma(g0v[e], g0v[e], v[3]);
          v[4] += fabs(xpe);
          v[5] = fmax(v[5], fabs(xpe));
          v[6] = fma(sub_nofma(x0v[e], xpe), sub_nofma(xpe, xav[e]), v[6]);
        }
      }
      s_x[k * FH_WG + tid] = xp;
      if constexpr (!XLDS) xq[XLDS ? 0 : k] = xp;
      if (team == 0 && c < p.ld2) {
        store_partial16(reinterpret_cast<d2*>(p.xhat), c, xh);
        store_partial16(reinterpret_cast<d2*>(xp_out), c, xp);
      }
    }
    block_reduce<7>(v, s_scr, 5);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) s_ctl[k] = v[k];
    }
    __syncthreads();
    const double dxg0 = s_ctl[0], dx2 = s_ctl[1], xh2 = s_ctl[2], g02 = s_ctl[3], gsum = s_ctl[4], gmax = s_ctl[5], rdot = s_ctl[6];
    // FISTA (:220-238): the restart test and the alpha recursion of THIS attempt
    double alpha0 = 0.0, coef = 0.0, alpha1_new = alpha1;
    if (o.accelerate) {
      alpha0 = alpha1;
      if (o.restart && rdot > 1E-30) alpha0 = 1.0;
      {
#pragma clang fp contract(off)
        alpha1_new = (1.0 + sqrt(1.0 + 4.0 * (alpha0 * alpha0))) / 2.0;
        coef = (alpha0 - 1.0) / alpha1_new;
      }
    }
    __syncthreads();                       // (s_ctl is reused below)
    FR_STAMP(0);

    // ---------------- phase A, rows of this workgroup
    d2 ga[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) ga[k] = (d2){0.0, 0.0};
    double fs = 0.0, fsa = 0.0;
    if (r_end > 0) {
      const uint32_t trips = ((r_end + NB - 1u) / NB) * NB;
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const uint32_t r = t + q;
          const bool live = r < r_end;
          const uint32_t gr = row_base + min(r, r_last);
          const double bi_ = bq[gr];
          double za = 0.0;
          if (o.accelerate) za = fr_sload_glc(zacc0 + gr);          // written by this workgroup in the previous iteration of THIS launch
          // (past the block's last row the prefetch wraps to its FIRST rows: what the next attempt starts with)
          load_row(B[(q + NB - 1) % NB], r + (NB - 1u) < trips ? min(r + (NB - 1u), r_last) : min(r + (NB - 1u) - trips, r_last));
          double part = 0.0;
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            const d2 xv = XLDS ? s_x[k * FH_WG + tid] : xq[XLDS ? 0 : k];
            part = fma(B[q][k].x, xv.x, part); part = fma(B[q][k].y, xv.y, part);
          }
          const double d = wave_sum(part);
          const int par = (t + q) & 1u;
          if (lane == 0) s_part2[par][wave] = d;
          ft_lds_barrier();
          const double zs = ((s_part2[par][0] + s_part2[par][1]) + s_part2[par][2]) + s_part2[par][3];
          const double zx = o.accelerate ? extrapolate(zs, za, coef) : zs;
          const double rv = live ? loss_grad(zx, bi_, p.loss) : 0.0;
          if (tid == 0 && live) {
            store_partial(z_out + gr, zs);
            if (gr < p.m) {
              if (p.loss == LOSS_LSQ) {
                if (o.accelerate) { fs = add_nofma(fs, loss_term(zs, bi_, LOSS_LSQ)); fsa = add_nofma(fsa, ft_sq(rv)); }
                else fs = add_nofma(fs, ft_sq(rv));
              } else {
                fs += loss_term(zs, bi_, p.loss);
                if (o.accelerate) fsa += loss_term(zx, bi_, p.loss);
              }
            }
          }
#pragma unroll
          for (int k = 0; k < PPT; ++k) { ga[k].x = fma(B[q][k].x, rv, ga[k].x); ga[k].y = fma(B[q][k].y, rv, ga[k].y); }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k)
      if (c0 + k * FH_WG < p.ld2) store_partial16(reinterpret_cast<d2*>(p.gpart) + (uint64_t)team * p.nv2, c0 + k * FH_WG, ga[k]);
    if (tid == 0) { store_partial(red + (uint64_t)team * 16, fs); store_partial(red + (uint64_t)team * 16 + 7, fsa); }
    FR_STAMP(1);
    if (p.hook_attempt && (int)attempt == p.hook_attempt && team == G - 1u && G > 1u) { stopped = 3; break; }      // (test hook: this workgroup never arrives)
    if (!grid_barrier2(p.bar, ++nbar, p.err, FT_SPIN_TICKS, s_flag)) { stopped = 3; break; }
    FR_STAMP(2);

    // ---------------- phase B: this workgroup's share of the columns (the split of k_fused_dense's finaliser)
    AdjP e;
    e.accel = o.accelerate; e.coef = coef; e.tau = tau;
    double u[5] = {0, 0, 0, 0, 0};                     // dxdg, dg2, xh2 (x1), gsum (x1), gmax (x1)
    const uint32_t share = (p.nv2 + G - 1) / G;
    const uint32_t slices = share < FH_WG ? min(FH_WG / max(share, 1u), G) : 1u;
    const uint32_t tps = (G + slices - 1) / slices;
    for (uint32_t t0 = 0; t0 < share; t0 += FH_WG) {
      const uint32_t col = slices > 1 ? tid % share : t0 + tid;
      const uint32_t slice = slices > 1 ? tid / share : 0u;
      const uint32_t c = team * share + col;
      const bool mine = col < share && slice < slices && c < p.nv2;
      d2 g = {0.0, 0.0};
      if (mine) {
        const uint32_t s1 = min((slice + 1u) * tps, G);
#pragma unroll 8
        for (uint32_t s = slice * tps; s < s1; ++s) g += load_partial16(reinterpret_cast<const d2*>(p.gpart), s * p.nv2 + c);
      }
      if (slices > 1) {
        __syncthreads();
        if (mine) s_fin[slice * share + col] = g;
        __syncthreads();
        if (mine && slice == 0) {
          for (uint32_t q = 1; q < slices; ++q) g += s_fin[q * share + col];
        }
      }
      if (!mine || slice != 0) continue;
      store_partial16(reinterpret_cast<d2*>(g1), c, g);
      const d2 x0v = load_partial16(reinterpret_cast<const d2*>(x0), c);
      const d2 xpv = load_partial16(reinterpret_cast<const d2*>(xp_out), c);
      const d2 xhv = load_partial16(reinterpret_cast<const d2*>(p.xhat), c);
      d2 xav = {0.0, 0.0};
      if (o.accelerate) xav = load_partial16(reinterpret_cast<const d2*>(xacc0), c);
      d2 x1v;
      x1v.x = bb_element(e, g.x, x0v.x, xpv.x, xav.x, xhv.x, 2u * c < p.n, u);
      x1v.y = bb_element(e, g.y, x0v.y, xpv.y, xav.y, xhv.y, 2u * c + 1u < p.n, u);
      if (o.accelerate) store_partial16(reinterpret_cast<d2*>(x1_out), c, x1v);
    }
    block_reduce<5>(u, s_scr, 4);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) store_partial(red + (uint64_t)team * 16 + 8 + k, u[k]);
    }
    FR_STAMP(3);
    if (!grid_barrier2(p.bar, ++nbar, p.err, FT_SPIN_TICKS, s_flag)) { stopped = 3; break; }
    FR_STAMP(4);

    // ---------------- phase C: every workgroup adds up all partial sums (workgroup order) and runs the controller
    double w[7] = {0, 0, 0, 0, 0, 0, 0};              // fs, fsa, dxdg, dg2, xh2', gsum', gmax'
    for (uint32_t i = tid; i < G; i += FH_WG) {
      w[0] += load_partial(red + (uint64_t)i * 16);
      w[1] += load_partial(red + (uint64_t)i * 16 + 7);
      w[2] += load_partial(red + (uint64_t)i * 16 + 8);
      w[3] += load_partial(red + (uint64_t)i * 16 + 9);
      w[4] += load_partial(red + (uint64_t)i * 16 + 10);
      w[5] += load_partial(red + (uint64_t)i * 16 + 11);
      w[6] = fmax(w[6], load_partial(red + (uint64_t)i * 16 + 12));
    }
    block_reduce<7>(w, s_scr, 6);
    if (tid == 0) {
      // ---- the host driver's decisions (fasta_python_amd/solver.py:step, line for line), one thread per workgroup, identical everywhere
#pragma clang fp contract(off)
      const double fsq = w[0], fsq_adj = o.accelerate ? w[1] : w[0];
      auto fval = [&](double s) -> double { if (p.loss != LOSS_LSQ) return s; const double q = sqrt(s); return .5 * (q * q); };
      double f1 = fval(fsq);
      bool retry = false;
      if (o.backtrack) {                                                          // :195-217
        const unsigned long long ita = it0 + it;
        const unsigned long long lo_ = ita + 1ull > (unsigned long long)o.window ? ita + 1ull - (unsigned long long)o.window : 0ull;
        double M = s_win[lo_ % FR_WINDOW_MAX];
        for (unsigned long long j = lo_ + 1ull; j <= ita; ++j) M = fr_pymax(M, s_win[j % FR_WINDOW_MAX]);
        const double dxn = sqrt(dx2);
        if (f1 - (M + dxg0 + (dxn * dxn) / (2.0 * tau)) > 1E-12 && bt < o.max_backtracks) retry = true;
      }
      double out[8];
      if (retry) {
        out[0] = 1.0; out[1] = tau * o.stepsize_shrink;
      } else {
        if (o.accelerate) f1 = fval(fsq_adj);                                    // :245
        const double xh2u = o.accelerate ? w[4] : xh2, gsu = o.accelerate ? w[5] : gsum, gmu = o.accelerate ? w[6] : gmax;
        double tau_nx = tau;                                                      // :249
        const double dx_norm = sqrt(dx2);
        if (o.adaptive) {                                                         // :253-270
          const double dot = w[2];
          const double tau_s = (dx_norm * dx_norm) / dot;
          const double sg = sqrt(w[3]);
          const double q = dot / (sg * sg);
          const double tau_m = 0.0 > q ? 0.0 : q;                                 // Python's max(q, 0)
          tau_nx = (2.0 * tau_m > tau_s) ? tau_m : tau_s - .5 * tau_m;
          if (tau_nx <= 0.0 || isinf(tau_nx) || isnan(tau_nx)) tau_nx = tau * 1.5;
        }
        const double resid = dx_norm / tau;                                       // :272
        const double a_ = sqrt(g02), b_ = sqrt(xh2u) / tau;
        const double normalizer = (b_ > a_ ? b_ : a_) + 1E-12;                    // max(a, b) + EPSILON  (:274)
        const double norm_resid = resid / normalizer;
        max_residual = fr_pymax(max_residual, resid);                             // :281
        double objective = 0.0, quality = resid;
        if (o.evaluate_objective) {                                               // :284-289
          const double gval = p.g_kind == 1 ? p.mu * gsu : (p.g_kind == 2 ? p.mu * gmu : 0.0);
          objective = f1 + gval;
          quality = objective;
        }
        const bool better = quality < best_quality;                               // :298-300
        if (better) best_quality = quality;
        bool stop = false;                                                        // stopping.py:6-51
        const bool ratio = resid / max_residual < o.tolerance, normed = norm_resid < o.tolerance;
        if (o.stop_rule == 0) stop = resid < o.tolerance;
        else if (o.stop_rule == 1) stop = normed;
        else if (o.stop_rule == 2) stop = ratio;
        else stop = ratio || normed;
        out[0] = 0.0; out[1] = tau_nx; out[2] = better ? 1.0 : 0.0; out[3] = stop ? 1.0 : 0.0; out[4] = f1;
        if (team == 0) {
          double* h = p.hist + (uint64_t)steps * FR_HIST;
          const bool restarted = o.accelerate && o.restart && rdot > 1E-30;      // (:231-233; the reference prints "Restarted acceleration.")
          h[0] = resid; h[1] = norm_resid; h[2] = tau; h[3] = f1; h[4] = objective; h[5] = (double)bt; h[6] = alpha0;
          h[7] = (better ? 1.0 : 0.0) + (restarted ? 2.0 : 0.0);
        }
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) s_ctl[k] = out[k];
      s_ctl[5] = max_residual; s_ctl[6] = best_quality;
    }
    __syncthreads();
    FR_STAMP(5);
    const bool retry = uni(s_ctl[0] != 0.0 ? 1 : 0) != 0;
    if (retry) {
      tau = s_ctl[1];
      bt += 1;
      __syncthreads();
      continue;                                       // same x0 / g0, smaller step (:204-215)
    }
    // ---- accepted: commit (fh_commit's pointer bookkeeping), by every workgroup alike
    const bool better = uni(s_ctl[2] != 0.0 ? 1 : 0) != 0;
    const double f1 = s_ctl[4];
    if (tid == 0) s_win[(it0 + it + 1ull) % FR_WINDOW_MAX] = f1;
    max_residual = s_ctl[5]; best_quality = s_ctl[6];
    total_bt += (unsigned)bt;
    bt = 0;
    // the n-side state for the next iteration, in this workgroup's LDS: x0 <- x1 (the extrapolation of bb_element, same arithmetic),
    // x_accel0 <- this prox output; the new gradient is re-read from memory (load_g)
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const d2 xpv = s_x[k * FH_WG + tid];
      d2 x1v = xpv;
      if (o.accelerate) {
        if constexpr (BIG) {
          // x1 itself, as phase B stored it (write-through, before barrier 2; the same extrapolate() arithmetic, zero past the row's end).
          // NOT re-derived from x_accel0 here: that buffer is the NEXT attempt's prox target, which workgroup 0 starts to overwrite in its
          // phase A with no grid barrier in between -- a workgroup that reached this line late would extrapolate from the new prox output
          // (round-5 advisor finding).  x1's buffer becomes x0 and is not written again before the next commit.
          x1v = load_partial16(reinterpret_cast<const d2*>(x1_out), pc[k]);
        } else {
          const d2 xav = s_xa[k * FH_WG + tid];
          x1v.x = extrapolate(xpv.x, xav.x, coef); x1v.y = extrapolate(xpv.y, xav.y, coef);
          const uint32_t c = c0 + k * FH_WG;
          if (!(2u * c < p.n)) x1v.x = 0.0;
          if (!(2u * c + 1u < p.n)) x1v.y = 0.0;
          s_xa[k * FH_WG + tid] = xpv;
        }
      }
      s_x0[k * FH_WG + tid] = x1v;
    }
    load_g = true;
    if (o.accelerate) { alpha1 = alpha1_new; pcx ^= 1; last_accel = 1; }
    else { const int a_ = perm[ti], b_ = perm[3 + (pcx ^ 1)]; perm[ti] = b_; perm[3 + (pcx ^ 1)] = a_; last_accel = 0; }   // std::swap(X[ti], P[pc ^ 1])
    xi = ti;
    if (better) bi = xi;
    for (int k = 0; k < 3; ++k) if (k != xi && k != bi) { ti = k; break; }
    zc ^= 1; gc ^= 1;
    tau = s_ctl[1];
    tau_iter = tau;
    it += 1u;
    steps += 1;
    if (uni(s_ctl[3] != 0.0 ? 1 : 0)) stopped = 1;
    __syncthreads();
  }

#ifdef FR_PROFILE
  if (team == 0 && tid == 0 && attempt)
    printf("run profile (workgroup 0, us per attempt over %u attempts): n-side %.2f | rows+publish %.2f | barrier1 %.2f | columns %.2f | barrier2 %.2f | controller %.2f | commit %.2f\n",
           attempt, fr_prof[0] * 0.01 / attempt, fr_prof[1] * 0.01 / attempt, fr_prof[2] * 0.01 / attempt, fr_prof[3] * 0.01 / attempt,
           fr_prof[4] * 0.01 / attempt, fr_prof[5] * 0.01 / attempt, fr_prof[7] * 0.01 / attempt);
#endif
  if (team == 0 && tid == 0) {
    {
      RunState* d = p.st_out;
      d->tau_next = stopped == 3 ? tau_iter : tau; d->alpha1 = alpha1; d->max_residual = max_residual; d->best_quality = best_quality;
      d->iteration = it0 + it; d->backtracks = bt0 + total_bt; d->stopped = stopped;
      d->xi = xi; d->ti = ti; d->bi = bi; d->pc = pcx; d->gc = gc; d->zc = zc; d->last_accel = last_accel;
#pragma unroll
      for (int k = 0; k < 5; ++k) d->perm[k] = perm[k];
      for (int k = 0; k < FR_WINDOW_MAX; ++k) d->f_window[k] = s_win[k];
    }
    // (steps done = iteration - the iteration count on entry: the host knows both)
  }
}
