// fh_fused.h -- ONE-PASS FBS iteration for the dense operator: z = A xprox AND g1 = A^T grad f(z) from a
// single read of A (the two-launch path reads A twice: 2*m*n*8 bytes per iteration; this reads m*n*8).
//
// Why it is possible: g1 = sum_i a_i * r_i with r_i = grad f(a_i . xprox) -- row i is needed twice, first
// whole (the dot product), then again for the rank-1 update.  A 512 KiB row does not fit one CU, so a TEAM of
// 8 co-resident workgroups splits the columns: each member keeps its 1/8 of the row (PPT 16-byte pieces per
// lane) IN REGISTERS, publishes its partial dot product as one write-through (`sc1`) 8-byte store into the
// row's 64-byte slot line, polls the other seven (bounded spin), sums the eight partials in member order,
// and applies r_i * (its row pieces) to its register-resident slice of g1.  Three row buffers rotate so that
// two rows of loads are in flight while a row's partials are being exchanged (hand-off latency ~1-3 us per
// the CDNA4 guide's price list vs ~2.4 us of streaming per row per CU).  The prox'd x slice also lives in
// registers, computed once per launch.
//
//   grid  = (#CUs / 8) teams x 8 members, 256 threads, 1 workgroup per CU (~330 VGPRs => 1 wave per SIMD), so
//           the whole grid is co-resident by construction; every spin is bounded by wall-clock and raises
//           p.err instead of hanging if that assumption is ever violated.
//   team t owns rows [t*rows_per_team, ...); member j owns 16-byte pieces [j*256*PPT, (j+1)*256*PPT).
//   After the rows: slice partials -> workspace, bounded grid barrier, then all workgroups sum the team
//   partials for their share of the columns in team order and run the n-side epilogue (same arithmetic as
//   K-adj's finaliser), last arriver sums the scalars.  No float atomics: bitwise repeatable.
//
// Used speculatively by the host driver (solver.py): the launch assumes the step will be accepted; if the
// backtracking test fails the driver re-runs K-fwd with the smaller step and K-adj as usual (identical
// results either way).  Requires no acceleration (the FISTA coefficient depends on this launch's own
// restart dot) and ld2 == 8*256*PPT (n = 4096*PPT, PPT in {1,2,4,8,16}); anything else uses the two-launch path.
#pragma once
#include "fh_dense.h"

#define FT_TEAM 8
#define FT_SENTINEL_HI 0x7FF8DEADu                  // slot filler: the NaN 0x7FF8DEAD7FF8DEAD (hipMemsetD32)
#define FT_SPIN_TICKS 50000000ull                   // 0.5 s of the 100 MHz s_memrealtime clock (grid barrier)
#define FT_SPIN_POLLS 400000u                       // slot-poll budget: ~1.2 us per poll (sc1 load + s_sleep) => ~0.5 s

// Workgroup barrier for LDS hand-offs inside the row loop.  `__syncthreads()` makes hipcc drain `vmcnt(0)` first,
// which would land every prefetched row before each of the two per-row barriers (pipeline depth 0); this waits
// for the LDS traffic only and leaves the row loads in flight.
__device__ __forceinline__ void ft_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FusedP {
  const double* A;
  uint64_t ld;
  uint32_t ld2, n, m, mp;
  uint32_t nteams, rows_per_team;
  const double* x0; const double* g0;
  double* xhat; double* xp;
  const double* b; double* z;
  double tau;
  int loss;
  int mode;              // 0 = full epilogue, 2 = row-sharded (g1 partial + local loss only)
  ProxP px;
  double* slots;         // [mp][8] partial dot products, pre-filled with the sentinel
  double* gpart;         // [nteams][ld]
  double* g1;
  double* red;           // [grid][16] reduction partials
  unsigned* bar;         // [0] grid barrier arrivals, [1] final arrivals   (zeroed before the launch)
  unsigned* err;         // set to 1 on a spin timeout
  int variant;           // bits: 2 = team members 32 blocks apart (one XCD), 4 = no s_sleep in the poll, 64 = fault injection (tests)
  double* out;
};

__device__ __forceinline__ bool ft_is_sentinel(double v) {
  return (unsigned)(__double_as_longlong(v) >> 32) == FT_SENTINEL_HI && (unsigned)__double_as_longlong(v) == FT_SENTINEL_HI;
}

template <int PPT, int NT, int KIND>
__global__ __launch_bounds__(FH_WG, 1) void k_fused_dense(const FusedP p) {
  __shared__ __attribute__((aligned(16))) double s_part[4];
  __shared__ __attribute__((aligned(16))) double s_bc[2];       // broadcast: r_i
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t team = (p.variant & 2) ? blockIdx.x % p.nteams : blockIdx.x / FT_TEAM;
  const uint32_t mem = (p.variant & 2) ? blockIdx.x / p.nteams : blockIdx.x % FT_TEAM;
  const uint32_t c0 = mem * (FH_WG * PPT) + tid;                // first 16-byte piece of this lane; next at +256
  const double level = (KIND == PX_LINF || KIND == PX_L1BALL) ? *p.px.level : 0.0;

  // ---------------- n-side: forward point + prox for this member's slice (registers); team 0 owns the outputs
  d2 xq[PPT];
  double v[7] = {0, 0, 0, 0, 0, 0, 0};   // dxg0, dx2, xh2, g02, gsum, gmax, (rdot unused: no acceleration here)
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const uint32_t c = c0 + k * FH_WG;
    const d2 x0v = reinterpret_cast<const d2*>(p.x0)[c];
    const d2 g0v = reinterpret_cast<const d2*>(p.g0)[c];
    d2 xh, xp;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool valid = (2u * c + e) < p.n;
      double xhe = fwd_point(x0v[e], g0v[e], p.tau);
      double xpe = prox_scalar<KIND>(xhe, p.px, level);
      if (!valid) { xhe = 0.0; xpe = 0.0; }
      xh[e] = xhe; xp[e] = xpe;
      if (valid && team == 0) {
        const double dx = sub_nofma(xpe, x0v[e]);
        const double dh = sub_nofma(xpe, xhe);
        v[0] = fma(dx, g0v[e], v[0]);
        v[1] = fma(dx, dx, v[1]);
        v[2] = fma(dh, dh, v[2]);
        v[3] = fma(g0v[e], g0v[e], v[3]);
        v[4] += fabs(xpe);
        v[5] = fmax(v[5], fabs(xpe));
      }
    }
    xq[k] = xp;
    if (team == 0) {            // write-through: other workgroups read these back after the grid barrier
      store_partial2(reinterpret_cast<d2*>(p.xhat) + c, xh);
      store_partial2(reinterpret_cast<d2*>(p.xp) + c, xp);
    }
  }

  // ---------------- rows of this team: one pass, three rotating register buffers ----------------------------
  const uint32_t r_begin = min(team * p.rows_per_team, p.mp);
  const uint32_t r_end = min(r_begin + p.rows_per_team, p.mp);
  const uint32_t r_last = r_end - 1u;                          // only used when the team has rows
  const d2* Abase = reinterpret_cast<const d2*>(p.A) + c0;
  d2 ga[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) ga[k] = (d2){0.0, 0.0};
  double fs = 0.0;
  bool dead = false;                                              // a spin timed out: stop exchanging, finish fast

  // Row loads are UNCONDITIONAL (callers clamp the row index to the team's last row; the two surplus reads per team are
  // noise): hipcc's waitcnt pass merges the pending-load state of both sides of any branch by taking the SMALLER
  // outstanding count, so a skipped prefetch on one path turns every later `s_waitcnt vmcnt(N)` into "wait for the
  // newest loads too" -- i.e. no prefetch distance at all.
  auto load_row = [&](d2 (&buf)[PPT], uint32_t r) {
    const d2* src = Abase + (uint64_t)r * p.ld2;
#pragma unroll
    for (int k = 0; k < PPT; ++k) buf[k] = load_stream<NT>(src + k * FH_WG);
  };
  // `vmcnt` retires in order: the polling wave (0) must not have a freshly issued row ahead of its poll loads, so it
  // issues the reload of the freed buffer AFTER the poll; waves 1-3 issue it up front (two rows in flight).
  auto process_row = [&](d2 (&buf)[PPT], uint32_t r, d2 (&nbuf)[PPT], uint32_t nr) {   // r < r_end, uniform over the workgroup
    load_row(nbuf, min(nr, r_last));
    // b[r] through the scalar cache (constant address space => s_load, counted by lgkmcnt): as a vector load inside the
    // lane-0 branch below it made hipcc drain vmcnt(0) -- all prefetched rows -- at the branch's join on every trip
    const double bi = ((const __attribute__((address_space(4))) double*)(uintptr_t)p.b)[r];
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      part = fma(buf[k].x, xq[k].x, part);
      part = fma(buf[k].y, xq[k].y, part);
    }
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    ft_lds_barrier();
    if (wave == 0) {
      double* line = p.slots + (uint64_t)r * FT_TEAM;
      // variant bit 64 = FAULT INJECTION for the test-suite: member 7 of team 0 never publishes its first row, so its
      // team-mates must hit the wall-clock bound, raise p.err and let the whole grid drain (no hang)
      const bool sabotage = (p.variant & 64) && team == 0 && mem == FT_TEAM - 1 && r == r_begin;
      if (lane == 0 && !sabotage) store_partial(line + mem, ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]);
      double val = 0.0;
      if (lane < FT_TEAM && !dead) {
        // The poll loop is written in asm on purpose: a C loop with loads inside this wave-0-only branch makes hipcc's
        // waitcnt pass lose count at the join and emit `s_waitcnt vmcnt(0)` before every later use of the row buffers in
        // ALL waves (measured: pipeline depth 0).  Hidden in asm, the compiler keeps exact counts for the row loads.
        // Bounded by an iteration budget (~0.5 s with the sleep) instead of the clock to stay within 32-bit scalar ops.
        const double* slot = line + lane;
        const unsigned long long sent = ((unsigned long long)FT_SENTINEL_HI << 32) | FT_SENTINEL_HI;
        unsigned long long tmp;
        unsigned cnt = 0u, timed_out;
        asm volatile(
            "s_mov_b32 %[to], 0\n"
            "1:\n\t"
            "global_load_dwordx2 %[val], %[addr], off sc1\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_cmp_ne_u64 vcc, %[sent], %[val]\n\t"
            "s_andn2_b64 %[tmp], exec, vcc\n\t"
            "s_cbranch_scc0 2f\n\t"
            "s_sleep 1\n\t"
            "s_add_u32 %[cnt], %[cnt], 1\n\t"
            "s_cmp_lt_u32 %[cnt], %[max]\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_mov_b32 %[to], 1\n"
            "2:\n"
            : [val] "=&v"(val), [tmp] "=&s"(tmp), [cnt] "+s"(cnt), [to] "=&s"(timed_out)
            : [addr] "v"(slot), [sent] "s"(sent), [max] "s"(FT_SPIN_POLLS)
            : "vcc", "scc", "memory");
        if (timed_out) {       // give up on the exchange for the rest of the launch (no p.err load in the loop:
          dead = true;         // a C-level load there would drain every prefetched row each trip)
          __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ft_is_sentinel(val)) val = 0.0;
        }
      }
      double zs = __shfl(val, 0, 64);
#pragma unroll
      for (int j = 1; j < FT_TEAM; ++j) zs += __shfl(val, j, 64);   // member order: deterministic
      if (lane == 0) {
        s_bc[0] = loss_grad(zs, bi, p.loss);
        if (mem == 0) store_partial(p.z + r, zs);                    // read back below by other lanes of this workgroup
      }
    }
    ft_lds_barrier();
    const double rv = s_bc[0];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      ga[k].x = fma(buf[k].x, rv, ga[k].x);
      ga[k].y = fma(buf[k].y, rv, ga[k].y);
    }
  };

  if (r_begin < r_end) {
    d2 b0[PPT], b1[PPT], b2[PPT];
    load_row(b0, r_begin);
    load_row(b1, min(r_begin + 1u, r_last));
    uint32_t r = r_begin;
    for (; r + 3u <= r_end; r += 3u) {              // branch-free body: exact vmcnt distances (two rows stay in flight)
      process_row(b0, r, b2, r + 2u);
      process_row(b1, r + 1u, b0, r + 3u);
      process_row(b2, r + 2u, b1, r + 4u);
    }
    if (r < r_end) {
      process_row(b0, r, b2, r + 2u);
      if (r + 1u < r_end) process_row(b1, r + 1u, b0, r + 3u);
    }
  }

  // ---------------- loss terms of this team's rows (member 0), off the exchange's critical path: keeping log/exp of the
  // logistic objective out of the row loop also keeps their constants out of its (full) register budget
  if (mem == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const auto* bq = (const __attribute__((address_space(4))) double*)(uintptr_t)p.b;
    for (uint32_t r = r_begin + tid; r < min(r_end, p.m); r += FH_WG) fs += loss_term(load_partial(p.z + r), bq[r], p.loss);
  }

  // ---------------- publish this member's slice partial, loss partial and (team 0) n-side partials -------------
#pragma unroll
  for (int k = 0; k < PPT; ++k)
    store_partial2(reinterpret_cast<d2*>(p.gpart) + (uint64_t)team * p.ld2 + c0 + k * FH_WG, ga[k]);
  {
    double w[8] = {fs, v[0], v[1], v[2], v[3], v[4], v[5], 0.0};
    block_reduce<8>(w, s_scr, 6);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) store_partial(p.red + (uint64_t)blockIdx.x * 16 + k, w[k]);
    }
  }

  // ---------------- bounded grid barrier (all workgroups are co-resident: one per CU) -----------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(p.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(p.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > FT_SPIN_TICKS) {
        __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();

  // ---------------- every workgroup finalises its share of the columns: team-ordered sum + n-side epilogue ----
  AdjP e;                                            // reuse K-adj's per-element epilogue
  e.accel = 0; e.coef = 0.0; e.tau = p.tau;
  double u[5] = {0, 0, 0, 0, 0};                     // dxdg, dg2, xh2, gsum, gmax
  const uint32_t share = (p.ld2 + gridDim.x - 1) / gridDim.x;
  for (uint32_t t = tid; t < share; t += FH_WG) {
    const uint32_t c = blockIdx.x * share + t;
    if (c >= p.ld2) continue;
    const d2* gp = reinterpret_cast<const d2*>(p.gpart) + c;
    d2 g = {0.0, 0.0};
#pragma unroll 8
    for (uint32_t s = 0; s < p.nteams; ++s) g += load_partial2(gp + (uint64_t)s * p.ld2);
    reinterpret_cast<d2*>(p.g1)[c] = g;
    if (p.mode == 0) {
      // xhat / xp were written by team 0 with plain stores earlier in THIS launch: read them back through sc1
      const d2 x0v = reinterpret_cast<const d2*>(p.x0)[c];
      const d2 xpv = load_partial2(reinterpret_cast<const d2*>(p.xp) + c);
      const d2 xhv = load_partial2(reinterpret_cast<const d2*>(p.xhat) + c);
      bb_element(e, g.x, x0v.x, xpv.x, 0.0, xhv.x, 2u * c < p.n, u);
      bb_element(e, g.y, x0v.y, xpv.y, 0.0, xhv.y, 2u * c + 1u < p.n, u);
    }
  }
  block_reduce<5>(u, s_scr, 4);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) store_partial(p.red + (uint64_t)blockIdx.x * 16 + 8 + k, u[k]);
  }
  if (!arrive_last(p.bar + 1, gridDim.x, s_flag)) return;
  double w[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const double q = load_partial(p.red + (uint64_t)i * 16 + k);
      if (k == 6 || k == 12) w[k] = fmax(w[k], q); else w[k] += q;
    }
  }
  {
    double a[8] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], 0.0};
    block_reduce<8>(a, s_scr, 6);
    double bq[5] = {w[8], w[9], w[10], w[11], w[12]};
    block_reduce<5>(bq, s_scr, 4);
    if (tid == 0) {
      p.out[S_FSQ] = a[0]; p.out[S_DXG0] = a[1]; p.out[S_DX2] = a[2]; p.out[S_XH2] = a[3]; p.out[S_G02] = a[4];
      p.out[S_GSUM] = a[5]; p.out[S_GMAX] = a[6]; p.out[S_RDOT] = 0.0;
      p.out[S_DXDG] = bq[0]; p.out[S_DG2] = bq[1]; p.out[S_XH2_ADJ] = bq[2]; p.out[S_GSUM_ADJ] = bq[3];
      p.out[S_GMAX_ADJ] = bq[4]; p.out[S_FSQ_ADJ] = a[0];
      p.out[S_ALPHA] = level;
      p.out[15] = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.0 : 0.0;   // spin timeout?
    }
  }
}
