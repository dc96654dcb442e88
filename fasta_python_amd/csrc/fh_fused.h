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
template <int PPT, int NT, int PIPE, int TEAM, int XLDS = 0, int NBO = 0, int F32 = 0>   // NBO: number of row buffers (0 = the schedule's default); F32: float32-storage A
__global__ __launch_bounds__(FH_WG, (fused_wpc<PPT, TEAM, XLDS, F32>())) void k_fused_dense(const FusedP p) {
  typedef typename PieceOf<F32>::type PT;
  constexpr int XD = xd2<F32>();                  // double pairs of x / g1 per 16-byte piece of A
  __shared__ __attribute__((aligned(16))) d2 s_x[XLDS ? PPT * XD * FH_WG : 1];
  __shared__ __attribute__((aligned(16))) d2 s_fin[FH_WG];       // finalise: per-slice partial sums of the team partials
  __shared__ __attribute__((aligned(16))) double s_part[4];
  __shared__ __attribute__((aligned(16))) double s_part2[2][4];  // TEAM == 1: wave partials, double-buffered by trip parity
  __shared__ __attribute__((aligned(16))) double s_bc[2];       // broadcast: r_i
  __shared__ __attribute__((aligned(16))) double s_scr[4 * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: scalar
  const uint32_t team = (p.variant & 2) ? blockIdx.x % p.nteams : blockIdx.x / TEAM;
  const uint32_t mem = (p.variant & 2) ? blockIdx.x / p.nteams : blockIdx.x % TEAM;
  const uint32_t c0 = mem * (FH_WG * PPT) + tid;                // first 16-byte piece of this lane; next at +256
  // the prox runs once per launch (n-side prologue), so its kind is a run-time switch here (K-fwd recomputes it per row
  // group and keeps it a template parameter)
  const int kind = p.px.kind;
  const double level = (kind == PX_LINF || kind == PX_L1BALL) ? *p.px.level : 0.0;
#ifdef FT_PROFILE
  unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  FT_PHASE(0);

  // ---------------- rows of this team (bookkeeping first: the row buffers are declared here so that PRELOAD can fill them early)
  // Row numbers below are TEAM-LOCAL (0 .. r_end-1); grow() maps them to rows of A.  Default: blocked (team t owns a
  // contiguous range of rows_per_team rows); variant bit 32: row-cyclic over the teams (t, t+nteams, ...), i.e. the whole
  // grid streams one contiguous window of nteams rows -- measured equal or a little slower (profiles/r01d_fused_tuning.txt).
  const bool blocked = (p.variant & 32) == 0;
  const uint32_t row_base = blocked ? min(team * p.rows_per_team, p.mp) : team;
  const uint32_t row_step = blocked ? 1u : p.nteams;
  const uint32_t r_begin = 0u;
  const uint32_t r_end = blocked ? min(row_base + p.rows_per_team, p.mp) - row_base
                                 : (team < p.mp ? (p.mp - team + p.nteams - 1u) / p.nteams : 0u);
  const uint32_t r_last = r_end - 1u;                          // only used when the team has rows
  auto grow = [&](uint32_t r) { return row_base + r * row_step; };
  // Row loads are UNCONDITIONAL (callers clamp the row index to the team's last row; the two surplus reads per team are
  // noise): hipcc's waitcnt pass merges the pending-load state of both sides of any branch by taking the SMALLER
  // outstanding count, so a skipped prefetch on one path turns every later `s_waitcnt vmcnt(N)` into "wait for the
  // newest loads too" -- i.e. no prefetch distance at all.
  uint32_t pc[PPT];                                  // this lane's piece indices, clamped to the row's last piece
#pragma unroll
  for (int k = 0; k < PPT; ++k) pc[k] = min(c0 + k * FH_WG, p.ld2 - 1u);
  auto load_row = [&](PT (&buf)[PPT], uint32_t r) {
    const PT* src = reinterpret_cast<const PT*>(p.A) + (uint64_t)grow(r) * p.ldp;
#pragma unroll
    for (int k = 0; k < PPT; ++k) buf[k] = load_stream<NT>(src + pc[k]);
  };
  // number of rotating row buffers of the schedule this instantiation runs (see the three loops below)
  constexpr int NB = NBO ? NBO : (TEAM == 1 ? (PPT >= 8 ? 5 : 6) : (!PIPE ? 3 : (PPT >= 16 ? 3 : (PPT >= 8 ? 5 : 6))));
  PT B[NB][PPT];
  // PRELOAD (round 4, teams of <= 8 members, i.e. n <= 32768 where a launch is short and its fixed cost shows): the first NB - 1 rows
  // do not depend on the n-side prologue, so their loads are issued BEFORE it -- behind the prologue's own x0 / g0 loads, which
  // vmcnt retires first -- and land while the forward point and the prox are computed (profiles/r04_sizes.txt).
  constexpr bool PRELOAD = TEAM <= 8 && !XLDS && PPT <= 8;
  d2 X0[PRELOAD ? PPT : 1][XD], G0[PRELOAD ? PPT : 1][XD], XA[PRELOAD ? PPT : 1][XD];
  if constexpr (PRELOAD) {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
#pragma unroll
      for (int h = 0; h < XD; ++h) {
        const uint32_t ci = pc[k] * XD + h;
        X0[k][h] = reinterpret_cast<const d2*>(p.x0)[ci];
        G0[k][h] = reinterpret_cast<const d2*>(p.g0)[ci];
        XA[k][h] = (d2){0.0, 0.0};
        if (p.accel) XA[k][h] = reinterpret_cast<const d2*>(p.xacc0)[ci];
      }
    }
    if (r_begin < r_end) {
#pragma unroll
      for (int k = 0; k < NB - 1; ++k) load_row(B[k], min(r_begin + k, r_last));
    }
  }

  // ---------------- n-side: forward point + prox for this member's slice (registers); team 0 owns the outputs
  d2 xq[XLDS ? 1 : PPT][XD];
  double v[7] = {0, 0, 0, 0, 0, 0, 0};   // dxg0, dx2, xh2, g02, gsum, gmax (team 0 only); [6]: restart dot (every team)
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const uint32_t c = c0 + k * FH_WG;
    const uint32_t cl = min(c, p.ld2 - 1u);          // lanes past the row's last piece re-read it (results masked out)
#pragma unroll
    for (int h = 0; h < XD; ++h) {
      const uint32_t ci = cl * XD + h, cr = c * XD + h;   // double-pair index: clamped (loads) / real (validity, stores)
      d2 x0v, g0v, xav = {0.0, 0.0};
      if constexpr (PRELOAD) { x0v = X0[k][h]; g0v = G0[k][h]; xav = XA[k][h]; }
      else {
        x0v = reinterpret_cast<const d2*>(p.x0)[ci];
        g0v = reinterpret_cast<const d2*>(p.g0)[ci];
        if (p.accel) xav = reinterpret_cast<const d2*>(p.xacc0)[ci];
      }
      d2 xh, xp;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool valid = (2u * cr + e) < p.n;
        double xhe = fwd_point(x0v[e], g0v[e], p.tau);
        double xpe = prox_scalar_rt(kind, xhe, p.px, level);
        if (!valid) { xhe = 0.0; xpe = 0.0; }
        xh[e] = xhe; xp[e] = xpe;
        if (valid) v[6] = fma(sub_nofma(x0v[e], xpe), sub_nofma(xpe, xav[e]), v[6]);
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
      if (XLDS) s_x[(k * XD + h) * FH_WG + tid] = xp; else xq[XLDS ? 0 : k][h] = xp;      // (each lane only ever reads back its own entries)
      if (team == 0 && c < p.ld2) {   // write-through: other workgroups read these back after the grid barrier
        store_partial16(reinterpret_cast<d2*>(p.xhat), cr, xh);
        store_partial16(reinterpret_cast<d2*>(p.xp), cr, xp);
      }
    }
  }

  FT_PHASE(1);
  // ---------------- rows of this team: one pass, NB rotating register buffers --------------------------------
  d2 ga[PPT][XD];
#pragma unroll
  for (int k = 0; k < PPT; ++k)
#pragma unroll
    for (int h = 0; h < XD; ++h) ga[k][h] = (d2){0.0, 0.0};
  double fs = 0.0, fsa = 0.0;
  bool dead = false;                                              // a spin timed out: stop exchanging, finish fast
  // Least-squares loss (round 4): the row's term of the scalar the host turns into f is the square of the gradient factor the
  // row loop has just formed (r = z - b), so the ONE lane that forms it (lane 0 of wave 0, member 0) adds it up in row order --
  // instead of a pass after the loop that drains vmcnt, barriers and reads z back through L2 (2-3 us of every member-0
  // workgroup, on the critical path into the grid barrier).  The logistic loss keeps that pass (log / exp stay out of the loop).
  const bool lsq_inline = p.loss == LOSS_LSQ;
  // (a macro, not a lambda: captured by reference inside the lane-0 branches of the row loops, fs / fsa ended up in scratch memory,
  // and their scratch loads' vmcnt(0) drained every prefetched row on every trip: +19 % at every size)
#define FT_LSQ_TERMS(zs_, rv_, bi_, gr_)                                                                   \
  do {                                                                                                     \
    if (lsq_inline && (gr_) < p.m) {                                                                       \
      if (p.accel) { fs = add_nofma(fs, loss_term((zs_), (bi_), LOSS_LSQ)); fsa = add_nofma(fsa, ft_sq((rv_))); }          \
      else fs = add_nofma(fs, ft_sq((rv_)));                                                               \
    }                                                                                                      \
  } while (0)
#ifdef FT_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
  // wave 0 (uniform): wait for the eight partials of row r (bounded), return their sum in member order.
  // The slot line is polled with SCALAR loads (`s_load_dwordx16 glc`: past the scalar cache, all eight slots at
  // once): they count on lgkmcnt, so the poll neither waits for this wave's prefetched rows nor for its stores --
  // a vector poll's `s_waitcnt vmcnt(0)` did both (vmcnt retires in order), which put one HBM latency into every
  // trip.  Measured hand-off (scripts/probes/bench_mem/handoff.hip): sc1 store -> s_load glc ~0.5 us within and across
  // XCDs, sc1 store -> sc1 vector load 0.6-0.9 us.  The loop itself is plain C around the asm load: it contains no
  // compiler-visible vector memory operation, so hipcc's vmcnt bookkeeping for the row buffers stays exact.
  typedef unsigned ft_line __attribute__((ext_vector_type(16)));
  constexpr int SL = TEAM < 8 ? 8 : TEAM;          // doubles per row in the slot array: whole 64-byte lines (teams of 2 / 4 use the first slots)
  constexpr int NL = SL / 8;                       // 64-byte slot lines per row
  constexpr int LG = NL < 2 ? NL : 2;              // lines per poll: at most two (32 SGPRs); 32 members poll twice
  constexpr int MG = TEAM < 8 ? TEAM : 8 * LG;     // members per poll
  // (A speculative read of the slot line at the top of the trip through the scalar cache -- no `glc`, compiler-visible --
  // was tried: it returned stale bytes from before the launch's sentinel fill now and then, i.e. WRONG RESULTS; every
  // slot read therefore stays a `glc` load inside the bounded loop below.  profiles/r01d_fused_tuning.txt, item 6.)
  auto poll_line = [&](uint32_t gl, bool live) -> double {     // gl: slot-line number = row of A (or mp + team)
    double zs = 0.0;
    if (live && !dead) {           // (the slot values live in SGPRs inside this branch only: no copies at the joins)
#pragma unroll
      for (int g = 0; g < NL / LG; ++g) {
        ft_line line[LG];
        const double* lp = p.slots + (uint64_t)gl * SL + g * (8 * LG);
        unsigned cnt = 0u;
        for (;;) {
          if (LG == 1) asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(line[0]) : "s"(lp) : "memory");
          else asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                            : "=&s"(line[0]), "=&s"(line[LG - 1]) : "s"(lp) : "memory");
          // (readfirstlane: inline-asm results count as divergent, which would put these compares on the vector ALU)
          unsigned pending = 0u;
#pragma unroll
          for (int j = 0; j < MG; ++j)
            pending |= (unsigned)__builtin_amdgcn_readfirstlane((int)line[j / 8][2 * (j % 8) + 1]) == FT_SENTINEL_HI ? 1u : 0u;
          if (pending == 0u) break;
          if (++cnt >= FT_SPIN_POLLS) {   // give up on the exchange for the rest of the launch: the launch is reported as
            dead = true;                  // timed out (p.err) and its results are discarded, so the values no longer matter
            if (lane == 0) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;                        // (no p.err LOAD anywhere in the loop: it would drain every prefetched row each trip)
          }
          if (!(p.variant & 4)) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < MG; ++j) {                              // member order: deterministic
          const double q = __hiloint2double(__builtin_amdgcn_readfirstlane((int)line[j / 8][2 * (j % 8) + 1]),
                                            __builtin_amdgcn_readfirstlane((int)line[j / 8][2 * (j % 8)]));
          zs = (g == 0 && j == 0) ? q : zs + q;
        }
      }
      if (dead) zs = 0.0;
    }
    return zs;
  };
  // lane 0 of ONE wave, after the barrier that follows the s_part writes: publish this member's partial of row r
  auto post_row = [&](uint32_t r, bool live) {
    // variant bit 64 = FAULT INJECTION for the test-suite: member 7 of team 0 never publishes its first row, so its
    // team-mates must hit the poll budget, raise p.err and let the whole grid drain (no hang)
    const bool sabotage = (p.variant & 64) && team == 0 && mem == TEAM - 1 && r == r_begin;
    if (lane == 0 && live && !sabotage) {
      const uint64_t at = (uint64_t)grow(r) * SL + mem;
      store_partial(p.slots + at, ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3]);
      store_partial(p.slots_next + at, ft_sentinel());      // fire and forget: read by the launch after this one
    }
  };
  auto dot_row = [&](const PT (&buf)[PPT]) -> double {
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      d2 xv[XD];
#pragma unroll
      for (int h = 0; h < XD; ++h) xv[h] = XLDS ? s_x[(k * XD + h) * FH_WG + tid] : xq[XLDS ? 0 : k][h];
      part = piece_dot(buf[k], xv, part);
    }
    return ft_wave_sum(part);
  };
  auto update_row = [&](const PT (&buf)[PPT], double rv) {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      PT a = buf[k];
      // two workgroups per CU (256 registers): make the piece opaque here, or hipcc keeps the float64 conversions that dot_row made of it
      // PIPE rows earlier alive until this update -- twice the registers of the row buffers themselves, i.e. scratch -- instead of converting again
      if constexpr (fused_wpc<PPT, TEAM, XLDS, F32>() == 2) asm volatile("" : "+v"(a));
      piece_axpy(a, rv, ga[k]);
    }
  };
  // ---------------- FISTA: every team needs this step's restart dot before its first row (the gradient is taken at the
  // extrapolated z): the members exchange their slice sums through the team's extra slot line, summed in member order
  double coef = 0.0, rdot = 0.0;
  if (TEAM > 1 && tid == 0)      // the restart-dot line of the next launch's array is re-armed whether or not this launch accelerates
    store_partial(p.slots_next + (uint64_t)(p.mp + team) * (TEAM < 8 ? 8 : TEAM) + mem, ft_sentinel());
  if (p.accel) {
    double w1[1] = {v[6]};
    block_reduce<1>(w1, s_scr, 1);
    if (TEAM == 1) {                // a team of one holds the whole dot already
      if (tid == 0) s_bc[1] = w1[0];
    } else {
      if (tid == 0) store_partial(p.slots + (uint64_t)(p.mp + team) * SL + mem, w1[0]);
      if (wave == 0) {
        const double t = poll_line(p.mp + team, true);
        if (lane == 0) s_bc[1] = t;
      }
    }
    ft_lds_barrier();
    rdot = s_bc[1];
    coef = (p.restart && rdot > 1E-30) ? 0.0 : p.coef;
  }
  const auto* zq = (const __attribute__((address_space(4))) double*)(uintptr_t)p.zacc0;   // z_accel0 (only read when accel)

  // b[r] through the scalar cache (constant address space => s_load, counted by lgkmcnt): as a vector load inside a
  // lane-0 branch it made hipcc drain vmcnt(0) -- all prefetched rows -- at the branch's join on every trip
  const auto* bq = (const __attribute__((address_space(4))) double*)(uintptr_t)p.b;

  if constexpr (TEAM == 1) {
    // ---- a workgroup owns whole rows (n <= 4096): no exchange at all.  One barrier per trip: the wave partials are
    // double-buffered by trip parity, every thread sums them and evaluates the row's gradient factor itself.
    if (r_begin < r_end) {
      const uint32_t trips = ((r_end - r_begin + NB - 1u) / NB) * NB;
      if constexpr (!PRELOAD) {
#pragma unroll
        for (int k = 0; k < NB - 1; ++k) load_row(B[k], min(r_begin + k, r_last));
      }
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint32_t r = r_begin + t + j;
          const bool live = r < r_end;
          const uint32_t gr = grow(min(r, r_last));
          const double bi = bq[gr];
          const double za = p.accel ? zq[gr] : 0.0;
          load_row(B[(j + NB - 1) % NB], min(r + (NB - 1u), r_last));
          const double d = dot_row(B[j]);
          const int par = (t + j) & 1u;
          if (lane == 0) s_part2[par][wave] = d;
          ft_lds_barrier();
          const double zs = ((s_part2[par][0] + s_part2[par][1]) + s_part2[par][2]) + s_part2[par][3];
          const double rv = live ? loss_grad(p.accel ? extrapolate(zs, za, coef) : zs, bi, p.loss) : 0.0;
          if (tid == 0 && live) { store_partial(p.z + gr, zs); FT_LSQ_TERMS(zs, rv, bi, gr); }
          update_row(B[j], rv);
        }
      }
    }
  } else if constexpr (!PIPE) {
    // ---- exchange in line: prefetch r+NB-1 | dot r | exchange r | update r  (NB-1 rows in flight during the exchange)
    auto process_row = [&](PT (&buf)[PPT], uint32_t r, PT (&nbuf)[PPT], uint32_t nr) {   // uniform over the workgroup; r >= r_end: phantom
      const bool live = r < r_end;
      const uint32_t gr = grow(min(r, r_last));
      load_row(nbuf, min(nr, r_last));
      const double bi = bq[gr];
      const double za = p.accel ? zq[gr] : 0.0;
      FT_T(0);
      const double part = dot_row(buf);
      FT_T(1);
      if (lane == 0) s_part[wave] = part;
      ft_lds_barrier();
      FT_T(2);
      if (wave == 0) {
        post_row(min(r, r_last), live);
        FT_T(3);
        const double zs = poll_line(gr, live);
        FT_T(4);
        if (lane == 0) {
          const double rv = live ? loss_grad(p.accel ? extrapolate(zs, za, coef) : zs, bi, p.loss) : 0.0;
          s_bc[0] = rv;
          if (mem == 0 && live) { store_partial(p.z + gr, zs); FT_LSQ_TERMS(zs, rv, bi, gr); }
        }
      }
      FT_T(5);
      ft_lds_barrier();
      FT_T(6);
      update_row(buf, s_bc[0]);
      FT_T(7);
    };
    if (r_begin < r_end) {
      // NB rotating buffers: one row is worked on, NB-1 rows of loads stay in flight across the exchange.  Three buffers of 16
      // pieces are all the registers hold next to the x slice; with the slice in LDS (XLDS) there is room for four or five.
      // Trips are padded to a multiple of NB with phantom rows (clamped loads, nothing posted or polled, factor 0).
      const uint32_t trips = ((r_end - r_begin + NB - 1u) / NB) * NB;
      if constexpr (!PRELOAD) {
#pragma unroll
        for (int k = 0; k < NB - 1; ++k) load_row(B[k], min(r_begin + k, r_last));
      }
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) process_row(B[j], r_begin + t + j, B[(j + NB - 1) % NB], r_begin + t + j + (NB - 1u));
      }
    }
  } else if (r_begin < r_end) {
    // ---- exchange one trip ahead: in trip t the team posts its partials of row t+1 and only then waits for row t's
    // (posted a whole trip earlier, so the poll normally hits at once): the hand-off latency leaves the critical path.
    // Wave 0 (the polling wave) and waves 1-3 run SEPARATE loops with the same barrier count: wave 0 issues its
    // prefetch after its poll (vmcnt retires in order: a fresh row ahead of the poll load would stall it).
    // NB register buffers rotate: row t is held until its update, row t+1 until the next trip, NB-2 rows prefetch.
    // Trips are padded to a multiple of NB with phantom rows (clamped loads, nothing posted or polled, factor 0).
    constexpr int D = PIPE;                                        // rows between a post and its poll
    static_assert(NB >= D + 2, "need at least one prefetching buffer");       // D+1 rows are held, NB-1-D rows prefetch
    const uint32_t trips = ((r_end - r_begin + NB - 1u) / NB) * NB;
    if constexpr (!PRELOAD) {
#pragma unroll
      for (int k = 0; k < NB - 1; ++k) load_row(B[k], min(r_begin + k, r_last));
    }
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < D; ++q) {                                // rows 0..D-1 are posted before the first trip
        const double d0 = dot_row(B[q]);
        if (lane == 0) s_part[0] = d0;
        ft_lds_barrier();
        ft_lds_barrier();
      }
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint32_t r = r_begin + t + j;
          const bool live = r < r_end;
          const double bi = bq[grow(min(r, r_last))];
          const double za = p.accel ? zq[grow(min(r, r_last))] : 0.0;
          load_row(B[(j + NB - 1) % NB], min(r + (NB - 1u), r_last));
          FT_T(0);
          const double d = dot_row(B[(j + D) % NB]);
          if (lane == 0) s_part[0] = d;
          FT_T(1);
          ft_lds_barrier();                                        // wave 1 posts row r+D from s_part[0..3]
          FT_T(2);
          FT_T(3);
          const double zs = poll_line(grow(min(r, r_last)), live);
          FT_T(4);
          if (lane == 0) {
            const double rv = live ? loss_grad(p.accel ? extrapolate(zs, za, coef) : zs, bi, p.loss) : 0.0;
            s_bc[0] = rv;
            if (mem == 0 && live) { store_partial(p.z + grow(r), zs); FT_LSQ_TERMS(zs, rv, bi, grow(r)); }
          }
          FT_T(5);
          ft_lds_barrier();
          FT_T(6);
          update_row(B[j], s_bc[0]);
          FT_T(7);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < D; ++q) {
        const double d0 = dot_row(B[q]);
        if (lane == 0) s_part[wave] = d0;
        ft_lds_barrier();
        if (wave == 1) post_row(r_begin + q, r_begin + q < r_end);
        ft_lds_barrier();
      }
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint32_t r = r_begin + t + j;
          load_row(B[(j + NB - 1) % NB], min(r + (NB - 1u), r_last));
          const double d = dot_row(B[(j + D) % NB]);
          if (lane == 0) s_part[wave] = d;
          ft_lds_barrier();
          // the post is wave 1's, the poll wave 0's: `vmcnt` also counts stores, and a write-through store is only
          // acknowledged ~0.5 us later -- issued by the polling wave it would hold up every poll's `vmcnt(0)`
          if (wave == 1) post_row(r + D, r + D < r_end);
          ft_lds_barrier();
          update_row(B[j], s_bc[0]);
        }
      }
    }
  }

#ifdef FT_PROFILE
  if (blockIdx.x == 0 && tid == 0)
    printf("fused profile (block 0 wave 0, %u rows; s_memtime ticks per row): top %.1f dot %.1f bar1 %.1f post %.1f poll %.1f bcast %.1f bar2 %.1f update %.1f\n",
           r_end - r_begin, (double)prof[0] / (r_end - r_begin), (double)prof[1] / (r_end - r_begin), (double)prof[2] / (r_end - r_begin),
           (double)prof[3] / (r_end - r_begin), (double)prof[4] / (r_end - r_begin), (double)prof[5] / (r_end - r_begin),
           (double)prof[6] / (r_end - r_begin), (double)prof[7] / (r_end - r_begin));
#endif
  FT_PHASE(2);
  // ---------------- loss terms of this team's rows (member 0), off the exchange's critical path: keeping log/exp of the
  // logistic objective out of the row loop also keeps their constants out of its (full) register budget
  if (mem == 0 && !lsq_inline) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t i = tid; i < r_end; i += FH_WG) {
      const uint32_t r = grow(i);
      if (r < p.m) {
        const double zr = load_partial(p.z + r);
        fs += loss_term(zr, bq[r], p.loss);
        if (p.accel) fsa += loss_term(extrapolate(zr, p.zacc0[r], coef), bq[r], p.loss);   // f at the extrapolated point (:245)
      }
    }
  }

  // ---------------- publish this member's slice partial, loss partial and (team 0) n-side partials -------------
#pragma unroll
  for (int k = 0; k < PPT; ++k)
    if (c0 + k * FH_WG < p.ld2) {
#pragma unroll
      for (int h = 0; h < XD; ++h)
        store_partial16(reinterpret_cast<d2*>(p.gpart) + (uint64_t)team * p.nv2, (c0 + k * FH_WG) * XD + h, ga[k][h]);
    }
  {
    double w[8] = {fs, v[0], v[1], v[2], v[3], v[4], v[5], fsa};
    block_reduce<8>(w, s_scr, 6);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) store_partial(p.red + (uint64_t)blockIdx.x * 16 + k, w[k]);
    }
  }

  FT_PHASE(3);
  // ---------------- bounded grid barrier (all workgroups are co-resident: one per CU) -----------------------
  // (two-level: arrivals spread over 32 counters, one release word polled -- 1.5 us instead of 3.6 us at 256 workgroups, 1.6 instead of 8.0
  // at 512; fh_device.h:grid_barrier2.  A timeout sets p.err and the launch runs on to its end: its results are discarded by the host.)
  (void)grid_barrier2(p.gbar, 1u, p.err, FT_SPIN_TICKS, s_flag);

  FT_PHASE(4);
  // ---------------- every workgroup finalises its share of the columns: team-ordered sum + n-side epilogue ----
  AdjP e;                                            // reuse K-adj's per-element epilogue
  e.accel = p.accel; e.coef = coef; e.tau = p.tau;
  double u[5] = {0, 0, 0, 0, 0};                     // dxdg, dg2, xh2, gsum, gmax
  const uint32_t share = (p.nv2 + gridDim.x - 1) / gridDim.x;
  // A workgroup's share is often far fewer columns than it has threads (n = 8192: 16 double pairs) while every column sums
  // nteams partials (up to 256) through L2: the threads split the TEAMS of a column between them (`slices` contiguous team
  // ranges per column, summed in team order, then added in slice order -- a fixed order, so still bitwise repeatable).
  const uint32_t slices = share < FH_WG ? min(FH_WG / max(share, 1u), p.nteams) : 1u;
  const uint32_t tps = (p.nteams + slices - 1) / slices;              // teams per slice
  for (uint32_t t0 = 0; t0 < share; t0 += FH_WG) {
    const uint32_t col = slices > 1 ? tid % share : t0 + tid;
    const uint32_t slice = slices > 1 ? tid / share : 0u;
    const uint32_t c = blockIdx.x * share + col;     // double-pair index into the n-side vectors
    const bool mine = col < share && slice < slices && c < p.nv2;
    d2 g = {0.0, 0.0};
    if (mine) {
      const uint32_t s1 = min((slice + 1u) * tps, p.nteams);
#pragma unroll 8
      for (uint32_t s = slice * tps; s < s1; ++s) g += load_partial16(reinterpret_cast<const d2*>(p.gpart), s * p.nv2 + c);
    }
    if (slices > 1) {                                // uniform over the workgroup
      __syncthreads();
      if (mine) s_fin[slice * share + col] = g;
      __syncthreads();
      if (mine && slice == 0) {
        for (uint32_t q = 1; q < slices; ++q) g += s_fin[q * share + col];
      }
    }
    if (!mine || slice != 0) continue;
    reinterpret_cast<d2*>(p.g1)[c] = g;
    if (p.mode == 0) {
      // xhat / xp were written by team 0 with plain stores earlier in THIS launch: read them back through sc1
      const d2 x0v = reinterpret_cast<const d2*>(p.x0)[c];
      const d2 xpv = load_partial16(reinterpret_cast<const d2*>(p.xp), c);
      const d2 xhv = load_partial16(reinterpret_cast<const d2*>(p.xhat), c);
      d2 xav = {0.0, 0.0};
      if (p.accel) xav = reinterpret_cast<const d2*>(p.xacc0)[c];
      d2 x1v;
      x1v.x = bb_element(e, g.x, x0v.x, xpv.x, xav.x, xhv.x, 2u * c < p.n, u);
      x1v.y = bb_element(e, g.y, x0v.y, xpv.y, xav.y, xhv.y, 2u * c + 1u < p.n, u);
      if (p.accel) reinterpret_cast<d2*>(p.x1)[c] = x1v;
    }
  }
  block_reduce<5>(u, s_scr, 4);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) store_partial(p.red + (uint64_t)blockIdx.x * 16 + 8 + k, u[k]);
  }
  FT_PHASE(5);
#ifdef FT_PROFILE
  if (blockIdx.x == 0 && tid == 0)
    printf("fused phases (block 0, us): prologue %.1f | rows %.1f | loss+publish %.1f | grid barrier %.1f | finalise %.1f\n",
           (phase[1] - phase[0]) * 0.01, (phase[2] - phase[1]) * 0.01, (phase[3] - phase[2]) * 0.01, (phase[4] - phase[3]) * 0.01, (phase[5] - phase[4]) * 0.01);
#endif
  if (!arrive_last2(p.gbar + GB_WORDS, s_flag)) return;
  double w[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
      const double q = load_partial(p.red + (uint64_t)i * 16 + k);
      if (k == 6 || k == 12) w[k] = fmax(w[k], q); else w[k] += q;
    }
  }
  {
    double a[8] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
    block_reduce<8>(a, s_scr, 6);
    double bq[5] = {w[8], w[9], w[10], w[11], w[12]};
    block_reduce<5>(bq, s_scr, 4);
    if (tid == 0) {
      // (system-scope stores: the block may be host-mapped memory that the host reads as soon as the sequence number below arrives)
      scal_store(p.out + S_FSQ, a[0]); scal_store(p.out + S_DXG0, a[1]); scal_store(p.out + S_DX2, a[2]); scal_store(p.out + S_XH2, a[3]);
      scal_store(p.out + S_G02, a[4]); scal_store(p.out + S_GSUM, a[5]); scal_store(p.out + S_GMAX, a[6]);
      scal_store(p.out + S_RDOT, rdot);   // every team computed the same restart dot
      scal_store(p.out + S_DXDG, bq[0]); scal_store(p.out + S_DG2, bq[1]); scal_store(p.out + S_XH2_ADJ, bq[2]); scal_store(p.out + S_GSUM_ADJ, bq[3]);
      scal_store(p.out + S_GMAX_ADJ, bq[4]); scal_store(p.out + S_FSQ_ADJ, p.accel ? a[7] : a[0]);
      scal_store(p.out + S_ALPHA, level);
      if (p.coef_out) *p.coef_out = coef;
      const double timed_out = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.0 : 0.0;   // spin timeout?
      scal_store(p.out + 15, timed_out);
      if (p.pack) { p.pack[0] = a[0]; p.pack[1] = timed_out; p.pack[2] = p.accel ? a[7] : a[0]; }
      publish_seq(p.out, p.px.seq);
      // leave the counters zero for the next launch (every workgroup is past the grid barrier and has taken its final ticket)
      __hip_atomic_store(p.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.err, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid < GB_GROUPS + 2) {      // both blocks: group counters, top counter, release word
    __hip_atomic_store(p.gbar + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.gbar + GB_WORDS + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
