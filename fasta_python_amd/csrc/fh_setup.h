// fh_setup.h -- the solver's whole SET-UP from ONE read of A (round 5).
//
// Before its first iteration the reference applies A and A^H three times (fasta/__init__.py:100-113: the two Lipschitz probes
// gradf(x1), gradf(x2) with x1, x2 ~ randn; :135-137: z = A x0, f(z), gradf at x0).  The right-hand sides are independent, so a
// multi-column instantiation of the one-pass scheme of fh_fused.h serves them from a single read of the matrix: a member of a team
// keeps its piece of rows t .. t+D in registers (the SAME team shapes, row ranges and buffers as k_fused_dense<PPT, 1, D, TEAM>),
// forms NR partial dot products per row against NR x slices held in LDS, the team exchanges them through NR slot lines per row, and
// every wave applies NR rank-1 updates to NR register-resident gradient slices.
//   * SHIPPED: NR = 2.  For the least-squares loss the probes enter only through their difference, grad(x1) - grad(x2) = A^T A (x1 - x2),
//     so the columns are d = x1 - x2 (homogeneous residual) and x0 (residual z - b); the first version carried all three columns
//     (NR = 3, still instantiable) and spilled into the row loop: 7.8 ms against 5.0-5.5 ms (profiles/r05_setup_cost.txt);
//   * the polls of a row run in PARALLEL: wave 1 posts the member's partials, waves 0 and 2 (and 3 for NR = 3) each poll one
//     right-hand side (scalar loads, bounded) and broadcast its gradient factor -- a trip is no longer than k_fused_dense's;
//   * every right-hand side is summed in exactly the order k_fused_dense sums it (pieces in lane order, DPP wave sum, waves
//     0..3, members 0..TEAM-1, teams 0..nteams-1 with the same slice split in the finaliser), so g0 / z / f come out BIT-IDENTICAL
//     to fh_init's one-pass launch in the 256-thread shapes (the two 512-thread shapes below: equal to summation-order rounding), and L
//     agrees with the three-pass value to ~1e-15 relative -- tests/test_gpu_setup.py;
//   * the finaliser also forms ||A^T A d||^2 and (team 0's prologue) ||x1 - x2||^2: the two norms of :110 come back with the scalar
//     block, no further launch.
// Shapes: PPT <= 8 pieces per lane (n <= 65536 in float64 storage, <= 131072 in float32 storage): NR x PPT x 4 (8) KiB of LDS for the x slices,
// NR x PPT x 4 (8) registers for the gradient slices.  Wider rows and the logistic loss keep the three-pass set-up (fh_setup falls back by itself).  Row blocks
// (round 6: a multi-device context, or a rank of a row-sharded run) launch it per block -- it is linear in the rows like least squares
// itself -- and sum A_k^T A_k d, the gradients and the loss sums in ONE exchange (csrc/fasta_hip.hip:setup_row_blocks).  The slots are filled with the sentinel by the host before the launch (one launch per solve: no re-arming).
#pragma once
#include "fh_fused.h"

#define FS_NR 3       // right-hand sides a SetupP can carry; a kernel uses the first NR of them

struct SetupP {
  const double* A;
  uint32_t ld2, n, m, mp, ldp, nv2;
  uint32_t nteams, rows_per_team;
  const double* x[FS_NR];      // x1, x2 (the probes), x0
  double* g[FS_NR];            // A^H grad f(A x_j)
  double* z;                   // A x0 (m-side, the solver's z_accel1)
  const double* b;
  double* slots;               // [mp][NR][max(TEAM, 8)], every double the sentinel on entry
  double* gpart;               // [nteams][FS_NR][nv2] double pairs
  double* red;                 // [grid][8]
  unsigned* bar;               // [1] final arrivals (zero on entry; the finaliser zeroes it again)
  unsigned* gbar;              // 2 x GB_WORDS words: the two-level grid barrier and final arrival (fh_device.h:grid_barrier2 / arrive_last2), zero on entry, zeroed again at the end
  unsigned* err;
  int variant;                 // bit 2: team members nteams blocks apart (one XCD), bit 4: no sleep between polls, bit 32: rows dealt cyclically (as FusedP.variant)
  double* out;                 // scalar block: [S_FSQ] loss sum at x0, [S_DX2] ||x1 - x2||^2, [S_DG2] ||grad1 - grad2||^2, [15] timeout
  double* pack;                // optional (row blocks, round 6): 2 doubles behind g[0] -- this block's loss sum and its timeout word -- so that the ONE sum
                               // over the row blocks of A_k^T A_k d carries them along; g[0], g[2] then hold this block's PARTIAL sums and [S_DG2] is void
};

// sum of NW wave partials in wave order (NW = 4: the order of k_fused_dense)
template <int NW>
__device__ __forceinline__ double fs_sum_waves(const double* v) {
  double s = v[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) s += v[w];
  return s;
}
// block_reduce of fh_device.h for NW waves (sums only); result valid in thread 0
template <int K, int NW>
__device__ __forceinline__ void fs_block_reduce(double (&v)[K], double* scr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    v[k] = wave_sum(v[k]);
    if (lane == 0) scr[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = scr[k];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += scr[w * K + k];
      v[k] = s;
    }
  }
  __syncthreads();
}

// MPP: 16-byte pieces of a row per member, in units of 256 (the PPT of the team shape: a member covers 256 x MPP pieces);
// NT: threads per workgroup, 256 or 512.  With 512 threads a lane holds ceil(MPP / 2) pieces: two waves per SIMD of at most 256
// registers each, where 256 lanes x 8 pieces x (3 gradient slices + 4 row buffers) does not fit one wave's registers without spilling
// into the loops.  With 256 threads every right-hand side is summed exactly as k_fused_dense sums it (bit-identical results); with
// 512 the lane -> piece map differs and the results agree to summation-order rounding.  SHIPPED with 512 threads: the full 8-piece shapes of 8 and
// 16 members (two right-hand sides double the issue slots per byte; the second wave per SIMD hides the first one's LDS reads, multiply-add chains
// and polls: 65536^2 5.13-5.28 -> 4.97-4.99 ms, at the speed of a step launch); everything else runs 256 threads.
// NR = 3: right-hand sides x1, x2, x0, each with the residual r = z - b (the reference's three passes verbatim).
// NR = 2: the two probes enter only through their DIFFERENCE -- grad(x1) - grad(x2) = A^T A (x1 - x2) for the least-squares loss -- so
//         right-hand side 0 is d = x1 - x2 with the homogeneous residual r = z, right-hand side 1 is x0 with r = z - b: one dot product,
//         one exchange and one rank-1 update fewer per row, 64 instead of 96 accumulator registers.  ||A^T A d|| equals
//         ||grad(x1) - grad(x2)|| up to rounding (no cancellation: the probes are independent), so L = that / ||d|| agrees with the
//         three-pass value to ~1e-15 relative; z, f and g0 do not depend on the probes at all.
// F32 = 1 (round 6): float32 storage of A -- a piece is four columns (fh_device.h: PieceOf), so a lane's x entries and gradient accumulators are
// XD = 2 double pairs per piece: twice the LDS (NR x PPT x 8 KiB) and twice the accumulator registers (NR x PPT x 8) of the float64 kernel;
// 256 threads, one wave per SIMD (512 registers), NR = 2.
template <int MPP, int PIPE, int TEAM, int NT, int NR, int F32 = 0>
__global__ __launch_bounds__(NT, 1) void k_setup_dense(const SetupP p) {
  typedef typename PieceOf<F32>::type PT;
  constexpr int XD = xd2<F32>();
  static_assert(NR == 2 || NR == 3, "two or three right-hand sides");
  static_assert(!F32 || (NR == 2 && NT == 256), "float32 storage: two right-hand sides, 256 threads");
  constexpr int NW = NT / 64;
  constexpr int PPT = (MPP * FH_WG + NT - 1) / NT;                      // pieces per lane
  __shared__ __attribute__((aligned(16))) d2 s_x[NR * PPT * XD * NT];
  __shared__ __attribute__((aligned(16))) d2 s_fin[FH_WG];
  __shared__ __attribute__((aligned(16))) double s_part[NR][NW];
  __shared__ __attribute__((aligned(16))) double s_part2[2][NR][NW];     // TEAM == 1: double-buffered by trip parity
  __shared__ __attribute__((aligned(16))) double s_bc[NR + 1];
  __shared__ __attribute__((aligned(16))) double s_scr[NW * 8];
  __shared__ __attribute__((aligned(16))) unsigned s_flag[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t team = (p.variant & 2) ? blockIdx.x % p.nteams : blockIdx.x / TEAM;
  const uint32_t mem = (p.variant & 2) ? blockIdx.x / p.nteams : blockIdx.x % TEAM;
  const uint32_t c0 = mem * (FH_WG * MPP) + tid;
  // piece k of this lane is piece c0 + k * NT of the row -- if it lies inside the member's range and inside the row
  auto piece_ok = [&](int k) { return tid + (uint32_t)k * NT < (uint32_t)(FH_WG * MPP) && c0 + (uint32_t)k * NT < p.ld2; };

  // rows of this team: exactly k_fused_dense's assignment -- variant bit 32 (the default): dealt cyclically (t, t + nteams, ...), else a contiguous block
  const bool blocked = (p.variant & 32) == 0;
  const uint32_t row_base = blocked ? min(team * p.rows_per_team, p.mp) : team;
  const uint32_t row_step = blocked ? 1u : p.nteams;
  const uint32_t r_begin = 0u;
  const uint32_t r_end = blocked ? min(row_base + p.rows_per_team, p.mp) - row_base
                                 : (team < p.mp ? (p.mp - team + p.nteams - 1u) / p.nteams : 0u);
  const uint32_t r_last = r_end - 1u;
  auto grow = [&](uint32_t r) { return row_base + r * row_step; };
  uint32_t pc[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) pc[k] = piece_ok(k) ? c0 + k * NT : p.ld2 - 1u;      // clamped: loads stay unconditional and in bounds
  auto load_row = [&](PT (&buf)[PPT], uint32_t r) {
    const PT* src = reinterpret_cast<const PT*>(p.A) + (uint64_t)grow(r) * p.ldp;
#pragma unroll
    for (int k = 0; k < PPT; ++k) buf[k] = load_stream<1>(src + pc[k]);
  };
  constexpr int NB = F32 ? (PPT >= 4 ? 4 : 6)       // float32 storage: a piece's accumulators are twice as wide -- four row buffers next to 4 pieces x 2 right-hand sides
                         : (NT == 512 ? 4 : (PPT >= 7 ? (NR == 2 && PIPE >= 2 ? 5 : 4) : (PPT >= 5 ? 5 : 6)));      // row buffers: what fits next to the gradient slices (512 threads: 256 registers per lane)
  PT B[NB][PPT];

  // ---------------- n-side: the three x slices into LDS (each lane reads back only its own entries); ||x1 - x2||^2 by team 0
  double dx2 = 0.0;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const bool ok = piece_ok(k);
#pragma unroll
    for (int h = 0; h < XD; ++h) {
      const uint32_t c = (c0 + k * NT) * XD + h;                    // double-pair index into the n-side vectors (validity); pc[k] * XD + h: clamped (loads)
      d2 xin[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        xin[j] = reinterpret_cast<const d2*>(p.x[j])[pc[k] * XD + h];
#pragma unroll
        for (int e = 0; e < 2; ++e) if (!(ok && (2u * c + e) < p.n)) xin[j][e] = 0.0;     // lanes past the member's / the row's last column carry 0
      }
      d2 dif;
      dif.x = sub_nofma(xin[0].x, xin[1].x); dif.y = sub_nofma(xin[0].y, xin[1].y);
      if (NR == 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) s_x[((j * PPT + k) * XD + h) * NT + tid] = xin[j];
      } else {
        s_x[((0 * PPT + k) * XD + h) * NT + tid] = dif;
        s_x[((1 * PPT + k) * XD + h) * NT + tid] = xin[2];
      }
      if (team == 0) { dx2 = fma(dif.x, dif.x, dx2); dx2 = fma(dif.y, dif.y, dx2); }
    }
  }

  d2 ga[NR][PPT][XD];
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int k = 0; k < PPT; ++k)
#pragma unroll
      for (int h = 0; h < XD; ++h) ga[j][k][h] = (d2){0.0, 0.0};
  double fs = 0.0;
  bool dead = false;
  // LEAST SQUARES ONLY (the host takes the three passes for the logistic loss: its exp / log constants do not fit next to three
  // gradient slices -- with them in the loop hipcc spills to scratch, and every scratch reload drains the prefetched rows).  The row's
  // loss term is the square of the gradient factor just formed, summed in row order by the lane that forms it, as k_fused_dense does.
  constexpr bool lsq_inline = true;

  typedef unsigned ft_line __attribute__((ext_vector_type(16)));
  constexpr int SL = TEAM < 8 ? 8 : TEAM;
  constexpr int NL = SL / 8;
  constexpr int LG = NL < 2 ? NL : 2;
  constexpr int MG = TEAM < 8 ? TEAM : 8 * LG;
  // one right-hand side's slot lines of one row: as k_fused_dense's poll_line (scalar loads past the scalar cache, bounded)
  auto poll_line = [&](uint64_t line, bool live) -> double {
    double zs = 0.0;
    if (live && !dead) {
#pragma unroll
      for (int g = 0; g < NL / LG; ++g) {
        ft_line ln[LG];
        const double* lp = p.slots + line * SL + g * (8 * LG);
        unsigned cnt = 0u;
        for (;;) {
          if (LG == 1) asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(ln[0]) : "s"(lp) : "memory");
          else asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                            : "=&s"(ln[0]), "=&s"(ln[LG - 1]) : "s"(lp) : "memory");
          unsigned pending = 0u;
#pragma unroll
          for (int q = 0; q < MG; ++q)
            pending |= (unsigned)__builtin_amdgcn_readfirstlane((int)ln[q / 8][2 * (q % 8) + 1]) == FT_SENTINEL_HI ? 1u : 0u;
          if (pending == 0u) break;
          if (++cnt >= FT_SPIN_POLLS) {
            dead = true;
            if (lane == 0) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          if (!(p.variant & 4)) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < MG; ++q) {
          const double v = __hiloint2double(__builtin_amdgcn_readfirstlane((int)ln[q / 8][2 * (q % 8) + 1]),
                                            __builtin_amdgcn_readfirstlane((int)ln[q / 8][2 * (q % 8)]));
          zs = (g == 0 && q == 0) ? v : zs + v;
        }
      }
      if (dead) zs = 0.0;
    }
    return zs;
  };
  // wave 1, lane 0, after the barrier that follows the s_part writes: this member's three partials of row r
  auto post_row = [&](uint32_t r, bool live) {
    const bool sabotage = (p.variant & 64) && team == 0 && mem == TEAM - 1 && r == r_begin;
    if (lane == 0 && live && !sabotage) {
#pragma unroll
      for (int j = 0; j < NR; ++j)
        store_partial(p.slots + ((uint64_t)grow(r) * NR + j) * SL + mem, fs_sum_waves<NW>(s_part[j]));
    }
  };
  // three partial dot products of one row buffer (per right-hand side: the order of k_fused_dense's dot_row)
  // (piece-major, the x pieces of the next piece in flight while this one is multiplied: right-hand-side-major, hipcc reads every x piece
  // into ONE register quad and waits for each of the 3 x PPT LDS round trips in turn -- the dot products are then LDS-latency bound)
  auto dot_row = [&](const PT (&buf)[PPT], double (&d)[NR]) {
    double part[NR] = {};
    d2 xa[NR][XD], xb[NR][XD];
    auto fetch = [&](d2 (&xv)[NR][XD], int k) {
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int h = 0; h < XD; ++h) xv[j][h] = s_x[((j * PPT + k) * XD + h) * NT + tid];
    };
    auto mul = [&](const PT& a, const d2 (&xv)[NR][XD]) {
#pragma unroll
      for (int j = 0; j < NR; ++j) part[j] = piece_dot(a, xv[j], part[j]);      // (float64: fma(a.x, x.x, .), fma(a.y, x.y, .) -- the order of k_fused_dense)
    };
    fetch(xa, 0);
#pragma unroll
    for (int k = 0; k < PPT; k += 2) {
      if (k + 1 < PPT) fetch(xb, k + 1);
      mul(buf[k], xa);
      if (k + 2 < PPT) fetch(xa, k + 2);
      if (k + 1 < PPT) mul(buf[k + 1], xb);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) d[j] = wave_sum(part[j]);
  };
  auto update_row = [&](const PT (&buf)[PPT], const double (&rv)[NR]) {
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int k = 0; k < PPT; ++k) piece_axpy(buf[k], rv[j], ga[j][k]);
  };
  const auto* bq = (const __attribute__((address_space(4))) double*)(uintptr_t)p.b;

  if constexpr (TEAM == 1) {
    if (r_begin < r_end) {
      const uint32_t trips = ((r_end - r_begin + NB - 1u) / NB) * NB;
#pragma unroll
      for (int k = 0; k < NB - 1; ++k) load_row(B[k], min(r_begin + k, r_last));
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const uint32_t r = r_begin + t + q;
          const bool live = r < r_end;
          const uint32_t gr = grow(min(r, r_last));
          const double bi = bq[gr];
          load_row(B[(q + NB - 1) % NB], min(r + (NB - 1u), r_last));
          double d[NR];
          dot_row(B[q], d);
          const int par = (t + q) & 1u;
          if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j) s_part2[par][j][wave] = d[j];
          }
          ft_lds_barrier();
          double rv[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            const double zs = fs_sum_waves<NW>(s_part2[par][j]);
            rv[j] = live ? ((NR == 2 && j == 0) ? zs : loss_grad(zs, bi, LOSS_LSQ)) : 0.0;
            if (j == NR - 1 && tid == 0 && live) {
              store_partial(p.z + gr, zs);
              if (lsq_inline && gr < p.m) fs = add_nofma(fs, ft_sq(rv[j]));
            }
          }
          update_row(B[q], rv);
        }
      }
    }
  } else if (r_begin < r_end) {
    constexpr int D = PIPE;
    static_assert(NB >= D + 2, "need at least one prefetching buffer");
    const uint32_t trips = ((r_end - r_begin + NB - 1u) / NB) * NB;
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) load_row(B[k], min(r_begin + k, r_last));
    if (wave != 1) {
      // ---- the three polling waves: wave 0 -> right-hand side 0, wave 2 -> 1, wave 3 -> 2 (waves 4.. of a 512-thread workgroup: none)
      const int mine = wave == 0 ? 0 : ((int)wave <= NR ? (int)wave - 1 : -1);
#pragma unroll
      for (int q = 0; q < D; ++q) {
        double d0[NR];
        dot_row(B[q], d0);
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < NR; ++j) s_part[j][wave] = d0[j];
        }
        ft_lds_barrier();
        ft_lds_barrier();
      }
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const uint32_t r = r_begin + t + q;
          const bool live = r < r_end;
          const uint32_t gr = grow(min(r, r_last));
          const double bi = bq[gr];
          load_row(B[(q + NB - 1) % NB], min(r + (NB - 1u), r_last));
          double d[NR];
          dot_row(B[(q + D) % NB], d);
          if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j) s_part[j][wave] = d[j];
          }
          ft_lds_barrier();                                        // wave 1 posts row r + D from s_part
          const double zs = poll_line((uint64_t)gr * NR + (mine < 0 ? 0 : mine), live && mine >= 0);
          if (lane == 0 && mine >= 0) {
            const double rvm = live ? ((NR == 2 && mine == 0) ? zs : loss_grad(zs, bi, LOSS_LSQ)) : 0.0;
            s_bc[mine] = rvm;
            if (mine == NR - 1 && mem == 0 && live) {
              store_partial(p.z + gr, zs);
              if (lsq_inline && gr < p.m) fs = add_nofma(fs, ft_sq(rvm));
            }
          }
          ft_lds_barrier();
          double rv[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) rv[j] = s_bc[j];
          update_row(B[q], rv);
        }
      }
    } else {
      // ---- wave 1: posts the member's three partials, D rows ahead of the polls
#pragma unroll
      for (int q = 0; q < D; ++q) {
        double d0[NR];
        dot_row(B[q], d0);
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < NR; ++j) s_part[j][wave] = d0[j];
        }
        ft_lds_barrier();
        post_row(r_begin + q, r_begin + q < r_end);
        ft_lds_barrier();
      }
      for (uint32_t t = 0; t < trips; t += NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const uint32_t r = r_begin + t + q;
          load_row(B[(q + NB - 1) % NB], min(r + (NB - 1u), r_last));
          double d[NR];
          dot_row(B[(q + D) % NB], d);
          if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j) s_part[j][wave] = d[j];
          }
          ft_lds_barrier();
          post_row(r + D, r + D < r_end);
          ft_lds_barrier();
          double rv[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) rv[j] = s_bc[j];
          update_row(B[q], rv);
        }
      }
    }
  }

  // ---------------- publish the three slice partials, the loss partial and (team 0) ||x1 - x2||^2 ----------------------------
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int k = 0; k < PPT; ++k)
      if (piece_ok(k)) {
#pragma unroll
        for (int h = 0; h < XD; ++h)
          store_partial16(reinterpret_cast<d2*>(p.gpart) + ((uint64_t)team * NR + j) * p.nv2, (c0 + k * NT) * XD + h, ga[j][k][h]);
      }
  {
    double w[2] = {fs, dx2};
    fs_block_reduce<2, NW>(w, s_scr);
    if (tid == 0) { store_partial(p.red + (uint64_t)blockIdx.x * 8, w[0]); store_partial(p.red + (uint64_t)blockIdx.x * 8 + 1, w[1]); }
  }

  // ---------------- bounded grid barrier (one workgroup per CU: all co-resident) ------------------------------------------------
  (void)grid_barrier2(p.gbar, 1u, p.err, FT_SPIN_TICKS, s_flag);

  // ---------------- every workgroup finalises its share of the columns: team-ordered sums (k_fused_dense's split), ||grad1 - grad2||^2
  double dg2 = 0.0;
  const uint32_t share = (p.nv2 + gridDim.x - 1) / gridDim.x;
  const uint32_t slices = share < FH_WG ? min(FH_WG / max(share, 1u), p.nteams) : 1u;
  const uint32_t tps = (p.nteams + slices - 1) / slices;
  for (uint32_t t0 = 0; t0 < share; t0 += FH_WG) {
    const uint32_t col = slices > 1 ? tid % share : t0 + tid;
    const uint32_t slice = slices > 1 ? tid / share : 0u;
    const uint32_t c = blockIdx.x * share + col;
    const bool mine = tid < FH_WG && col < share && slice < slices && c < p.nv2;      // (a 512-thread workgroup finalises with its first 256)
    d2 g[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      g[j] = (d2){0.0, 0.0};
      if (mine) {
        const uint32_t s1 = min((slice + 1u) * tps, p.nteams);
#pragma unroll 8
        for (uint32_t s = slice * tps; s < s1; ++s) g[j] += load_partial16(reinterpret_cast<const d2*>(p.gpart), (s * NR + j) * p.nv2 + c);
      }
      if (slices > 1) {                                // uniform over the workgroup
        __syncthreads();
        if (mine) s_fin[slice * share + col] = g[j];
        __syncthreads();
        if (mine && slice == 0) {
          for (uint32_t q = 1; q < slices; ++q) g[j] += s_fin[q * share + col];
        }
      }
    }
    if (!mine || slice != 0) continue;
    if (NR == 3) {
#pragma unroll
      for (int j = 0; j < 3; ++j) reinterpret_cast<d2*>(p.g[j])[c] = g[j];
    } else {
      reinterpret_cast<d2*>(p.g[0])[c] = g[0];          // A^T A (x1 - x2)
      reinterpret_cast<d2*>(p.g[2])[c] = g[NR - 1];     // the gradient at x0
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (2u * c + e < p.n) { const double d = NR == 3 ? sub_nofma(g[0][e], g[1][e]) : g[0][e]; dg2 = fma(d, d, dg2); }
  }
  {
    double w[1] = {dg2};
    fs_block_reduce<1, NW>(w, s_scr);
    if (tid == 0) store_partial(p.red + (uint64_t)blockIdx.x * 8 + 2, w[0]);
  }
  if (!arrive_last2(p.gbar + GB_WORDS, s_flag)) return;
  double w[3] = {0, 0, 0};
  if (tid < FH_WG) {
    for (uint32_t i = tid; i < gridDim.x; i += FH_WG) {
#pragma unroll
      for (int k = 0; k < 3; ++k) w[k] += load_partial(p.red + (uint64_t)i * 8 + k);
    }
  }
  fs_block_reduce<3, NW>(w, s_scr);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) p.out[k] = 0.0;
    p.out[S_FSQ] = w[0]; p.out[S_DX2] = w[1]; p.out[S_DG2] = w[2];
    p.out[15] = __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.0 : 0.0;
    if (p.pack) { p.pack[0] = w[0]; p.pack[1] = p.out[15]; }
    __hip_atomic_store(p.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.err, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < GB_GROUPS + 2) {      // both blocks: group counters, top counter, release word
    __hip_atomic_store(p.gbar + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.gbar + GB_WORDS + tid * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
