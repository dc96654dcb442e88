// fh_device.h -- device-side building blocks shared by the FBS kernels (gfx950 / wave64 only).
//
// Conventions
//   * 256-thread workgroups = 4 wavefronts of 64 lanes.
//   * Elementwise solver arithmetic (forward step, prox, BB terms) is written with FMA contraction
//     OFF so each operation rounds exactly like the NumPy expression it replaces
//     (fasta/__init__.py:181, :242, :254); dot products and matvec accumulations use explicit fma().
//   * Cross-workgroup hand-offs (partials -> last-arriving workgroup) use write-through `sc1` atomic
//     stores/loads + an agent-scope ticket (CDNA4 guide, Guideline 16); see arrive_last below.  All
//     reductions are summed in index order, never arrival order, so results are bitwise repeatable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d2 __attribute__((ext_vector_type(2)));   // one 16-byte global access
typedef float f4 __attribute__((ext_vector_type(4)));    // one 16-byte piece of a float32-storage matrix: four columns

// ---- storage type of A ---------------------------------------------------------------------------------------------------
// A "piece" is one aligned 16-byte access to a row of A: two float64 columns (F32 = 0) or four float32 columns (F32 = 1, the
// opt-in storage mode: half the bytes per pass; every vector, every accumulation and every scalar stays float64).  The matching
// stretch of an n-side vector is XD2 = 1 or 2 double pairs.
template <int F32> struct PieceOf { typedef d2 type; };
template <> struct PieceOf<1> { typedef f4 type; };
template <int F32> __device__ __forceinline__ constexpr int xd2() { return F32 ? 2 : 1; }

__device__ __forceinline__ double piece_dot(d2 a, const d2 (&x)[1], double acc) {
  acc = fma(a.x, x[0].x, acc);
  return fma(a.y, x[0].y, acc);
}
__device__ __forceinline__ double piece_dot(f4 a, const d2 (&x)[2], double acc) {
  acc = fma((double)a.x, x[0].x, acc);
  acc = fma((double)a.y, x[0].y, acc);
  acc = fma((double)a.z, x[1].x, acc);
  return fma((double)a.w, x[1].y, acc);
}
__device__ __forceinline__ void piece_axpy(d2 a, double r, d2 (&g)[1]) {
  g[0].x = fma(a.x, r, g[0].x);
  g[0].y = fma(a.y, r, g[0].y);
}
__device__ __forceinline__ void piece_axpy(f4 a, double r, d2 (&g)[2]) {
  g[0].x = fma((double)a.x, r, g[0].x);
  g[0].y = fma((double)a.y, r, g[0].y);
  g[1].x = fma((double)a.z, r, g[1].x);
  g[1].y = fma((double)a.w, r, g[1].y);
}

#define FH_WG 256

enum { PX_PLAIN = -1, PX_IDENTITY = 0, PX_SHRINK = 1, PX_NONNEG = 2, PX_LINF = 3, PX_L1BALL = 4, PX_TVBALL = 5, PX_BOX = 6 };

struct ProxP {
  int kind;
  unsigned seq;        // (in the struct's padding) != 0: the launch's finaliser publishes this number behind the scalar block (publish_seq below)
  double thr;          // shrink: tau*mu (rounded on the host exactly like `t*self.mu`)
  double lo, hi;       // box
  const double* level; // LINF / L1BALL: device scalar holding the clipping level alpha
};

// Cross-lane reductions by DPP (VALU) instead of `__shfl_down` (ds_bpermute: an LDS-crossbar round trip of ~100 cycles per step,
// six dependent steps, twice per double): every lane ends up with the wave's result, in a fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);      // row_half_mirror
  v += dpp_f64<0x140>(v);      // row_mirror: every lane of a 16-lane row holds the row's sum
  return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// sign(x) as NumPy defines it for finite x (np.sign): -1, 0 or +1
__device__ __forceinline__ double sgn(double x) { return (double)((x > 0.0) - (x < 0.0)); }

// proxg(x, t) for the scalar-separable kinds.  fasta/proximal.py:58-67 (shrink), :28-31 (linf tail),
// :41 (l1 ball), examples/nn_least_squares.py:42 (non-negativity), examples/svm.py:71 (box).
template <int KIND>
__device__ __forceinline__ double prox_scalar(double x, const ProxP& px, double level) {
#pragma clang fp contract(off)
  if (KIND == PX_SHRINK) return sgn(x) * fmax(fabs(x) - px.thr, 0.0);
  if (KIND == PX_NONNEG) return fmax(x, 0.0);
  if (KIND == PX_BOX)    return fmin(fmax(x, px.lo), px.hi);
  // (a level <= 0 means "the zero vector", proximal.py:28-31; a NaN level -- a search that could not finish -- is PROPAGATED, so that it
  //  reaches every sum of the step and the host's check instead of passing for "level <= 0")
  if (KIND == PX_LINF)   return level > 0.0 ? fmin(fabs(x), level) * sgn(x) : (level != level ? level : 0.0);
  if (KIND == PX_L1BALL) return x - (level > 0.0 ? fmin(fabs(x), level) * sgn(x) : (level != level ? level : 0.0));
  return x;
}

// same, kind chosen at run time (kernels that evaluate the prox once per launch)
__device__ __forceinline__ double prox_scalar_rt(int kind, double x, const ProxP& px, double level) {
  switch (kind) {
    case PX_SHRINK: return prox_scalar<PX_SHRINK>(x, px, level);
    case PX_NONNEG: return prox_scalar<PX_NONNEG>(x, px, level);
    case PX_BOX:    return prox_scalar<PX_BOX>(x, px, level);
    case PX_LINF:   return prox_scalar<PX_LINF>(x, px, level);
    case PX_L1BALL: return prox_scalar<PX_L1BALL>(x, px, level);
    default:        return x;
  }
}

// forward (gradient) step x0 - tau*g0, two roundings as in fasta/__init__.py:181
__device__ __forceinline__ double fwd_point(double x0, double g0, double tau) {
#pragma clang fp contract(off)
  double s = tau * g0;
  return x0 - s;
}

// z + coef*(z - zprev), fasta/__init__.py:242-243 evaluation order
__device__ __forceinline__ double extrapolate(double v, double vprev, double coef) {
#pragma clang fp contract(off)
  double d = v - vprev;
  double s = coef * d;
  return v + s;
}

// Dg = g1 + (xhat - x0)/tau, fasta/__init__.py:254
__device__ __forceinline__ double bb_dgrad(double g1, double xhat, double x0, double tau) {
#pragma clang fp contract(off)
  double d = xhat - x0;
  double q = d / tau;
  return g1 + q;
}

// The same with the quotient by a launch-uniform tau formed from its reciprocal (rtau = 1/tau, correctly rounded, computed once
// per thread): q = d*rtau, one exact-remainder correction (Markstein).  The result is the correctly rounded quotient except in
// rare corner cases (tau's significand all ones, subnormal quotients), where it is off by one ulp -- used ONLY where Dg feeds the
// Barzilai-Borwein SUMS (<Dx,Dg>, ||Dg||^2), whose summation order already differs from NumPy's by far more than that; it
// replaces two ~25-instruction IEEE divisions per pixel in the stencil sweeps, which are otherwise ALU-co-limited.
__device__ __forceinline__ double bb_dgrad_rcp(double g1, double xhat, double x0, double tau, double rtau) {
#pragma clang fp contract(off)
  const double d = xhat - x0;
  double q = d * rtau;
  const double e = __builtin_fma(-q, tau, d);
  q = __builtin_fma(e, rtau, q);
  return g1 + q;
}

__device__ __forceinline__ double sub_nofma(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double add_nofma(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}

// scalar slots (mirror include/fasta_hip.h enum fh_scalar)
enum { S_FSQ = 0, S_DXG0 = 1, S_DX2 = 2, S_XH2 = 3, S_G02 = 4, S_GSUM = 5, S_GMAX = 6, S_RDOT = 7,
       S_DXDG = 8, S_DG2 = 9, S_FSQ_ADJ = 10, S_XH2_ADJ = 11, S_GSUM_ADJ = 12, S_GMAX_ADJ = 13, S_ALPHA = 14 };

// Reduce K running values over the workgroup; slot `maxslot` (or -1) uses max instead of +.
// Result valid in thread 0.  `scr` = 4*K doubles of LDS.
template <int K>
__device__ __forceinline__ void block_reduce(double (&v)[K], double* scr, int maxslot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    v[k] = (k == maxslot) ? wave_max(v[k]) : wave_sum(v[k]);
    if (lane == 0) scr[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k == maxslot) v[k] = fmax(fmax(scr[k], scr[K + k]), fmax(scr[2 * K + k], scr[3 * K + k]));
      else v[k] = ((scr[k] + scr[K + k]) + scr[2 * K + k]) + scr[3 * K + k];
    }
  }
  __syncthreads();
}

// ---- smooth term on the m-side: per-row value and derivative ----------------------------------------------
// LOSS_LSQ      f = .5*sum (z-b)^2 (the kernels return sum (z-b)^2), grad = z - b   (examples/sparse_least_squares.py:41-42)
// LOSS_LOGISTIC f = sum log(1+exp(z)) - (b==1)*z,  grad = -b / (1 + exp(b*z)),  b in {-1,+1}
//               (examples/sparse_logistic.py:47-48).  exp/log are the device libm: last-ulp differences from NumPy.
enum { LOSS_LSQ = 0, LOSS_LOGISTIC = 1 };

__device__ __forceinline__ double loss_grad(double z, double b, int kind) {
#pragma clang fp contract(off)
  if (kind == LOSS_LOGISTIC) return -b / (1.0 + exp(b * z));
  return z - b;
}
// contribution of one row to the scalar the host turns into f (sum of squares for LSQ, the loss itself otherwise)
__device__ __forceinline__ double loss_term(double z, double b, int kind) {
#pragma clang fp contract(off)
  if (kind == LOSS_LOGISTIC) return log(1.0 + exp(z)) - (b == 1.0 ? z : 0.0);
  const double r = z - b;
  return r * r;
}

// ---- cross-workgroup hand-off of partial results ("last workgroup to arrive finishes the reduction") ------
// Fence-free form of the CDNA4 guide (Guideline 16, visibility table row 1): EVERY byte that is handed off
// is stored write-through by an agent-scope relaxed atomic store (`global_store_dwordx2 sc1`), every storing
// wave drains `vmcnt`, the workgroup barriers, ONE lane takes a ticket with an agent-scope atomic add; the
// workgroup whose add came last is the consumer and reads every handed-off byte with agent-scope relaxed
// atomic loads (`sc1`, served past this CU's L1).  No `buffer_wbl2` / `buffer_inv`, so the kernels' bulk
// outputs are neither flushed nor waited for.  Partials are always combined in index order => repeatable.
__device__ __forceinline__ void store_partial(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_partial(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_partial2(d2* p, d2 v) {
  store_partial(reinterpret_cast<double*>(p), v.x);
  store_partial(reinterpret_cast<double*>(p) + 1, v.y);
}
// The same hand-off store as ONE 16-byte write-through instruction (`buffer_store_dwordx4 ... offen sc1`), for bulk partials (a slice of
// g1 per workgroup: tens of KiB).  As two 8-byte atomic stores each half is a fabric write of its own -- publishing a 32-KiB slice that
// way took ~10 us of every launch (round 5: csrc/fh_run.h's phase table; CDNA4 guide, Guideline 16 pitfall 7).  Through the raw-buffer
// builtin (aux 16 = sc1) hipcc counts the store in its vmcnt bookkeeping and may take the data straight from accumulator registers (an
// asm store with "v" operands forced the slices into VGPRs and cost the 16-piece shape 13 %).  `base` must be wave-uniform (it becomes
// the buffer descriptor: readfirstlane spares hipcc a waterfall loop), `idx` is this lane's double-pair index from it (< 2^27).
typedef unsigned fh_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_partial16(d2* base, uint32_t idx, d2 v) {
#ifdef FH_NARROW_PARTIALS      // A/B builds only: the two 8-byte atomic stores of rounds 1-4
  store_partial2(base + idx, v);
  return;
#endif
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xFFFFFFFFu));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)u, (short)0, 0x7FFFFFFF, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fh_u4, v), rs, idx * 16u, 0, 16);
}
__device__ __forceinline__ d2 load_partial2(const d2* p);
// ... and the matching 16-byte load past L1 (`buffer_load_dwordx4 ... offen sc1`): the consumer's side of the same hand-off (CDNA4 guide,
// Guideline 16: every load of handed-off bytes is an sc1 load to registers; 16-byte loads are in the measured set).  Half the load
// instructions of load_partial2 in the finalisers, which add up tens to hundreds of team partials per column.
__device__ __forceinline__ d2 load_partial16(const d2* base, uint32_t idx) {
#ifdef FH_NARROW_PARTIALS
  return load_partial2(base + idx);
#endif
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xFFFFFFFFu));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)u, (short)0, 0x7FFFFFFF, 0x00020000);
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16u, 0, 16));
}
__device__ __forceinline__ d2 load_partial2(const d2* p) {
  d2 v;
  v.x = load_partial(reinterpret_cast<const double*>(p));
  v.y = load_partial(reinterpret_cast<const double*>(p) + 1);
  return v;
}
// all waves may have stored partials: returns true in EVERY thread of the last-arriving workgroup
__device__ __forceinline__ bool arrive_last(unsigned* counter, unsigned total, volatile unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == total - 1u) ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}

// ---- two-level grid barrier (round 5) ------------------------------------------------------------------------------------------------------
// A generation barrier for co-resident workgroups.  G atomic adds on ONE address cost 3.6 us at G = 256 and 8.0 us at G = 512 (same-address
// atomics serialise at the memory side); with the arrivals spread over GB_GROUPS counters on separate 128-byte lines -- workgroup b arrives at
// counter b % GB_GROUPS, the last arriver of a group at the top counter, the last of those publishes the generation in a release word that
// everybody polls with plain write-through-coherent loads -- the same barrier costs 1.5 / 1.6 us (scripts/probes/bench_mem/gridbar.hip,
// profiles/r05_gridbar.txt; polling the release word with SCALAR glc loads instead: 17-45 us, hundreds of pollers on one line).
// All counters only grow: `gen` = 1, 2, 3, ... counts the barriers of the launch; the host zeroes the GB_WORDS words before the launch.
// Bounded: after FT_SPIN_TICKS-like `budget` ticks of the 100 MHz clock the caller's `err` word is set and false is returned (every
// workgroup then times out by itself, so the grid drains).
#define GB_GROUPS 32
#define GB_WORDS (GB_GROUPS * 32 + 64)
__device__ __forceinline__ bool grid_barrier2(unsigned* base, unsigned gen, unsigned* err, unsigned long long budget, volatile unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned G = gridDim.x, g = blockIdx.x % GB_GROUPS;
    const unsigned ngroups = G < GB_GROUPS ? G : GB_GROUPS;
    const unsigned gsize = (G - g + GB_GROUPS - 1u) / GB_GROUPS;                  // workgroups that arrive at my group's counter
    unsigned* top = base + GB_GROUPS * 32;
    unsigned* rel = top + 32;
    const unsigned old = __hip_atomic_fetch_add(base + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == gen * gsize) {                                                  // last of my group in this generation
      const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 + 1u == gen * ngroups) __hip_atomic_store(rel, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned ok = 1u;
    while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0u;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    *flag = ok;
  }
  __syncthreads();
  return *flag != 0u;
}

// "Who is last?" over the same two levels (a second block of GB_WORDS words): true in EVERY thread of the last-arriving workgroup of the grid.
// One-shot (the counters are zero on entry; the caller's last workgroup zeroes them again).  The flat form -- arrive_last on one counter --
// makes the last arriver queue behind up to G - 1 same-address atomics when the workgroups finish together, as they do after a grid barrier.
__device__ __forceinline__ bool arrive_last2(unsigned* base, volatile unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned G = gridDim.x, g = blockIdx.x % GB_GROUPS;
    const unsigned ngroups = G < GB_GROUPS ? G : GB_GROUPS;
    const unsigned gsize = (G - g + GB_GROUPS - 1u) / GB_GROUPS;
    unsigned last = 0u;
    const unsigned old = __hip_atomic_fetch_add(base + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == gsize) {
      const unsigned o2 = __hip_atomic_fetch_add(base + GB_GROUPS * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (o2 + 1u == ngroups) ? 1u : 0u;
    }
    *flag = last;
  }
  __syncthreads();
  return *flag != 0u;
}

// ---- "the scalars are there" without waiting for the end of the launch (round 6) -----------------------------------------------------------
// A single-GPU launch writes its FH_S_* block straight into host-mapped memory.  The host used to learn that it may read it from
// hipStreamSynchronize -- i.e. from the completion signal of the whole launch, 14.8 us after its last instruction for an empty kernel; a
// word the launch itself writes behind the block and the host spins on arrives after 10.0 (scripts/probes/bench_mem/launchgap.hip,
// profiles/r06_launchgap.txt): 4-5 us off EVERY iteration of every operator.  The finalising thread calls this after its last store to the
// block.  Visibility: every entry of the block is stored with a SYSTEM-scope store (write-through past L2: scal_store below) and the
// sequence number follows as a system-scope RELEASE store (hipcc: buffer_wbl2 sc0 sc1, s_waitcnt vmcnt(0), then the store) -- the first
// form of this, plain stores + vmcnt(0) + a relaxed store, let the host see the number BEFORE the block's entries (a stale FH_S_ALPHA in
// tests/test_gpu_faults.py): plain stores to the mapped block may sit in L2 until the launch's end-of-kernel release.
#define FH_SEQ_SLOT 28                       // in doubles from the start of the scalar block (FH_NSCALARS + 16 doubles are allocated)
__device__ __forceinline__ void scal_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void publish_seq(double* out, unsigned seq) {
  if (seq) __hip_atomic_store(reinterpret_cast<unsigned*>(out + FH_SEQ_SLOT), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// K doubles per workgroup, all held by thread 0 (the usual case after block_reduce)
template <int K>
__device__ __forceinline__ bool publish_partials(double* slot, const double (&v)[K], unsigned* counter, unsigned total,
                                                 volatile unsigned* flag) {
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) store_partial(slot + k, v[k]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == total - 1u) ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}

// 16-byte streaming load of A.  NT=1 marks it non-temporal (read-once stream; keeps x0/g0 in L2).
// (-DFH_PLAIN_LOADS: an A/B build whose streams all use the default cache policy -- make OUT=../libfasta_hip_plain.so BUILD=build_plain EXTRA=-DFH_PLAIN_LOADS)
#ifdef FH_PLAIN_LOADS
#define FH_NT_LOAD(p) (*(p))
#else
#define FH_NT_LOAD(p) __builtin_nontemporal_load(p)
#endif
template <int NT>
__device__ __forceinline__ d2 load_stream(const d2* p) {
  if (NT) return FH_NT_LOAD(p);
  return *p;
}
template <int NT>
__device__ __forceinline__ f4 load_stream(const f4* p) {
  if (NT) return FH_NT_LOAD(p);
  return *p;
}

// ---- synthetic generator (device twin of oracle/problems.py:synth_values) --------------------
__host__ __device__ __forceinline__ uint64_t fh_mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double fh_ihall(uint64_t key, uint64_t idx, double coef) {
  uint64_t h1 = fh_mix(key + 2ull * idx), h2 = fh_mix(key + 2ull * idx + 1ull);
  int64_t s = (int64_t)((h1 & 0xFFFF) + ((h1 >> 16) & 0xFFFF) + ((h1 >> 32) & 0xFFFF) + (h1 >> 48)
                        + (h2 & 0xFFFF) + ((h2 >> 16) & 0xFFFF) + ((h2 >> 32) & 0xFFFF) + (h2 >> 48));
  return (double)(s - 262140) * coef;
}
