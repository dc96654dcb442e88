// fasta_hip.hip -- host side of libfasta_hip.so (C ABI declared in include/fasta_hip.h).
// gfx950 only.  No PyTorch, no rocBLAS: every device operation is a kernel from fh_dense.h / fh_tv.h,
// plus RCCL (dlopen'ed on first use) for the row-sharded adjoint.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/fasta_hip.h"
#include "fh_dense.h"
#include "fh_tv.h"
#include "fh_prox.h"
#include "fh_fused.h"

// The one-pass kernel's variants: ONE table, fh_fused_instances.inc, drives (a) the explicit instantiations, compiled in four
// parallel groups by fh_fused_part.hip, (b) their `extern template` declarations here and (c) the host dispatch table kFusedTable
// below -- a shape that fused_shape() can produce but the table lacks is an error at launch (and a failure of
// tests/test_cabi_cpu.py, which walks every n), never a silent fall to some default instantiation.
// (-DFH_SINGLE_TU, used by the asm / resources / prof targets, instantiates everything in this unit instead.)
#ifndef FH_SINGLE_TU
#define FH_FUSED_DECLARE(P, PI, T, X, NB, F) extern template __global__ void k_fused_dense<P, 1, PI, T, X, NB, F>(const FusedP);
#define FUSED_INST_0 FH_FUSED_DECLARE
#define FUSED_INST_1 FH_FUSED_DECLARE
#define FUSED_INST_2 FH_FUSED_DECLARE
#define FUSED_INST_3 FH_FUSED_DECLARE
#include "fh_fused_instances.inc"
#undef FUSED_INST_0
#undef FUSED_INST_1
#undef FUSED_INST_2
#undef FUSED_INST_3
#endif
struct FusedEntry { int ppt, pipe, team, xlds, nbo, f32; void (*kernel)(const FusedP); };
#define FH_FUSED_ROW(P, PI, T, X, NB, F) {P, PI, T, X, NB, F, k_fused_dense<P, 1, PI, T, X, NB, F>},
#define FUSED_INST_0 FH_FUSED_ROW
#define FUSED_INST_1 FH_FUSED_ROW
#define FUSED_INST_2 FH_FUSED_ROW
#define FUSED_INST_3 FH_FUSED_ROW
static const FusedEntry kFusedTable[] = {
#include "fh_fused_instances.inc"
};
#undef FUSED_INST_0
#undef FUSED_INST_1
#undef FUSED_INST_2
#undef FUSED_INST_3

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code ? code : FH_E_ARG;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define FH_TRY(expr)          \
  do {                        \
    int r_ = (expr);          \
    if (r_ != 0) return r_;   \
  } while (0)

extern "C" const char* fh_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// RCCL through dlopen (no link-time dependency; the single-GPU path never touches it)
// ------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } fh_nccl_uid;
typedef void* fh_nccl_comm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(fh_nccl_uid*) = nullptr;
  int (*CommInitRank)(fh_nccl_comm*, int, fh_nccl_uid, int) = nullptr;
  int (*CommInitAll)(fh_nccl_comm*, int, const int*) = nullptr;
  int (*CommDestroy)(fh_nccl_comm) = nullptr;
  int (*CommCount)(const fh_nccl_comm, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, fh_nccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static const int kNcclFloat64 = 8;   // ncclDouble
static const int kNcclSum = 0;       // ncclSum

static int rccl_load() {
  if (g_rccl.lib) return 0;
  const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* nm : names) {
    g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) return fail(FH_E_RCCL, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                      \
  *(void**)(&g_rccl.field) = dlsym(g_rccl.lib, name);                         \
  if (!g_rccl.field) return fail(FH_E_RCCL, "librccl lacks symbol %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return 0;
}
#define NCCL_TRY(expr)                                                                        \
  do {                                                                                        \
    int r_ = (expr);                                                                          \
    if (r_ != 0) return fail(20000 + r_, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));  \
  } while (0)

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
enum { OP_NONE = 0, OP_DENSE = 1, OP_STENCIL = 2 };

struct fh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int op = OP_NONE;
  bool op_pending_stencil = false;
  uint64_t m = 0, n = 0;     // logical (local) rows / columns of A   (stencil: m = H*W, n = 2*H*W)
  uint64_t mp = 0, ld = 0;   // padded rows, device leading dimension in elements (dense)
  int f32 = 0;               // storage of A: 0 = float64, 1 = float32 (opt-in, fh_create_ex; vectors and arithmetic stay float64)
  uint64_t nv = 0, mv = 0;   // allocated n-side / m-side vector lengths (doubles)
  uint64_t H = 0, W = 0;
  double* A = nullptr;
  // n-side
  // iterate pool: X[xi] = x0, X[ti] = where the next x1 lands, X[bi] = best-quality iterate (may alias X[xi]):
  // the best iterate is tracked by index, never copied (a 1 GiB copy per improving iteration at 8192^2 TV)
  double* X[3] = {nullptr, nullptr, nullptr};
  int xi = 0, ti = 1, bi = 0;
  double* P[2] = {nullptr, nullptr};   // prox outputs: x_accel1 / x_accel0
  double* G[2] = {nullptr, nullptr};   // g0 / g1
  double* xhat = nullptr;
  double* T[4] = {nullptr, nullptr, nullptr, nullptr};
  int pc = 0, gc = 0, zc = 0;
  bool last_accel = false;
  // m-side
  double* b = nullptr;
  double* Z[2] = {nullptr, nullptr};
  double* zt = nullptr;
  double* ZX[2] = {nullptr, nullptr};   // stencil + FISTA: extrapolated z' (the residual source of the next g0)
  int zxc = 0;
  const double* zcur = nullptr;         // stencil: z at the current x0 (Z[zc], or ZX[zxc] after an accelerated step)
  // stencil + FISTA in ONE pass (k_fused_tv_accel): the iterate and its image are kept LAZILY as (P1, P0, c) and (Z1, Z0, c):
  // x0 = P1 + c*(P1 - P0), z(x0) = Z1 + c*(Z1 - Z0) are formed inside the next sweep and never written.  Buffers are taken
  // from the n-side pool {X[0], X[1], X[2], P[0], P[1]} and the m-side pool {Z[0], Z[1], ZX[0]} by index.
  bool lazy = false;
  int lq1 = 0, lq0 = 0, lqn = 0;        // last prox output, the one before, target of the next launch
  int lz1 = 0, lz0 = 0, lzn = 0;        // their images
  int lb1 = 0, lb0 = 0;                 // best-quality iterate = nq(lb1) + lbc*(nq(lb1) - nq(lb0)), by reference
  double lbc = 0.0, lc = 0.0, lc_pending = 0.0;   // coefficient of the best iterate / of x0 / decided by the launch awaiting fh_commit
  uint64_t commits = 0;                 // fh_commit calls since fh_init
  // z-free one-pass stencil kernels (k_tv_onepass, the default): they neither read nor write z, so after such a step is
  // committed the stored image of x0 is stale; the two-launch kernels recompute it on demand (one plain div pass)
  int tv_zfree = 1;                     // FH_TUNE_TV_ZFREE
  bool tvz_pending = false;             // the latest launch was z-free (its z_new exists only inside the kernel)
  bool zcur_stale = false;
  bool has_b = false;
  int loss_kind = LOSS_LSQ;
  // prox
  int prox_kind = FH_PROX_IDENTITY;
  double mu = 0.0, lo = 0.0, hi = 0.0;
  // workspace
  double* ws = nullptr;
  size_t ws_bytes = 0;
  unsigned* counters = nullptr;      // 4096 words, zeroed at creation; kernels leave them zero
  double* dscal = nullptr;           // FH_NSCALARS + 16 doubles on device
  double* hscal = nullptr;           // pinned, device-mapped host block: single-GPU launches write their scalars here
  double* hscal_dev = nullptr;       // device-side address of hscal
  bool scal_mirrored = false;        // row-sharded runs: the last launch already copied the scalar block into hscal
  // tuning
  int fwd_rows = 0;          // 0 = auto
  long long fwd_cap = 0;     // 0 = auto (4 workgroups per CU, grid-stride over row groups)
  int adj_slab = 0;          // 0 = auto
  int adj_cpt = 0;           // 0 = auto
  int ld_pad = 0;
  int nt_loads = 1;
  // stencil defaults measured on MI355X at 8192^2 (profiles/r01_tune_tv.txt): plain (not nt) accesses,
  // 8 rows in flight, 32 rows per workgroup for K-fwd and 128 for the read-only K-adj
  int tv_u = 0;              // 0 = auto: 8 for the kernels that stream z, 2 / 4 for the z-free one-pass sweeps (profiles/r02_tune_tv.txt)
  int tv_rows = 0;           // 0 = auto (32 fwd / 128 adj)
  int tv_nt = 0;
  int tv_pipe = 0;           // FH_TUNE_TV_PIPE: rotating trip buffers of the one-pass sweep (0 = auto, 1 = burst, 2, 3)
  int fused_variant = 2;     // team members 32 blocks apart (one XCD): best in profiles/r01b_tune_fused.txt
  int fused_min_rows = 16;   // use fewer teams when m is small: at least this many rows per team (scripts/fused_small_m.py)
  // one-pass kernel hand-off slots: two arrays alternate between launches, each launch re-arms the other one in passing;
  // the host fills both with the sentinel only when this signature (workspace, layout) changes or a launch timed out
  double* slotbuf = nullptr;     // dedicated allocation: the shared workspace `ws` is scribbled over by every other kernel
  size_t slotbuf_bytes = 0;
  uint64_t slots_sig = 0;
  int slots_parity = 0;
  // timing
  bool timing = false;
  hipEvent_t ev[FH_NKERNELS][2];
  bool ev_pending[FH_NKERNELS] = {false, false, false, false, false};
  double tot_ms[FH_NKERNELS] = {0, 0, 0, 0, 0};
  uint64_t launches[FH_NKERNELS] = {0, 0, 0, 0, 0};
  // comm
  fh_nccl_comm comm = nullptr;
  int nranks = 1, rank = 0;
  int ncu = 0;               // compute units of the device (fused one-pass kernel: one workgroup per CU)
  // ---- in-process row sharding (fh_create_ex with ndev > 1; SURVEY.md 8(b)/(e): one host thread, one context per device) ----
  // A context created over several devices is a SHELL: it owns one child context per entry of dev_ids (`shards`), each holding
  // a contiguous block of rows of A and the matching slice of b / z, while x, g, xhat are replicated.  Every entry point of the
  // C ABI runs on a shell as: local launches on every shard -> sum over the shards -> n-side epilogue on every shard -> ONE host
  // synchronisation, scalars from shard 0.  The sum is one grouped ncclAllReduce per shard (communicators from
  // ncclCommInitAll) when the device ids differ; when they REPEAT (several shards on one GPU: what a one-GPU box can run) all
  // shards share one stream and k_sum_shards adds their buffers in shard order.
  std::vector<fh_ctx*> shards;       // non-empty: this context is a shell
  std::vector<uint64_t> shard_row0;  // first row of every shard, plus the total (size shards + 1)
  fh_ctx* owner = nullptr;           // set in a shard
  bool emulated = false;             // shell / shard: the device ids repeat (one device, one stream, k_sum_shards)
  bool owns_stream = true;           // false in shards 1.. of an emulated group (they run on shard 0's stream)
};

#define FH_MAX_SHARDS 64
static inline int nshards(fh_ctx* c) { return c->shards.empty() ? 1 : (int)c->shards.size(); }
static inline fh_ctx* shard_of(fh_ctx* c, int k) { return c->shards.empty() ? c : c->shards[k]; }
// a context whose launches leave the sums over rows to an exchange step: a rank of a multi-process run, or a shard of a shell
static inline bool row_sharded(const fh_ctx* c) { return c->comm != nullptr || c->owner != nullptr; }

static const int kCounterWords = 8192;
enum { CNT_FWD = 0, CNT_ADJ_FIN = 1, CNT_AUX = 2, CNT_FUSED_BAR = 4, CNT_FUSED_ERR = 8, CNT_ADJ_CC = 16 };

static inline uint64_t round_up(uint64_t v, uint64_t q) { return (v + q - 1) / q * q; }

// device scratch that is released on every exit path (the HIP_TRY macros return early)
struct DevBuf {
  double* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

static int use_device(fh_ctx* c) {
  HIP_TRY(hipSetDevice(c->device));
  return 0;
}

static void free_operator(fh_ctx* c) {
  for (fh_ctx* s : c->shards) { (void)hipSetDevice(s->device); free_operator(s); }
  auto fr = [](double*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  fr(c->A);
  for (int i = 0; i < 2; ++i) { fr(c->P[i]); fr(c->G[i]); fr(c->Z[i]); }
  for (int i = 0; i < 3; ++i) fr(c->X[i]);
  fr(c->xhat); fr(c->b); fr(c->zt); fr(c->ZX[0]); fr(c->ZX[1]);
  for (int i = 0; i < 4; ++i) fr(c->T[i]);
  fr(c->ws); c->ws_bytes = 0;
  fr(c->slotbuf); c->slotbuf_bytes = 0; c->slots_sig = 0;
  c->op = OP_NONE; c->has_b = false;
}

static int alloc_zero(fh_ctx* c, double** p, uint64_t elems) {
  HIP_TRY(hipMalloc((void**)p, elems * sizeof(double)));
  HIP_TRY(hipMemsetAsync(*p, 0, elems * sizeof(double), c->stream));
  return 0;
}

static int alloc_vectors(fh_ctx* c) {
  // +16 slack doubles on the n-side so sharded runs can append scalars to the all-reduce buffer
  for (int i = 0; i < 2; ++i) {
    FH_TRY(alloc_zero(c, &c->P[i], c->nv + 16));
    FH_TRY(alloc_zero(c, &c->G[i], c->nv + 16));
    FH_TRY(alloc_zero(c, &c->Z[i], c->mv + 16));
  }
  FH_TRY(alloc_zero(c, &c->xhat, c->nv + 16));
  for (int i = 0; i < 3; ++i) FH_TRY(alloc_zero(c, &c->X[i], c->nv + 16));
  for (int i = 0; i < 4; ++i) FH_TRY(alloc_zero(c, &c->T[i], c->nv + 16));
  FH_TRY(alloc_zero(c, &c->b, c->mv + 16));
  FH_TRY(alloc_zero(c, &c->zt, c->mv + 16));
  if (c->op_pending_stencil) { FH_TRY(alloc_zero(c, &c->ZX[0], c->mv + 16)); FH_TRY(alloc_zero(c, &c->ZX[1], c->mv + 16)); }
  c->pc = c->gc = c->zc = c->zxc = 0;
  c->xi = 0; c->ti = 1; c->bi = 0;
  c->zcur = nullptr;
  return 0;
}

static int ensure_ws(fh_ctx* c, size_t bytes) {
  if (bytes <= c->ws_bytes) return 0;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->ws) { HIP_TRY(hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
  bytes = round_up(bytes, 1 << 20);
  HIP_TRY(hipMalloc((void**)&c->ws, bytes));
  c->ws_bytes = bytes;
  return 0;
}

// ---- timing helpers ------------------------------------------------------------------------------
static inline void t_begin(fh_ctx* c, int k) {
  if (c->timing) { (void)hipEventRecord(c->ev[k][0], c->stream); }
}
static inline void t_end(fh_ctx* c, int k) {
  if (c->timing) { (void)hipEventRecord(c->ev[k][1], c->stream); c->ev_pending[k] = true; }
}
static int finish(fh_ctx* c) {   // synchronise the stream and harvest pending event pairs
  if (!c->shards.empty()) {        // shell: all shards (an emulated group shares one stream; its first shard's sync covers the rest)
    for (fh_ctx* s : c->shards) { HIP_TRY(hipSetDevice(s->device)); FH_TRY(finish(s)); }
    return 0;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->timing) {
    for (int k = 0; k < FH_NKERNELS; ++k) {
      if (!c->ev_pending[k]) continue;
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, c->ev[k][0], c->ev[k][1]));
      c->tot_ms[k] += ms;
      c->launches[k] += 1;
      c->ev_pending[k] = false;
    }
  }
  return 0;
}

// Where kernels write the FH_S_* block: straight into the mapped host block on one GPU (no D2H copy, the
// stream sync alone publishes it); device memory when row-sharded, because RCCL reduces scalars in place.
static inline double* scalar_out(fh_ctx* c) { return row_sharded(c) ? c->dscal : c->hscal_dev; }

// row-sharded runs: the block lives in device memory (RCCL reduces into it); a 16-lane kernel forwards it to the mapped
// host block -- a hipMemcpyAsync D2H of 128 bytes costs ~10 us more per iteration than this launch
__global__ void k_forward_scalars(const double* src, double* dst) {
  if (threadIdx.x < FH_NSCALARS) dst[threadIdx.x] = src[threadIdx.x];
}

static int fetch_scalars(fh_ctx* c, double* scalars) {
  for (int k = 0; k < nshards(c); ++k) {
    fh_ctx* s = shard_of(c, k);
    const bool mirrored = s->scal_mirrored;
    s->scal_mirrored = false;
    if (row_sharded(s) && !mirrored) {
      HIP_TRY(hipSetDevice(s->device));
      k_forward_scalars<<<dim3(1), dim3(64), 0, s->stream>>>(s->dscal, s->hscal_dev);
      HIP_TRY(hipGetLastError());
    }
  }
  FH_TRY(finish(c));               // ONE host synchronisation per call (per device of a shell)
  // every shard holds the same block: each entry is either a sum over all shards or computed from replicated vectors
  if (scalars) memcpy(scalars, shard_of(c, 0)->hscal, FH_NSCALARS * sizeof(double));
  return 0;
}

// ---- sums over the row blocks ------------------------------------------------------------------------------------------
// out[i] = ((v0[i] + v1[i]) + v2[i]) + ... written back to every shard's buffer: the in-library, fixed-order replacement for the
// all-reduce when several shards live on ONE device (device ids repeat; all shards share a stream, so plain ordering suffices)
struct SumShardsP { double* v[FH_MAX_SHARDS]; int n; };
__global__ __launch_bounds__(FH_WG) void k_sum_shards(const SumShardsP p, uint64_t count) {
  for (uint64_t i = (uint64_t)blockIdx.x * FH_WG + threadIdx.x; i < count; i += (uint64_t)gridDim.x * FH_WG) {
    double acc = p.v[0][i];
    for (int k = 1; k < p.n; ++k) acc += p.v[k][i];
    for (int k = 0; k < p.n; ++k) p.v[k][i] = acc;
  }
}

// Sum `count` doubles at sel(shard) -- and, in the same exchange, `count2` doubles at sel2(shard) -- over all row blocks, in place,
// on every shard:
//   plain context with a communicator (one process per GPU) -> ncclAllReduce on its stream;
//   shell over distinct devices -> one grouped ncclAllReduce per shard (ncclCommInitAll communicators, one host thread);
//   shell over a repeated device -> k_sum_shards;      plain context without a communicator -> nothing to do.
template <typename Sel, typename Sel2>
static int sum_over_shards(fh_ctx* c, Sel sel, size_t count, Sel2 sel2, size_t count2) {
  if (c->shards.empty()) {
    if (!c->comm) return 0;
    t_begin(c, FH_K_COMM);
    if (count2) NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY(g_rccl.AllReduce(sel(c), sel(c), count, kNcclFloat64, kNcclSum, c->comm, c->stream));
    if (count2) {
      NCCL_TRY(g_rccl.AllReduce(sel2(c), sel2(c), count2, kNcclFloat64, kNcclSum, c->comm, c->stream));
      NCCL_TRY(g_rccl.GroupEnd());
    }
    t_end(c, FH_K_COMM);
    return 0;
  }
  if (c->emulated) {
    fh_ctx* s0 = c->shards[0];
    HIP_TRY(hipSetDevice(s0->device));
    t_begin(s0, FH_K_COMM);
    for (int pass = 0; pass < (count2 ? 2 : 1); ++pass) {
      SumShardsP sp;
      sp.n = (int)c->shards.size();
      for (int k = 0; k < sp.n; ++k) sp.v[k] = pass ? sel2(c->shards[k]) : sel(c->shards[k]);
      const uint64_t cnt = pass ? count2 : count;
      const unsigned grid = (unsigned)std::min<uint64_t>((cnt + FH_WG - 1) / FH_WG, 1024);
      k_sum_shards<<<dim3(grid), dim3(FH_WG), 0, s0->stream>>>(sp, cnt);
    }
    t_end(s0, FH_K_COMM);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  for (fh_ctx* s : c->shards) { HIP_TRY(hipSetDevice(s->device)); t_begin(s, FH_K_COMM); }
  NCCL_TRY(g_rccl.GroupStart());
  for (fh_ctx* s : c->shards) {
    NCCL_TRY(g_rccl.AllReduce(sel(s), sel(s), count, kNcclFloat64, kNcclSum, s->comm, s->stream));
    if (count2) NCCL_TRY(g_rccl.AllReduce(sel2(s), sel2(s), count2, kNcclFloat64, kNcclSum, s->comm, s->stream));
  }
  NCCL_TRY(g_rccl.GroupEnd());
  for (fh_ctx* s : c->shards) { HIP_TRY(hipSetDevice(s->device)); t_end(s, FH_K_COMM); }
  return 0;
}
template <typename Sel>
static int sum_over_shards(fh_ctx* c, Sel sel, size_t count) {
  return sum_over_shards(c, sel, count, [](fh_ctx*) { return (double*)nullptr; }, 0);
}

// ------------------------------------------------------------------------------------------------
// library / context API
// ------------------------------------------------------------------------------------------------
extern "C" int fh_device_count(int* count) {
  if (!count) return fail(FH_E_ARG, "fh_device_count: null pointer");
  HIP_TRY(hipGetDeviceCount(count));
  return 0;
}

static int create_body(fh_ctx* c, int device, hipStream_t shared_stream = nullptr) {
  c->device = device;
  HIP_TRY(hipSetDevice(device));
  (void)hipSetDeviceFlags(hipDeviceScheduleSpin);   // spin on stream syncs: the host waits ~2x per iteration
  (void)hipGetLastError();
  if (shared_stream) { c->stream = shared_stream; c->owns_stream = false; }     // shards 1.. of a group on one device
  else HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  { hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device)); c->ncu = prop.multiProcessorCount; }
  HIP_TRY(hipMalloc((void**)&c->counters, kCounterWords * sizeof(unsigned)));
  HIP_TRY(hipMemsetAsync(c->counters, 0, kCounterWords * sizeof(unsigned), c->stream));
  HIP_TRY(hipMalloc((void**)&c->dscal, (FH_NSCALARS + 16) * sizeof(double)));
  HIP_TRY(hipMemsetAsync(c->dscal, 0, (FH_NSCALARS + 16) * sizeof(double), c->stream));
  HIP_TRY(hipHostMalloc((void**)&c->hscal, (FH_NSCALARS + 16) * sizeof(double), hipHostMallocMapped));
  memset(c->hscal, 0, (FH_NSCALARS + 16) * sizeof(double));
  HIP_TRY(hipHostGetDevicePointer((void**)&c->hscal_dev, c->hscal, 0));
  for (int k = 0; k < FH_NKERNELS; ++k) {
    HIP_TRY(hipEventCreate(&c->ev[k][0]));
    HIP_TRY(hipEventCreate(&c->ev[k][1]));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int fh_destroy(fh_ctx* c);

// destroys a partially built context while keeping the error text of what failed
static int create_failed(fh_ctx* c, int rc) {
  char keep[sizeof(g_err)];
  memcpy(keep, g_err, sizeof(keep));
  (void)fh_destroy(c);
  memcpy(g_err, keep, sizeof(keep));
  return rc;
}

static int create_one(int device, hipStream_t shared_stream, fh_ctx** out) {
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(FH_E_ARG, "fh_create: device %d out of range (have %d)", device, ndev);
  fh_ctx* c = new fh_ctx();
  for (int k = 0; k < FH_NKERNELS; ++k) c->ev[k][0] = c->ev[k][1] = nullptr;
  const int rc = create_body(c, device, shared_stream);
  if (rc != 0) return create_failed(c, rc);          // release whatever was created (the error text is already set)
  *out = c;
  return 0;
}

extern "C" int fh_create(int device, fh_ctx** out) {
  if (!out) return fail(FH_E_ARG, "fh_create: null out pointer");
  return create_one(device, nullptr, out);
}

// SURVEY.md 8(b) form of the constructor: device list + storage type of A.
//   ndev == 1: a plain context (as fh_create; one process per GPU attaches it to a communicator with fh_comm_init).
//   ndev  > 1: IN-PROCESS row sharding -- a shell over one shard per entry of dev_ids.  Distinct ids: one GPU each, RCCL
//              communicators from ncclCommInitAll, grouped all-reduce.  A repeated id (all entries equal): every shard on that one
//              GPU, one stream, sums by k_sum_shards in shard order -- the form a one-GPU box can run and test.
extern "C" int fh_create_ex(int ndev, const int* dev_ids, int dtype, fh_ctx** out) {
  if (!out || !dev_ids) return fail(FH_E_ARG, "fh_create_ex: null argument");
  if (ndev < 1 || ndev > FH_MAX_SHARDS) return fail(FH_E_ARG, "fh_create_ex: ndev must be in [1,%d] (got %d)", FH_MAX_SHARDS, ndev);
  if (dtype != FH_DTYPE_F64 && dtype != FH_DTYPE_F32_STORAGE) return fail(FH_E_ARG, "fh_create_ex: unknown dtype %d", dtype);
  if (ndev == 1) {
    FH_TRY(fh_create(dev_ids[0], out));
    (*out)->f32 = dtype == FH_DTYPE_F32_STORAGE ? 1 : 0;
    return 0;
  }
  bool distinct = true, equal = true;
  for (int i = 0; i < ndev; ++i) {
    if (dev_ids[i] != dev_ids[0]) equal = false;
    for (int j = 0; j < i; ++j) if (dev_ids[i] == dev_ids[j]) distinct = false;
  }
  if (!distinct && !equal)
    return fail(FH_E_ARG, "fh_create_ex: device ids must be all different (one GPU per shard, RCCL) or all equal (every shard on one GPU)");
  fh_ctx* shell = new fh_ctx();
  for (int k = 0; k < FH_NKERNELS; ++k) shell->ev[k][0] = shell->ev[k][1] = nullptr;
  shell->device = dev_ids[0];
  shell->emulated = equal;
  shell->f32 = dtype == FH_DTYPE_F32_STORAGE ? 1 : 0;
  for (int i = 0; i < ndev; ++i) {
    fh_ctx* s = nullptr;
    const int rc = create_one(dev_ids[i], (equal && i > 0) ? shell->shards[0]->stream : nullptr, &s);
    if (rc != 0) return create_failed(shell, rc);
    s->owner = shell; s->emulated = equal; s->f32 = shell->f32;
    s->nranks = ndev; s->rank = i;
    shell->shards.push_back(s);
  }
  shell->ncu = shell->shards[0]->ncu;
  if (distinct) {
    int rc = rccl_load();
    if (rc == 0) {
      std::vector<fh_nccl_comm> comms((size_t)ndev, nullptr);
      const int nr = g_rccl.CommInitAll(comms.data(), ndev, dev_ids);
      if (nr != 0) rc = fail(20000 + nr, "ncclCommInitAll over %d devices failed: %s", ndev, g_rccl.GetErrorString(nr));
      else for (int i = 0; i < ndev; ++i) shell->shards[i]->comm = comms[(size_t)i];
    }
    if (rc != 0) return create_failed(shell, rc);
  }
  *out = shell;
  return 0;
}

extern "C" int fh_comm_destroy(fh_ctx* c);

extern "C" int fh_destroy(fh_ctx* c) {
  if (!c) return 0;
  if (!c->shards.empty()) {                         // shell: the shards in reverse order (shard 0 owns an emulated group's stream)
    for (fh_ctx* s : c->shards) { (void)hipSetDevice(s->device); if (s->stream) (void)hipStreamSynchronize(s->stream); }
    for (size_t k = c->shards.size(); k-- > 0;) { c->shards[k]->owner = nullptr; (void)fh_destroy(c->shards[k]); }
    c->shards.clear();
    delete c;
    return 0;
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->owner = nullptr;
  (void)fh_comm_destroy(c);
  free_operator(c);
  if (c->counters) (void)hipFree(c->counters);
  if (c->dscal) (void)hipFree(c->dscal);
  if (c->hscal) (void)hipHostFree(c->hscal);
  for (int k = 0; k < FH_NKERNELS; ++k) { if (c->ev[k][0]) (void)hipEventDestroy(c->ev[k][0]); if (c->ev[k][1]) (void)hipEventDestroy(c->ev[k][1]); }
  if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int fh_sync(fh_ctx* c) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (c->shards.empty()) FH_TRY(use_device(c));
  return finish(c);
}

static int set_tuning_one(fh_ctx* c, int key, long long value);
extern "C" int fh_set_tuning(fh_ctx* c, int key, long long value) {
  if (!c) return fail(FH_E_ARG, "null context");
  for (fh_ctx* s : c->shards) FH_TRY(set_tuning_one(s, key, value));      // a shell forwards to every shard (and keeps a copy)
  return set_tuning_one(c, key, value);
}

static int set_tuning_one(fh_ctx* c, int key, long long value) {
  switch (key) {
    case FH_TUNE_FWD_ROWS:
      if (value != 0 && value != 4 && value != 8 && value != 16) return fail(FH_E_ARG, "FWD_ROWS must be 0 (auto), 4, 8 or 16");
      c->fwd_rows = (int)value; return 0;
    case FH_TUNE_FWD_GRID_CAP:
      if (value < 0) return fail(FH_E_ARG, "FWD_GRID_CAP must be >= 0");
      c->fwd_cap = value; return 0;
    case FH_TUNE_ADJ_SLAB_ROWS:
      if (value < 0 || value > ADJ_MAX_SLAB || value % 8) return fail(FH_E_ARG, "ADJ_SLAB_ROWS must be a multiple of 8 in [0,%d]", ADJ_MAX_SLAB);
      c->adj_slab = (int)value; return 0;
    case FH_TUNE_ADJ_CPT:
      if (value != 0 && value != 1 && value != 2 && value != 4) return fail(FH_E_ARG, "ADJ_CPT must be 0 (auto), 1, 2 or 4");
      c->adj_cpt = (int)value; return 0;
    case FH_TUNE_LD_PAD:
      if (value < 0 || value % 32) return fail(FH_E_ARG, "LD_PAD must be a non-negative multiple of 32");
      if (c->op != OP_NONE) return fail(FH_E_STATE, "LD_PAD must be set before the matrix");
      c->ld_pad = (int)value; return 0;
    case FH_TUNE_NT_LOADS:
      c->nt_loads = value ? 1 : 0; return 0;
    case FH_TUNE_TV_U:
      if (value != 0 && value != 2 && value != 4 && value != 8) return fail(FH_E_ARG, "TV_U must be 0 (auto), 2, 4 or 8");
      c->tv_u = (int)value; return 0;
    case FH_TUNE_TV_ROWS:
      if (value < 0 || value > 4096) return fail(FH_E_ARG, "TV_ROWS must be in [0,4096] (0 = auto)");
      c->tv_rows = (int)value; return 0;
    case FH_TUNE_TV_NT:
      if (value < 0 || value > 3) return fail(FH_E_ARG, "TV_NT must be 0 (default), 1 (non-temporal loads and stores), 2 (non-temporal stores) or 3 (plain)");
      c->tv_nt = (int)value; return 0;
    case FH_TUNE_TV_PIPE:
      if (value < 0 || value > 3) return fail(FH_E_ARG, "TV_PIPE must be 0 (auto), 1 (load a trip, consume it) or 3 (three rotating trip buffers; 2 is taken as 3)");
      c->tv_pipe = (int)value; return 0;
    case FH_TUNE_TV_ZFREE:
      // while the iterate is kept lazily (one-pass FISTA on the stencil), z-free steps rotate their image buffers without ever
      // writing them: the z-streaming kernel would read stale images after a switch
      if (c->lazy && (value ? 1 : 0) != c->tv_zfree)
        return fail(FH_E_STATE, "TV_ZFREE cannot change while a one-pass accelerated stencil solve is in flight (call fh_init / fh_set_vector(X0) first)");
      c->tv_zfree = value ? 1 : 0; return 0;
    case FH_TUNE_FUSED_VARIANT:
      c->fused_variant = (int)(value & 0xFFFF);      // bits: see FusedP.variant (csrc/fh_fused.h) and fused_shape() below (8, 16: A/B shapes)
      if (value >> 16) c->fused_min_rows = (int)(value >> 16) == 0xFFFF ? 0 : (int)(value >> 16);   // high half: rows-per-team floor (0xFFFF = none)
      return 0;
    default: return fail(FH_E_ARG, "unknown tuning key %d", key);
  }
}

// ------------------------------------------------------------------------------------------------
// operator set-up
// ------------------------------------------------------------------------------------------------
static int setup_dense(fh_ctx* c, uint64_t m, uint64_t n) {
  if (m == 0 || n == 0) return fail(FH_E_ARG, "matrix must be non-empty (got %llu x %llu)", (unsigned long long)m, (unsigned long long)n);
  if (m >= (1ull << 31) || n >= (1ull << 31)) return fail(FH_E_ARG, "matrix dimension exceeds 2^31-1");
  FH_TRY(use_device(c));
  HIP_TRY(hipStreamSynchronize(c->stream));
  free_operator(c);
  c->m = m; c->n = n;
  c->mp = round_up(m, 16);
  c->ld = round_up(n, c->f32 ? 32 : 16) + (uint64_t)c->ld_pad;      // rows stay 128-byte aligned in either storage
  c->nv = c->ld; c->mv = c->mp;
  HIP_TRY(hipMalloc((void**)&c->A, c->mp * c->ld * (c->f32 ? sizeof(float) : sizeof(double))));
  FH_TRY(alloc_vectors(c));
  c->op = OP_DENSE;
  return 0;
}

// Row blocks of a shell: contiguous, the first (m mod p) shards hold ceil(m/p) rows and the others floor(m/p) -- SURVEY.md 8(e)'s
// ceil(m/p) blocks whenever p divides m, and never an empty shard otherwise.
static int shell_partition(fh_ctx* c, uint64_t m, uint64_t n) {
  const uint64_t p = c->shards.size();
  if (m < p) return fail(FH_E_ARG, "a matrix of %llu rows cannot be split over %llu shards", (unsigned long long)m, (unsigned long long)p);
  c->shard_row0.assign(p + 1, 0);
  for (uint64_t k = 0; k < p; ++k) c->shard_row0[k + 1] = c->shard_row0[k] + m / p + (k < m % p ? 1 : 0);
  c->m = m; c->n = n; c->op = OP_NONE; c->has_b = false;
  return 0;
}
static void shell_adopt(fh_ctx* c) {        // after every shard holds its block
  c->op = OP_DENSE;
  c->mp = c->m; c->ld = c->shards[0]->ld; c->nv = c->shards[0]->nv; c->mv = c->m;
}
static inline uint64_t shard_rows(fh_ctx* c, int k) { return c->shard_row0[(size_t)k + 1] - c->shard_row0[(size_t)k]; }

extern "C" int fh_set_matrix(fh_ctx* c, const double* A, uint64_t m, uint64_t n, uint64_t ld_host) {
  if (!c || !A) return fail(FH_E_ARG, "fh_set_matrix: null argument");
  if (ld_host < n) return fail(FH_E_ARG, "fh_set_matrix: ld_host %llu < n %llu", (unsigned long long)ld_host, (unsigned long long)n);
  if (!c->shards.empty()) {          // shell: each shard copies its own row block H2D
    FH_TRY(shell_partition(c, m, n));
    for (int k = 0; k < nshards(c); ++k) FH_TRY(fh_set_matrix(c->shards[k], A + c->shard_row0[k] * ld_host, shard_rows(c, k), n, ld_host));
    shell_adopt(c);
    return 0;
  }
  FH_TRY(setup_dense(c, m, n));
  HIP_TRY(hipMemsetAsync(c->A, 0, c->mp * c->ld * (c->f32 ? sizeof(float) : sizeof(double)), c->stream));
  if (!c->f32) {
    HIP_TRY(hipMemcpy2DAsync(c->A, c->ld * sizeof(double), A, ld_host * sizeof(double), n * sizeof(double), m,
                             hipMemcpyHostToDevice, c->stream));
    return finish(c);
  }
  // float32 storage: float64 row blocks go through a staging buffer and are rounded on the device (round to nearest even)
  const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(m, ((uint64_t)64 << 20) / (n * sizeof(double))));
  DevBuf staging;
  HIP_TRY(hipMalloc((void**)&staging.p, chunk * n * sizeof(double)));
  double* stage = staging.p;
  for (uint64_t r0 = 0; r0 < m; r0 += chunk) {
    const uint64_t rows = std::min<uint64_t>(chunk, m - r0);
    HIP_TRY(hipMemcpy2DAsync(stage, n * sizeof(double), A + r0 * ld_host, ld_host * sizeof(double), n * sizeof(double), rows,
                             hipMemcpyHostToDevice, c->stream));
    k_rows_to_f32<<<dim3(2048), dim3(FH_WG), 0, c->stream>>>(stage, n, reinterpret_cast<float*>(c->A) + r0 * c->ld, c->ld, (uint32_t)rows, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));      // the host block may be pageable: finish before the next chunk reuses `stage`
  }
  return finish(c);
}

// float32 host matrix straight into a float32-storage context (no float64 detour on either side)
extern "C" int fh_set_matrix_f32(fh_ctx* c, const float* A, uint64_t m, uint64_t n, uint64_t ld_host) {
  if (!c || !A) return fail(FH_E_ARG, "fh_set_matrix_f32: null argument");
  if (!c->f32) return fail(FH_E_STATE, "fh_set_matrix_f32 needs a float32-storage context (fh_create_ex with FH_DTYPE_F32_STORAGE)");
  if (ld_host < n) return fail(FH_E_ARG, "fh_set_matrix_f32: ld_host %llu < n %llu", (unsigned long long)ld_host, (unsigned long long)n);
  if (!c->shards.empty()) {
    FH_TRY(shell_partition(c, m, n));
    for (int k = 0; k < nshards(c); ++k) FH_TRY(fh_set_matrix_f32(c->shards[k], A + c->shard_row0[k] * ld_host, shard_rows(c, k), n, ld_host));
    shell_adopt(c);
    return 0;
  }
  FH_TRY(setup_dense(c, m, n));
  HIP_TRY(hipMemsetAsync(c->A, 0, c->mp * c->ld * sizeof(float), c->stream));
  HIP_TRY(hipMemcpy2DAsync(c->A, c->ld * sizeof(float), A, ld_host * sizeof(float), n * sizeof(float), m, hipMemcpyHostToDevice, c->stream));
  return finish(c);
}

extern "C" int fh_generate_matrix(fh_ctx* c, uint64_t m, uint64_t n, uint64_t row0, uint64_t seed, double coef) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (!c->shards.empty()) {          // shell: every shard generates its own rows of the same counter-based matrix
    FH_TRY(shell_partition(c, m, n));
    for (int k = 0; k < nshards(c); ++k) FH_TRY(fh_generate_matrix(c->shards[k], shard_rows(c, k), n, row0 + c->shard_row0[k], seed, coef));
    shell_adopt(c);
    return 0;
  }
  FH_TRY(setup_dense(c, m, n));
  const uint64_t key = fh_mix(seed);
  if (c->f32) k_gen_matrix<1><<<dim3(8192), dim3(FH_WG), 0, c->stream>>>(c->A, (uint32_t)(c->ld / 4), (uint32_t)m, (uint32_t)c->mp, (uint32_t)n, row0, key, coef);
  else k_gen_matrix<0><<<dim3(8192), dim3(FH_WG), 0, c->stream>>>(c->A, (uint32_t)(c->ld / 2), (uint32_t)m, (uint32_t)c->mp, (uint32_t)n, row0, key, coef);
  HIP_TRY(hipGetLastError());
  return finish(c);
}

extern "C" int fh_get_matrix_rows(fh_ctx* c, uint64_t row0, uint64_t nrows, double* out) {
  if (!c || !out) return fail(FH_E_ARG, "null argument");
  if (c->op != OP_DENSE) return fail(FH_E_STATE, "no dense matrix set");
  if (row0 + nrows > c->m) return fail(FH_E_ARG, "rows [%llu,%llu) out of range (m=%llu)", (unsigned long long)row0,
                                       (unsigned long long)(row0 + nrows), (unsigned long long)c->m);
  if (!c->shards.empty()) {          // shell: gather from the shards that hold the rows
    for (int k = 0; k < nshards(c); ++k) {
      const uint64_t lo = std::max(row0, c->shard_row0[k]), hi = std::min(row0 + nrows, c->shard_row0[k + 1]);
      if (lo < hi) FH_TRY(fh_get_matrix_rows(c->shards[k], lo - c->shard_row0[k], hi - lo, out + (lo - row0) * c->n));
    }
    return 0;
  }
  FH_TRY(use_device(c));
  if (!c->f32) {
    HIP_TRY(hipMemcpy2DAsync(out, c->n * sizeof(double), c->A + row0 * c->ld, c->ld * sizeof(double), c->n * sizeof(double),
                             nrows, hipMemcpyDeviceToHost, c->stream));
    return finish(c);
  }
  // float32 storage: widen on the device (exact), row blocks through a staging buffer
  const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(nrows, ((uint64_t)64 << 20) / (c->n * sizeof(double))));
  DevBuf staging;
  HIP_TRY(hipMalloc((void**)&staging.p, chunk * c->n * sizeof(double)));
  double* stage = staging.p;
  for (uint64_t r0 = 0; r0 < nrows; r0 += chunk) {
    const uint64_t rows = std::min<uint64_t>(chunk, nrows - r0);
    k_rows_from_f32<<<dim3(2048), dim3(FH_WG), 0, c->stream>>>(reinterpret_cast<const float*>(c->A) + (row0 + r0) * c->ld, c->ld, stage, c->n, (uint32_t)rows, (uint32_t)c->n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out + r0 * c->n, stage, rows * c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return finish(c);
}

extern "C" int fh_set_stencil(fh_ctx* c, uint64_t H, uint64_t W) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (H < 1 || W < 1) return fail(FH_E_ARG, "stencil needs H>=1 and W>=1 (got %llu x %llu)", (unsigned long long)H, (unsigned long long)W);
  if (H * W >= (1ull << 31)) return fail(FH_E_ARG, "image too large");
  if (!c->shards.empty() || c->owner) return fail(FH_E_STATE, "row sharding is implemented for the dense operator only");
  FH_TRY(use_device(c));
  HIP_TRY(hipStreamSynchronize(c->stream));
  free_operator(c);
  c->H = H; c->W = W;
  c->m = H * W; c->n = 2 * H * W;
  c->mp = c->m; c->ld = c->n;
  c->nv = round_up(c->n, 16); c->mv = round_up(c->m, 16);
  c->op_pending_stencil = true;
  FH_TRY(alloc_vectors(c));
  c->op_pending_stencil = false;
  c->op = OP_STENCIL;
  return finish(c);
}

extern "C" int fh_shape(fh_ctx* c, uint64_t* m, uint64_t* n) {
  if (!c || !m || !n) return fail(FH_E_ARG, "null argument");
  if (c->op == OP_NONE) return fail(FH_E_STATE, "no operator set");
  *m = c->m; *n = c->n;
  return 0;
}

static int set_loss(fh_ctx* c, int kind, const double* b, uint64_t len) {
  if (!c || !b) return fail(FH_E_ARG, "null argument");
  if (c->op == OP_NONE) return fail(FH_E_STATE, "set the operator before the loss");
  if (len != c->m) return fail(FH_E_ARG, "b has %llu entries, operator has %llu rows", (unsigned long long)len, (unsigned long long)c->m);
  if (kind != LOSS_LSQ && c->op != OP_DENSE) return fail(FH_E_STATE, "the logistic loss is implemented for the dense operator");
  if (!c->shards.empty()) {          // shell: b is sharded like the rows
    for (int k = 0; k < nshards(c); ++k) FH_TRY(set_loss(c->shards[k], kind, b + c->shard_row0[k], shard_rows(c, k)));
    c->has_b = true; c->loss_kind = kind;
    return 0;
  }
  FH_TRY(use_device(c));
  HIP_TRY(hipMemcpyAsync(c->b, b, len * sizeof(double), hipMemcpyHostToDevice, c->stream));
  c->has_b = true;
  c->loss_kind = kind;
  return finish(c);
}

extern "C" int fh_set_loss_lsq(fh_ctx* c, const double* b, uint64_t len) { return set_loss(c, LOSS_LSQ, b, len); }

extern "C" int fh_set_loss_logistic(fh_ctx* c, const double* labels, uint64_t len) {
  if (labels) for (uint64_t i = 0; i < len; ++i)
    if (labels[i] != 1.0 && labels[i] != -1.0) return fail(FH_E_ARG, "logistic labels must be -1 or +1 (entry %llu is %g)", (unsigned long long)i, labels[i]);
  return set_loss(c, LOSS_LOGISTIC, labels, len);
}

extern "C" int fh_set_prox(fh_ctx* c, int kind, double mu, double lo, double hi) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (kind < FH_PROX_IDENTITY || kind > FH_PROX_BOX) return fail(FH_E_ARG, "unknown prox kind %d", kind);
  if (kind == FH_PROX_BOX && !(lo <= hi)) return fail(FH_E_ARG, "box prox needs lo <= hi");
  for (fh_ctx* s : c->shards) { s->prox_kind = kind; s->mu = mu; s->lo = lo; s->hi = hi; }      // the prox is replicated work
  c->prox_kind = kind; c->mu = mu; c->lo = lo; c->hi = hi;
  return 0;
}

// ---- vector access --------------------------------------------------------------------------------
static double* vec_ptr(fh_ctx* c, int which, uint64_t* len) {
  const bool acc = c->last_accel;
  *len = c->n;
  // the stencil path never materialises the gradient or xhat (fh_tv.h): those ids are not addressable there
  if (c->op == OP_STENCIL && (which == FH_VEC_G0 || which == FH_VEC_G1 || which == FH_VEC_XHAT)) return nullptr;
  switch (which) {
    case FH_VEC_X0: return c->X[c->xi];
    case FH_VEC_G0: return c->G[c->gc];
    case FH_VEC_XHAT: return c->xhat;
    case FH_VEC_XPROX: return c->P[c->pc ^ 1];
    case FH_VEC_X1: return acc ? c->X[c->ti] : c->P[c->pc ^ 1];
    case FH_VEC_G1: return c->G[c->gc ^ 1];
    case FH_VEC_BEST: return c->X[c->bi];
    case FH_VEC_B: *len = c->m; return c->b;
    case FH_VEC_Z: *len = c->m; return c->Z[c->zc ^ 1];
    case FH_VEC_T0: case FH_VEC_T1: case FH_VEC_T2: case FH_VEC_T3: return c->T[which - FH_VEC_T0];
    default: return nullptr;
  }
}

static int launch_fwd_tv(fh_ctx* c, int mode, double tau, const double* x0, const double* g0, const double* xacc0,
                         double* xhat, double* xp, double* z, int sub_b);
// z = div(x) into `z` (plain stencil pass; the scalar block is scratch afterwards)
static int tv_image(fh_ctx* c, const double* x, double* z) { return launch_fwd_tv(c, 1, 0.0, x, nullptr, nullptr, nullptr, nullptr, z, 0); }
// the two-launch stencil kernels read the stored image of x0: bring it up to date after z-free steps
static int tv_refresh_zcur(fh_ctx* c) {
  if (c->op != OP_STENCIL || !c->zcur_stale) return 0;
  FH_TRY(tv_image(c, c->X[c->xi], c->Z[c->zc]));
  c->zcur = c->Z[c->zc];
  c->zcur_stale = false;
  return 0;
}

// ---- lazily-kept stencil iterate (one-pass FISTA) ---------------------------------------------------
static inline double* nq(fh_ctx* c, int i) { return i < 3 ? c->X[i] : c->P[i - 3]; }
static inline double* mq(fh_ctx* c, int i) { return i < 2 ? c->Z[i] : c->ZX[0]; }
static void lazy_pick_targets(fh_ctx* c) {
  for (int k = 0; k < 5; ++k) if (k != c->lq1 && k != c->lq0 && k != c->lb1 && k != c->lb0) { c->lqn = k; break; }
  for (int k = 0; k < 3; ++k) if (k != c->lz1 && k != c->lz0) { c->lzn = k; break; }
}
static int not_lazy(fh_ctx* c, const char* what) {
  if (c->lazy) return fail(FH_E_STATE, "%s: this solve runs the one-pass accelerated stencil step (fh_step_accel), whose iterate is kept "
                           "in extrapolated-on-the-fly form; call fh_init before switching kernels", what);
  return 0;
}
// device pointer for fh_get_vector while the iterate is lazy: x0 / x1 / best are materialised into scratch T[2]
static int lazy_vec(fh_ctx* c, int which, double** out) {
  int a = -1, b = -1; double coef = 0.0;
  switch (which) {
    case FH_VEC_X0: case FH_VEC_X1: a = c->lq1; b = c->lq0; coef = c->lc; break;
    case FH_VEC_BEST: a = c->lb1; b = c->lb0; coef = c->lbc; break;
    case FH_VEC_XPROX: *out = nq(c, c->lqn); return 0;
    default: *out = nullptr; return 0;
  }
  const unsigned grid = (unsigned)std::min<uint64_t>((c->n + FH_WG - 1) / FH_WG, 4096);
  k_extrapolate_vec<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(c->T[2], nq(c, a), nq(c, b), coef, c->n);
  HIP_TRY(hipGetLastError());
  *out = c->T[2];
  return 0;
}

static inline bool m_side(int which) { return which == FH_VEC_B || which == FH_VEC_Z; }

extern "C" int fh_set_vector(fh_ctx* c, int which, const double* host, uint64_t len) {
  if (!c || !host) return fail(FH_E_ARG, "null argument");
  if (c->op == OP_NONE) return fail(FH_E_STATE, "no operator set");
  if (!c->shards.empty()) {          // shell: n-side vectors are replicated, m-side vectors sharded like the rows
    const uint64_t want = m_side(which) ? c->m : c->n;
    if (len != want) return fail(FH_E_ARG, "vector %d has length %llu, got %llu", which, (unsigned long long)want, (unsigned long long)len);
    for (int k = 0; k < nshards(c); ++k)
      FH_TRY(m_side(which) ? fh_set_vector(c->shards[k], which, host + c->shard_row0[k], shard_rows(c, k)) : fh_set_vector(c->shards[k], which, host, len));
    if (which == FH_VEC_B) c->has_b = true;
    return 0;
  }
  if (which == FH_VEC_X0) c->lazy = false;          // a new start: fh_init follows
  uint64_t want = 0;
  double* d = vec_ptr(c, which, &want);
  if (!d) return fail(FH_E_ARG, "unknown vector id %d", which);
  if (len != want) return fail(FH_E_ARG, "vector %d has length %llu, got %llu", which, (unsigned long long)want, (unsigned long long)len);
  FH_TRY(use_device(c));
  HIP_TRY(hipMemcpyAsync(d, host, len * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (which == FH_VEC_B) c->has_b = true;
  return finish(c);
}

extern "C" int fh_get_vector(fh_ctx* c, int which, double* host, uint64_t len) {
  if (!c || !host) return fail(FH_E_ARG, "null argument");
  if (c->op == OP_NONE) return fail(FH_E_STATE, "no operator set");
  if (!c->shards.empty()) {          // shell: replicated vectors from shard 0, sharded ones gathered
    const uint64_t want = m_side(which) ? c->m : c->n;
    if (len != want) return fail(FH_E_ARG, "vector %d has length %llu, got %llu", which, (unsigned long long)want, (unsigned long long)len);
    if (!m_side(which)) return fh_get_vector(c->shards[0], which, host, len);
    for (int k = 0; k < nshards(c); ++k) FH_TRY(fh_get_vector(c->shards[k], which, host + c->shard_row0[k], shard_rows(c, k)));
    return 0;
  }
  uint64_t want = 0;
  double* d = vec_ptr(c, which, &want);
  FH_TRY(use_device(c));
  if (c->lazy && (which == FH_VEC_X0 || which == FH_VEC_X1 || which == FH_VEC_BEST || which == FH_VEC_XPROX)) FH_TRY(lazy_vec(c, which, &d));
  if (c->lazy && which == FH_VEC_Z) d = mq(c, c->lzn);
  if (c->op == OP_STENCIL && which == FH_VEC_Z && c->tvz_pending) {      // z1 = div(xprox) was never written: form it now
    FH_TRY(tv_image(c, c->lazy ? nq(c, c->lqn) : c->P[c->pc ^ 1], c->zt));
    d = c->zt;
  }
  if (!d) return fail(FH_E_ARG, "unknown vector id %d", which);
  if (len != want) return fail(FH_E_ARG, "vector %d has length %llu, got %llu", which, (unsigned long long)want, (unsigned long long)len);
  HIP_TRY(hipMemcpyAsync(host, d, len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return finish(c);
}

// ------------------------------------------------------------------------------------------------
// kernel launchers
// ------------------------------------------------------------------------------------------------
static ProxP make_prox(fh_ctx* c, double tau) {
  ProxP px;
  px.kind = c->prox_kind;
  px.thr = tau * c->mu;               // `t*self.mu`, examples/sparse_least_squares.py:44
  px.lo = c->lo; px.hi = c->hi;
  px.level = c->dscal + FH_NSCALARS;  // device scalar written by the level search
  return px;
}

template <int R, int KIND>
static void launch_fwd_rk(fh_ctx* c, const FwdP& p, unsigned grid) {
  if (c->f32) {                                   // float32 storage: non-temporal loads only, R = 4 or 8
    if constexpr (R == 16) k_fwd_dense<8, 1, KIND, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
    else k_fwd_dense<R, 1, KIND, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  }
  else if (c->nt_loads) k_fwd_dense<R, 1, KIND><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  else k_fwd_dense<R, 0, KIND><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
}
template <int R>
static void launch_fwd_r(fh_ctx* c, const FwdP& p, unsigned grid, int kind) {
  switch (kind) {
    case PX_PLAIN:  launch_fwd_rk<R, PX_PLAIN>(c, p, grid); break;
    case PX_SHRINK: launch_fwd_rk<R, PX_SHRINK>(c, p, grid); break;
    case PX_NONNEG: launch_fwd_rk<R, PX_NONNEG>(c, p, grid); break;
    case PX_LINF:   launch_fwd_rk<R, PX_LINF>(c, p, grid); break;
    case PX_L1BALL: launch_fwd_rk<R, PX_L1BALL>(c, p, grid); break;
    case PX_BOX:    launch_fwd_rk<R, PX_BOX>(c, p, grid); break;
    default:        launch_fwd_rk<R, PX_IDENTITY>(c, p, grid); break;
  }
}

// z := A * (mode 0: prox(x0 - tau g0) ; mode 1: x0) on the dense operator
static int launch_fwd_dense(fh_ctx* c, int mode, double tau, const double* x0, const double* g0, const double* xacc0,
                            double* xhat, double* xp, double* z, int sub_b) {
  // rows per pass (sweep, profiles/r01_tune_sizes.txt): 4 up to n = 32768, 8 beyond
  // float32 storage: the x0/g0 pieces of a trip are twice as many per byte of A, so it takes 8 rows per pass from n = 32768 on
  // to keep as many bytes of A in flight (4 rows: 4.7 TB/s at 65536^2, profiles/r02_f32_storage.txt)
  int R = c->fwd_rows ? c->fwd_rows : (c->ld <= (c->f32 ? 16384u : 32768u) ? 4 : 8);
  if (c->f32 && R == 16) R = 8;
  if (mode == 0 && c->prox_kind == FH_PROX_TVBALL) return fail(FH_E_STATE, "TV-ball prox needs the stencil operator");
  FwdP p;
  p.A = c->A; p.ld = c->ld; p.ld2 = (uint32_t)(c->ld / (c->f32 ? 4 : 2)); p.nv2 = (uint32_t)(c->nv / 2); p.n = (uint32_t)c->n; p.m = (uint32_t)c->m;
  p.nrg = (uint32_t)(c->mp / R);
  p.nchunks = (p.nv2 + FH_WG - 1) / FH_WG;
  p.x0 = x0; p.g0 = g0; p.xacc0 = xacc0; p.xhat = xhat; p.xp = xp;
  p.b = c->b; p.z = z; p.tau = tau; p.sub_b = sub_b; p.loss = c->loss_kind;
  p.px = make_prox(c, tau);
  const int kind = mode == 0 ? c->prox_kind : (int)PX_PLAIN;
  unsigned grid = std::max(p.nrg, mode == 0 ? p.nchunks : 1u);
  // measured on MI355X (profiles/r01_tune_dense.txt, r01_tune_sizes.txt): 2 persistent workgroups per CU
  // grid-striding over the row groups beat one workgroup per row group by 5-12 %
  grid = (unsigned)std::min<long long>(grid, c->fwd_cap > 0 ? c->fwd_cap : 512);
  const size_t need = ((size_t)p.nchunks * 8 + grid) * sizeof(double);
  FH_TRY(ensure_ws(c, need));
  p.red_n = c->ws; p.red_m = c->ws + (size_t)p.nchunks * 8;
  p.counter = c->counters + CNT_FWD;
  p.out = scalar_out(c);
  t_begin(c, FH_K_FWD);
  switch (R) {
    case 4: launch_fwd_r<4>(c, p, grid, kind); break;
    case 16: launch_fwd_r<16>(c, p, grid, kind); break;
    default: launch_fwd_r<8>(c, p, grid, kind); break;
  }
  t_end(c, FH_K_FWD);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int CPT>
static void launch_adj_c(fh_ctx* c, const AdjP& p, unsigned grid) {
  if (c->f32) k_adj_dense<CPT, 1, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  else if (c->nt_loads) k_adj_dense<CPT, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  else k_adj_dense<CPT, 0><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
}

struct AdjIO {
  const double* z; const double* zacc0; int sub_b; int accel; double coef; int mode; double tau;
  const double* x0; const double* xp; const double* xacc0; const double* xhat; double* x1; double* g1;
  const double* g0;   // stencil path only: K-adj recomputes xhat = x0 - tau*g0
};

static int launch_adj_dense(fh_ctx* c, const AdjIO& io) {
  AdjP p;
  p.A = c->A; p.ld = c->ld; p.ld2 = (uint32_t)(c->ld / (c->f32 ? 4 : 2)); p.nv2 = (uint32_t)(c->nv / 2);
  p.n = (uint32_t)c->n; p.mp = (uint32_t)c->mp; p.m = (uint32_t)c->m;
  // auto rules from the MI355X sweeps (profiles/r01_tune_dense.txt, r01_tune_sizes.txt): about 32 slabs
  // (more when there are few column chunks, so that >= 128 workgroups exist), slabs of 32..2048 rows, and
  // column chunks of 2 x 16 B per lane below n = 32768, 4 x 16 B from there on (1 x for n <= 1024).
  int CPT = c->adj_cpt;
  if (CPT == 0) CPT = p.ld2 <= 512 ? 1 : (p.ld2 < 16384 ? 2 : 4);
  p.ncc = (p.ld2 + FH_WG * CPT - 1) / (FH_WG * CPT);
  uint32_t slab = (uint32_t)c->adj_slab;
  if (slab == 0) {
    // (float32 storage has half the column chunks per row: aim for the same ~1024 workgroups the float64 matrix gets at C2)
    const uint64_t target_slabs = std::max<uint64_t>(32, ((c->f32 ? 1024 : 128) + p.ncc - 1) / p.ncc);
    const uint64_t slab_min = p.ncc >= 8 ? 128 : 32;
    uint64_t s = round_up((c->mp + target_slabs - 1) / target_slabs, 8);
    slab = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(s, slab_min), ADJ_MAX_SLAB);
  }
  p.slab_rows = slab;
  p.nslab = (uint32_t)((c->mp + slab - 1) / slab);
  if (p.ncc + CNT_ADJ_CC > (uint32_t)kCounterWords) return fail(FH_E_ARG, "too many column chunks (%u)", p.ncc);
  p.z = io.z; p.zacc0 = io.zacc0; p.b = c->b; p.sub_b = io.sub_b; p.loss = c->loss_kind; p.accel = io.accel; p.coef = io.coef;
  p.mode = io.mode; p.tau = io.tau;
  p.x0 = io.x0; p.xp = io.xp; p.xacc0 = io.xacc0; p.xhat = io.xhat; p.x1 = io.x1; p.g1 = io.g1;
  const size_t gpart_elems = (size_t)p.nslab * c->ld;
  const size_t need = (gpart_elems + (size_t)p.ncc * 8 + p.nslab) * sizeof(double);
  FH_TRY(ensure_ws(c, need));
  p.gpart = c->ws; p.red_bb = c->ws + gpart_elems; p.red_f = p.red_bb + (size_t)p.ncc * 8;
  p.cc_counter = c->counters + CNT_ADJ_CC; p.fin_counter = c->counters + CNT_ADJ_FIN;
  p.out = scalar_out(c);
  const unsigned grid = p.ncc * p.nslab;
  t_begin(c, FH_K_ADJ);
  switch (CPT) {
    case 1: launch_adj_c<1>(c, p, grid); break;
    case 4: launch_adj_c<4>(c, p, grid); break;
    default: launch_adj_c<2>(c, p, grid); break;
  }
  t_end(c, FH_K_ADJ);
  HIP_TRY(hipGetLastError());
  return 0;
}

// n-side epilogue as its own launch (row-sharded runs, after the all-reduce of g1)
static int bb_epilogue_only(fh_ctx* c, const AdjIO& io, const double* fsq_src, const double* coef_src = nullptr, const double* pack = nullptr) {
  AdjP p;
  memset(&p, 0, sizeof(p));
  p.ld = c->nv; p.ld2 = (uint32_t)(c->nv / 2); p.nv2 = p.ld2; p.n = (uint32_t)c->n;
  p.accel = io.accel; p.coef = io.coef; p.mode = 0; p.tau = io.tau;
  p.x0 = io.x0; p.xp = io.xp; p.xacc0 = io.xacc0; p.xhat = io.xhat; p.x1 = io.x1; p.g1 = io.g1;
  const uint32_t nchunks = (p.ld2 + FH_WG - 1) / FH_WG;
  FH_TRY(ensure_ws(c, (size_t)nchunks * 8 * sizeof(double)));
  p.red_bb = c->ws; p.fin_counter = c->counters + CNT_AUX; p.out = c->dscal;
  t_begin(c, FH_K_AUX);
  k_bb_epilogue<<<dim3(nchunks), dim3(FH_WG), 0, c->stream>>>(p, nchunks, fsq_src, coef_src, pack, c->hscal_dev);
  c->scal_mirrored = true;         // (always the last launch before the caller's fetch_scalars)
  t_end(c, FH_K_AUX);
  HIP_TRY(hipGetLastError());
  return 0;
}

// sum|x_i| and max|x_i| of an n-length device vector -> dscal[GSUM], dscal[GMAX]  (g(x0) for objective_hist[0], :143)
static int launch_gterms(fh_ctx* c, const double* x) {
  const unsigned grid = (unsigned)std::min<uint64_t>((c->n + FH_WG - 1) / FH_WG, 1024);
  FH_TRY(ensure_ws(c, (size_t)grid * 2 * sizeof(double)));
  t_begin(c, FH_K_AUX);
  k_gterms<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(x, (uint32_t)c->n, c->ws, c->counters + CNT_AUX, scalar_out(c));
  t_end(c, FH_K_AUX);
  HIP_TRY(hipGetLastError());
  return 0;
}

// clipping level alpha for FH_PROX_LINF (radius tau*mu) / FH_PROX_L1BALL (radius mu) -> dscal[FH_NSCALARS]
static int launch_level_search(fh_ctx* c, double tau) {
  if (c->op != OP_DENSE) return fail(FH_E_STATE, "LINF / L1BALL prox need the dense operator");
  const double radius = c->prox_kind == FH_PROX_L1BALL ? c->mu : tau * c->mu;
  const double* x0 = c->X[c->xi];
  const double* g0 = c->G[c->gc];
  double* out = c->dscal + FH_NSCALARS;
  const uint32_t n = (uint32_t)c->n;
  t_begin(c, FH_K_AUX);
  if (n <= 1u * LVL_WG) k_level_search<1><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 4u * LVL_WG) k_level_search<4><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 16u * LVL_WG) k_level_search<16><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 64u * LVL_WG) k_level_search<64><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else k_level_search<0><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  t_end(c, FH_K_AUX);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_fwd_tv(fh_ctx* c, int mode, double tau, const double* x0, const double* g0, const double* xacc0,
                         double* xhat, double* xp, double* z, int sub_b) {
  (void)xhat; (void)g0;   // the stencil path materialises neither xhat nor the gradient (fh_tv.h)
  if (mode == 0 && c->prox_kind != FH_PROX_TVBALL && c->prox_kind != FH_PROX_IDENTITY)
    return fail(FH_E_STATE, "the stencil operator supports the TV-ball prox or no prox (got kind %d)", c->prox_kind);
  const uint32_t H = (uint32_t)c->H, W = (uint32_t)c->W;
  const uint32_t rows_wg = (uint32_t)(c->tv_rows > 0 ? c->tv_rows : 32);
  const uint32_t row_chunks = (H + rows_wg - 1) / rows_wg;
  if (mode == 0) {
    if (!c->zcur) return fail(FH_E_STATE, "fh_fwd on the stencil operator before fh_init");
    TvStepFwdP p;
    p.H = H; p.W = W; p.rows_wg = rows_wg;
    p.strip_groups = ((W + TVS_FWD_OWN - 1) / TVS_FWD_OWN + 3) / 4;
    p.x0 = x0; p.xacc0 = xacc0; p.xp = xp; p.zc = c->zcur; p.b = c->b; p.zn = z; p.tau = tau;
    const unsigned grid = p.strip_groups * row_chunks;
    FH_TRY(ensure_ws(c, (size_t)grid * 8 * sizeof(double)));
    p.red = c->ws; p.counter = c->counters + CNT_FWD; p.out = scalar_out(c);
    t_begin(c, FH_K_FWD);
#define TV_STEP(U, NT)                                                                                           \
  do {                                                                                                           \
    if (c->prox_kind == FH_PROX_TVBALL) k_fwd_tv_step<0, U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);   \
    else k_fwd_tv_step<1, U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);                                  \
  } while (0)
    if (c->tv_nt == 1) { if (c->tv_u == 2) TV_STEP(2, 1); else if (c->tv_u == 4) TV_STEP(4, 1); else TV_STEP(8, 1); }
    else { if (c->tv_u == 2) TV_STEP(2, 0); else if (c->tv_u == 4) TV_STEP(4, 0); else TV_STEP(8, 0); }
#undef TV_STEP
    t_end(c, FH_K_FWD);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  TvFwdP p;
  p.H = H; p.W = W; p.rows_wg = rows_wg;
  p.strip_groups = ((W + TV_SW - 1) / TV_SW + 3) / 4;
  (void)xp; (void)tau;
  p.x0 = x0; p.b = c->b; p.z = z; p.sub_b = sub_b;
  const unsigned grid = p.strip_groups * row_chunks;
  FH_TRY(ensure_ws(c, (size_t)grid * 8 * sizeof(double)));
  p.red = c->ws; p.counter = c->counters + CNT_FWD; p.out = scalar_out(c);
  t_begin(c, FH_K_FWD);
  if (c->tv_nt == 1) k_fwd_tv<4, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  else k_fwd_tv<4, 0><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  t_end(c, FH_K_FWD);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_adj_tv(fh_ctx* c, const AdjIO& io) {
  const uint32_t H = (uint32_t)c->H, W = (uint32_t)c->W;
  const uint32_t rows_wg = (uint32_t)(c->tv_rows > 0 ? c->tv_rows : 128);
  const uint32_t row_chunks = (H + rows_wg - 1) / rows_wg;
  if (io.mode == 0) {          // FBS step: reductions only, the gradient is recomputed from z and b
    if (!c->zcur) return fail(FH_E_STATE, "fh_adj on the stencil operator before fh_init");
    TvStepAdjP p;
    p.H = H; p.W = W; p.rows_wg = rows_wg;
    p.strip_groups = ((W + TVS_ADJ_OWN - 1) / TVS_ADJ_OWN + 3) / 4;
    p.zn = io.z; p.zacc0 = io.zacc0; p.zc = c->zcur; p.b = c->b;
    p.accel = io.accel; p.coef = io.coef; p.tau = io.tau;
    p.x0 = io.x0; p.xp = io.xp; p.xacc0 = io.xacc0; p.x1 = io.x1; p.zx = c->ZX[c->zxc ^ 1];
    const unsigned grid = p.strip_groups * row_chunks;
    FH_TRY(ensure_ws(c, (size_t)grid * 8 * sizeof(double)));
    p.red = c->ws; p.counter = c->counters + CNT_ADJ_FIN; p.out = scalar_out(c);
    t_begin(c, FH_K_ADJ);
#define TV_STEP(U, NT) k_adj_tv_step<U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p)
    if (c->tv_nt == 1) { if (c->tv_u == 2) TV_STEP(2, 1); else if (c->tv_u == 4) TV_STEP(4, 1); else TV_STEP(8, 1); }
    else { if (c->tv_u == 2) TV_STEP(2, 0); else if (c->tv_u == 4) TV_STEP(4, 0); else TV_STEP(8, 0); }
#undef TV_STEP
    t_end(c, FH_K_ADJ);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  TvAdjP p;                    // plain gradient (Lipschitz probes, fh_apply): materialises g1 = grad(z - b)
  p.H = H; p.W = W; p.rows_wg = rows_wg;
  p.strip_groups = ((W + TV_SW - 1) / TV_SW + 3) / 4;
  p.z = io.z; p.b = c->b; p.sub_b = io.sub_b; p.g1 = io.g1;
  const unsigned grid = p.strip_groups * row_chunks;
  FH_TRY(ensure_ws(c, (size_t)grid * 8 * sizeof(double)));
  p.red = c->ws; p.counter = c->counters + CNT_ADJ_FIN; p.out = scalar_out(c);
  t_begin(c, FH_K_ADJ);
  if (c->tv_nt == 1) k_adj_tv<4, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  else k_adj_tv<4, 0><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  t_end(c, FH_K_ADJ);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- fused one-pass iteration (fh_fused.h) ---------------------------------------------------------------
// Shape of the one-pass launch: TEAM members x 256 lanes x PPT 16-byte pieces cover one row; lanes past the row's last
// piece are masked (clamped loads, zero x), so any n up to 262144 fits the next shape up.  A member's piece of a row is
// kept at 5..8 pieces per lane (20-32 KiB per workgroup per row) by choosing the team size -- fewer members means more
// teams, i.e. fewer rows (trips of ~0.7-1.2 us) per team:
//   n <= 4096  : 1 member  (a workgroup owns whole rows: no exchange), PPT = ceil(n/512) rounded up to 1, 2, 4, 5..8
//   n <= 8192  : 2 members x PPT = ceil(n/1024) in 5..8, posts one row ahead
//   n <= 16384 : 4 members x PPT = ceil(n/2048) in 5..8, posts one row ahead
//   n <= 32768 : 8 members x PPT = ceil(n/4096) in 5..8, posts one row ahead
//   n <= 65536 : 16 members x PPT = ceil(n/8192) in 5..8, posts two rows ahead
//   n <= 131072: 16 members x PPT = ceil(n/8192) in 9..16, x slice in LDS, posts one row ahead (3-4 row buffers)
//   n <= 262144: 32 members x PPT = ceil(n/16384) in 9..16, same schedule (6.1 TB/s at n = 262144: a trip with 32 members is
//                slower, but still 1.8x the two-launch path)
// FH_TUNE_FUSED_VARIANT bit 8 (A/B, tests): 8 members for every n <= 32768 and 8 members x 16 pieces in line at n = 65536;
// bit 16: n in (65536, 131072] as in round 1 (16 members x 16 pieces in registers, exchange in line).
// FusedShape = the template key of k_fused_dense (PPT, PIPE, TEAM, XLDS, NBO) for a row of n columns: a pure function of
// (n, row stride, storage, FH_TUNE_FUSED_VARIANT, #CUs), exported as fh_fused_shape so that it can be checked without a GPU.
struct FusedShape { int ppt, team, pipe, xlds, nbo; };
static FusedShape fused_shape_for(uint64_t n, uint64_t ld, int f32, int variant, int ncu) {
  const FusedShape none = {0, 0, 0, 0, 0};
  if (ld % 2 || n == 0) return none;
  // 16-byte pieces per row that hold data (the row stride ld may be padded): 2 columns each, 4 in float32 storage
  const uint64_t pieces = f32 ? round_up(n, 32) / 4 : round_up(n, 16) / 2;
  if (pieces > ld / (f32 ? 4 : 2)) return none;
  FusedShape sh = none;
  if (f32) {
    // float32 storage: the same byte rule (a member's piece of a row is 5..8 pieces per lane = 20-32 KiB per workgroup per
    // row), i.e. twice the columns per team size: n <= 8192 one member, then 2 / 4 / 8 / 16 members up to n = 131072.  A piece
    // carries four columns, so the x and g1 slices cost twice the registers per piece: from 5 pieces on the x slice lives in
    // LDS and 4 (5-6 pieces) or 3 (7-8 pieces) row buffers rotate, posting one row ahead -- the spill-free combinations
    // (-Rpass-analysis=kernel-resource-usage)
    for (int team = 1; team <= 16; team *= 2) {
      if (pieces > (uint64_t)team * FH_WG * 8) continue;
      int ppt = (int)((pieces + (uint64_t)team * FH_WG - 1) / ((uint64_t)team * FH_WG));
      if (ppt == 3) ppt = 4;
      if (team > 1 && ppt < 5) ppt = 5;          // (cannot happen: pieces > (team/2)*256*8 already means ppt >= 5)
      const int xl = ppt >= 5 ? 1 : 0;
      sh = {ppt, team, 1, xl, xl ? (ppt <= 6 ? 4 : 3) : 0};
      break;
    }
  } else if (pieces <= (uint64_t)1 * FH_WG * 8 && !(variant & 8)) {
    int ppt = (int)((pieces + FH_WG - 1) / FH_WG);                       // n <= 4096: a workgroup owns whole rows, 256 "teams" of one
    if (ppt == 3) ppt = 4;
    sh = {ppt, 1, 1, 0, 0};
  } else if (pieces > (uint64_t)1 * FH_WG * 8 && pieces <= (uint64_t)2 * FH_WG * 8 && !(variant & 8)) {
    sh = {(int)((pieces + 2 * FH_WG - 1) / (2 * FH_WG)), 2, 1, 0, 0};  // n in (4096, 8192]: 2 members x 5..8 pieces, 128 teams
  } else if (pieces > (uint64_t)2 * FH_WG * 8 && pieces <= (uint64_t)4 * FH_WG * 8 && !(variant & 8)) {
    sh = {(int)((pieces + 4 * FH_WG - 1) / (4 * FH_WG)), 4, 1, 0, 0};  // n in (8192, 16384]: 4 members x 5..8 pieces, 64 teams
  } else if (pieces <= (uint64_t)8 * FH_WG * 8) {
    int ppt = (int)((pieces + 8 * FH_WG - 1) / (8 * FH_WG));
    if (ppt == 3) ppt = 4;
    sh = {ppt, 8, 1, 0, 0};
  } else if (pieces == (uint64_t)8 * FH_WG * 16 && (variant & 8)) {
    sh = {16, 8, 0, 0, 0};
  } else if (pieces <= (uint64_t)16 * FH_WG * 8) {
    sh = {(int)((pieces + 16 * FH_WG - 1) / (16 * FH_WG)), 16, 2, 0, 0};       // 16 members: posts run two rows ahead of the polls
  } else if (pieces <= (uint64_t)16 * FH_WG * 16) {
    // n in (65536, 131072]: 16 members x 9..16 pieces POSTING ONE ROW AHEAD, made possible by keeping the x slice in LDS (the
    // registers hold 3-4 row buffers -- the largest count hipcc allocates without spilling -- and the g1 slice); measured against
    // the round-1 in-line shape in profiles/r02_fused_wide.txt: 131072 columns 6.41 -> 4.80 ms (7.16 TB/s), 70000: 5.80 -> 2.87 ms
    // (variant bit 16: that round-1 shape -- 16 pieces, x slice in registers, 3 row buffers, exchange in line)
    const int ppt = (int)((pieces + 16 * FH_WG - 1) / (16 * FH_WG));
    sh = (variant & 16) ? FusedShape{16, 16, 0, 0, 0} : FusedShape{ppt, 16, 1, 1, ppt <= 10 ? 4 : 3};
  } else if (pieces <= (uint64_t)32 * FH_WG * 16) {
    // n in (131072, 262144]: 32 members (a whole XCD per team, 8 teams) x 9..16 pieces, same schedule
    const int ppt = (int)((pieces + 32 * FH_WG - 1) / (32 * FH_WG));
    sh = {ppt, 32, 1, 1, ppt <= 10 ? 4 : 3};
  }
  if (!sh.ppt || ncu < sh.team || ncu % sh.team) return none;     // one workgroup per CU, whole teams only
  return sh;
}
static const FusedEntry* fused_lookup(const FusedShape& sh, int f32) {
  for (const FusedEntry& e : kFusedTable)
    if (e.ppt == sh.ppt && e.pipe == sh.pipe && e.team == sh.team && e.xlds == sh.xlds && e.nbo == sh.nbo && e.f32 == f32) return &e;
  return nullptr;
}
static FusedShape fused_shape(fh_ctx* c) {
  if (c->op != OP_DENSE || c->prox_kind == FH_PROX_TVBALL) return FusedShape{0, 0, 0, 0, 0};
  return fused_shape_for(c->n, c->ld, c->f32, c->fused_variant, c->ncu);
}
static int fused_ppt(fh_ctx* c) { return fused_shape(c).ppt; }
// diagnostic / test entry: the shape chosen for n columns and whether its kernel is instantiated (no device needed)
extern "C" int fh_fused_shape(uint64_t n, int dtype, int variant, int ncu, int* shape5, int* instantiated) {
  if (!shape5 || !instantiated) return fail(FH_E_ARG, "null argument");
  const int f32 = dtype == FH_DTYPE_F32_STORAGE ? 1 : 0;
  const FusedShape sh = fused_shape_for(n, round_up(n, f32 ? 32 : 16), f32, variant, ncu);
  shape5[0] = sh.ppt; shape5[1] = sh.pipe; shape5[2] = sh.team; shape5[3] = sh.xlds; shape5[4] = sh.nbo;
  *instantiated = sh.ppt && fused_lookup(sh, f32) ? 1 : 0;
  return 0;
}
// the one-pass launch beats K-fwd + K-adj once its fixed cost is amortised: wide rows, or at least 8 Mi elements
// (profiles/r02_fused_crossover.txt; 32 Mi in round 1, when every launch still refilled its hand-off slots from the host)
static bool fused_pays(fh_ctx* c) { return c->n >= 16384 || (uint64_t)c->m * c->n >= ((uint64_t)1 << 23); }

// (the prox kind travels in p.px.kind: a run-time switch in the kernel's n-side prologue; FH_PROX_* == PX_* numerically.  The
// one-pass kernels always stream A with non-temporal loads, +10 % in the dense sweeps: only NT = 1 is built.)
// operands of one fused launch; fh_step takes them from the solver state, fh_init / fh_gradient_at pass their own
struct FusedIO {
  const double* x0; const double* g0; double* xhat; double* xp; double* z; double* g1;
  int kind;     // prox kind (FH_PROX_IDENTITY with tau = 0 gives the plain pair z = A x0, g1 = A^T grad f(z))
  int mode;     // 0 = with the n-side epilogue, 2 = g1 (+ loss) only
  // FISTA (zero-initialised = off): x1 = xp + c*(xp - xacc0), gradient at z + c*(z - zacc0), c = coef or 0 after a restart
  int accel = 0, restart = 0; double coef = 0.0;
  const double* xacc0 = nullptr; const double* zacc0 = nullptr; double* x1 = nullptr; double* coef_out = nullptr;
  double* pack = nullptr;      // row-sharded: where the launch appends its loss sums and timeout word (behind g1)
};

// after the synchronisation that follows a one-pass launch: a launch that timed out has left slots un-posted / un-armed
static inline void fused_after(fh_ctx* c) { if (c->hscal[15] != 0.0) c->slots_sig = 0; }

static int launch_fused_dense(fh_ctx* c, double tau, const FusedIO& io) {
  const FusedShape sh = fused_shape(c);
  if (!sh.ppt) return fail(FH_E_STATE, "fused one-pass step: unsupported operator shape (needs a dense A with n <= 262144 and a scalar-separable prox)");
  const FusedEntry* k_fused_dense_entry = fused_lookup(sh, c->f32);
  if (!k_fused_dense_entry)
    return fail(FH_E_STATE, "fused one-pass step: no instantiation for PPT %d, PIPE %d, TEAM %d, XLDS %d, NBO %d, F32 %d (fh_fused_instances.inc)",
                sh.ppt, sh.pipe, sh.team, sh.xlds, sh.nbo, c->f32);
  FusedP p;
  p.A = c->A; p.ld = c->ld; p.n = (uint32_t)c->n; p.m = (uint32_t)c->m; p.mp = (uint32_t)c->mp;
  p.ld2 = (uint32_t)(c->f32 ? round_up(c->n, 32) / 4 : round_up(c->n, 16) / 2);
  p.ldp = (uint32_t)(c->ld / (c->f32 ? 4 : 2));
  p.nv2 = p.ld2 * (c->f32 ? 2u : 1u);
  p.nteams = (uint32_t)(c->ncu / sh.team);
  // few rows: fewer teams (at least FUSED_MIN_ROWS rows each when possible, and a multiple of 8 teams so that the members of
  // a team stay on one XCD): a smaller grid barrier and fewer g1 partials to sum in the epilogue
  if (c->fused_min_rows > 0) {
    const uint64_t want = std::max<uint64_t>(8, round_up((c->mp + c->fused_min_rows - 1) / c->fused_min_rows, 8));
    p.nteams = (uint32_t)std::min<uint64_t>(p.nteams, want);
  }
  p.rows_per_team = (uint32_t)((c->mp + p.nteams - 1) / p.nteams);
  p.x0 = io.x0; p.g0 = io.g0; p.xhat = io.xhat; p.xp = io.xp;
  p.b = c->b; p.z = io.z; p.tau = tau; p.loss = c->loss_kind; p.mode = io.mode;
  p.px = make_prox(c, tau);
  p.px.kind = io.kind;
  p.accel = io.accel; p.restart = io.restart; p.coef = io.coef; p.xacc0 = io.xacc0; p.zacc0 = io.zacc0; p.x1 = io.x1; p.coef_out = io.coef_out;
  p.pack = io.pack;
  const unsigned grid = p.nteams * sh.team;
  const size_t slots_elems = ((size_t)c->mp + p.nteams) * (sh.team < 8 ? 8 : sh.team);     // (32 members: four 64-byte lines per row)   // whole 64-byte lines; + one line per team for the restart dot
  const size_t gpart_elems = (size_t)p.nteams * p.nv2 * 2;
  FH_TRY(ensure_ws(c, (gpart_elems + (size_t)grid * 16) * sizeof(double)));
  p.gpart = c->ws; p.red = p.gpart + gpart_elems;
  if (2 * slots_elems * sizeof(double) > c->slotbuf_bytes) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->slotbuf) { HIP_TRY(hipFree(c->slotbuf)); c->slotbuf = nullptr; c->slotbuf_bytes = 0; }
    const size_t bytes = round_up(2 * slots_elems * sizeof(double), 1 << 20);
    HIP_TRY(hipMalloc((void**)&c->slotbuf, bytes));
    c->slotbuf_bytes = bytes;
    c->slots_sig = 0;
  }
  p.g1 = io.g1;
  p.bar = c->counters + CNT_FUSED_BAR; p.err = c->counters + CNT_FUSED_ERR; p.variant = c->fused_variant;
  p.out = scalar_out(c);
  t_begin(c, FH_K_FUSED);
  {
    // signature of everything the slot layout depends on; 0 = "refill" (set after a timed-out launch, see fused_after)
    uint64_t sig = fh_mix((uint64_t)(uintptr_t)c->slotbuf ^ fh_mix(slots_elems * 131 + (uint64_t)sh.team * 7 + p.nteams)) | 1ull;
    if (sig != c->slots_sig) {
      // (a team of one exchanges nothing through the slots, but its grid barrier and error word are the same counters: a launch
      // that timed out in another shape must not leave them armed for it)
      if (sh.team > 1) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->slotbuf, (int)FT_SENTINEL_HI, 2 * slots_elems * 2, c->stream));
      HIP_TRY(hipMemsetAsync(c->counters + CNT_FUSED_BAR, 0, 8 * sizeof(unsigned), c->stream));
      c->slots_sig = sig;
      c->slots_parity = 0;
    }
    p.slots = c->slotbuf + (size_t)c->slots_parity * slots_elems;
    p.slots_next = c->slotbuf + (size_t)(c->slots_parity ^ 1) * slots_elems;
    c->slots_parity ^= 1;
  }
  k_fused_dense_entry->kernel<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  return 0;
}

// z = A x, g = A^T grad f(z) from ONE read of A when the one-pass kernel pays off (single GPU, fused_pays()):
// identity prox and tau = 0 make xprox = x.  `xhat` and the prox target serve as the launch's scratch outputs.
static bool cu_masked() { return getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK"); }
static bool plain_pair_fused_ok(fh_ctx* c) { return c->op == OP_DENSE && !row_sharded(c) && c->shards.empty() && fused_ppt(c) && fused_pays(c) && !cu_masked(); }
// returns 0 and sets *ok = false when the launch reported a spin timeout (caller falls back to two launches)
static int plain_pair_fused(fh_ctx* c, const double* x, double* z, double* g, bool* ok) {
  const FusedIO fio = {x, x, c->xhat, c->P[c->pc ^ 1], z, g, FH_PROX_IDENTITY, 2};
  FH_TRY(launch_fused_dense(c, 0.0, fio));
  FH_TRY(finish(c));                          // single GPU: the scalar block (incl. the timeout word) is in mapped host memory
  *ok = c->hscal[15] == 0.0;
  fused_after(c);
  return 0;
}

// ---- operator-generic wrappers ---------------------------------------------------------------------
static int op_fwd(fh_ctx* c, int mode, double tau, const double* x0, const double* g0, const double* xacc0,
                  double* xhat, double* xp, double* z, int sub_b) {
  if (c->op == OP_DENSE) return launch_fwd_dense(c, mode, tau, x0, g0, xacc0, xhat, xp, z, sub_b);
  if (c->op == OP_STENCIL) return launch_fwd_tv(c, mode, tau, x0, g0, xacc0, xhat, xp, z, sub_b);
  return fail(FH_E_STATE, "no operator set");
}

// ---- the adjoint launch in three stages, so that a shell can run stage 1 on every shard, ONE exchange, stage 3 on every shard ----
// stage 1, local: row-sharded contexts leave the n-side epilogue (mode 0) to adj_tail, which needs the summed g1
static int adj_local(fh_ctx* c, const AdjIO& io_in) {
  AdjIO io = io_in;
  if (row_sharded(c) && c->op != OP_DENSE) return fail(FH_E_STATE, "row sharding is implemented for the dense operator only");
  if (row_sharded(c) && io.mode == 0) io.mode = 2;
  if (c->op == OP_DENSE) return launch_adj_dense(c, io);
  if (c->op == OP_STENCIL) return launch_adj_tv(c, io);
  return fail(FH_E_STATE, "no operator set");
}
// stage 2, exchange: A_k^T r_k partials (nv doubles at g1(shard)) and the local loss sums (FH_S_FSQ_ADJ) summed over the row blocks
template <typename Sel>
static int adj_sum(fh_ctx* c, Sel g1) {
  return sum_over_shards(c, g1, (size_t)c->nv, [](fh_ctx* s) { return s->dscal + FH_S_FSQ_ADJ; }, 1);
}
// stage 3: the n-side epilogue on the summed g1
static int adj_tail(fh_ctx* c, const AdjIO& io) {
  if (!row_sharded(c) || io.mode != 0) return 0;
  return bb_epilogue_only(c, io, c->dscal + FH_S_FSQ_ADJ);
}
// all three on a plain context (fh_init, fh_gradient_at, fh_apply of a single context)
static int op_adj(fh_ctx* c, const AdjIO& io) {
  FH_TRY(adj_local(c, io));
  double* g1 = io.g1;
  FH_TRY(adj_sum(c, [g1](fh_ctx*) { return g1; }));
  return adj_tail(c, io);
}

// local ||r_k||^2 (or logistic loss sum) of the forward launch summed over the row blocks, before the host's line-search test
static int reduce_fsq_over_ranks(fh_ctx* c) {
  return sum_over_shards(c, [](fh_ctx* s) { return s->dscal + FH_S_FSQ; }, 1);
}

static int check_ready(fh_ctx* c, bool need_b) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (c->op == OP_NONE) return fail(FH_E_STATE, "no operator set (call fh_set_matrix / fh_generate_matrix / fh_set_stencil)");
  if (need_b && !c->has_b) return fail(FH_E_STATE, "no loss set (call fh_set_loss_lsq)");
  return c->shards.empty() ? use_device(c) : 0;      // (a shell selects the device shard by shard)
}

// the solver-state operands of K-adj / the n-side epilogue (fh_adj, fh_fwd_adj, fh_step on a row-sharded context)
static AdjIO solver_adj_io(fh_ctx* c, double tau, int accel, double coef) {
  AdjIO io;
  io.z = c->Z[c->zc ^ 1]; io.zacc0 = c->Z[c->zc]; io.sub_b = 1; io.accel = accel ? 1 : 0; io.coef = coef;
  io.mode = 0; io.tau = tau;
  io.x0 = c->X[c->xi]; io.xp = c->P[c->pc ^ 1]; io.xacc0 = c->P[c->pc]; io.xhat = c->xhat;
  io.x1 = c->X[c->ti]; io.g1 = c->G[c->gc ^ 1]; io.g0 = c->G[c->gc];
  return io;
}
// K-fwd of the solver state (level search for the two sort-free prox kinds first)
static int solver_fwd_local(fh_ctx* c, double tau, const char* who) {
  FH_TRY(use_device(c));
  FH_TRY(not_lazy(c, who));
  FH_TRY(tv_refresh_zcur(c));
  c->tvz_pending = false;
  if (c->prox_kind == FH_PROX_LINF || c->prox_kind == FH_PROX_L1BALL) FH_TRY(launch_level_search(c, tau));
  return op_fwd(c, 0, tau, c->X[c->xi], c->G[c->gc], c->P[c->pc], c->xhat, c->P[c->pc ^ 1], c->Z[c->zc ^ 1], 1);
}

// ------------------------------------------------------------------------------------------------
// solver steps
// ------------------------------------------------------------------------------------------------
extern "C" int fh_init(fh_ctx* c, double* scalars) {
  FH_TRY(check_ready(c, true));
  const int ns = nshards(c);
  // z_accel1 := A x0 lands in Z[zc] so the first iteration finds it as z_accel0 (fasta/__init__.py:154-157)
  bool fused_done = false;
  for (int k = 0; k < ns; ++k) { fh_ctx* s = shard_of(c, k); s->lazy = false; s->commits = 0; s->tvz_pending = false; s->zcur_stale = false; }
  if (plain_pair_fused_ok(c)) FH_TRY(plain_pair_fused(c, c->X[c->xi], c->Z[c->zc], c->G[c->gc], &fused_done));   // one pass: z, f, gradient
  if (!fused_done) {
    for (int k = 0; k < ns; ++k) {
      fh_ctx* s = shard_of(c, k);
      FH_TRY(use_device(s));
      FH_TRY(op_fwd(s, 1, 0.0, s->X[s->xi], nullptr, nullptr, nullptr, nullptr, s->Z[s->zc], 1));
    }
    FH_TRY(reduce_fsq_over_ranks(c));
    for (int k = 0; k < ns; ++k) {
      fh_ctx* s = shard_of(c, k);
      if (s->op == OP_STENCIL) { s->zcur = s->Z[s->zc]; continue; }     // g0 = grad(zcur - b) is recomputed inside the step kernels
      FH_TRY(use_device(s));
      AdjIO io = {s->Z[s->zc], nullptr, 1, 0, 0.0, 1, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr, s->G[s->gc]};
      FH_TRY(adj_local(s, io));
    }
    if (shard_of(c, 0)->op == OP_DENSE) FH_TRY(adj_sum(c, [](fh_ctx* s) { return s->G[s->gc]; }));
  }
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    FH_TRY(use_device(s));
    double* x0 = s->X[s->xi];
    // x_accel1 := x0, best := x0 ; g(x0) terms for objective_hist[0] (:143) come from the host wrapper via FH_VEC ops
    HIP_TRY(hipMemcpyAsync(s->P[s->pc], x0, s->nv * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    s->bi = s->xi;                         // best iterate := x0 (fasta/__init__.py:167), by reference
    for (int q = 0; q < 3; ++q) if (q != s->xi) { s->ti = q; break; }
    s->last_accel = false;
    FH_TRY(launch_gterms(s, x0));
  }
  return fetch_scalars(c, scalars);
}

extern "C" int fh_gradient_at(fh_ctx* c, int src_vec, int dst_vec) {
  FH_TRY(check_ready(c, true));
  const int ns = nshards(c);
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    uint64_t l1 = 0, l2 = 0;
    double* src = vec_ptr(s, src_vec, &l1);
    double* dst = vec_ptr(s, dst_vec, &l2);
    if (!src || !dst || l1 != s->n || l2 != s->n) return fail(FH_E_ARG, "fh_gradient_at needs two n-length vectors");
    if (plain_pair_fused_ok(s)) {
      bool ok = false;
      FH_TRY(plain_pair_fused(s, src, s->zt, dst, &ok));
      if (ok) return 0;
    }
    FH_TRY(use_device(s));
    FH_TRY(op_fwd(s, 1, 0.0, src, nullptr, nullptr, nullptr, nullptr, s->zt, 1));
    AdjIO io = {s->zt, nullptr, 1, 0, 0.0, 1, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr, dst};
    FH_TRY(adj_local(s, io));
  }
  FH_TRY(adj_sum(c, [dst_vec](fh_ctx* s) { uint64_t l = 0; return vec_ptr(s, dst_vec, &l); }));
  return finish(c);
}

extern "C" int fh_diff_norm(fh_ctx* c, int vec_a, int vec_b, double* out) {
  FH_TRY(check_ready(c, false));
  if (!out) return fail(FH_E_ARG, "null out");
  if (!c->shards.empty()) {                        // n-side vectors are replicated: any shard has the answer
    if (m_side(vec_a) || m_side(vec_b)) return fail(FH_E_ARG, "fh_diff_norm on a multi-device context takes n-side vectors");
    return fh_diff_norm(c->shards[0], vec_a, vec_b, out);
  }
  uint64_t l1 = 0, l2 = 0;
  double* a = vec_ptr(c, vec_a, &l1);
  double* b = vec_ptr(c, vec_b, &l2);
  if (!a || !b || l1 != l2) return fail(FH_E_ARG, "fh_diff_norm needs two vectors of equal length");
  const unsigned grid = (unsigned)std::min<uint64_t>((l1 + FH_WG - 1) / FH_WG, 1024);
  FH_TRY(ensure_ws(c, grid * sizeof(double)));
  k_diff_sq<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(a, b, (uint32_t)l1, c->ws, c->counters + CNT_AUX, c->dscal + FH_NSCALARS + 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->hscal + FH_NSCALARS + 1, c->dscal + FH_NSCALARS + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  FH_TRY(finish(c));
  *out = sqrt(c->hscal[FH_NSCALARS + 1]);
  return 0;
}

extern "C" int fh_fwd(fh_ctx* c, double tau, double* scalars) {
  FH_TRY(check_ready(c, true));
  for (int k = 0; k < nshards(c); ++k) FH_TRY(solver_fwd_local(shard_of(c, k), tau, "fh_fwd"));
  FH_TRY(reduce_fsq_over_ranks(c));
  return fetch_scalars(c, scalars);
}

// K-adj of the solver state on every row block, one exchange, the n-side epilogue on every row block
static int solver_adj(fh_ctx* c, double tau, int accel, double coef, const char* who) {
  const int ns = nshards(c);
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    FH_TRY(use_device(s));
    FH_TRY(not_lazy(s, who));
    s->last_accel = accel != 0;
    FH_TRY(adj_local(s, solver_adj_io(s, tau, accel, coef)));
  }
  FH_TRY(adj_sum(c, [](fh_ctx* s) { return s->G[s->gc ^ 1]; }));
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    if (!row_sharded(s)) continue;
    FH_TRY(use_device(s));
    FH_TRY(adj_tail(s, solver_adj_io(s, tau, accel, coef)));
  }
  return 0;
}

extern "C" int fh_adj(fh_ctx* c, double tau, int accel, double coef, double* scalars) {
  FH_TRY(check_ready(c, true));
  FH_TRY(solver_adj(c, tau, accel, coef, "fh_adj"));
  return fetch_scalars(c, scalars);
}

// K-fwd and K-adj of the same tau enqueued back to back, ONE synchronisation (no acceleration): for callers that
// speculate on the step being accepted when the launches are short and the host round trip is what costs
// (fasta/__init__.py:181-188 and :248-260 in one call; a rejected step has wasted the K-adj launch).
extern "C" int fh_fwd_adj(fh_ctx* c, double tau, double* scalars) {
  FH_TRY(check_ready(c, true));
  for (int k = 0; k < nshards(c); ++k) FH_TRY(solver_fwd_local(shard_of(c, k), tau, "fh_fwd_adj"));
  FH_TRY(reduce_fsq_over_ranks(c));
  FH_TRY(solver_adj(c, tau, 0, 0.0, "fh_fwd_adj"));
  return fetch_scalars(c, scalars);
}

extern "C" int fh_fused_supported(fh_ctx* c, int* yes) {
  if (!c || !yes) return fail(FH_E_ARG, "null argument");
  if (!c->shards.empty()) {          // shell: what every shard supports (row blocks may differ by one row; the kinds rarely differ)
    int all = -1;
    for (fh_ctx* s : c->shards) {
      int one = 0;
      FH_TRY(fh_fused_supported(s, &one));
      all = all < 0 ? one : (all == one ? all : ((all && one) ? 3 : 0));
    }
    *yes = all;
    return 0;
  }
  // 0 = unsupported; 1 = dense one-pass kernel, recommended;
  // 2 = stencil one-pass kernel (costs no more than K-fwd alone: it simply replaces both launches);
  // 3 = dense one-pass kernel available but NOT recommended: its launch has ~35-50 us of fixed cost (slot fill, n-side
  //     prologue, grid barrier, epilogue), which two short launches under one sync beat on a small matrix
  //     (profiles/r02_fused_crossover.txt: 512 x 1024 35.6 vs 34.9 us, 2048^2 46 vs 53 us, 1024 x 8192 60 vs 63 us, 4096^2 77 vs 76 us)
  int ppt = c->op == OP_DENSE ? fused_ppt(c) : 0;
  // The dense one-pass kernel needs its whole grid (one workgroup per CU the device REPORTS) co-resident.  A CU mask hides
  // CUs from the dispatcher without changing that count: say "unsupported" up front instead of running into the bounded-spin
  // timeout on the first launch (the timeout stays as the safety net for partition modes this check cannot see).
  if (ppt && (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK"))) ppt = 0;
  *yes = c->op == OP_STENCIL ? (row_sharded(c) ? 0 : 2) : (ppt ? (fused_pays(c) ? 1 : 3) : 0);
  return 0;
}

static int launch_fused_tv(fh_ctx* c, double tau) {
  if (c->prox_kind != FH_PROX_TVBALL && c->prox_kind != FH_PROX_IDENTITY)
    return fail(FH_E_STATE, "the stencil operator supports the TV-ball prox or no prox (got kind %d)", c->prox_kind);
  if (!c->zcur) return fail(FH_E_STATE, "fh_step on the stencil operator before fh_init");
  TvStepFwdP p;
  p.H = (uint32_t)c->H; p.W = (uint32_t)c->W;
  p.rows_wg = (uint32_t)(c->tv_rows > 0 ? c->tv_rows : 32);
  p.strip_groups = ((p.W + TVF_OWN - 1) / TVF_OWN + 3) / 4;
  p.x0 = c->X[c->xi]; p.xacc0 = nullptr; p.xp = c->P[c->pc ^ 1]; p.zc = c->zcur; p.b = c->b; p.zn = c->Z[c->zc ^ 1];
  p.tau = tau;
  const unsigned grid = p.strip_groups * ((p.H + p.rows_wg - 1) / p.rows_wg);
  FH_TRY(ensure_ws(c, (size_t)grid * 16 * sizeof(double)));
  p.red = c->ws; p.counter = c->counters + CNT_FWD; p.out = scalar_out(c);
  t_begin(c, FH_K_FUSED);
#define TV_FUSED(U, NT)                                                                                            \
  do {                                                                                                             \
    if (c->prox_kind == FH_PROX_TVBALL) k_fused_tv_step<0, U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);   \
    else k_fused_tv_step<1, U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);                                  \
  } while (0)
  if (c->tv_nt == 1) { if (c->tv_u == 2) TV_FUSED(2, 1); else if (c->tv_u == 4) TV_FUSED(4, 1); else TV_FUSED(8, 1); }
  else { if (c->tv_u == 2) TV_FUSED(2, 0); else if (c->tv_u == 4) TV_FUSED(4, 0); else TV_FUSED(8, 0); }
#undef TV_FUSED
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_fused_tv_accel(fh_ctx* c, double tau, double coef, int restart) {
  if (c->prox_kind != FH_PROX_TVBALL && c->prox_kind != FH_PROX_IDENTITY)
    return fail(FH_E_STATE, "the stencil operator supports the TV-ball prox or no prox (got kind %d)", c->prox_kind);
  TvAccelP p;
  p.H = (uint32_t)c->H; p.W = (uint32_t)c->W;
  p.rows_wg = (uint32_t)(c->tv_rows > 0 ? c->tv_rows : 32);
  p.strip_groups = ((p.W + TVF_OWN - 1) / TVF_OWN + 3) / 4;
  p.p1 = nq(c, c->lq1); p.p0 = nq(c, c->lq0); p.pn = nq(c, c->lqn);
  p.z1 = mq(c, c->lz1); p.z0 = mq(c, c->lz0); p.zn = mq(c, c->lzn);
  p.b = c->b; p.tau = tau; p.cprev = c->lc; p.coef = coef; p.restart = restart;
  const unsigned grid = p.strip_groups * ((p.H + p.rows_wg - 1) / p.rows_wg);
  FH_TRY(ensure_ws(c, (size_t)grid * 16 * sizeof(double)));
  p.red = c->ws; p.counter = c->counters + CNT_FWD; p.out = scalar_out(c);
  t_begin(c, FH_K_FUSED);
#define TV_ACCEL(U, NT)                                                                                             \
  do {                                                                                                              \
    if (c->prox_kind == FH_PROX_TVBALL) k_fused_tv_accel<0, U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);   \
    else k_fused_tv_accel<1, U, NT><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);                                  \
  } while (0)
  if (c->tv_nt == 1) { if (c->tv_u == 2) TV_ACCEL(2, 1); else if (c->tv_u == 4) TV_ACCEL(4, 1); else TV_ACCEL(8, 1); }
  else { if (c->tv_u == 2) TV_ACCEL(2, 0); else if (c->tv_u == 4) TV_ACCEL(4, 0); else TV_ACCEL(8, 0); }
#undef TV_ACCEL
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  return 0;
}

// z-free one-pass stencil step (k_tv_onepass): accel = 0 -> x0 = X[xi]; accel = 1 -> the lazily-kept (P1, P0, c) state
static int launch_tv_onepass(fh_ctx* c, double tau, int accel, double coef, int restart) {
  if (c->prox_kind != FH_PROX_TVBALL && c->prox_kind != FH_PROX_IDENTITY)
    return fail(FH_E_STATE, "the stencil operator supports the TV-ball prox or no prox (got kind %d)", c->prox_kind);
  TvZP p;
  p.H = (uint32_t)c->H; p.W = (uint32_t)c->W;
  p.strip_groups = ((p.W + TVZ_OWN - 1) / TVZ_OWN + 3) / 4;
  if (c->tv_rows > 0) p.rows_wg = (uint32_t)c->tv_rows;
  else {
    // auto: as many row chunks as make the grid just FILL the resident capacity (5 workgroups per CU at 88 registers), so that all
    // workgroups run side by side and finish together -- 2240 workgroups of 128 rows on 1280 slots ran 1.75 rounds, the last one
    // three-quarters empty (8192^2: 128 rows 0.607 ms, 228-235 rows 0.588; profiles/r02_tune_tv.txt, r03_tune_tv.txt).  At least
    // 32 rows per chunk (rows + 4 are read and computed), at most the image.
    const uint32_t slots = (uint32_t)std::max(1, c->ncu) * 5u;
    const uint32_t chunks = std::max(1u, slots / p.strip_groups);
    p.rows_wg = std::min(p.H, std::max(32u, (p.H + chunks - 1) / chunks));
  }
  // rows per trip / rotating trip buffers: 2 rows, load-then-consume for the plain sweep; 4 rows x 3 rotating buffers with FISTA
  // (two streams to read): profiles/r03_tune_tv.txt.  Every combination produces the same bits (scripts/tune_tvz.py).
  const int tvu = c->tv_u ? c->tv_u : (accel ? 4 : 2);
  if (accel) { p.p1 = nq(c, c->lq1); p.p0 = nq(c, c->lq0); p.pn = nq(c, c->lqn); p.cprev = c->lc; }
  else { p.p1 = c->X[c->xi]; p.p0 = c->X[c->xi]; p.pn = c->P[c->pc ^ 1]; p.cprev = 0.0; }
  p.b = c->b; p.tau = tau; p.coef = coef; p.restart = restart;
  const unsigned grid = p.strip_groups * ((p.H + p.rows_wg - 1) / p.rows_wg);
  FH_TRY(ensure_ws(c, (size_t)grid * 16 * sizeof(double)));
  p.red = c->ws; p.counter = c->counters + CNT_FWD; p.out = scalar_out(c);
  t_begin(c, FH_K_FUSED);
  // tunables -> template parameters.  Non-temporal LOADS lose 12 % here (the halo columns and rows are re-read by the neighbouring
  // waves and workgroups through L2), so this sweep only distinguishes non-temporal (default) and plain STORES (FH_TUNE_TV_NT = 3).
  const int nb = c->tv_pipe ? c->tv_pipe : (accel ? 3 : 1);
  const bool nts = c->tv_nt != 3;       // stores are non-temporal unless FH_TUNE_TV_NT = 3 asks for plain ones (+2-3 %: xprox is not re-read by this launch)
  const bool ident = c->prox_kind != FH_PROX_TVBALL;
#define TVZ(ID, AC, U, NT, NB) k_tv_onepass<ID, AC, U, NT, NB><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p)
#define TVZ_NB(AC, U, NT) do { if (nb >= 2) TVZ(0, AC, U, NT, 3); else TVZ(0, AC, U, NT, 1); } while (0)
#define TVZ_U(AC, NT) do { if (tvu <= 2) TVZ_NB(AC, 2, NT); else if (tvu == 8) TVZ_NB(AC, 8, NT); else TVZ_NB(AC, 4, NT); } while (0)
#define TVZ_NT(AC) do { if (nts) TVZ_U(AC, 2); else TVZ_U(AC, 0); } while (0)
  if (ident) {        // no prox (g = None): the round-2 burst form
    if (accel) { if (tvu == 2) TVZ(1, 1, 2, 0, 1); else if (tvu == 8) TVZ(1, 1, 8, 0, 1); else TVZ(1, 1, 4, 0, 1); }
    else { if (tvu == 2) TVZ(1, 0, 2, 0, 1); else if (tvu == 8) TVZ(1, 0, 8, 0, 1); else TVZ(1, 0, 4, 0, 1); }
  } else if (accel) TVZ_NT(1);
  else TVZ_NT(0);
#undef TVZ_NB
#undef TVZ_NT
#undef TVZ_U
#undef TVZ
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  c->tvz_pending = true;
  return 0;
}

// One-pass iteration of the dense operator on every row block (`accel` = 0: fh_step; 1: fh_step_accel): local launch on every
// shard -> ONE sum over the row blocks of g1 with the local loss sums and the timeout word appended (every shard / rank then sees
// the same verdict, so all of them drop to the two-launch path together if a hand-off ever times out) -> the n-side epilogue on
// every shard -> one host synchronisation.  A plain single-GPU context is the one-shard case without the exchange.
static int dense_step(fh_ctx* c, double tau, int accel, double coef, int restart, double* scalars) {
  const int ns = nshards(c);
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    FH_TRY(use_device(s));
    if (s->prox_kind == FH_PROX_LINF || s->prox_kind == FH_PROX_L1BALL) FH_TRY(launch_level_search(s, tau));
    const bool sharded = row_sharded(s);
    double* g1 = s->G[s->gc ^ 1];
    FusedIO fio = {s->X[s->xi], s->G[s->gc], s->xhat, s->P[s->pc ^ 1], s->Z[s->zc ^ 1], g1, s->prox_kind, sharded ? 2 : 0};
    if (accel) {
      fio.accel = 1; fio.restart = restart ? 1 : 0; fio.coef = coef;
      fio.xacc0 = s->P[s->pc]; fio.zacc0 = s->Z[s->zc]; fio.x1 = s->X[s->ti];
      fio.coef_out = sharded ? s->dscal + FH_NSCALARS + 3 : nullptr;     // the applied coefficient, for the separate epilogue
    }
    if (sharded) fio.pack = g1 + s->nv;          // slack behind every n-side vector (alloc_vectors)
    FH_TRY(launch_fused_dense(s, tau, fio));
    s->last_accel = accel != 0;
  }
  FH_TRY(sum_over_shards(c, [](fh_ctx* s) { return s->G[s->gc ^ 1]; }, (size_t)shard_of(c, 0)->nv + 3));
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    if (!row_sharded(s)) continue;
    FH_TRY(use_device(s));
    AdjIO io = solver_adj_io(s, tau, accel, coef);
    io.z = nullptr; io.zacc0 = nullptr;
    double* pack = io.g1 + s->nv;
    FH_TRY(bb_epilogue_only(s, io, accel ? pack + 2 : pack, accel ? s->dscal + FH_NSCALARS + 3 : nullptr, pack));
  }
  FH_TRY(fetch_scalars(c, scalars));
  for (int k = 0; k < ns; ++k) fused_after(shard_of(c, k));
  return 0;
}

// One-pass FBS iteration: K-fwd and K-adj of the same tau from a single read of A (no acceleration).
// Writes the complete FH_S_* block; scalars[15] != 0 reports a spin timeout (results invalid: use the two-launch path).
extern "C" int fh_step(fh_ctx* c, double tau, double* scalars) {
  FH_TRY(check_ready(c, true));
  for (int k = 0; k < nshards(c); ++k) FH_TRY(not_lazy(shard_of(c, k), "fh_step"));
  if (c->op == OP_STENCIL) {
    if (row_sharded(c)) return fail(FH_E_STATE, "row sharding is implemented for the dense operator only");
    if (c->tv_zfree) {
      if (!c->zcur) return fail(FH_E_STATE, "fh_step on the stencil operator before fh_init");
      FH_TRY(launch_tv_onepass(c, tau, 0, 0.0, 0));
    } else {
      FH_TRY(tv_refresh_zcur(c));
      c->tvz_pending = false;
      FH_TRY(launch_fused_tv(c, tau));
    }
    c->last_accel = false;
    return fetch_scalars(c, scalars);
  }
  return dense_step(c, tau, 0, 0.0, 0, scalars);
}

// One-pass iteration WITH acceleration (fasta/__init__.py:220-248): the launch computes this step's restart dot before
// its first row, applies `coef` unless (restart != 0 and the dot > 1e-30, :231) and reports the dot in FH_S_RDOT so that
// the caller can update alpha the same way.  Dense operator.  Row-sharded runs work the same way: x, xprox and x_accel0
// are replicated, so every rank computes the same dot and takes the same restart decision; the applied coefficient
// travels to the separate n-side epilogue through a device scalar.
extern "C" int fh_step_accel(fh_ctx* c, double tau, double coef, int restart, double* scalars) {
  FH_TRY(check_ready(c, true));
  if (c->op == OP_STENCIL) {
    if (row_sharded(c)) return fail(FH_E_STATE, "row sharding is implemented for the dense operator only");
    if (!c->lazy) {      // first accelerated one-pass step after fh_init: x0 = X[xi] (c = 0), z(x0) = Z[zc], best = x0
      if (!c->zcur) return fail(FH_E_STATE, "fh_step_accel on the stencil operator before fh_init");
      if (c->commits) return fail(FH_E_STATE, "fh_step_accel on the stencil operator must drive the solve from the first iteration after fh_init");
      c->lazy = true;
      c->lq1 = c->lq0 = c->lb1 = c->lb0 = c->xi;
      c->lz1 = c->lz0 = c->zc;
      c->lc = c->lbc = c->lc_pending = 0.0;
      lazy_pick_targets(c);
    }
    if (c->tv_zfree) FH_TRY(launch_tv_onepass(c, tau, 1, coef, restart ? 1 : 0));
    else FH_TRY(launch_fused_tv_accel(c, tau, coef, restart ? 1 : 0));
    c->last_accel = true;
    FH_TRY(fetch_scalars(c, scalars));
    c->lc_pending = (restart && c->hscal[FH_S_RDOT] > 1E-30) ? 0.0 : coef;      // what the launch applied (:231); adopted by fh_commit
    return 0;
  }
  return dense_step(c, tau, 1, coef, restart, scalars);
}

extern "C" int fh_commit(fh_ctx* c, int save_best) {
  FH_TRY(check_ready(c, false));
  if (!c->shards.empty()) {          // shell: the same rotation on every shard (pointer bookkeeping only)
    for (fh_ctx* s : c->shards) FH_TRY(fh_commit(s, save_best));
    return 0;
  }
  c->commits += 1;
  if (c->op == OP_STENCIL && c->tvz_pending && !c->lazy) c->zcur_stale = true;     // the adopted x0 has no stored image
  c->tvz_pending = false;
  if (c->lazy) {                                    // rotate the (P1, P0, c) / (Z1, Z0, c) state; nothing is copied
    c->lq0 = c->lq1; c->lq1 = c->lqn;
    c->lz0 = c->lz1; c->lz1 = c->lzn;
    c->lc = c->lc_pending;
    if (save_best) { c->lb1 = c->lq1; c->lb0 = c->lq0; c->lbc = c->lc; }
    lazy_pick_targets(c);
    return 0;
  }
  if (c->last_accel) {
    c->pc ^= 1;                                   // x_accel0 <- this iteration's prox output (P ping-pong)
  } else {
    // x1 is the prox output itself: adopt its buffer into the pool slot ti; the slot's old buffer becomes the
    // next prox target
    std::swap(c->X[c->ti], c->P[c->pc ^ 1]);
  }
  c->xi = c->ti;                                  // x0 <- x1
  if (save_best) c->bi = c->xi;                   // best iterate by reference (fasta/__init__.py:298-300)
  for (int k = 0; k < 3; ++k) if (k != c->xi && k != c->bi) { c->ti = k; break; }   // a slot that is neither x0 nor best
  c->zc ^= 1;     // z_accel0 <- z1
  c->gc ^= 1;     // g0 <- g1
  if (c->op == OP_STENCIL) {
    if (c->last_accel) { c->zxc ^= 1; c->zcur = c->ZX[c->zxc]; }   // residual source = extrapolated z'
    else c->zcur = c->Z[c->zc];                                      // residual source = z1 itself
  }
  return 0;
}

extern "C" int fh_apply(fh_ctx* c, int adjoint, const double* in, double* out) {
  FH_TRY(check_ready(c, false));
  if (!in || !out) return fail(FH_E_ARG, "null argument");
  const int ns = nshards(c);
  const bool shell = !c->shards.empty();
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    const uint64_t r0 = shell ? c->shard_row0[k] : 0;       // rows of the whole operator this shard holds: [r0, r0 + s->m)
    FH_TRY(use_device(s));
    if (!adjoint) {
      HIP_TRY(hipMemcpyAsync(s->T[3], in, s->n * sizeof(double), hipMemcpyHostToDevice, s->stream));
      FH_TRY(op_fwd(s, 1, 0.0, s->T[3], nullptr, nullptr, nullptr, nullptr, s->zt, 0));
      HIP_TRY(hipMemcpyAsync(out + r0, s->zt, s->m * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(s->zt, in + r0, s->m * sizeof(double), hipMemcpyHostToDevice, s->stream));
      AdjIO io = {s->zt, nullptr, 0, 0, 0.0, 1, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr, s->T[3]};
      FH_TRY(adj_local(s, io));
    }
  }
  if (adjoint) {
    FH_TRY(adj_sum(c, [](fh_ctx* s) { return s->T[3]; }));
    fh_ctx* s0 = shard_of(c, 0);
    FH_TRY(use_device(s0));
    HIP_TRY(hipMemcpyAsync(out, s0->T[3], s0->n * sizeof(double), hipMemcpyDeviceToHost, s0->stream));
  }
  return finish(c);
}

// ------------------------------------------------------------------------------------------------
// row sharding
// ------------------------------------------------------------------------------------------------
extern "C" int fh_comm_unique_id(void* id128) {
  if (!id128) return fail(FH_E_ARG, "null id buffer");
  FH_TRY(rccl_load());
  fh_nccl_uid id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int fh_comm_init(fh_ctx* c, int nranks, int rank, const void* id128) {
  if (!c || !id128) return fail(FH_E_ARG, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(FH_E_ARG, "bad rank %d of %d", rank, nranks);
  if (!c->shards.empty() || c->owner)
    return fail(FH_E_STATE, "fh_comm_init: a multi-device context (fh_create_ex, ndev > 1) already shards the rows in-process");
  FH_TRY(rccl_load());
  FH_TRY(use_device(c));
  if (c->comm) FH_TRY(fh_comm_destroy(c));
  fh_nccl_uid id;
  memcpy(&id, id128, sizeof(id));
  NCCL_TRY(g_rccl.CommInitRank(&c->comm, nranks, id, rank));
  c->nranks = nranks; c->rank = rank;
  return 0;
}

extern "C" int fh_comm_count(fh_ctx* c, int* nranks) {
  if (!c || !nranks) return fail(FH_E_ARG, "null argument");
  *nranks = 1;                       // no communicator: a single-GPU context
  if (!c->shards.empty()) {          // shell: what RCCL reports for shard 0's communicator; shards on one device: their number
    if (c->emulated) { *nranks = (int)c->shards.size(); return 0; }
    c = c->shards[0];
  }
  if (c->comm) NCCL_TRY(g_rccl.CommCount(c->comm, nranks));   // what RCCL itself reports, not what the caller asked for
  return 0;
}

extern "C" int fh_comm_destroy(fh_ctx* c) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (!c->shards.empty() || c->owner) return fail(FH_E_STATE, "fh_comm_destroy: the communicators of a multi-device context live as long as it does (fh_destroy)");
  if (c->comm) {
    (void)hipStreamSynchronize(c->stream);
    NCCL_TRY(g_rccl.CommDestroy(c->comm));
    c->comm = nullptr; c->nranks = 1; c->rank = 0;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------
extern "C" int fh_timing_enable(fh_ctx* c, int on) {
  if (!c) return fail(FH_E_ARG, "null context");
  for (fh_ctx* s : c->shards) s->timing = on != 0;
  c->timing = on != 0;
  return 0;
}
// a multi-device context reports the SUM over its shards of each kernel's time and launches (shards on one device run one after
// the other; on separate devices the per-launch average total_ms / launches is the mean over the devices)
extern "C" int fh_timing_get(fh_ctx* c, int k, double* total_ms, uint64_t* launches) {
  if (!c || k < 0 || k >= FH_NKERNELS) return fail(FH_E_ARG, "bad kernel id");
  double ms = c->tot_ms[k];
  uint64_t cnt = c->launches[k];
  for (fh_ctx* s : c->shards) { ms += s->tot_ms[k]; cnt += s->launches[k]; }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = cnt;
  return 0;
}
extern "C" int fh_timing_reset(fh_ctx* c) {
  if (!c) return fail(FH_E_ARG, "null context");
  for (fh_ctx* s : c->shards) FH_TRY(fh_timing_reset(s));
  for (int k = 0; k < FH_NKERNELS; ++k) { c->tot_ms[k] = 0; c->launches[k] = 0; c->ev_pending[k] = false; }
  return 0;
}

// number of row blocks of a context (1 for a plain one) and borrowed access to one of them -- the shard is a complete context
// (diagnostics and tests read its replicated vectors with fh_get_vector; it stays owned by the multi-device context)
extern "C" int fh_shard_count(fh_ctx* c, int* count) {
  if (!c || !count) return fail(FH_E_ARG, "null argument");
  *count = nshards(c);
  return 0;
}
extern "C" int fh_shard(fh_ctx* c, int k, fh_ctx** shard, uint64_t* row0, uint64_t* rows) {
  if (!c || !shard) return fail(FH_E_ARG, "null argument");
  if (k < 0 || k >= nshards(c)) return fail(FH_E_ARG, "shard %d out of range (have %d)", k, nshards(c));
  *shard = shard_of(c, k);
  const bool laid_out = !c->shards.empty() && c->shard_row0.size() == c->shards.size() + 1;
  if (row0) *row0 = laid_out ? c->shard_row0[(size_t)k] : 0;
  if (rows) *rows = laid_out ? shard_rows(c, k) : c->m;
  return 0;
}

extern "C" int fh_stream_read_ms(fh_ctx* c, int reps, double* ms_per_pass, uint64_t* bytes_per_pass) {
  FH_TRY(check_ready(c, false));
  if (!c->shards.empty()) return fh_stream_read_ms(c->shards[0], reps, ms_per_pass, bytes_per_pass);     // shard 0's block on its device
  if (c->op != OP_DENSE) return fail(FH_E_STATE, "stream-read ceiling needs a dense matrix");
  if (reps < 1) reps = 1;
  // k_stream_probe<16,1> as described in include/fasta_hip.h: persistent workgroups, 1 per CU by default, three rotating buffers of 16 nt loads per lane
  const uint64_t npieces = c->mp * (c->ld / (c->f32 ? 4 : 2));
  const unsigned grid = (unsigned)(c->fwd_cap > 0 ? c->fwd_cap : (c->ncu > 0 ? c->ncu : 256));
  double* sink = c->dscal + FH_NSCALARS + 2;
  k_stream_probe<16, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(c->A, npieces, sink);   // warm-up
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, c->stream));
  for (int i = 0; i < reps; ++i) k_stream_probe<16, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(c->A, npieces, sink);
  HIP_TRY(hipEventRecord(e1, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_per_pass) *ms_per_pass = ms / reps;
  if (bytes_per_pass) *bytes_per_pass = npieces * 16;
  return 0;
}
