// fh_host_ctx.h -- host-side state of libfasta_hip.so: error reporting, RCCL through dlopen, the context (fh_ctx: one device,
// or a shell over one shard per device), buffer management, HIP-event timing, the scalar block's way back to the host and the
// sums over row blocks.  Included by fasta_hip.hip only (one translation unit); the C ABI itself is in fasta_hip.hip.
#pragma once
#include <atomic>
#include <chrono>
#include <thread>
#include <chrono>

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
  int (*GetVersion)(int*) = nullptr;          // optional (diagnostics only)
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static const int kNcclFloat64 = 8;   // ncclDouble
static const int kNcclSum = 0;       // ncclSum

static char g_rccl_path[512] = "";     // what rccl_load() opened (fh_comm_library)
static int rccl_load() {
  if (g_rccl.lib) return 0;
  // FASTA_RCCL_LIB names another library with RCCL's entry points (a different RCCL build; the tests' multi-process stand-in).
  // A substitution is never silent: one line on stderr, and fh_comm_library() reports the path that was opened.
  const char* sub = getenv("FASTA_RCCL_LIB");
  const char* names[] = {sub, "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.lib) { snprintf(g_rccl_path, sizeof g_rccl_path, "%s", nm); break; }
    if (nm == sub) return fail(FH_E_RCCL, "FASTA_RCCL_LIB=%s cannot be opened: %s (unset it to use the system's RCCL)", nm, dlerror());
  }
  if (!g_rccl.lib) return fail(FH_E_RCCL, "cannot dlopen librccl: %s", dlerror());
  if (sub && *sub) fprintf(stderr, "libfasta_hip: collectives come from FASTA_RCCL_LIB=%s, not from the system's RCCL\n", sub);
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
  *(void**)(&g_rccl.GetVersion) = dlsym(g_rccl.lib, "ncclGetVersion");
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
  size_t a_block_bytes = 0;  // size of the block A lives in (>= the matrix: a kept block may be up to twice as large, see acquire_matrix_block)
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
  unsigned* counters = nullptr;      // kCounterWords (8192) words, zeroed at creation; kernels leave them zero
  double* dscal = nullptr;           // FH_NSCALARS + 16 doubles on device
  double* hscal = nullptr;           // pinned, device-mapped host block: single-GPU launches write their scalars here
  double* hscal_dev = nullptr;       // device-side address of hscal
  bool scal_mirrored = false;        // row-sharded runs: the last launch already copied the scalar block into hscal
  // tuning
  int fwd_rows = 0;          // 0 = auto
  long long fwd_cap = 0;     // 0 = auto (4 workgroups per CU, grid-stride over row groups)
  int adj_slab = 0;          // 0 = auto
  int adj_cpt = 0;           // 0 = auto
  int adj_cyclic = 0;        // 0 = auto = 2 = contiguous slabs, 1 = rows dealt cyclically to the slabs (A/B)
  int ld_pad = 0;
  int nt_loads = -1;         // K-fwd / K-adj stream A with non-temporal loads: 1 / 0, -1 = auto (nt_for below: plain loads for a matrix of at most 256 MiB)
  // stencil defaults measured on MI355X at 8192^2 (profiles/r01_tune_tv.txt): plain (not nt) accesses,
  // 8 rows in flight, 32 rows per workgroup for K-fwd and 128 for the read-only K-adj
  int tv_u = 0;              // 0 = auto: 8 for the kernels that stream z, 2 / 4 for the z-free one-pass sweeps (profiles/r02_tune_tv.txt)
  int tv_rows = 0;           // 0 = auto (32 fwd / 128 adj)
  int tv_nt = 0;
  int tv_lds_pad = 0;        // FH_TUNE_TV_LDS_PAD
  int tv_xcd = 0;            // FH_TUNE_TV_XCD: workgroup ids of the one-pass sweep dealt out XCD by XCD (0 / 1 = on, 2 = off)
  int tv_slots = 0;          // FH_TUNE_TV_SLOTS: persistent one-pass sweep, workgroups per CU (0 = one workgroup per chunk)
  int tv_ring = 0;           // FH_TUNE_TV_RING: LDS-DMA trip ring of the one-pass sweep (0 = auto, 1 = off, 2 / 3 = slots per wave)
  int tv_pipe = 0;           // FH_TUNE_TV_PIPE: rotating trip buffers of the one-pass sweep (0 = auto, 1 = burst, 2, 3)
  bool fused_variant_auto = true;   // cleared by FH_TUNE_FUSED_VARIANT: the caller's word is taken as it is (fused_variant_for)
  int fused_variant = 2 | 32;   // 2: team members 32 blocks apart (one XCD), best in profiles/r01b_tune_fused.txt; 32 (round 6): rows dealt cyclically to the teams -- the
                             // whole grid streams ONE window of nteams x (rows in flight) consecutive rows instead of nteams windows spread over the matrix: 1-2 % faster on a
                             // well-placed matrix (65536^2: 4.76 vs 4.84-4.88 ms), 9 % at 20000 x 30000, and INSENSITIVE to where the allocator put the matrix (blocked: 5.2-5.45 ms
                             // for a 32 GiB matrix that got the far part of the device's memory, cyclic 4.79-4.81; profiles/r06_placement.txt)
  int test_hooks = 0;        // FH_TUNE_TEST_HOOKS (csrc/fh_experimental.h): fault injection, set by the test-suite only
  int fused_min_rows = 16;   // use fewer teams when m is small: at least this many rows per team (scripts/probes/fused_small_m.py)
  // one-pass kernel hand-off slots: two arrays alternate between launches, each launch re-arms the other one in passing;
  // the host fills both with the sentinel only when this signature (workspace, layout) changes or a launch timed out
  double* slotbuf = nullptr;     // dedicated allocation: the shared workspace `ws` is scribbled over by every other kernel
  size_t slotbuf_bytes = 0;
  uint64_t slots_sig = 0;
  int slots_parity = 0;
  // timing
  bool timing = false;
  // two event pairs per kernel id, used alternately: a pair is read (hipEventElapsedTime) when it is about to be re-recorded, i.e. two launches
  // after its own -- long complete, so the harvest never waits even when the host did not synchronise the stream in between (seq_wait below)
  hipEvent_t ev[FH_NKERNELS][2][2];
  bool ev_pending[FH_NKERNELS][2] = {};
  int ev_cur[FH_NKERNELS] = {};
  // "the scalars are there" by sequence number (fh_device.h:publish_seq): the launcher of a kernel whose finaliser publishes one sets seq_wait
  // to the number it passed; collect_scalars then spins on the word behind the mapped scalar block instead of synchronising the stream
  unsigned seq = 0, seq_wait = 0;
  int seq_poll = 1;          // FH_TUNE_SEQ_POLL: 0 = always hipStreamSynchronize (A/B)
  // FH_K_HOST_ISSUE: host time of a one-pass dense step from its entry to the start of its final synchronisation
  bool timing_skip_kernels = false;
  double host_issue_ms = 0.0; uint64_t host_issue_calls = 0;
  std::chrono::steady_clock::time_point issue_t0; bool issue_open = false;
  double tot_ms[FH_NKERNELS] = {};
  uint64_t launches[FH_NKERNELS] = {};
  // comm
  fh_nccl_comm comm = nullptr;
  int nranks = 1, rank = 0;
  int ncu = 0;               // compute units of the device (fused one-pass kernel: one workgroup per CU)
  int coresident = -1;       // -1 = not probed yet; 1 / 0 = fused_ncu() workgroups can / cannot run side by side (co_resident())
  int run_max_n = 0;         // FH_TUNE_RUN_MAX_N: widest row fh_run takes (0 = kRunDefaultMaxN)
  int fused_cus = 0;         // FH_TUNE_FUSED_CUS: the one-pass dense kernel uses at most this many CUs (0 = all the device reports)
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
  // fh_run (csrc/fh_run.h): the host-mapped block the launch writes its final solver state to, and the host-mapped history block
  void* run_st_host = nullptr; void* run_st_host_dev = nullptr;
  double* run_hist = nullptr; double* run_hist_dev = nullptr; size_t run_hist_steps = 0;
  unsigned* gridbar = nullptr;       // 3 x GB_WORDS words: two-level grid barrier, two-level final arrival of the dense one-pass / set-up / persistent-loop kernels, final arrival of the stencil sweep (fh_device.h); zero between launches
  double* lvl_rec = nullptr;         // multi-workgroup level search: per-pass records and counters (allocated on first use, counters kept zero)
  unsigned* lvl_cnt = nullptr;
  double* selftest_buf = nullptr;    // fh_comm_selftest's scratch (freed before it returns)
  void* chain_state = nullptr;       // device block of the chained form of fh_run (csrc/fh_fused.h: ChainState)
  int run_chain_on = 0;              // FH_TUNE_RUN_CHAIN = 1: outside the persistent launch's window fh_run takes the chained form (opt-in: measured equal to the host-side loop)
  uint64_t run_timeouts = 0;         // persistent launches of fh_run that ended in a grid-barrier timeout (fh_recovered_count)
  bool pending_step = false;         // fh_step_begin has issued a step whose fh_step_end is still to come (every other entry point refuses)
  bool emulated = false;             // shell / shard: the device ids repeat (one device, one stream, k_sum_shards)
  bool owns_stream = true;           // false in shards 1.. of an emulated group (they run on shard 0's stream)
};

#define FH_MAX_SHARDS 64
static inline int nshards(fh_ctx* c) { return c->shards.empty() ? 1 : (int)c->shards.size(); }
static inline fh_ctx* shard_of(fh_ctx* c, int k) { return c->shards.empty() ? c : c->shards[k]; }
// a context whose launches leave the sums over rows to an exchange step: a rank of a multi-process run, or a shard of a shell
static inline bool row_sharded(const fh_ctx* c) { return c->comm != nullptr || c->owner != nullptr; }

static const int kCounterWords = 8192;
enum { CNT_FWD = 0, CNT_ADJ_FIN = 1, CNT_AUX = 2, CNT_FUSED_BAR = 4, CNT_FUSED_ERR = 8, CNT_PROBE = 12, CNT_RUN_BAR = 14, CNT_ADJ_CC = 16,
       CNT_DIAG = kCounterWords - 8 };     // the last 8 words only ever grow: [0] level searches that fell back to one workgroup, [1] ... that found no level (fh_recovered_count)

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

// ---- large frees and the allocations behind them -------------------------------------------------------------------------------------
// The driver clears freed device memory in the background, ~30 ms per GiB on MI355X.  Meanwhile a read-only stream over OTHER memory runs
// 0.7-3 % slow -- and a large allocation made while the clearing is still going on is mapped less favourably FOR ITS WHOLE LIFETIME.  What that costs
// depends on the kernel's access pattern (round 6, profiles/r06_placement.txt): with the rows dealt to the teams in contiguous blocks (rounds 1-5) the
// one-pass kernel ran 9-14 % slow on such a matrix (6 of 6 interleaved cycles: 5.49-5.54 against 4.85-4.89 ms per step, profiles/r06_alloc_settle.txt),
// and just as slow on a matrix that simply got the far part of the device's memory; with the rows dealt cyclically (the default now: the whole grid
// streams one window of the matrix) it does not care: 4.82-4.85 ms right behind a 128 GiB free, waited for or not (3 of 3).  What is left:
//   1. re-use (deterministic: no clearing, no new mapping, no sleep; worth ~1 % and the allocation's time): the matrix block (>= 1 GiB) a context gives up
//      is not returned to the driver but kept -- one block per device -- and handed to the next matrix on that device that fits it and fills at least
//      half of it; fh_release_cached() / fh_alloc_cache(0) return it to the driver.
//   2. settle (OFF by default since the cyclic dealing; fh_alloc_settle(1) for callers that run the blocked dealing or live on K-adj, which still
//      depend on the mapping): a matrix that cannot be served by 1. is allocated only after the device's earlier large frees have presumably been
//      cleared (35 ms per GiB behind the free); fh_alloc_settle_waited reports what it has cost so far.
#include <mutex>
#define FH_MAX_DEVICES 64
static std::atomic<long long> g_clear_until_ns[FH_MAX_DEVICES];       // steady clock, per device
static std::atomic<long long> g_settle_waited_ns{0};
static std::atomic<int> g_settle_on{0};
static std::atomic<int> g_cache_on{1};
static std::atomic<unsigned long long> g_cache_hits{0};
struct CachedBlock { double* p = nullptr; size_t bytes = 0; };
static std::mutex g_cache_mu;
static CachedBlock g_cache[FH_MAX_DEVICES];
static const size_t kSettleMinBytes = (size_t)1 << 30;
static const double kClearNsPerByte = 35.0e6 / (double)((size_t)1 << 30);     // 35 ms per GiB (measured: 2.9 s for 96 GiB)
static long long steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline int dev_slot(int device) { return device >= 0 && device < FH_MAX_DEVICES ? device : 0; }
static void note_large_free(int device, size_t bytes) {
  if (bytes < kSettleMinBytes) return;
  std::atomic<long long>& until = g_clear_until_ns[dev_slot(device)];
  const long long now = steady_ns(), add = (long long)(kClearNsPerByte * (double)bytes);
  long long cur = until.load();
  while (!until.compare_exchange_weak(cur, std::max(cur, now) + add)) { }
}
static void settle_before_large_alloc(int device, size_t bytes) {
  if (!g_settle_on.load() || bytes < kSettleMinBytes) return;
  const long long wait = g_clear_until_ns[dev_slot(device)].load() - steady_ns();
  if (wait <= 0) return;
  std::this_thread::sleep_for(std::chrono::nanoseconds(wait));
  g_settle_waited_ns.fetch_add(wait);
}
// a context gives up its matrix block: kept for the next matrix on this device (the block it displaces goes back to the driver), or freed
static void release_matrix_block(int device, double* p, size_t bytes) {
  if (!p) return;
  if (bytes >= kSettleMinBytes && g_cache_on.load()) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CachedBlock& cb = g_cache[dev_slot(device)];
    if (cb.p) { (void)hipFree(cb.p); note_large_free(device, cb.bytes); }
    cb.p = p; cb.bytes = bytes;
    return;
  }
  (void)hipFree(p);
  note_large_free(device, bytes);
}
// a block for a matrix of `bytes` on `device` (the current device): the kept one if it fits and is at most twice as large, else a fresh one
// behind the settle wait.  *got = the size of the block handed out.
static hipError_t acquire_matrix_block(int device, size_t bytes, double** out, size_t* got) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CachedBlock& cb = g_cache[dev_slot(device)];
    if (cb.p && cb.bytes >= bytes && cb.bytes / 2 <= bytes) {
      *out = cb.p; *got = cb.bytes;
      cb = CachedBlock();
      g_cache_hits.fetch_add(1);
      return hipSuccess;
    }
    if (cb.p && bytes >= kSettleMinBytes) {            // it cannot serve this matrix and would only stand in its way: back to the driver
      (void)hipFree(cb.p); note_large_free(device, cb.bytes);
      cb = CachedBlock();
    }
  }
  settle_before_large_alloc(device, bytes);
  *got = bytes;
  return hipMalloc((void**)out, bytes);
}
static void release_cached_blocks(int device /* -1 = all */) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (int d = 0; d < FH_MAX_DEVICES; ++d) {
    if ((device >= 0 && d != device) || !g_cache[d].p) continue;
    (void)hipSetDevice(d);
    (void)hipFree(g_cache[d].p); note_large_free(d, g_cache[d].bytes);
    g_cache[d] = CachedBlock();
  }
}

static void free_operator(fh_ctx* c) {
  for (fh_ctx* s : c->shards) { (void)hipSetDevice(s->device); free_operator(s); }
  auto fr = [](double*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  if (c->A && c->op == OP_DENSE) { release_matrix_block(c->device, c->A, c->a_block_bytes); c->A = nullptr; c->a_block_bytes = 0; }
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
// (timing_skip_kernels: a block of a same-device multi-block context that is not the sampled one -- fh_timing_enable)
static inline bool t_on(fh_ctx* c, int k) { return c->timing && !(c->timing_skip_kernels && k != FH_K_COMM); }
// read a recorded pair; wait = true: wait for its stop event first (only fh_timing_get and a pair reused before it completed need that)
static void t_harvest(fh_ctx* c, int k, int s, bool wait) {
  if (!c->ev_pending[k][s]) return;
  float ms = 0.f;
  hipError_t e = hipEventElapsedTime(&ms, c->ev[k][s][0], c->ev[k][s][1]);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    if (!wait) return;
    (void)hipEventSynchronize(c->ev[k][s][1]);
    e = hipEventElapsedTime(&ms, c->ev[k][s][0], c->ev[k][s][1]);
  }
  if (e == hipSuccess) { c->tot_ms[k] += ms; c->launches[k] += 1; }
  else (void)hipGetLastError();
  c->ev_pending[k][s] = false;
}
static inline void t_begin(fh_ctx* c, int k) {
  c->seq_wait = 0;                   // (every launcher passes here: only the LAST launch before collect_scalars may offer a sequence number)
  if (t_on(c, k)) {
    const int s = c->ev_cur[k] ^ 1;
    t_harvest(c, k, s, true);
    c->ev_cur[k] = s;
    (void)hipEventRecord(c->ev[k][s][0], c->stream);
  }
}
static inline void t_end(fh_ctx* c, int k) {
  if (t_on(c, k)) { (void)hipEventRecord(c->ev[k][c->ev_cur[k]][1], c->stream); c->ev_pending[k][c->ev_cur[k]] = true; }
}
// the sequence number the next launch's finaliser is to publish (0 = none: the caller will synchronise the stream)
static inline unsigned seq_offer(fh_ctx* c) {
  if (!c->seq_poll || !c->shards.empty() || c->comm != nullptr || c->owner != nullptr) return 0u;
  c->seq = c->seq + 1u ? c->seq + 1u : 1u;
  c->seq_wait = c->seq;
  return c->seq;
}
static int finish(fh_ctx* c) {   // synchronise the stream and harvest pending event pairs
  if (!c->shards.empty()) {        // shell: all shards (an emulated group shares one stream; its first shard's sync covers the rest)
    for (fh_ctx* s : c->shards) { HIP_TRY(hipSetDevice(s->device)); FH_TRY(finish(s)); }
    return 0;
  }
  c->seq_wait = 0;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->timing) {
    for (int k = 0; k < FH_NKERNELS; ++k) { t_harvest(c, k, 0, true); t_harvest(c, k, 1, true); }
  }
  return 0;
}
// Wait until the scalar block of the latest launch is on the host: by its sequence number when the launch publishes one (the host
// returns ~5 us before the launch's completion signal would have let it; what follows on the stream is ordered behind the launch as
// always), else -- or if the stream goes idle without the number having arrived, which a healthy launch cannot do -- by synchronising.
static int wait_scalars(fh_ctx* c) {
  const unsigned want = c->seq_wait;
  if (!want) return finish(c);
  c->seq_wait = 0;
  volatile unsigned* word = reinterpret_cast<volatile unsigned*>(c->hscal + FH_SEQ_SLOT);
  for (unsigned spins = 1;; ++spins) {
    if (*word == want) return 0;
    if ((spins & 0xFFFu) == 0u) {                            // every few tens of microseconds: is the stream still busy at all?
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) { if (*word == want) return 0; return finish(c); }     // (idle: the block is as complete as it will get)
      if (q != hipErrorNotReady) return fail((int)q, "hipStreamQuery failed while waiting for a launch: %s", hipGetErrorString(q));
      (void)hipGetLastError();
    }
    __builtin_ia32_pause();
  }
}

// Where kernels write the FH_S_* block: straight into the mapped host block on one GPU (no D2H copy, the
// stream sync alone publishes it); device memory when row-sharded, because RCCL reduces scalars in place.
static inline double* scalar_out(fh_ctx* c) { return row_sharded(c) ? c->dscal : c->hscal_dev; }

// row-sharded runs: the block lives in device memory (RCCL reduces into it); a 16-lane kernel forwards it to the mapped
// host block -- a hipMemcpyAsync D2H of 128 bytes costs ~10 us more per iteration than this launch
__global__ void k_forward_scalars(const double* src, double* dst) {
  if (threadIdx.x < FH_NSCALARS) dst[threadIdx.x] = src[threadIdx.x];
}

// The scalar block's way back, in two halves so that a step can be issued now and waited for later (fh_step_begin / fh_step_end):
// issue_scalars enqueues what is still missing (row-sharded contexts forward the block from device memory to the mapped host block)
// and closes the host-issue stopwatch; collect_scalars is the call's ONE host synchronisation (per device of a shell) and the copy out.
static int issue_scalars(fh_ctx* c) {
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
  if (c->issue_open) {             // everything of this call has been issued: what follows is the wait
    c->host_issue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->issue_t0).count();
    c->host_issue_calls += 1;
    c->issue_open = false;
  }
  return 0;
}
static int collect_scalars(fh_ctx* c, double* scalars) {
  FH_TRY(c->shards.empty() ? wait_scalars(c) : finish(c));               // ONE host wait per call (per device of a shell)
  // every shard holds the same block: each entry is either a sum over all shards or computed from replicated vectors
  fh_ctx* s0 = shard_of(c, 0);
  if (scalars) memcpy(scalars, s0->hscal, FH_NSCALARS * sizeof(double));
  // The clipping level of the l-infinity prox / l1-ball projection travels with every forward launch (FH_S_ALPHA).  NaN = the level
  // search could not finish (csrc/fh_prox.h: a hand-off timed out AND the single-workgroup fall-back was not available); prox_scalar
  // has propagated it into every output of the launch.  Typed, so that a caller can tell it from a device fault.
  if ((s0->prox_kind == FH_PROX_LINF || s0->prox_kind == FH_PROX_L1BALL) && s0->hscal[FH_S_ALPHA] != s0->hscal[FH_S_ALPHA])
    return fail(FH_E_TIMEOUT, "the clipping-level search of the l-infinity prox / l1-ball projection did not finish (hand-off between its workgroups "
                              "timed out and no fall-back was possible): the step's outputs are NaN");
  return 0;
}
static int fetch_scalars(fh_ctx* c, double* scalars) {
  FH_TRY(issue_scalars(c));
  return collect_scalars(c, scalars);
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
    // (an error inside a group must still close it: the thread's group depth would otherwise stay open and every later collective,
    // the fallback path's included, would be queued and never run)
    if (count2) NCCL_TRY(g_rccl.GroupStart());
    int r1 = g_rccl.AllReduce(sel(c), sel(c), count, kNcclFloat64, kNcclSum, c->comm, c->stream);
    if (count2) {
      if (r1 == 0) r1 = g_rccl.AllReduce(sel2(c), sel2(c), count2, kNcclFloat64, kNcclSum, c->comm, c->stream);
      const int r2 = g_rccl.GroupEnd();
      if (r1 == 0) r1 = r2;
    }
    if (r1 != 0) return fail(20000 + r1, "ncclAllReduce over the row blocks failed: %s", g_rccl.GetErrorString(r1));
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
  int rg = 0;                                            // first error; the group is closed whatever happens
  for (fh_ctx* s : c->shards) {
    if (rg == 0) rg = g_rccl.AllReduce(sel(s), sel(s), count, kNcclFloat64, kNcclSum, s->comm, s->stream);
    if (rg == 0 && count2) rg = g_rccl.AllReduce(sel2(s), sel2(s), count2, kNcclFloat64, kNcclSum, s->comm, s->stream);
  }
  { const int re = g_rccl.GroupEnd(); if (rg == 0) rg = re; }
  if (rg != 0) return fail(20000 + rg, "grouped ncclAllReduce over the row blocks failed: %s", g_rccl.GetErrorString(rg));
  for (fh_ctx* s : c->shards) { HIP_TRY(hipSetDevice(s->device)); t_end(s, FH_K_COMM); }
  return 0;
}
template <typename Sel>
static int sum_over_shards(fh_ctx* c, Sel sel, size_t count) {
  return sum_over_shards(c, sel, count, [](fh_ctx*) { return (double*)nullptr; }, 0);
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
                         double* xhat, double* xp, double* z, int sub_b);      // fh_host_launch.h
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

