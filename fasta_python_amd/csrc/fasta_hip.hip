// fasta_hip.hip -- the C ABI of libfasta_hip.so (declared in include/fasta_hip.h): entry points only.  The context and its helpers are
// in fh_host_ctx.h, the kernel launchers in fh_host_launch.h, the kernels in fh_dense.h / fh_tv.h / fh_prox.h / fh_fused.h.
// gfx950 only.  No PyTorch, no rocBLAS: every device operation is a kernel from fh_dense.h / fh_tv.h,
// plus RCCL (dlopen'ed on first use) for the row-sharded adjoint.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/fasta_hip.h"
#include "fh_experimental.h"
#include "fh_dense.h"
#include "fh_tv.h"
#include "fh_prox.h"
#include "fh_fused.h"
#include "fh_setup.h"
#include "fh_run.h"

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

// the chained form of the one-pass launch (k_fused_chain, csrc/fh_fused.h): group 0 of the same table -- float64, teams of 1 / 2 / 4 (fh_chain_part.hip)
#ifndef FH_SINGLE_TU
#define FH_CHAIN_DECLARE(P, PI, T, X, NB, F) extern template __global__ void k_fused_chain<P, 1, PI, T, X, NB, F>(const FusedP, const ChainP);
#define FUSED_INST_0 FH_CHAIN_DECLARE
#define FUSED_INST_1(...)
#define FUSED_INST_2(...)
#define FUSED_INST_3(...)
#include "fh_fused_instances.inc"
#undef FUSED_INST_0
#undef FUSED_INST_1
#undef FUSED_INST_2
#undef FUSED_INST_3
#endif
struct ChainEntry { int ppt, pipe, team, xlds, nbo, f32; void (*kernel)(const FusedP, const ChainP); };
#define FH_CHAIN_ROW(P, PI, T, X, NB, F) {P, PI, T, X, NB, F, k_fused_chain<P, 1, PI, T, X, NB, F>},
#define FUSED_INST_0 FH_CHAIN_ROW
#define FUSED_INST_1(...)
#define FUSED_INST_2(...)
#define FUSED_INST_3(...)
static const ChainEntry kChainTable[] = {
#include "fh_fused_instances.inc"
};
#undef FUSED_INST_0
#undef FUSED_INST_1
#undef FUSED_INST_2
#undef FUSED_INST_3

// the set-up kernel's variants: the same scheme, fh_setup_instances.inc / fh_setup_part.hip
#ifndef FH_SINGLE_TU
#define FH_SETUP_DECLARE(P, PI, T, NT, NR) extern template __global__ void k_setup_dense<P, PI, T, NT, NR>(const SetupP);
#define FH_SETUP_DECLARE32(P, PI, T, NT, NR) extern template __global__ void k_setup_dense<P, PI, T, NT, NR, 1>(const SetupP);
#define SETUP_INST_0 FH_SETUP_DECLARE
#define SETUP_INST_1 FH_SETUP_DECLARE
#define SETUP_INST_2 FH_SETUP_DECLARE32
#include "fh_setup_instances.inc"
#undef SETUP_INST_0
#undef SETUP_INST_1
#undef SETUP_INST_2
#endif
struct SetupEntry { int ppt, pipe, team, threads, nr, f32; void (*kernel)(const SetupP); };
#define FH_SETUP_ROW(P, PI, T, NT, NR) {P, PI, T, NT, NR, 0, k_setup_dense<P, PI, T, NT, NR>},
#define FH_SETUP_ROW32(P, PI, T, NT, NR) {P, PI, T, NT, NR, 1, k_setup_dense<P, PI, T, NT, NR, 1>},
#define SETUP_INST_0 FH_SETUP_ROW
#define SETUP_INST_1 FH_SETUP_ROW
#define SETUP_INST_2 FH_SETUP_ROW32
static const SetupEntry kSetupTable[] = {
#include "fh_setup_instances.inc"
};
#undef SETUP_INST_0
#undef SETUP_INST_1
#undef SETUP_INST_2

#include "fh_host_ctx.h"
#include "fh_host_launch.h"

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
  HIP_TRY(hipMalloc((void**)&c->gridbar, 3 * GB_WORDS * sizeof(unsigned)));
  HIP_TRY(hipMemsetAsync(c->gridbar, 0, 3 * GB_WORDS * sizeof(unsigned), c->stream));
  HIP_TRY(hipMalloc((void**)&c->dscal, (FH_NSCALARS + 16) * sizeof(double)));
  HIP_TRY(hipMemsetAsync(c->dscal, 0, (FH_NSCALARS + 16) * sizeof(double), c->stream));
  HIP_TRY(hipHostMalloc((void**)&c->hscal, (FH_NSCALARS + 16) * sizeof(double), hipHostMallocMapped));
  memset(c->hscal, 0, (FH_NSCALARS + 16) * sizeof(double));
  HIP_TRY(hipHostGetDevicePointer((void**)&c->hscal_dev, c->hscal, 0));
  for (int k = 0; k < FH_NKERNELS; ++k)
    for (int q = 0; q < 4; ++q) HIP_TRY(hipEventCreate(&c->ev[k][q / 2][q % 2]));
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
  for (int k = 0; k < FH_NKERNELS; ++k) for (int q = 0; q < 4; ++q) c->ev[k][q / 2][q % 2] = nullptr;
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
  // FH_CREATE_RCCL_SHELL or'ed into dtype: build the multi-device form even for ONE device -- a shell with one shard whose exchange is
  // the grouped ncclAllReduce on a communicator from ncclCommInitAll.  This is how a one-GPU box runs the RCCL branch of the
  // in-process form for real (tests); it computes what a plain context computes.
  const bool rccl_shell = (dtype & FH_CREATE_RCCL_SHELL) != 0;
  dtype &= ~FH_CREATE_RCCL_SHELL;
  if (dtype != FH_DTYPE_F64 && dtype != FH_DTYPE_F32_STORAGE) return fail(FH_E_ARG, "fh_create_ex: unknown dtype %d", dtype);
  if (ndev == 1 && !rccl_shell) {
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
  // shards on ONE device always share a stream (their one-pass launches each need every CU: they must not overlap); their exchange
  // is k_sum_shards -- unless FH_CREATE_RCCL_SHELL asks for the RCCL form anyway (needs an RCCL that accepts several ranks on one
  // device: real RCCL does not, the tests' stand-in does)
  const bool one_device = equal;
  const bool use_rccl = distinct || rccl_shell;
  fh_ctx* shell = new fh_ctx();
  for (int k = 0; k < FH_NKERNELS; ++k) for (int q = 0; q < 4; ++q) shell->ev[k][q / 2][q % 2] = nullptr;
  shell->device = dev_ids[0];
  shell->emulated = !use_rccl;
  shell->f32 = dtype == FH_DTYPE_F32_STORAGE ? 1 : 0;
  for (int i = 0; i < ndev; ++i) {
    fh_ctx* s = nullptr;
    const int rc = create_one(dev_ids[i], (one_device && i > 0) ? shell->shards[0]->stream : nullptr, &s);
    if (rc != 0) return create_failed(shell, rc);
    s->owner = shell; s->emulated = shell->emulated; s->f32 = shell->f32;
    s->nranks = ndev; s->rank = i;
    shell->shards.push_back(s);
  }
  shell->ncu = shell->shards[0]->ncu;
  if (use_rccl) {
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
  if (c->run_st_host) (void)hipHostFree(c->run_st_host);
  if (c->run_hist) (void)hipHostFree(c->run_hist);
  if (c->chain_state) (void)hipFree(c->chain_state);
  if (c->gridbar) (void)hipFree(c->gridbar);
  if (c->lvl_rec) (void)hipFree(c->lvl_rec);
  if (c->lvl_cnt) (void)hipFree(c->lvl_cnt);
  if (c->counters) (void)hipFree(c->counters);
  if (c->dscal) (void)hipFree(c->dscal);
  if (c->hscal) (void)hipHostFree(c->hscal);
  for (int k = 0; k < FH_NKERNELS; ++k) for (int q = 0; q < 4; ++q) if (c->ev[k][q / 2][q % 2]) (void)hipEventDestroy(c->ev[k][q / 2][q % 2]);
  if (c->stream && c->owns_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

// process-wide switches of the two remedies for "a large allocation made while the driver is still clearing a large free is slow" (fh_host_ctx.h)
extern "C" int fh_alloc_settle(int enable) { g_settle_on.store(enable ? 1 : 0); return 0; }
extern "C" int fh_alloc_cache(int enable) {
  g_cache_on.store(enable ? 1 : 0);
  if (!enable) release_cached_blocks(-1);
  return 0;
}
extern "C" int fh_release_cached(int device) {
  if (device < -1 || device >= FH_MAX_DEVICES) return fail(FH_E_ARG, "fh_release_cached: device must be -1 (all) or a device id");
  release_cached_blocks(device);
  return 0;
}
extern "C" int fh_alloc_cache_hits(uint64_t* hits) {
  if (!hits) return fail(FH_E_ARG, "null argument");
  *hits = g_cache_hits.load();
  return 0;
}
extern "C" int fh_alloc_settle_waited(double* seconds) {
  if (!seconds) return fail(FH_E_ARG, "null argument");
  *seconds = (double)g_settle_waited_ns.load() * 1e-9;
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
#ifdef FH_EXPERIMENTAL
    case FH_TUNE_TV_SLOTS:
      if (value < 0 || value > 8) return fail(FH_E_ARG, "TV_SLOTS must be in [0, 8] workgroups per CU (0 = one workgroup per chunk)");
      c->tv_slots = (int)value; return 0;
    case FH_TUNE_TV_RING:
      if (value < 0 || value > 3) return fail(FH_E_ARG, "TV_RING must be 0 (auto), 1 (register-staged trips), 2 or 3 (LDS-DMA ring slots per wave)");
      c->tv_ring = (int)value; return 0;
    case FH_TUNE_TV_LDS_PAD:
      if (value < 0 || value > 65536) return fail(FH_E_ARG, "TV_LDS_PAD must be in [0, 65536] bytes");
      c->tv_lds_pad = (int)value; return 0;
    case FH_TUNE_TV_ZFREE:
      // while the iterate is kept lazily (one-pass FISTA on the stencil), z-free steps rotate their image buffers without ever
      // writing them: the z-streaming kernel would read stale images after a switch
      if (c->lazy && (value ? 1 : 0) != c->tv_zfree)
        return fail(FH_E_STATE, "TV_ZFREE cannot change while a one-pass accelerated stencil solve is in flight (call fh_init / fh_set_vector(X0) first)");
      c->tv_zfree = value ? 1 : 0; return 0;
#else
    case FH_TUNE_TV_SLOTS: case FH_TUNE_TV_RING: case FH_TUNE_TV_LDS_PAD: case FH_TUNE_TV_ZFREE:
      return fail(FH_E_ARG, "tuning key %d is an experimental form of the stencil sweep: only in libfasta_hip_experimental.so (make experimental, csrc/fh_experimental.h)", key);
#endif
    case FH_TUNE_TV_XCD:
      if (value < 0 || value > 2) return fail(FH_E_ARG, "TV_XCD must be 0 (auto), 1 (on) or 2 (off)");
      c->tv_xcd = (int)value; return 0;
    case FH_TUNE_FUSED_CUS:
      if (value < 0 || value > 65536) return fail(FH_E_ARG, "FUSED_CUS must be in [0, 65536] (0 = every CU the device reports)");
      if ((int)value != c->fused_cus) { c->fused_cus = (int)value; c->coresident = -1; c->slots_sig = 0; }
      return 0;
    case FH_TUNE_RUN_CHAIN:
      c->run_chain_on = value ? 1 : 0; return 0;
    case FH_TUNE_SEQ_POLL:
      c->seq_poll = value ? 1 : 0; return 0;
    case FH_TUNE_ADJ_CYCLIC:
      if (value < 0 || value > 2) return fail(FH_E_ARG, "ADJ_CYCLIC must be 0 (auto), 1 (on) or 2 (off)");
      c->adj_cyclic = (int)value; return 0;
    case FH_TUNE_RUN_MAX_N:
      if (value < 0 || value > 7168) return fail(FH_E_ARG, "RUN_MAX_N must be in [0, 7168] (0 = the measured default; 7168 = the widest row fh_run has a kernel for)");
      c->run_max_n = (int)value; return 0;
    case FH_TUNE_FUSED_VARIANT:
      // (bits 64 / 128 were test hooks until round 4: a caller's variant word must not be able to switch the one-pass kernel off)
      if (value & 0xFFC1) return fail(FH_E_ARG, "FUSED_VARIANT: only the scheduling bits 2, 4, 8, 16, 32 are defined (got 0x%llx)", (unsigned long long)(value & 0xFFFF));
      c->fused_variant_auto = false;
      c->fused_variant = (int)(value & 0xFFFF);      // bits: see FusedP.variant (csrc/fh_fused.h) and fused_shape() below (8, 16: A/B shapes)
      if (value >> 16) c->fused_min_rows = (int)(value >> 16) == 0xFFFF ? 0 : (int)(value >> 16);   // high half: rows-per-team floor (0xFFFF = none)
      return 0;
    case FH_TUNE_TEST_HOOKS:        // not in the public header (csrc/fh_experimental.h): fault injection for the test-suite
      if (value & ~0xFF0Fll) return fail(FH_E_ARG, "TEST_HOOKS: bits 1, 2, 4, 8 and the attempt byte (bits 8..15) only");
      if ((int)value != c->test_hooks) { c->test_hooks = (int)value; c->coresident = -1; }
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
  const size_t a_bytes = (size_t)c->mp * c->ld * (c->f32 ? sizeof(float) : sizeof(double));
  HIP_TRY(acquire_matrix_block(c->device, a_bytes, &c->A, &c->a_block_bytes));      // the block this device kept from an earlier matrix, or a fresh one behind the settle wait (fh_host_ctx.h)
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

// The solver's whole set-up (fasta/__init__.py:100-113 and :135-137) in ONE call: the two Lipschitz probes live in FH_VEC_T0 / FH_VEC_T1,
// x0 in FH_VEC_X0.  Leaves the state fh_init leaves (z, g0, best iterate, acceleration history) and returns fh_init's scalars plus
// FH_S_DG2 = ||A^H grad f(A T0) - A^H grad f(A T1)||^2 and FH_S_DX2 = ||T0 - T1||^2 (the two norms of :110); FH_VEC_T2 / FH_VEC_T3 are
// scratch afterwards (the one-read kernel forms the gradient difference as A^T A (T0 - T1): csrc/fh_setup.h).
// Where the set-up kernel of csrc/fh_setup.h has a shape for this operator (dense float64, least squares, n <= 65536, single context, large
// enough for the one-pass kernel to pay) all of it comes from ONE read of A; everywhere else -- and after a hand-off timeout -- it is
// the three passes fh_gradient_at x 2 + fh_init, so the call is always available and its results agree to summation-order rounding
// (bit for bit where the one-pass kernel serves the three passes too).
static const SetupEntry* setup_entry(fh_ctx* c);
// sharded = true: a row block (shard of a multi-device context, or a rank): scalars to device memory, partial sums in T[2] / G[gc], pack behind T[2]
static int launch_setup_dense(fh_ctx* c, bool* launched, bool sharded = false) {
  *launched = false;
  if (!sharded && (row_sharded(c) || !c->shards.empty())) return 0;
  const SetupEntry* e = setup_entry(c);
  if (!e || !co_resident(c)) return 0;
  struct { int team; } sh = {e->team};               // (the set-up kernel's own team size: float32 storage does not follow the step kernel's shape)
  SetupP p;
  p.A = c->A; p.n = (uint32_t)c->n; p.m = (uint32_t)c->m; p.mp = (uint32_t)c->mp;
  p.ld2 = (uint32_t)(c->f32 ? round_up(c->n, 32) / 4 : round_up(c->n, 16) / 2);
  p.ldp = (uint32_t)(c->ld / (c->f32 ? 4 : 2));
  p.nv2 = p.ld2 * (c->f32 ? 2u : 1u);
  p.nteams = (uint32_t)(fused_ncu(c) / sh.team);
  if (c->fused_min_rows > 0) {
    const uint64_t want = std::max<uint64_t>(8, round_up((c->mp + c->fused_min_rows - 1) / c->fused_min_rows, 8));
    p.nteams = (uint32_t)std::min<uint64_t>(p.nteams, want);
  }
  p.rows_per_team = (uint32_t)((c->mp + p.nteams - 1) / p.nteams);
  p.x[0] = c->T[0]; p.x[1] = c->T[1]; p.x[2] = c->X[c->xi];
  p.g[0] = c->T[2]; p.g[1] = c->T[3]; p.g[2] = c->G[c->gc];
  p.z = c->Z[c->zc]; p.b = c->b;
  const unsigned grid = p.nteams * sh.team;
  const size_t sl = sh.team < 8 ? 8 : sh.team;
  const size_t slots_elems = (size_t)c->mp * e->nr * sl;
  const size_t gpart_elems = (size_t)p.nteams * e->nr * p.nv2 * 2;
  FH_TRY(ensure_ws(c, (gpart_elems + (size_t)grid * 8) * sizeof(double)));
  p.gpart = c->ws; p.red = p.gpart + gpart_elems;
  if (slots_elems * sizeof(double) > c->slotbuf_bytes) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->slotbuf) { HIP_TRY(hipFree(c->slotbuf)); c->slotbuf = nullptr; c->slotbuf_bytes = 0; }
    const size_t bytes = round_up(slots_elems * sizeof(double), 1 << 20);
    HIP_TRY(hipMalloc((void**)&c->slotbuf, bytes));
    c->slotbuf_bytes = bytes;
  }
  c->slots_sig = 0;                       // the step kernel's two slot arrays share this buffer: it refills them on its next launch
  p.slots = c->slotbuf;
  if (sh.team > 1) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->slotbuf, (int)FT_SENTINEL_HI, slots_elems * 2, c->stream));
  HIP_TRY(hipMemsetAsync(c->counters + CNT_FUSED_BAR, 0, 8 * sizeof(unsigned), c->stream));
  HIP_TRY(hipMemsetAsync(c->gridbar, 0, 2 * GB_WORDS * sizeof(unsigned), c->stream));
  p.bar = c->counters + CNT_FUSED_BAR; p.gbar = c->gridbar; p.err = c->counters + CNT_FUSED_ERR;
  // (rows dealt cyclically -- bit 32, the step kernel's default: 65536^2 float64 4.98-5.01 ms wherever the matrix lies, blocked 4.95-5.32; the float32 shapes
  // are faster blocked, 4.39 vs 4.69 ms: profiles/r06_placement.txt)
  p.variant = (fused_variant_for(c, p.rows_per_team) & ~(c->f32 && c->fused_variant_auto ? 32 : 0)) | ((c->test_hooks & FH_HOOK_WITHHOLD_PARTIAL) ? 64 : 0);
  p.out = scalar_out(c);
  p.pack = sharded ? c->T[2] + c->nv : nullptr;     // (slack behind every n-side vector: alloc_vectors)
  t_begin(c, FH_K_FUSED);
  e->kernel<<<dim3(grid), dim3(e->threads), 0, c->stream>>>(p);
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  *launched = true;
  return 0;
}

static __global__ void k_set_pair(double* w, double a, double b) { w[0] = a; w[1] = b; }
// the one-read set-up kernel of this context's shape, or nullptr (the three passes then): dense float64 least squares, n <= 65536, large enough to pay
static const SetupEntry* setup_entry(fh_ctx* c) {
  if (c->op != OP_DENSE || c->loss_kind != LOSS_LSQ || !fused_pays(c)) return nullptr;
  if (c->f32) {
    // float32 storage (round 6): a member covers 256 x (at most 4) four-column pieces -- two right-hand sides cost 16 accumulator registers per
    // piece, five pieces spill into the row loop -- so the team is as large as the row needs (n <= 65536: up to 16 members), ONE workgroup per CU;
    // its own rows of the table (fh_setup_instances.inc, group 2), not the step kernel's shape
    if (c->ld % 4 || fused_ncu(c) < 1) return nullptr;
    const uint64_t pieces = round_up(c->n, 32) / 4;
    for (int team = 1; team <= 16; team *= 2) {
      if (pieces > (uint64_t)team * FH_WG * 4) continue;
      if (fused_ncu(c) % team) return nullptr;
      int mpp = (int)((pieces + (uint64_t)team * FH_WG - 1) / ((uint64_t)team * FH_WG));
      if (mpp == 3 || team > 1) mpp = 4;
      for (const SetupEntry& k : kSetupTable) if (k.f32 == 1 && k.ppt == mpp && k.team == team) return &k;
      return nullptr;
    }
    return nullptr;
  }
  const FusedShape sh = fused_shape(c);
  if (!sh.ppt || sh.xlds) return nullptr;
  // full 8-piece shapes of 8 / 16 members: 512-thread workgroups (fh_setup_instances.inc); FH_TUNE_FUSED_VARIANT bit 16 keeps the 256-thread form (A/B)
  const int want_threads = (sh.ppt == 8 && sh.team >= 8 && sh.team <= 16 && !(c->fused_variant & 16)) ? 512 : 256;
  for (const SetupEntry& k : kSetupTable) if (!k.f32 && k.ppt == sh.ppt && k.team == sh.team && k.pipe == (sh.team == 1 ? 1 : sh.pipe) && k.threads == want_threads) return &k;
  if (want_threads == 512) for (const SetupEntry& k : kSetupTable) if (!k.f32 && k.ppt == sh.ppt && k.team == sh.team && k.pipe == (sh.team == 1 ? 1 : sh.pipe)) return &k;
  return nullptr;
}

static int diff_norm_sq(fh_ctx* c, int vec_a, int vec_b, double* sumsq);
extern "C" int fh_gradient_at(fh_ctx* c, int src_vec, int dst_vec);

// The one-read set-up over ROW BLOCKS (round 6): least squares is linear in the rows, so every block runs the two-right-hand-side kernel on its
// own rows -- A_k^T A_k (x1 - x2) into T[2], A_k^T (A_k x0 - b_k) into g0, z_k = A_k x0, its loss sum -- and ONE exchange sums both n-vectors
// with the loss sums and the timeout words riding behind the first (as a step's n + 3 exchange does); ||sum_k A_k^T A_k d||^2 is then one small
// reduction on replicated data.  One launch per block instead of three.  *done = false: not taken (a block has no shape for it, the ranks do
// not agree, a hand-off timed out somewhere -- every block then sees the same summed timeout word): the caller runs the three passes.
// Ranks of a communicator settle the decision first (fh_fused_agree is collective: every rank is inside fh_setup at this point).
static int setup_row_blocks(fh_ctx* c, double* scalars, bool* done) {
  *done = false;
  const int ns = nshards(c);
  bool all = true;
  for (int k = 0; k < ns; ++k) { fh_ctx* s = shard_of(c, k); if (!setup_entry(s) || !co_resident(s)) all = false; }
  if (c->shards.empty()) {
    // a rank: EVERY rank must take the same way (the two ways issue different collectives), and a rank's own verdict depends on its local rows
    // and its own co-residency probe: the "no" verdicts are summed over the communicator -- whatever this rank found, it enters the exchange
    double* w = c->dscal + FH_NSCALARS + 8;              // scratch behind the scalar block (as fh_fused_agree)
    double* back = c->hscal + FH_NSCALARS + 8;
    FH_TRY(use_device(c));
    k_set_pair<<<dim3(1), dim3(1), 0, c->stream>>>(w, all ? 0.0 : 1.0, 0.0);
    HIP_TRY(hipGetLastError());
    FH_TRY(sum_over_shards(c, [w](fh_ctx*) { return w; }, 2));
    HIP_TRY(hipMemcpyAsync(back, w, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    all = back[0] == 0.0;
  }
  if (!all) return 0;
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    FH_TRY(use_device(s));
    bool launched = false;
    FH_TRY(launch_setup_dense(s, &launched, true));
    if (!launched) return fail(FH_E_STATE, "fh_setup: a row block lost its one-read set-up kernel between the check and the launch");
  }
  FH_TRY(sum_over_shards(c, [](fh_ctx* s) { return s->T[2]; }, (size_t)shard_of(c, 0)->nv + 2, [](fh_ctx* s) { return s->G[s->gc]; }, (size_t)shard_of(c, 0)->nv));
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    FH_TRY(use_device(s));
    s->lazy = false; s->commits = 0; s->tvz_pending = false; s->zcur_stale = false;
    double* x0 = s->X[s->xi];
    HIP_TRY(hipMemcpyAsync(s->P[s->pc], x0, s->nv * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    s->bi = s->xi;
    for (int q = 0; q < 3; ++q) if (q != s->xi) { s->ti = q; break; }
    s->last_accel = false;
    s->slots_sig = 0;                   // (the set-up kernel's slots share the step kernel's buffer)
    FH_TRY(launch_gterms(s, x0));
  }
  // ||sum_k A_k^T A_k d||^2 on block 0 (replicated data), and the summed loss / timeout words from behind T[2]
  fh_ctx* s0 = shard_of(c, 0);
  FH_TRY(use_device(s0));
  HIP_TRY(hipMemsetAsync(s0->T[3], 0, s0->nv * sizeof(double), s0->stream));
  double dg2 = 0.0;
  FH_TRY(diff_norm_sq(s0, FH_VEC_T2, FH_VEC_T3, &dg2));
  HIP_TRY(hipMemcpyAsync(s0->hscal + FH_NSCALARS + 2, s0->T[2] + s0->nv, 2 * sizeof(double), hipMemcpyDeviceToHost, s0->stream));
  FH_TRY(fetch_scalars(c, scalars));
  const double fsq = s0->hscal[FH_NSCALARS + 2], timed_out = s0->hscal[FH_NSCALARS + 3];
  if (timed_out != 0.0) return 0;       // (summed over the blocks: every block / rank comes to the same conclusion)
  scalars[FH_S_FSQ] = fsq;
  scalars[FH_S_DG2] = dg2;
  scalars[15] = 0.0;
  *done = true;
  return 0;
}

extern "C" int fh_setup(fh_ctx* c, double* scalars) {
  FH_TRY(check_ready(c, true));
  if (!scalars) return fail(FH_E_ARG, "fh_setup: null scalars");
  if (!c->shards.empty() || c->comm) {
    bool done = false;
    FH_TRY(setup_row_blocks(c, scalars, &done));
    if (done) return 0;
  }
  if (c->shards.empty()) {
    bool launched = false;
    FH_TRY(use_device(c));
    FH_TRY(launch_setup_dense(c, &launched));
    if (launched) {
      // what fh_init does besides its pass over A (fasta/__init__.py:154-167), behind the one launch, under one synchronisation
      c->lazy = false; c->commits = 0; c->tvz_pending = false; c->zcur_stale = false;
      double* x0 = c->X[c->xi];
      HIP_TRY(hipMemcpyAsync(c->P[c->pc], x0, c->nv * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      c->bi = c->xi;
      for (int q = 0; q < 3; ++q) if (q != c->xi) { c->ti = q; break; }
      c->last_accel = false;
      FH_TRY(launch_gterms(c, x0));
      FH_TRY(fetch_scalars(c, scalars));
      if (scalars[15] == 0.0) return 0;
      c->slots_sig = 0;                   // a hand-off timed out (results invalid): the three passes below redo everything
    }
  }
  double dg2 = 0.0, dx2 = 0.0;
  FH_TRY(fh_gradient_at(c, FH_VEC_T0, FH_VEC_T2));
  FH_TRY(fh_gradient_at(c, FH_VEC_T1, FH_VEC_T3));
  FH_TRY(diff_norm_sq(c, FH_VEC_T2, FH_VEC_T3, &dg2));
  FH_TRY(diff_norm_sq(c, FH_VEC_T0, FH_VEC_T1, &dx2));
  FH_TRY(fh_init(c, scalars));
  scalars[FH_S_DG2] = dg2;
  scalars[FH_S_DX2] = dx2;
  return 0;
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

// sum of (a_i - b_i)^2 of two device vectors of equal length
static int diff_norm_sq(fh_ctx* c, int vec_a, int vec_b, double* sumsq) {
  FH_TRY(check_ready(c, false));
  if (!sumsq) return fail(FH_E_ARG, "null out");
  if (!c->shards.empty()) {                        // n-side vectors are replicated: any shard has the answer
    if (m_side(vec_a) || m_side(vec_b)) return fail(FH_E_ARG, "fh_diff_norm on a multi-device context takes n-side vectors");
    return diff_norm_sq(c->shards[0], vec_a, vec_b, sumsq);
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
  *sumsq = c->hscal[FH_NSCALARS + 1];
  return 0;
}
extern "C" int fh_diff_norm(fh_ctx* c, int vec_a, int vec_b, double* out) {
  if (!out) return fail(FH_E_ARG, "null out");
  double sumsq = 0.0;
  FH_TRY(diff_norm_sq(c, vec_a, vec_b, &sumsq));
  *out = sqrt(sumsq);
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

// A purely LOCAL query (no exchange, safe to call from one rank alone, e.g. for logging): what THIS context's shape and THIS
// context's co-residency probe say.  Ranks of a row-sharded run settle on one verdict with fh_fused_agree below.
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
  // The dense one-pass kernel needs its whole grid (one workgroup per CU it uses) co-resident: probed once per context
  // (co_resident above); "unsupported" up front instead of a bounded-spin timeout on the first launch (the timeout stays as the
  // safety net for CUs that disappear later).
  if (ppt && !co_resident(c)) ppt = 0;
  int kind = c->op == OP_STENCIL ? (row_sharded(c) ? 0 : 2) : (ppt ? (fused_pays(c) ? 1 : 3) : 0);
  *yes = kind;
  return 0;
}

// One process per GPU: the verdict must be the SAME on every rank -- a rank whose probe said no would otherwise take fh_fwd / fh_adj
// (exchanges of 1, then n + 1 doubles) while its peers take fh_step (one exchange of n + 3): mismatched collectives, i.e. a hang.
// fh_fused_agree is the explicitly COLLECTIVE form: every rank of the communicator calls it at the same point (FBSolver.setup
// does); the counts of "0" and "3" verdicts are summed over the ranks, any 0 makes it 0 everywhere, else any 3 makes it 3.  A rank
// whose local query FAILED still enters the exchange, contributing a "0" verdict, so that its peers get an answer instead of a
// hang; the failure is then returned on that rank.  Never cached: a cache keyed on local state could make one rank skip an
// exchange its peers enter.  Without a communicator (plain context, or a multi-device context, whose shards are combined in one
// process) it is the local query.
extern "C" int fh_fused_agree(fh_ctx* c, int* yes) {
  if (!c || !yes) return fail(FH_E_ARG, "null argument");
  int kind = 0;
  const int rc_local = fh_fused_supported(c, &kind);
  if (rc_local != 0) kind = 0;
  if (!c->comm || c->owner || !c->shards.empty() || c->op != OP_DENSE) { *yes = kind; return rc_local; }
  char keep[sizeof(g_err)];
  memcpy(keep, g_err, sizeof(keep));                   // (the text of a local failure survives the exchange)
  double* w = c->dscal + FH_NSCALARS + 8;              // scratch behind the scalar block
  double* back = c->hscal + FH_NSCALARS + 8;           // pinned: no stack buffer is ever handed to an asynchronous copy
  (void)hipSetDevice(c->device);
  k_set_pair<<<dim3(1), dim3(1), 0, c->stream>>>(w, kind == 0 ? 1.0 : 0.0, kind == 3 ? 1.0 : 0.0);   // (values travel as kernel arguments)
  (void)hipGetLastError();
  const int rc_sum = sum_over_shards(c, [w](fh_ctx*) { return w; }, 2);
  if (rc_local != 0) { memcpy(g_err, keep, sizeof(keep)); return rc_local; }
  FH_TRY(rc_sum);
  HIP_TRY(hipMemcpyAsync(back, w, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *yes = back[0] != 0.0 ? 0 : (back[1] != 0.0 ? 3 : kind);
  return 0;
}

// CUs the device reports and CUs the one-pass dense kernel is launched on (FH_TUNE_FUSED_CUS)
extern "C" int fh_cu_count(fh_ctx* c, int* device_cus, int* one_pass_cus) {
  if (!c || !device_cus || !one_pass_cus) return fail(FH_E_ARG, "null argument");
  if (!c->shards.empty()) c = c->shards[0];
  *device_cus = c->ncu;
  *one_pass_cus = fused_ncu(c);
  return 0;
}
// the library the RCCL entry points were taken from ("" before the first communicator): the system's RCCL or $FASTA_RCCL_LIB
extern "C" const char* fh_comm_library(void) { return g_rccl_path; }

// One-pass iteration of the dense operator on every row block (`accel` = 0: fh_step; 1: fh_step_accel): local launch on every
// shard -> ONE sum over the row blocks of g1 with the local loss sums and the timeout word appended (every shard / rank then sees
// the same verdict, so all of them drop to the two-launch path together if a hand-off ever times out) -> the n-side epilogue on
// every shard -> one host synchronisation.  A plain single-GPU context is the one-shard case without the exchange.
static void step_collected(fh_ctx* c) { for (int k = 0; k < nshards(c); ++k) fused_after(shard_of(c, k)); }
static int dense_step(fh_ctx* c, double tau, int accel, double coef, int restart, double* scalars, bool wait = true) {
  const int ns = nshards(c);
  if (c->timing) { c->issue_t0 = std::chrono::steady_clock::now(); c->issue_open = true; }
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
  FH_TRY(issue_scalars(c));
  if (!wait) { c->pending_step = true; return 0; }
  FH_TRY(collect_scalars(c, scalars));
  step_collected(c);
  return 0;
}

// One-pass FBS iteration: K-fwd and K-adj of the same tau from a single read of A (no acceleration).
// Writes the complete FH_S_* block; scalars[15] != 0 reports a spin timeout (results invalid: use the two-launch path).
static int step_body(fh_ctx* c, double tau, double* scalars, bool wait) {
  FH_TRY(check_ready(c, true));
  for (int k = 0; k < nshards(c); ++k) FH_TRY(not_lazy(shard_of(c, k), "fh_step"));
  if (c->op == OP_STENCIL) {
    if (row_sharded(c)) return fail(FH_E_STATE, "row sharding is implemented for the dense operator only");
#ifdef FH_EXPERIMENTAL
    if (!c->tv_zfree) {
      FH_TRY(tv_refresh_zcur(c));
      c->tvz_pending = false;
      FH_TRY(launch_fused_tv(c, tau));
    } else
#endif
    {
      if (!c->zcur) return fail(FH_E_STATE, "fh_step on the stencil operator before fh_init");
      FH_TRY(launch_tv_onepass(c, tau, 0, 0.0, 0));
    }
    c->last_accel = false;
    FH_TRY(issue_scalars(c));
    if (!wait) { c->pending_step = true; return 0; }
    return collect_scalars(c, scalars);
  }
  return dense_step(c, tau, 0, 0.0, 0, scalars, wait);
}
extern "C" int fh_step(fh_ctx* c, double tau, double* scalars) { return step_body(c, tau, scalars, true); }
// fh_step in two halves: fh_step_begin issues the launch(es) on the context's stream and returns WITHOUT waiting; fh_step_end is the
// wait and delivers the scalar block.  Between the two the host is free -- in particular to issue a step on ANOTHER context, which
// is how one host thread keeps two solves (FH_TUNE_FUSED_CUS each) in flight on one device.  Every other entry point of the
// context refuses (FH_E_STATE) until fh_step_end has been called.
extern "C" int fh_step_begin(fh_ctx* c, double tau) { return step_body(c, tau, nullptr, false); }
extern "C" int fh_step_end(fh_ctx* c, double* scalars) {
  if (!c) return fail(FH_E_ARG, "null context");
  if (!c->pending_step) return fail(FH_E_STATE, "fh_step_end without fh_step_begin");
  c->pending_step = false;
  if (c->shards.empty()) FH_TRY(use_device(c));
  FH_TRY(collect_scalars(c, scalars));
  if (c->op == OP_DENSE) step_collected(c);
  return 0;
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
#ifdef FH_EXPERIMENTAL
    if (!c->tv_zfree) FH_TRY(launch_fused_tv_accel(c, tau, coef, restart ? 1 : 0));
    else
#endif
    FH_TRY(launch_tv_onepass(c, tau, 1, coef, restart ? 1 : 0));
    c->last_accel = true;
    FH_TRY(fetch_scalars(c, scalars));
    c->lc_pending = (restart && c->hscal[FH_S_RDOT] > 1E-30) ? 0.0 : coef;      // what the launch applied (:231); adopted by fh_commit
    return 0;
  }
  return dense_step(c, tau, 1, coef, restart, scalars);
}

// ---- the loop on the device (csrc/fh_run.h) --------------------------------------------------------------------------------------------
static const int kRunDefaultMaxN = 6144;
struct RunEntry { int ppt; void (*kernel)(const RunP); };
static const RunEntry kRunTable[] = {{1, k_run_dense<1>}, {2, k_run_dense<2>}, {4, k_run_dense<4>}, {5, k_run_dense<5>},
                                     {6, k_run_dense<6>}, {7, k_run_dense<7>}, {8, k_run_dense<8>},
                                     {10, k_run_dense<10>}, {12, k_run_dense<12>}, {14, k_run_dense<14>}};
// (16 pieces per lane -- n in (7168, 8192] -- were measured and left out: 8192^2 158 us per iteration against 140 us on the per-iteration path,
//  profiles/r05_device_loop.txt)
static const RunEntry* run_entry(fh_ctx* c) {
  if (c->op != OP_DENSE || c->f32 || row_sharded(c) || !c->shards.empty()) return nullptr;
  if (c->prox_kind != FH_PROX_IDENTITY && c->prox_kind != FH_PROX_SHRINK && c->prox_kind != FH_PROX_NONNEG && c->prox_kind != FH_PROX_BOX) return nullptr;
  // a workgroup owns whole rows: 16-byte pieces per lane = the first table entry that covers the row (lanes past the row's end re-read its last piece)
  const uint64_t pieces = round_up(c->n, 16) / 2;
  if (c->ld % 2 || pieces == 0 || pieces > (uint64_t)FH_WG * 14) return nullptr;
  // ... and it is OFFERED only where it beats one launch per iteration issued by fh_iterate (profiles/r06_device_loop.txt: a grid of 36 shapes, ratio
  // device / library loop): whole rows in LDS (n <= 4096) up to 32 Mi elements -- beyond, the launches are long enough to hide their fixed cost and the
  // team kernel streams faster (4096 x 16384: 0.96, 2048 x 32768: 0.95); the wide shapes (n <= 6144) from 4096 rows on -- below, the redundant n-side
  // work of every workgroup costs more than a launch (1024 x 6144: 0.94) -- up to 40 Mi elements (8192 x 6144: 0.97; 6144^2: 1.08, 4096 x 6144: 1.19).
  // FH_TUNE_RUN_MAX_N = N replaces the window by "n <= N, any m" (tests, probes).
  if (c->run_max_n > 0) { if (c->n > (uint64_t)c->run_max_n) return nullptr; }
  else if (c->n <= 4096) { if (c->m * c->n > ((uint64_t)1 << 25)) return nullptr; }
  else if (c->n <= (uint64_t)kRunDefaultMaxN) { if (c->m < 4096 || c->m * c->n > ((uint64_t)40 << 20)) return nullptr; }
  else return nullptr;
  const int need = (int)((pieces + FH_WG - 1) / FH_WG);
  for (const RunEntry& e : kRunTable) if (e.ppt >= need) return &e;
  return nullptr;
}
// The chained form (k_fused_chain, csrc/fh_fused.h) can serve what the persistent launch does not: float64 shapes of 1 / 2 / 4 team members
// (n <= 16384) outside fh_run's window -- same prox kinds, single context.  OPT-IN (FH_TUNE_RUN_CHAIN = 1): it removes the gap between two
// launches altogether (rocprofv3: 0.0 us against 11.6 us host-driven) but each launch is 5-7 us longer (the dependent dispatch, the state
// block's way in, the controller), which is what the host-side loop with its sequence-number wait costs too -- 8192^2: 116.9 vs 115.6 us per
// iteration, 16384^2: 351.6 vs 352.2 (profiles/r06_chain.txt).  The floor of DEPENDENT launches is the hardware's, not the host's.
static bool chain_ok(fh_ctx* c) {
  if (c->op != OP_DENSE || c->f32 || row_sharded(c) || !c->shards.empty() || !c->run_chain_on) return false;
  if (c->prox_kind != FH_PROX_IDENTITY && c->prox_kind != FH_PROX_SHRINK && c->prox_kind != FH_PROX_NONNEG && c->prox_kind != FH_PROX_BOX) return false;
  const FusedShape sh = fused_shape(c);
  return sh.ppt && sh.team <= 4 && !sh.xlds && chain_lookup(sh, 0) != nullptr;
}
extern "C" int fh_abi_sizes(uint64_t sizes[4]) {
  if (!sizes) return fail(FH_E_ARG, "null argument");
  sizes[0] = sizeof(fh_run_opts); sizes[1] = sizeof(fh_run_state); sizes[2] = FH_RUN_HIST; sizes[3] = FH_RUN_WINDOW_MAX;
  return 0;
}
extern "C" int fh_run_supported(fh_ctx* c, int* yes) {
  if (!c || !yes) return fail(FH_E_ARG, "null argument");
  *yes = ((run_entry(c) || chain_ok(c)) && co_resident(c)) ? 1 : 0;
  return 0;
}
static int run_adopt(fh_ctx* c, const fh_run_opts* o, const RunState* hs, double* const (&nb)[5], int max_steps, fh_run_state* state, double* history, int* steps_done, const char* what);
static int run_chain(fh_ctx* c, int max_steps, const fh_run_opts* o, fh_run_state* state, double* history, int* steps_done);
extern "C" int fh_run(fh_ctx* c, int max_steps, const fh_run_opts* o, fh_run_state* state, double* history, int* steps_done) {
  FH_TRY(check_ready(c, true));
  if (!o || !state || !history || !steps_done) return fail(FH_E_ARG, "fh_run: null argument");
  if (max_steps < 0 || max_steps > 65536) return fail(FH_E_ARG, "fh_run: max_steps must be in [0, 65536]");
  if (o->window < 1 || o->window > FH_RUN_WINDOW_MAX) return fail(FH_E_ARG, "fh_run: window must be in [1, %d]", FH_RUN_WINDOW_MAX);
  if (o->stop_rule < 0 || o->stop_rule > 3) return fail(FH_E_ARG, "fh_run: stop_rule must be 0..3 (the four rules of fasta/stopping.py)");
  const RunEntry* e = run_entry(c);
  if (!e && chain_ok(c) && co_resident(c)) return run_chain(c, max_steps, o, state, history, steps_done);
  if (!e || !co_resident(c)) return fail(FH_E_STATE, "fh_run: no device-side loop for this operator / loss / prox (see fh_run_supported)");
  FH_TRY(use_device(c));
  FH_TRY(not_lazy(c, "fh_run"));
  if (!c->run_st_host) {
    HIP_TRY(hipHostMalloc(&c->run_st_host, 2 * sizeof(ChainState), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(&c->run_st_host_dev, c->run_st_host, 0));
  }
  if ((size_t)max_steps > c->run_hist_steps) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->run_hist) { HIP_TRY(hipHostFree(c->run_hist)); c->run_hist = nullptr; c->run_hist_steps = 0; }
    const size_t steps = round_up((size_t)max_steps, 256);
    HIP_TRY(hipHostMalloc((void**)&c->run_hist, steps * FR_HIST * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&c->run_hist_dev, c->run_hist, 0));
    c->run_hist_steps = steps;
  }
  RunP p;
  RunState* hs = &p.init;                               // the state on entry travels as a kernel argument
  memset(hs, 0, sizeof(RunState));
  hs->tau_next = state->tau_next; hs->alpha1 = state->alpha1; hs->max_residual = state->max_residual; hs->best_quality = state->best_quality;
  hs->iteration = state->iteration; hs->backtracks = state->backtracks; hs->stopped = 0;
  hs->xi = c->xi; hs->ti = c->ti; hs->bi = c->bi; hs->pc = c->pc; hs->gc = c->gc; hs->zc = c->zc; hs->last_accel = c->last_accel ? 1 : 0;
  for (int q = 0; q < 5; ++q) hs->perm[q] = q;
  memcpy(hs->f_window, state->f_window, sizeof(hs->f_window));
  p.A = c->A; p.n = (uint32_t)c->n; p.m = (uint32_t)c->m; p.mp = (uint32_t)c->mp;
  p.ld2 = (uint32_t)(round_up(c->n, 16) / 2); p.ldp = (uint32_t)(c->ld / 2); p.nv2 = p.ld2;
  p.nteams = (uint32_t)fused_ncu(c);
  if (c->fused_min_rows > 0) {
    const uint64_t want = std::max<uint64_t>(8, round_up((c->mp + c->fused_min_rows - 1) / c->fused_min_rows, 8));
    p.nteams = (uint32_t)std::min<uint64_t>(p.nteams, want);
  }
  p.rows_per_team = (uint32_t)((c->mp + p.nteams - 1) / p.nteams);
  double* nb[5] = {c->X[0], c->X[1], c->X[2], c->P[0], c->P[1]};
  for (int q = 0; q < 5; ++q) p.nbuf[q] = nb[q];
  p.G[0] = c->G[0]; p.G[1] = c->G[1]; p.Z[0] = c->Z[0]; p.Z[1] = c->Z[1]; p.xhat = c->xhat; p.b = c->b;
  p.loss = c->loss_kind; p.prox_kind = c->prox_kind; p.mu = c->mu; p.lo = c->lo; p.hi = c->hi;
  p.nt = nt_for(c);
  p.g_kind = c->prox_kind == FH_PROX_SHRINK ? 1 : 0;
  p.o.adaptive = o->adaptive; p.o.accelerate = o->accelerate; p.o.backtrack = o->backtrack; p.o.restart = o->restart;
  p.o.evaluate_objective = o->evaluate_objective; p.o.stop_rule = o->stop_rule; p.o.window = o->window; p.o.max_backtracks = o->max_backtracks;
  p.o.stepsize_shrink = o->stepsize_shrink; p.o.tolerance = o->tolerance;
  p.max_steps = max_steps;
  p.st_out = (RunState*)c->run_st_host_dev; p.hist = c->run_hist_dev;
  memset(c->run_st_host, 0xFF, sizeof(RunState));     // (poisoned: a launch that never wrote its state back cannot pass for one that did)
  const unsigned grid = p.nteams;
  const size_t gpart_elems = (size_t)grid * p.nv2 * 2;
  FH_TRY(ensure_ws(c, (gpart_elems + (size_t)2 * grid * 16) * sizeof(double)));
  p.gpart = c->ws; p.red = p.gpart + gpart_elems;
  HIP_TRY(hipMemsetAsync(c->gridbar, 0, GB_WORDS * sizeof(unsigned), c->stream));
  HIP_TRY(hipMemsetAsync(c->counters + CNT_RUN_BAR, 0, 2 * sizeof(unsigned), c->stream));
  p.bar = c->gridbar; p.err = c->counters + CNT_RUN_BAR + 1;
  p.hook_attempt = FH_HOOK_RUN_ATTEMPT(c->test_hooks);
  t_begin(c, FH_K_FUSED);
  e->kernel<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  FH_TRY(finish(c));
  return run_adopt(c, o, (const RunState*)c->run_st_host, nb, max_steps, state, history, steps_done, "the persistent launch");
}

// What a device-side loop wrote back -- fh_run's persistent launch, or the last launch of a chain -- is checked before a single index of it
// is used (the block was poisoned before the launch) and then adopted: buffer roles and indices exactly as fh_commit would have left them.
// After a timeout (stopped == 3) that is the state of the last COMPLETED iteration: an attempt writes only buffers that are not x0 / g0 /
// x_accel0 / z_accel0, so everything the next iteration reads is intact, and tau_next is the step the interrupted iteration began with.
static int run_adopt(fh_ctx* c, const fh_run_opts* o, const RunState* hs, double* const (&nb)[5], int max_steps, fh_run_state* state, double* history, int* steps_done, const char* what) {
  {
    bool sane = (hs->stopped == 0 || hs->stopped == 1 || hs->stopped == 3) && hs->iteration >= state->iteration &&
                hs->iteration - state->iteration <= (uint64_t)max_steps && hs->backtracks >= state->backtracks &&
                hs->xi >= 0 && hs->xi < 3 && hs->ti >= 0 && hs->ti < 3 && hs->bi >= 0 && hs->bi < 3 && hs->xi != hs->ti &&
                (hs->pc | 1) == 1 && (hs->gc | 1) == 1 && (hs->zc | 1) == 1;
    unsigned seen = 0;
    for (int q = 0; q < 5 && sane; ++q) { if (hs->perm[q] < 0 || hs->perm[q] >= 5) sane = false; else seen |= 1u << hs->perm[q]; }
    if (!sane || seen != 31u) {
      c->slots_sig = 0;
      return fail(FH_E_STATE, "fh_run: %s did not write a valid solver state back (stopped = %d) -- the state is undefined; call fh_init", what, hs->stopped);
    }
  }
  double* nx[5];
  for (int q = 0; q < 5; ++q) nx[q] = nb[hs->perm[q]];
  c->X[0] = nx[0]; c->X[1] = nx[1]; c->X[2] = nx[2]; c->P[0] = nx[3]; c->P[1] = nx[4];
  c->xi = hs->xi; c->ti = hs->ti; c->bi = hs->bi; c->pc = hs->pc; c->gc = hs->gc; c->zc = hs->zc; c->last_accel = hs->last_accel != 0;
  const int done = (int)(hs->iteration - state->iteration);
  c->commits += (uint64_t)done;
  c->slots_sig = 0;
  state->tau_next = hs->tau_next; state->alpha1 = hs->alpha1; state->max_residual = hs->max_residual; state->best_quality = hs->best_quality;
  state->iteration = hs->iteration; state->backtracks = hs->backtracks; state->stopped = hs->stopped;
  memcpy(state->f_window, hs->f_window, sizeof(hs->f_window));
  memcpy(history, c->run_hist, (size_t)done * FR_HIST * sizeof(double));
  *steps_done = done;
  if (hs->stopped == 3) {
    c->run_timeouts += 1;
    // One corner: a workgroup whose wait ended in time may have begun the NEXT attempt before it ran into the missing one, and then its rows of
    // that attempt's z target -- the completed iteration's z_accel0, which only FISTA reads -- are overwritten.  z_accel0 = A x_accel0: form it again.
    if (o->accelerate) FH_TRY(op_fwd(c, 1, 0.0, c->P[c->pc], nullptr, nullptr, nullptr, nullptr, c->Z[c->zc], 0));
    FH_TRY(finish(c));
    return fail(FH_E_TIMEOUT, "fh_run: a grid barrier / team hand-off of %s timed out after %d completed iterations (workgroups not co-resident?); "
                              "the state of the last completed iteration is in place -- continue with fh_iterate / fh_step", what, done);
  }
  return 0;
}

// fh_run by a CHAIN of one-pass launches (csrc/fh_fused.h: k_fused_chain): the state goes to a device block, max_steps launches are enqueued back to
// back -- each one attempt: it reads its step size and buffer roles from the block, its finaliser runs the controller and rewrites the block; launches
// behind the stop rule, the step budget or a timeout return at once -- then the block comes back in one copy and is adopted as after the persistent launch.
static int run_chain(fh_ctx* c, int max_steps, const fh_run_opts* o, fh_run_state* state, double* history, int* steps_done) {
  FH_TRY(use_device(c));
  FH_TRY(not_lazy(c, "fh_run"));
  if (!c->run_st_host) {
    HIP_TRY(hipHostMalloc(&c->run_st_host, 2 * sizeof(ChainState), hipHostMallocMapped));     // [0]: what goes up, [1]: what comes back
    HIP_TRY(hipHostGetDevicePointer(&c->run_st_host_dev, c->run_st_host, 0));
  }
  if (!c->chain_state) HIP_TRY(hipMalloc((void**)&c->chain_state, sizeof(ChainState)));
  if ((size_t)max_steps > c->run_hist_steps) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->run_hist) { HIP_TRY(hipHostFree(c->run_hist)); c->run_hist = nullptr; c->run_hist_steps = 0; }
    const size_t steps = round_up((size_t)max_steps, 256);
    HIP_TRY(hipHostMalloc((void**)&c->run_hist, steps * FR_HIST * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&c->run_hist_dev, c->run_hist, 0));
    c->run_hist_steps = steps;
  }
  ChainState* hst = (ChainState*)c->run_st_host;
  memset(hst, 0, sizeof(ChainState));
  RunState* hs = &hst->rs;
  hs->tau_next = state->tau_next; hs->alpha1 = state->alpha1; hs->max_residual = state->max_residual; hs->best_quality = state->best_quality;
  hs->iteration = state->iteration; hs->backtracks = state->backtracks; hs->stopped = 0;
  hs->xi = c->xi; hs->ti = c->ti; hs->bi = c->bi; hs->pc = c->pc; hs->gc = c->gc; hs->zc = c->zc; hs->last_accel = c->last_accel ? 1 : 0;
  for (int q = 0; q < 5; ++q) hs->perm[q] = q;
  memcpy(hs->f_window, state->f_window, sizeof(hs->f_window));
  hst->tau_iter = state->tau_next;
  HIP_TRY(hipMemcpyAsync(c->chain_state, hst, sizeof(ChainState), hipMemcpyHostToDevice, c->stream));
  double* nb[5] = {c->X[0], c->X[1], c->X[2], c->P[0], c->P[1]};
  ChainP ch;
  for (int q = 0; q < 5; ++q) ch.nbuf[q] = nb[q];
  ch.G[0] = c->G[0]; ch.G[1] = c->G[1]; ch.Z[0] = c->Z[0]; ch.Z[1] = c->Z[1];
  ch.mu = c->mu;
  ch.o.adaptive = o->adaptive; ch.o.accelerate = o->accelerate; ch.o.backtrack = o->backtrack; ch.o.restart = o->restart;
  ch.o.evaluate_objective = o->evaluate_objective; ch.o.stop_rule = o->stop_rule; ch.o.window = o->window; ch.o.max_backtracks = o->max_backtracks;
  ch.o.stepsize_shrink = o->stepsize_shrink; ch.o.tolerance = o->tolerance;
  ch.g_kind = c->prox_kind == FH_PROX_SHRINK ? 1 : 0;
  ch.max_steps = max_steps;
  ch.st = (ChainState*)c->chain_state;
  ch.hist = c->run_hist_dev;
  // (operands the launch takes from the state block are placeholders here; xhat is the one n-side buffer whose role never changes)
  FusedIO fio = {c->X[c->xi], c->G[c->gc], c->xhat, c->P[c->pc ^ 1], c->Z[c->zc ^ 1], c->G[c->gc ^ 1], c->prox_kind, 0};
  c->seq_wait = 0;
  const bool timed = t_on(c, FH_K_FUSED);
  if (timed) t_begin(c, FH_K_FUSED);                       // ONE event pair around the chain: records between dependent launches would serialise the host with them
  ChainState* back = hst + 1;
  // One launch = one attempt, so max_steps launches complete max_steps iterations only when nothing backtracks.  The call never returns in the
  // MIDDLE of an iteration (its retries so far and the step it began with live in the device block only): while the block says "retrying", the
  // launches the remaining budget allows are enqueued on top -- the block stays where it is, only its copy comes back.  That is also what makes
  // the solve independent of the chain length (K = 1: every chain is one attempt).
  for (int enqueue = max_steps;;) {
    for (int j = 0; j < enqueue; ++j) FH_TRY(launch_fused_dense(c, 0.0, fio, &ch));
    if (timed) t_end(c, FH_K_FUSED);
    memset(back, 0xFF, sizeof(ChainState));               // (poisoned: a copy that never landed cannot pass for a state)
    HIP_TRY(hipMemcpyAsync(back, c->chain_state, sizeof(ChainState), hipMemcpyDeviceToHost, c->stream));
    if (timed) { const int keep = c->ev_cur[FH_K_FUSED]; c->ev_pending[FH_K_FUSED][keep] = false; }       // (harvested below, once, after the last round)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (back->rs.stopped != 0 || back->bt <= 0 || back->steps_done < 0 || back->steps_done >= max_steps) break;
    enqueue = max_steps - back->steps_done;
  }
  if (timed) c->ev_pending[FH_K_FUSED][c->ev_cur[FH_K_FUSED]] = true;
  FH_TRY(finish(c));
  if (timed && back->attempts > 1 && back->attempts <= max_steps) c->launches[FH_K_FUSED] += (uint64_t)(back->attempts - 1);     // the pair timed `attempts` launches (+ the no-op ones, ~4 us each)
  return run_adopt(c, o, &back->rs, nb, max_steps, state, history, steps_done, "a chained one-pass launch");
}

// How often this context recovered from an in-launch timeout: what = 0 level searches that fell back to one workgroup, 1 level searches
// that found no level (FH_E_TIMEOUT), 2 persistent launches of fh_run that ended in a barrier timeout (FH_E_TIMEOUT, state recovered).
extern "C" int fh_recovered_count(fh_ctx* c, int what, uint64_t* count) {
  if (!c || !count || what < 0 || what > 2) return fail(FH_E_ARG, "fh_recovered_count: bad argument");
  *count = 0;
  for (int k = 0; k < nshards(c); ++k) {
    fh_ctx* s = shard_of(c, k);
    if (what == 2) { *count += s->run_timeouts; continue; }
    FH_TRY(use_device(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    unsigned w[2] = {0, 0};
    HIP_TRY(hipMemcpy(w, s->counters + CNT_DIAG, sizeof(w), hipMemcpyDeviceToHost));
    *count += w[what];
  }
  return 0;
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

// RCCL's version code (ncclGetVersion), or -1 when the loaded library does not export it / none is loaded yet
extern "C" int fh_comm_version(int* version) {
  if (!version) return fail(FH_E_ARG, "null argument");
  *version = -1;
  if (g_rccl.lib && g_rccl.GetVersion) { int v = 0; if (g_rccl.GetVersion(&v) == 0) *version = v; }
  return 0;
}

// The exchange of this context -- the all-reduce of a rank's communicator, the grouped all-reduce or the in-library sum of a
// multi-device context -- run on a KNOWN pattern and checked against its closed form: row block r (0-based) contributes
// (r + 1) * (i mod 1021 + 1) at index i, so the sum over N blocks must be N (N + 1) / 2 * (i mod 1021 + 1) exactly.
// Collective on a context with a communicator (every rank calls it with the same count).  For the multi-GPU preflight
// (fasta_python_amd/preflight.py): the first contact with RCCL happens on a few KiB, not behind a 16-GiB allocation.
static __global__ void k_selftest_fill(double* v, uint64_t count, double rank1) {
  for (uint64_t i = (uint64_t)blockIdx.x * FH_WG + threadIdx.x; i < count; i += (uint64_t)gridDim.x * FH_WG) v[i] = rank1 * (double)(i % 1021u + 1u);
}
extern "C" int fh_comm_selftest(fh_ctx* c, uint64_t count, double* max_abs_err, int* nblocks) {
  if (!c || !max_abs_err || count == 0 || count > ((uint64_t)1 << 26)) return fail(FH_E_ARG, "fh_comm_selftest: bad argument");
  if (c->pending_step) return fail(FH_E_STATE, "a step issued by fh_step_begin is still in flight on this context");
  const int ns = nshards(c);
  const bool shell = !c->shards.empty();
  const int N = shell ? ns : (c->comm ? c->nranks : 1);
  int rc = 0;
  for (int k = 0; k < ns && rc == 0; ++k) {
    fh_ctx* s = shard_of(c, k);
    rc = use_device(s);
    if (rc == 0 && hipMalloc((void**)&s->selftest_buf, count * sizeof(double)) != hipSuccess) rc = fail(FH_E_STATE, "fh_comm_selftest: hipMalloc of %llu doubles failed", (unsigned long long)count);
    if (rc == 0) {
      const unsigned grid = (unsigned)std::min<uint64_t>((count + FH_WG - 1) / FH_WG, 1024);
      k_selftest_fill<<<dim3(grid), dim3(FH_WG), 0, s->stream>>>(s->selftest_buf, count, (double)((shell ? k : s->rank) + 1));
      if (hipGetLastError() != hipSuccess) rc = fail(FH_E_STATE, "fh_comm_selftest: fill launch failed");
    }
  }
  // (a rank that failed locally still enters the exchange when it can: its peers must not be left waiting)
  const bool can_exchange = [&] { for (int k = 0; k < ns; ++k) if (!shard_of(c, k)->selftest_buf) return false; return true; }();
  if (can_exchange) { const int rs = sum_over_shards(c, [](fh_ctx* s) { return s->selftest_buf; }, (size_t)count); if (rc == 0) rc = rs; }
  double worst = 0.0;
  std::vector<double> host(rc == 0 ? count : 0);
  for (int k = 0; k < ns; ++k) {
    fh_ctx* s = shard_of(c, k);
    (void)hipSetDevice(s->device);
    if (rc == 0) {
      if (hipMemcpyAsync(host.data(), s->selftest_buf, count * sizeof(double), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
          hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(FH_E_STATE, "fh_comm_selftest: reading the result back failed");
      else for (uint64_t i = 0; i < count; ++i) worst = std::max(worst, fabs(host[i] - 0.5 * N * (N + 1) * (double)(i % 1021u + 1u)));
    }
    if (s->selftest_buf) { (void)hipStreamSynchronize(s->stream); (void)hipFree(s->selftest_buf); s->selftest_buf = nullptr; }
  }
  if (rc != 0) return rc;
  *max_abs_err = worst;
  if (nblocks) *nblocks = N;
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
  // Blocks that share ONE device (a repeated device id) share one stream: two event records around EVERY launch of every block
  // cost more than the plumbing they are meant to measure (8 blocks: 34 records = 0.14 ms per iteration against 0.08 ms of
  // plumbing, profiles/r04_inproc_issue.txt).  There only ONE block's launches are timed -- a middle one: the
  // first block starts on an idle device after the host's synchronisation and reads high -- and fh_timing_get scales it by the number
  // of blocks; the sum over the blocks is one launch, timed on the first block's context as always.
  for (size_t k = 0; k < c->shards.size(); ++k) {
    c->shards[k]->timing = on != 0;
    c->shards[k]->timing_skip_kernels = c->emulated && k != c->shards.size() / 2;
  }
  c->timing = on != 0;
  return 0;
}
// a multi-device context reports the SUM over its shards of each kernel's time and launches (shards on one device run one after
// the other; on separate devices the per-launch average total_ms / launches is the mean over the devices)
extern "C" int fh_timing_get(fh_ctx* c, int k, double* total_ms, uint64_t* launches) {
  if (!c || k < 0 || k >= FH_NKERNELS) return fail(FH_E_ARG, "bad kernel id");
  if (k == FH_K_HOST_ISSUE) {
    if (total_ms) *total_ms = c->host_issue_ms;
    if (launches) *launches = c->host_issue_calls;
    return 0;
  }
  // (event pairs recorded since the last stream synchronisation are read now: a step that was waited for by its sequence number leaves its pair pending)
  if (c->shards.empty()) { (void)hipSetDevice(c->device); t_harvest(c, k, 0, true); t_harvest(c, k, 1, true); }
  for (fh_ctx* s : c->shards) { (void)hipSetDevice(s->device); t_harvest(s, k, 0, true); t_harvest(s, k, 1, true); }
  double ms = c->tot_ms[k];
  uint64_t cnt = c->launches[k];
  if (c->emulated && !c->shards.empty()) {       // one block sampled, scaled (fh_timing_enable); the sum over the blocks runs once
    const size_t ns = c->shards.size();
    fh_ctx* s = k == FH_K_COMM ? c->shards[0] : c->shards[ns / 2];
    const double scale = k == FH_K_COMM ? 1.0 : (double)ns;
    ms += s->tot_ms[k] * scale;
    cnt += (uint64_t)(s->launches[k] * scale);
  } else
  for (fh_ctx* s : c->shards) { ms += s->tot_ms[k]; cnt += s->launches[k]; }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = cnt;
  return 0;
}
extern "C" int fh_timing_reset(fh_ctx* c) {
  if (!c) return fail(FH_E_ARG, "null context");
  for (fh_ctx* s : c->shards) FH_TRY(fh_timing_reset(s));
  for (int k = 0; k < FH_NKERNELS; ++k) { c->tot_ms[k] = 0; c->launches[k] = 0; c->ev_pending[k][0] = c->ev_pending[k][1] = false; }
  c->host_issue_ms = 0.0; c->host_issue_calls = 0; c->issue_open = false;
  return 0;
}

// Did the latest timed launch of kernel k on context a and the latest one on context b (two plain contexts on one device, timing
// enabled, both launches waited for) run at the same time?  a_ms / b_ms: their durations; overlap_ms: the length of the interval
// both were running (<= 0: one had ended before the other began), from the HIP events that bracket each launch on its own stream.
extern "C" int fh_timing_overlap(fh_ctx* a, fh_ctx* b, int k, double* a_ms, double* b_ms, double* overlap_ms) {
  if (!a || !b || k < 0 || k >= FH_NKERNELS || k == FH_K_HOST_ISSUE) return fail(FH_E_ARG, "fh_timing_overlap: bad argument");
  if (!a->shards.empty() || !b->shards.empty() || a->device != b->device) return fail(FH_E_ARG, "fh_timing_overlap takes two plain contexts on one device");
  FH_TRY(use_device(a));
  // the LATEST recorded pair of each context (its launch has been waited for; the pair itself may not have been read yet)
  const int sa = a->ev_cur[k], sb = b->ev_cur[k];
  if (a->pending_step || b->pending_step || (!a->launches[k] && !a->ev_pending[k][sa]) || (!b->launches[k] && !b->ev_pending[k][sb]))
    return fail(FH_E_STATE, "fh_timing_overlap: both contexts need a completed, timed launch of kernel %d", k);
  HIP_TRY(hipEventSynchronize(a->ev[k][sa][1]));
  HIP_TRY(hipEventSynchronize(b->ev[k][sb][1]));
  // signed time from event e1 to event e2 (whichever way round the runtime is willing to subtract them)
  auto between = [](hipEvent_t e1, hipEvent_t e2, float* ms) -> hipError_t {
    hipError_t r = hipEventElapsedTime(ms, e1, e2);
    if (r != hipSuccess) { (void)hipGetLastError(); r = hipEventElapsedTime(ms, e2, e1); *ms = -*ms; }
    return r;
  };
  float da = 0.f, db0 = 0.f, db1 = 0.f;               // everything relative to the start of a's launch
  HIP_TRY(between(a->ev[k][sa][0], a->ev[k][sa][1], &da));
  HIP_TRY(between(a->ev[k][sa][0], b->ev[k][sb][0], &db0));
  HIP_TRY(between(a->ev[k][sa][0], b->ev[k][sb][1], &db1));
  if (a_ms) *a_ms = da;
  if (b_ms) *b_ms = db1 - db0;
  if (overlap_ms) *overlap_ms = std::min(da, db1) - std::max(0.f, db0);
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

#include "fh_host_iterate.h"
