/* fasta_hip.h -- C ABI of libfasta_hip.so: the MI355X (gfx950) forward-backward-splitting hot path.
 *
 * The reference (phasepack/fasta-python) is pure Python and has no FFI; its plugin boundary for this
 * path is the Python call fasta.fasta(A[, At], f, gradf, g, proxg, x0, ...) (fasta/__init__.py:38-53).
 * This header is the boundary a binding for that call sits on: plain pointers and sizes, no torch
 * types.  Each entry point cites the reference arithmetic it replaces (paths relative to the
 * reference tree).  The ctypes binding that consumes it is fasta_python_amd/hip.py; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, otherwise a non-zero code (hipError_t / ncclResult_t
 *     value, or FH_E_*); fh_last_error() then describes it (thread-local string).
 *   - a context is NOT thread-safe; calls are synchronous from the caller's view (fh_fwd / fh_adj
 *     return once their scalars are on the host).  A context drives one device (fh_create), or --
 *     fh_create_ex with ndev > 1 -- several row blocks of A in ONE process from one host thread.
 *   - the library copies on every set_* and never keeps a host pointer past the call; it owns all
 *     device memory behind fh_ctx until fh_destroy.
 *   - all arithmetic is IEEE float64 (the reference is float64 throughout).
 */
#ifndef FASTA_HIP_H
#define FASTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fh_ctx fh_ctx;

#define FH_E_ARG      10001   /* bad argument / shape mismatch                      */
#define FH_E_STATE    10002   /* call out of order (e.g. fh_fwd before fh_set_*)     */
#define FH_E_RCCL     10003   /* librccl could not be loaded / symbol missing        */
#define FH_E_TIMEOUT  10004   /* a bounded in-launch hand-off ran out (workgroups not co-resident?) and no fall-back inside the library
                                 could finish the call: the clipping-level search of FH_PROX_LINF / FH_PROX_L1BALL (its outputs are NaN,
                                 never a finite wrong prox), or a grid barrier of fh_run's persistent launch (the state of the last
                                 completed iteration is adopted and the completed history returned: continue with fh_iterate / fh_step) */

/* prox operators g(x) <-> proxg(x, t)                                                   */
enum fh_prox_kind {
  FH_PROX_IDENTITY = 0,  /* g = None branch, fasta/__init__.py:88-90                     */
  FH_PROX_SHRINK   = 1,  /* proximal.shrink(x, t*mu), fasta/proximal.py:58-67            */
  FH_PROX_NONNEG   = 2,  /* np.maximum(x, 0), examples/nn_least_squares.py:42            */
  FH_PROX_LINF     = 3,  /* proximal.project_Linf_ball(x, t*mu), fasta/proximal.py:12-31 */
  FH_PROX_L1BALL   = 4,  /* proximal.project_L1_ball(x, mu), fasta/proximal.py:34-41     */
  FH_PROX_TVBALL   = 5,  /* per-pixel 2-vector / max(norm,1), examples/tv_denoising.py:89-96 */
  FH_PROX_BOX      = 6   /* clip to [lo, hi], examples/svm.py:71                         */
};

/* device vectors addressable through fh_set_vector / fh_get_vector                      */
enum fh_vec {
  FH_VEC_X0 = 0,    /* current iterate x0 (n)            fasta/__init__.py:176            */
  FH_VEC_G0 = 1,    /* gradient at x0: A^H grad f(A x0)  :177                             */
  FH_VEC_XHAT = 2,  /* forward point x0 - tau*g0         :181                             */
  FH_VEC_XPROX = 3, /* prox output (pre-acceleration x1) :184                             */
  FH_VEC_X1 = 4,    /* next iterate (post-acceleration)  :242                             */
  FH_VEC_G1 = 5,    /* next gradient                     :248                             */
  FH_VEC_BEST = 6,  /* best-quality iterate so far       :298-300                         */
  FH_VEC_B = 7,     /* least-squares target b (m)        examples/sparse_least_squares.py:41 */
  FH_VEC_Z = 8,     /* z1 = A xprox (m)                  :187                             */
  FH_VEC_T0 = 9, FH_VEC_T1 = 10, FH_VEC_T2 = 11, FH_VEC_T3 = 12   /* n-length scratch     */
};

/* layout of the scalar block written by fh_init / fh_fwd / fh_adj (FH_NSCALARS doubles)  */
enum fh_scalar {
  FH_S_FSQ = 0,     /* ||z1 - b||^2 ; f1 = .5*sqrt(.)**2      :188                        */
  FH_S_DXG0 = 1,    /* <Dx, g0>                                 :200                        */
  FH_S_DX2 = 2,     /* ||Dx||^2 , Dx = xprox - x0               :200, :258, :272            */
  FH_S_XH2 = 3,     /* ||xprox - xhat||^2 (normaliser, no accel):274                        */
  FH_S_G02 = 4,     /* ||g0||^2                                 :274                        */
  FH_S_GSUM = 5,    /* sum |xprox_i|   (g for shrink = mu*this) examples/sparse_least_squares.py:43 */
  FH_S_GMAX = 6,    /* max |xprox_i|   (g for linf  = mu*this)  examples/democratic_representation.py:41 */
  FH_S_RDOT = 7,    /* (x0 - xprox).(xprox - xacc0)  restart    :231                        */
  FH_S_DXDG = 8,    /* <Dx, Dg>, Dg = g1 + (xhat - x0)/tau      :254-255                    */
  FH_S_DG2 = 9,     /* ||Dg||^2                                 :260                        */
  FH_S_FSQ_ADJ = 10,/* ||z1' - b||^2 at the extrapolated z1'    :243-245                    */
  FH_S_XH2_ADJ = 11,/* ||x1 - xhat||^2 with extrapolated x1     :274                        */
  FH_S_GSUM_ADJ = 12, FH_S_GMAX_ADJ = 13,   /* g terms at the extrapolated x1 :285           */
  FH_S_ALPHA = 14,  /* threshold level used by LINF / L1BALL prox (diagnostic)              */
  FH_NSCALARS = 16
};

/* FH_K_HOST_ISSUE is not a kernel: fh_timing_get reports under it the HOST time a one-pass step of the dense operator (fh_step /
 * fh_step_accel) spends issuing its launches and exchanges -- from the call's entry to the start of its one final synchronisation --
 * and the number of such calls: what one host thread pays to drive all the row blocks of a multi-device context per iteration.   */
/* FH_K_LEVEL: the clipping-level search that precedes every forward launch of the LINF / L1BALL prox kinds (csrc/fh_prox.h). */
enum fh_kernel_id { FH_K_FWD = 0, FH_K_ADJ = 1, FH_K_AUX = 2, FH_K_COMM = 3, FH_K_FUSED = 4, FH_K_HOST_ISSUE = 5, FH_K_LEVEL = 6, FH_NKERNELS = 7 };

enum fh_tuning_key {
  FH_TUNE_FWD_ROWS = 0,      /* rows per workgroup pass in K-fwd: 4, 8, 16 (0 = auto)        */
  FH_TUNE_FWD_GRID_CAP = 1,  /* max workgroups of K-fwd (0 = auto: 1024, grid-stride over row groups) */
  FH_TUNE_ADJ_SLAB_ROWS = 2, /* rows per K-adj slab (multiple of 8; 0 = auto)               */
  FH_TUNE_ADJ_CPT = 3,       /* 16-byte column pairs per thread in K-adj: 1, 2, 4 (0 = auto) */
  FH_TUNE_LD_PAD = 4,        /* extra doubles appended to each device row of A (multiple of 16; set before the matrix) */
  FH_TUNE_NT_LOADS = 5,      /* K-fwd / K-adj / the device loop: 1 = stream A with non-temporal loads, 0 = default cache policy; unset = auto: plain
                                loads for a matrix of at most 256 MiB (it stays in the last-level cache between launches), non-temporal above  */
  FH_TUNE_TV_U = 6,          /* stencil kernels: rows of loads per trip and lane (2, 4, 8; 0 = auto)  */
  FH_TUNE_TV_ROWS = 7,       /* stencil kernels: image rows per workgroup (0 = auto)                 */
  FH_TUNE_TV_NT = 8,         /* stencil kernels: 0 = default, 1 = non-temporal loads and stores (two-launch kernels and the z-streaming
                                one-pass kernels; the z-free one-pass sweep never loads non-temporally: its halo columns and rows are
                                re-read through L2), 2 = non-temporal stores (the z-free sweep's default), 3 = plain accesses      */
  FH_TUNE_FUSED_VARIANT = 9, /* fused one-pass kernel: scheduling variant bits (see csrc/fh_fused.h: 2 = team members on one XCD, 4 = no
                                sleep between polls, 8 / 16 = A/B team shapes, 32 = rows dealt cyclically; default 2 | 32); other bits are FH_E_ARG */
  /* key 10 (the round-1 one-pass stencil kernels that stream z) is only present in -DFH_EXPERIMENTAL builds                    */
  FH_TUNE_TV_PIPE = 11,      /* z-free one-pass stencil sweep: 1 = load a trip of FH_TUNE_TV_U rows, consume it; 3 (2 is taken as 3) = three
                                rotating trip buffers (two trips of loads stay in flight behind the one being consumed); 0 = auto   */
  FH_TUNE_TV_XCD = 12,       /* z-free one-pass stencil sweep: deal the workgroup ids out XCD by XCD, so that strips that share halo
                                cache lines share an L2 (0 = auto = 1 = on, 2 = off: plain blockIdx order)                    */
  /* keys 13-15 (occupancy limiter, LDS-DMA trip ring, persistent chunk walk of the stencil sweep: measured flat twice,
     profiles/r04_tune_tv.txt) are only present in -DFH_EXPERIMENTAL builds (csrc/fh_experimental.h); FH_E_ARG otherwise        */
  FH_TUNE_RUN_MAX_N = 17,    /* fh_run: the widest row (columns) the device-side loop is offered for (fh_run_supported), whatever the number of
                                rows; 0 = the measured window (profiles/r06_device_loop.txt): n <= 4096 up to 32 Mi elements, n <= 6144 from 4096 rows
                                on up to 40 Mi elements -- elsewhere one launch per iteration issued by fh_iterate is as fast or faster; at most
                                7168 = the widest row it has a kernel for                                                          */
  FH_TUNE_RUN_CHAIN = 19,    /* 0 (default) / 1: outside the persistent launch's window fh_run takes the CHAINED form where it exists (float64, n <= 16384,
                                separable prox: max_steps one-pass launches enqueued back to back, step size and buffer roles from a device state
                                block, the loop's controller in each launch's finaliser).  Opt-in: measured equal to the host-side loop
                                (profiles/r06_chain.txt) -- the gap between launches disappears, the launches grow by as much             */
  FH_TUNE_ADJ_CYCLIC = 20,   /* K-adj: 1 = the rows are dealt cyclically to the slabs (slab s: rows s, s + nslab, ...), as bit 32 of FH_TUNE_FUSED_VARIANT does for
                                the one-pass kernel; 0 = auto = 2 = contiguous slabs.  Measured mixed (faster on small and mid shapes, 6 % slower at
                                65536^2; profiles/r06_placement.txt), so it stays an A/B switch                                                   */
  FH_TUNE_SEQ_POLL = 18,     /* 1 (default): a single-device step waits for its scalar block by the sequence number the launch writes behind
                                it into host-mapped memory (~5 us sooner than the launch's completion signal); 0: hipStreamSynchronize (A/B)  */
  FH_TUNE_FUSED_CUS = 16     /* dense one-pass kernel: launch it on at most this many CUs (one workgroup each; 0 = every CU the device
                                reports).  The co-residency probe then asks for that many.  Lets several one-pass grids run side by side
                                on one device: two solves at once, partitioned devices, ranks of a row-sharded run that share a GPU
                                (the tests: world x (CUs / world)).  The team shape needs a multiple of the team size (<= 32)        */
};

/* ---- library / context -------------------------------------------------------------- */
const char* fh_last_error(void);
int fh_device_count(int* count);
int fh_create(int device, fh_ctx** out);
/* SURVEY.md 8(b) form: device list + storage type of A.
 *   ndev == 1: a plain single-device context (what fh_create returns).
 *   ndev  > 1: IN-PROCESS ROW SHARDING (the op being sharded is `A @ x` / `A.T @ x`, fasta/linalg.py:41).  The
 *     context owns one shard per entry of dev_ids; fh_set_matrix / fh_set_matrix_f32 / fh_generate_matrix split A into contiguous
 *     row blocks (the first m mod ndev shards hold ceil(m/ndev) rows, the rest floor(m/ndev)), fh_set_loss_* and the m-side
 *     vectors (FH_VEC_B, FH_VEC_Z) are split the same way, n-side vectors are replicated.  Every solver step then runs as: local
 *     launch on each shard -> ONE sum over the shards -> n-side epilogue on each shard -> one host synchronisation; the scalar
 *     block is identical on all shards and is returned from shard 0.  The call sequence, arguments and results of every other
 *     entry point are those of a plain context (fh_set_stencil and fh_comm_* refuse: dense operator only, and the rows are
 *     already sharded).
 *       all dev_ids different: one GPU per shard; RCCL communicators from ncclCommInitAll, the sum is one grouped
 *                              ncclAllReduce(n + 3 doubles) per shard over xGMI;
 *       all dev_ids equal:     every shard on that one GPU, on one stream; the sum is an in-library kernel that adds the shards'
 *                              buffers in shard order -- same arithmetic structure, runnable (and tested) on a one-GPU box.
 *     (a mixture is FH_E_ARG.)  ndev <= 64.
 * dtype FH_DTYPE_F32_STORAGE keeps the device copy of A in float32 (rounded to nearest on
 * upload / generation): half the bytes per pass; every vector, accumulation and scalar stays float64.  OPT-IN: the iterates
 * are those of the reference run on the ROUNDED matrix, i.e. they differ from the float64-matrix run by the rounding of A
 * (relative 6e-8 per entry; SURVEY.md section 7: <= 3e-7 on the iterates away from the chaotic regime).                  */
enum fh_dtype { FH_DTYPE_F64 = 0, FH_DTYPE_F32_STORAGE = 1 };
/* or'ed into dtype: use the RCCL form of the multi-device context (ncclCommInitAll, grouped ncclAllReduce) where the default would
 * not: with ndev == 1 (one shard, a 1-rank communicator: same results as a plain context), or with a repeated device id (needs an
 * RCCL that accepts several ranks on one device -- real RCCL refuses, the tests' stand-in accepts).  Lets a one-GPU box run the
 * RCCL branch.                                                                                                               */
#define FH_CREATE_RCCL_SHELL 0x100
int fh_create_ex(int ndev, const int* dev_ids, int dtype, fh_ctx** out);
/* row blocks of a context (1 for a plain one); fh_shard lends shard k -- a complete single-device context that stays owned by
 * `ctx` -- with the rows of the whole operator it holds: [row0, row0 + rows).  For diagnostics and tests (e.g. reading a
 * replicated vector from every shard).                                                                                   */
int fh_shard_count(fh_ctx* ctx, int* count);
int fh_shard(fh_ctx* ctx, int k, fh_ctx** shard, uint64_t* row0, uint64_t* rows);
int fh_destroy(fh_ctx* ctx);
int fh_sync(fh_ctx* ctx);
int fh_set_tuning(fh_ctx* ctx, int key, long long value);

/* ---- operator A (replaces LinearMap.from_matrix closures `A @ x`, `A.T @ x`, fasta/linalg.py:37-41) */
/* dense row-major host matrix, m rows, n columns, leading dimension ld_host (doubles).          */
int fh_set_matrix(fh_ctx* ctx, const double* A, uint64_t m, uint64_t n, uint64_t ld_host);
/* float32 host matrix into a float32-storage context (fh_create_ex, FH_DTYPE_F32_STORAGE): copied as is, no rounding step  */
int fh_set_matrix_f32(fh_ctx* ctx, const float* A, uint64_t m, uint64_t n, uint64_t ld_host);
/* synthetic rows [row0, row0+m) of a (.., n) matrix: element (i,j) = ihall(seed, (row0+i)*n + j) * coef
 * (device twin of oracle/problems.py:synth_values).                                            */
int fh_generate_matrix(fh_ctx* ctx, uint64_t m, uint64_t n, uint64_t row0, uint64_t seed, double coef);
int fh_get_matrix_rows(fh_ctx* ctx, uint64_t row0, uint64_t nrows, double* out /* nrows*n */);
/* periodic difference stencil pair: A = div: (H,W,2)->(H,W), A^H = grad (examples/tv_denoising.py:26-63) */
int fh_set_stencil(fh_ctx* ctx, uint64_t H, uint64_t W);
int fh_shape(fh_ctx* ctx, uint64_t* m, uint64_t* n);

/* ---- smooth term f(z) = .5||z - b||^2, grad f(z) = z - b (examples/sparse_least_squares.py:41-42) */
int fh_set_loss_lsq(fh_ctx* ctx, const double* b, uint64_t len);
/* ---- smooth term f(z) = sum log(1+exp(z)) - (b==1)*z, grad f(z) = -b/(1+exp(b*z)), labels b in {-1,+1}
 *      (examples/sparse_logistic.py:47-48); FH_S_FSQ / FH_S_FSQ_ADJ then carry f itself. Dense operator only.   */
int fh_set_loss_logistic(fh_ctx* ctx, const double* labels, uint64_t len);
/* ---- prox term (kinds above); mu as in the closures, lo/hi for FH_PROX_BOX only              */
int fh_set_prox(fh_ctx* ctx, int kind, double mu, double lo, double hi);

int fh_set_vector(fh_ctx* ctx, int which, const double* host, uint64_t len);
int fh_get_vector(fh_ctx* ctx, int which, double* host, uint64_t len);

/* ---- solver steps -------------------------------------------------------------------- */
/* fasta/__init__.py:132-137: z1 = A x0, f1 = f(z1), g0 = A^H grad f(z1); arms the acceleration
 * state (x_accel1 = x0, z_accel1 = z1, :154-157).  scalars: FH_S_FSQ, FH_S_GSUM, FH_S_GMAX.       */
int fh_init(fh_ctx* ctx, double* scalars);
/* The whole set-up of a solve in one call (fasta/__init__.py:100-113 Lipschitz probes + :135-137 initial pass): with the two probes
 * in FH_VEC_T0 / FH_VEC_T1 and x0 in FH_VEC_X0 it leaves the state fh_init leaves and returns fh_init's scalars plus
 * FH_S_DG2 = ||A^H grad f(A T0) - A^H grad f(A T1)||^2 and FH_S_DX2 = ||T0 - T1||^2 (L = sqrt of their quotient, :110).
 * FH_VEC_T2 / FH_VEC_T3 are scratch afterwards.  A dense least-squares operator with n <= 65536 (either storage) on a single-device context
 * is read ONCE for the whole set-up (csrc/fh_setup.h: two dot products and two rank-1 updates per row buffer -- x0, and the probes'
 * difference, since grad(T0) - grad(T1) = A^T A (T0 - T1) there); ROW BLOCKS (a multi-device context, a rank with a communicator) read
 * their block once each and sum the two gradients, the loss sums and the timeout words in one exchange -- a rank first settles the decision
 * with its peers: fh_setup is COLLECTIVE on a context with a communicator; every other operator takes the three passes fh_gradient_at x 2 +
 * fh_init inside the call.  z, f, g0 are bit-identical either way; the norm of the gradient difference agrees to ~1e-15 relative.  */
int fh_setup(fh_ctx* ctx, double* scalars);
/* dst = A^H grad f(A src) for n-length device vectors (Lipschitz probes, fasta/__init__.py:106-107) */
int fh_gradient_at(fh_ctx* ctx, int src_vec, int dst_vec);
/* ||a - b||_2 of two n-length device vectors (fasta/__init__.py:110)                             */
int fh_diff_norm(fh_ctx* ctx, int vec_a, int vec_b, double* out);
/* K-fwd, one launch: xhat = x0 - tau*g0; xprox = prox(xhat, tau); z1 = A xprox; reductions
 * FH_S_FSQ..FH_S_RDOT (fasta/__init__.py:181-188, 200, 231, 272-274).  May be called repeatedly
 * with a smaller tau (backtracking, :204-213).                                                   */
int fh_fwd(fh_ctx* ctx, double tau, double* scalars);
/* K-adj, one launch (+ one RCCL all-reduce when row-sharded): z1' = z1 + coef*(z1 - z_accel0),
 * x1 = xprox + coef*(xprox - x_accel0) (coef = 0 and accel = 0 without acceleration, :242-243);
 * g1 = A^H (z1' - b) (:248); reductions FH_S_DXDG..FH_S_GMAX_ADJ (:254-260, 274, 285).          */
int fh_adj(fh_ctx* ctx, double tau, int accel, double coef, double* scalars);
/* fh_fwd followed by fh_adj (no acceleration) under ONE synchronisation: the complete FH_S_* block comes back in one round
 * trip.  For short launches, where the host round trip dominates; the caller speculates on the step being accepted.      */
int fh_fwd_adj(fh_ctx* ctx, double tau, double* scalars);
/* ONE-PASS iteration (dense operator, no acceleration): K-fwd and K-adj of the same tau from a SINGLE read of A
 * (teams of 1 to 16 co-resident workgroups cover a row and exchange partial dot products; see csrc/fh_fused.h).  Writes the complete
 * FH_S_* block (both halves); scalars[15] != 0 reports a bounded-spin timeout (results invalid).  The caller uses it
 * speculatively: if the backtracking test on FH_S_FSQ fails it re-runs fh_fwd (smaller tau) + fh_adj.
 * fh_fused_supported: 0 = no (n > 262144, TV prox on a dense operator); 1 = dense, recommended (n >= 16384 or at least
 * 8 Mi elements); 3 = dense, available but no faster than two short launches; 2 = stencil operator (one sweep replaces both). */
int fh_fused_supported(fh_ctx* ctx, int* yes);
/* fh_fused_supported is a purely LOCAL query: safe to call from one rank alone.  The ranks of a row-sharded run (one process per GPU)
 * must all take the same kind of step, so they settle on ONE verdict with fh_fused_agree: a COLLECTIVE call -- every rank of the
 * communicator calls it at the same point (FBSolver.setup does) -- that sums the ranks' "0" and "3" verdicts; any 0 makes it 0
 * everywhere, else any 3 makes it 3.  A rank whose local query failed still enters the exchange (contributing "0") before it
 * returns its error, so its peers are never left waiting.  Without a communicator it equals fh_fused_supported.               */
int fh_fused_agree(fh_ctx* ctx, int* yes);
/* The dense one-pass kernel needs one workgroup on every compute unit at the same time.  The library checks that once per context
 * with a ~20 us probe launch (fh_fused_supported then reports 0 for the dense operator if CUs are hidden by a mask, a partition
 * mode or a co-tenant); this entry runs the same probe for `workgroups` whole-CU workgroups and reports whether they all ran side
 * by side (at most 2 ms when they cannot).                                                                                  */
int fh_coresident_probe(fh_ctx* ctx, int workgroups, int* ok);
/* Which variant of the one-pass kernel serves rows of n columns -- shape5 = {pieces per lane, posting distance, team members,
 * x slice in LDS, row buffers} (all 0: no one-pass kernel for this width) -- and whether that variant is compiled into the
 * library (csrc/fh_fused_instances.inc).  A pure host function: no device needed.  dtype: fh_dtype; variant: FH_TUNE_FUSED_VARIANT
 * bits; ncu: compute units (256 on MI355X).                                                                                */
int fh_fused_shape(uint64_t n, int dtype, int variant, int ncu, int* shape5, int* instantiated);
int fh_step(fh_ctx* ctx, double tau, double* scalars);
/* fh_step in two halves (the loop body of fasta/__init__.py:171-188 issued now, waited for later): fh_step_begin enqueues the launch
 * on the context's stream and returns at once; fh_step_end waits and delivers the scalar block.  In between the host may drive
 * OTHER contexts -- two solves side by side on one device (FH_TUNE_FUSED_CUS each) from one host thread.  Every other entry point
 * of `ctx` returns FH_E_STATE until fh_step_end has been called.                                                              */
int fh_step_begin(fh_ctx* ctx, double tau);
int fh_step_end(fh_ctx* ctx, double* scalars);
/* ONE-PASS iteration with acceleration (fasta/__init__.py:220-248; dense operator, also row-sharded): as fh_step, plus
 * x1 = xprox + c*(xprox - x_accel0) and the gradient taken at z1 + c*(z1 - z_accel0) with c = coef, or 0 when restart != 0
 * and this step's restart dot <x0 - xprox, xprox - x_accel0> (:231) exceeds 1e-30; the dot is returned in FH_S_RDOT and
 * f at the extrapolated point in FH_S_FSQ_ADJ, so the caller updates alpha exactly as after fh_fwd + fh_adj.
 * Stencil operator: one sweep as well (csrc/fh_tv.h, k_fused_tv_accel) -- it carries both candidates of the coefficient and
 * keeps the extrapolated iterate in (prox output, previous prox output, coefficient) form, formed on the fly by the next
 * sweep; a solve that uses it must use it from the first iteration after fh_init (FH_E_STATE otherwise, and fh_fwd /
 * fh_adj / fh_step refuse until the next fh_init).  fh_get_vector(X0 | X1 | BEST) materialises the iterate.              */
int fh_step_accel(fh_ctx* ctx, double tau, double coef, int restart, double* scalars);
/* ---- the loop itself on the device (opt-in: fasta(..., device_iters=K)) -----------------------------------------------------------
 * fh_run executes up to max_steps FBS iterations -- fasta/__init__.py:171-312: forward step, prox, both matvecs, the non-monotone
 * backtracking test with its retries (:195-217), FISTA restart and alpha recursion (:220-238), Barzilai-Borwein step (:253-270),
 * residuals, best iterate (:272-300) and ONE OF THE FOUR BUILT-IN stop rules (fasta/stopping.py:6-51) -- in ONE persistent launch
 * (csrc/fh_run.h), and returns the iterations' histories in one block.  For short launches, where the fixed cost of a launch and the
 * host round trip between two launches dominate (a workgroup owns whole rows: n <= 6144).  Arithmetic and decisions are those of the
 * per-iteration path (fh_step + fh_iterate), so iteration and backtrack counts are the same and histories agree to rounding.
 *   opts    the options of fasta() the loop reads; stop_rule: 0 residual, 1 norm_residual, 2 ratio_residual, 3 hybrid_residual
 *   state   in/out, carried from call to call: after fh_init / fh_setup set tau_next = tau0, alpha1 = 1, max_residual = -inf,
 *           best_quality = +inf, iteration = backtracks = 0, f_window[0] = f(x0) (f_window[j % 64] holds f_hist[j]; window <= 64)
 *   history max_steps records of FH_RUN_HIST doubles: residual, norm_residual, stepsize, f_hist[i+1], objective, backtracks of the
 *           iteration, alpha0, 1 if the iterate became the best one (+ 2 if the acceleration was restarted, :231-233)
 *   steps_done  iterations executed (fewer than max_steps when the stop rule fired: state->stopped = 1)
 * fh_run_supported: 1 if this context's operator, loss and prox have a kernel for it (dense float64 operator inside the measured window of FH_TUNE_RUN_MAX_N -- n <= 6144, at most 32-40 Mi elements -- on a
 * single-device context, a scalar-separable prox without level search, every CU free for one resident workgroup).                 */
#define FH_RUN_HIST 8
#define FH_RUN_WINDOW_MAX 64
/* how fh_iterate's forward launches reach the device (fh_run ignores it).  The caller settles it once per solve -- over the ranks of a
 * row-sharded run with fh_fused_agree -- from fh_fused_agree's verdict: 1 / 2 -> ALWAYS, 3 -> SPECULATIVE, else PAIR (small dense
 * operator, no acceleration) or SEPARATE.  A policy changes which launches produce the results, never the results.                    */
enum fh_launch_mode {
  FH_LAUNCH_SEPARATE = 0,            /* fh_fwd, decide, fh_adj                                                                      */
  FH_LAUNCH_ONEPASS_ALWAYS = 1,      /* fh_step / fh_step_accel for every launch of the loop, backtracking retries included          */
  FH_LAUNCH_ONEPASS_SPECULATIVE = 2, /* fh_step / fh_step_accel, except for 8 iterations after a backtrack and in the retries         */
  FH_LAUNCH_PAIR = 3                 /* fh_fwd_adj (one synchronisation), same exceptions; no acceleration                            */
};
typedef struct fh_run_opts {
  int adaptive, accelerate, backtrack, restart, evaluate_objective, stop_rule, window, max_backtracks;
  double stepsize_shrink, tolerance;
  int launch_mode;                   /* fh_iterate only: enum fh_launch_mode                                                        */
  int reserved;
} fh_run_opts;
typedef struct fh_run_state {
  double tau_next, alpha1, max_residual, best_quality;
  uint64_t iteration, backtracks;
  int stopped, reserved;             /* stopped: 0 = ran out of steps, 1 = the stop rule fired, 3 = fh_run's launch timed out (FH_E_TIMEOUT) */
  double f_window[FH_RUN_WINDOW_MAX];
  /* fh_iterate only -- the launch policy's memory, carried from call to call like the rest (set spec_cooldown = 0, onepass_off_until = -1,
   * onepass_backoff = 64 and the three counters to 0 before the first call):                                                       */
  int spec_cooldown;                 /* iterations left before the one-pass kernel / the pair is speculated on again after a backtrack */
  int onepass_backoff;               /* iterations to stay off the one-pass kernel after its NEXT hand-off timeout (doubles per failure) */
  int64_t onepass_off_until;         /* -1, or the iteration from which the one-pass kernel is tried again after a hand-off timeout    */
  uint64_t onepass_launches, pair_launches, onepass_timeouts;   /* totals: one-pass launches that delivered, pairs, hand-off timeouts  */
} fh_run_state;
/* sizeof(fh_run_opts), sizeof(fh_run_state), FH_RUN_HIST, FH_RUN_WINDOW_MAX as THIS build of the library sees them: a binding checks its own
 * struct layouts against them before the first call (fasta_python_amd/hip.py:load_library does).  No device needed.                           */
int fh_abi_sizes(uint64_t sizes[4]);
int fh_run_supported(fh_ctx* ctx, int* yes);
/* After a grid-barrier timeout of the persistent launch (workgroups not co-resident) fh_run returns FH_E_TIMEOUT with state->stopped = 3:
 * the context and `state` are those of the last COMPLETED iteration (state->tau_next = the step the interrupted iteration started with),
 * `history` / `steps_done` describe the completed ones -- the caller continues with fh_iterate or fh_step.  Any other failure: FH_E_STATE. */
int fh_run(fh_ctx* ctx, int max_steps, const fh_run_opts* opts, fh_run_state* state, double* history, int* steps_done);
/* The same loop driven from the HOST side of the library (csrc/fh_host_iterate.h), for EVERY operator, loss, prox and sharding form the
 * step entry points serve -- a dense matrix of any width, the level-search prox kinds, the stencil, float32 storage, row blocks in this
 * process, a rank of a row-sharded run (every rank calls it with the same arguments).  Per iteration it issues what the Python driver of
 * round 1-5 issued (fh_step / fh_step_accel / fh_fwd / fh_adj / fh_fwd_adj by opts->launch_mode, then fh_commit), synchronises once and takes
 * the reference's decisions (fasta/__init__.py:195-217, :220-238, :253-270, :272-300; fasta/stopping.py:6-51) in float64 exactly as that
 * driver does -- results are bit-identical to it -- without the 17-19 us of interpreter time per iteration.  Arguments, state, history
 * records (record[7] = 1 if the iterate became the best one, + 2 if the acceleration was restarted) and steps_done as fh_run.  On an
 * error the completed iterations stay committed and are reported (state, history, steps_done); the failed one has changed nothing.   */
int fh_iterate(fh_ctx* ctx, int max_steps, const fh_run_opts* opts, fh_run_state* state, double* history, int* steps_done);
/* how often this context got over an in-launch timeout by itself: what = 0 clipping-level searches finished by the single-workgroup
 * fall-back, 1 = searches that found no level (reported as FH_E_TIMEOUT), 2 = fh_run launches that ended in a barrier timeout      */
int fh_recovered_count(fh_ctx* ctx, int what, uint64_t* count);
/* x0 <- x1, g0 <- g1, acceleration history rotates (:176-177, :222-226); save_best != 0 also
 * copies x1 into FH_VEC_BEST (:298-300).                                                         */
int fh_commit(fh_ctx* ctx, int save_best);
/* plain operator application on host vectors (tests, synthetic b): adjoint=0: out(m) = A in(n);
 * adjoint=1: out(n) = A^H in(m).                                                                 */
int fh_apply(fh_ctx* ctx, int adjoint, const double* in, double* out);

/* ---- row sharding across PROCESSES: one process per GPU, RCCL communicator by rank (the in-process form is fh_create_ex
 *      with ndev > 1; the two are not combined: these refuse a multi-device context, except fh_comm_count, which reports
 *      its number of shards) ------------------------------------------------------------------------------------------ */
/* writes an ncclUniqueId (128 bytes) -- rank 0 calls it and ships the bytes to the other ranks     */
int fh_comm_unique_id(void* id128);
int fh_comm_init(fh_ctx* ctx, int nranks, int rank, const void* id128);
/* ranks of the attached communicator as RCCL reports them (ncclCommCount); 1 without a communicator */
int fh_comm_count(fh_ctx* ctx, int* nranks);
int fh_comm_destroy(fh_ctx* ctx);
/* Large frees and the allocations behind them.  The driver clears freed device memory in the background (~30 ms per GiB on MI355X); a large
 * allocation made meanwhile is mapped less favourably for its whole lifetime.  The one-pass kernel of rounds 1-5 (rows dealt to the teams in blocks) ran
 * 9-14 % slow on such a matrix (profiles/r06_alloc_settle.txt: 6 of 6 cycles); with the rows dealt cyclically (default since round 6) it does not care
 * (profiles/r06_placement.txt: 3 of 3).  Per device:
 *   - the matrix block (>= 1 GiB) a context gives up (fh_destroy, a new fh_set_matrix / fh_generate_matrix / fh_set_stencil) is KEPT, one block
 *     per device, and handed to the next matrix on that device that fits it and fills at least half of it: no clearing, no new mapping, no
 *     waiting.  The memory stays allocated until then: fh_release_cached(device | -1 = all) returns it to the driver, fh_alloc_cache(0)
 *     switches the keeping off (and releases), fh_alloc_cache_hits counts the re-uses;
 *   - fh_alloc_settle(1) (default 0 since the cyclic dealing): a matrix of >= 1 GiB that no kept block serves is allocated only after the device's
 *     earlier large frees have presumably been cleared (35 ms per GiB behind the free) -- for callers that select the blocked dealing or live on the
 *     two-launch path, whose K-adj still depends on the mapping; fh_alloc_settle_waited returns the seconds spent in it.                     */
int fh_alloc_settle(int enable);
int fh_alloc_settle_waited(double* seconds);
int fh_alloc_cache(int enable);
int fh_release_cached(int device);
int fh_alloc_cache_hits(uint64_t* hits);

/* RCCL's version code as ncclGetVersion reports it; -1 when no library is loaded yet or it does not export the symbol           */
int fh_comm_version(int* version);
/* The context's exchange (the rank's all-reduce, or a multi-device context's grouped all-reduce / in-library sum) on a known
 * pattern of `count` doubles, checked against its closed-form sum: max_abs_err must come back 0, nblocks = the ranks / row blocks
 * summed over.  COLLECTIVE on a context with a communicator.  The multi-GPU preflight's first contact with RCCL
 * (fasta_python_amd/preflight.py); the op being sharded is `A.T @ x`, fasta/linalg.py:41.                                      */
int fh_comm_selftest(fh_ctx* ctx, uint64_t count, double* max_abs_err, int* nblocks);
/* path of the library the collectives' entry points were taken from: the system's RCCL, or what $FASTA_RCCL_LIB names (a
 * substitution is also announced once on stderr); "" before the first communicator                                         */
const char* fh_comm_library(void);
/* CUs the device reports, and CUs the dense one-pass kernel is launched on (FH_TUNE_FUSED_CUS)                               */
int fh_cu_count(fh_ctx* ctx, int* device_cus, int* one_pass_cus);

/* ---- measurement: HIP-event timing of each launch on the context's stream ---------------------
 * A multi-device context reports sums over its row blocks.  When the blocks share ONE device (a repeated device id: one stream) only
 * ONE block's launches carry event records -- the MIDDLE block's (index nblocks / 2; the first one starts on an idle device and reads
 * high) -- and the figures are that block's scaled by the number of blocks: an ESTIMATE (row blocks may differ by one row), taken
 * because two records around each of the 8 x 2 launches cost more than the plumbing being measured (profiles/r04_inproc_issue.txt).
 * The sum over the blocks (FH_K_COMM) is one launch and is timed as such.                                                        */
int fh_timing_enable(fh_ctx* ctx, int on);
int fh_timing_get(fh_ctx* ctx, int kernel_id, double* total_ms, uint64_t* launches);
int fh_timing_reset(fh_ctx* ctx);
/* Did the latest timed launch of `kernel_id` on context a and on context b (plain contexts on one device, timing enabled, both
 * waited for) run at the same time?  a_ms / b_ms: the two launches' durations; overlap_ms: how long both were running
 * (<= 0: one had ended before the other began).  From the HIP events that bracket each launch on its own stream.              */
int fh_timing_overlap(fh_ctx* a, fh_ctx* b, int kernel_id, double* a_ms, double* b_ms, double* overlap_ms);
/* streaming-read ceiling: one read-only pass over the device copy of A by k_stream_probe<16,1> (csrc/fh_dense.h): one persistent
 * workgroup per CU (FH_TUNE_FWD_GRID_CAP overrides), three rotating register buffers of 16 non-temporal 16-byte loads per lane
 * (32 loads in flight behind the buffer being summed), loads + adds only; returns ms per pass.  A multi-device context
 * measures shard 0's row block.                                                                                           */
int fh_stream_read_ms(fh_ctx* ctx, int reps, double* ms_per_pass, uint64_t* bytes_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* FASTA_HIP_H */
