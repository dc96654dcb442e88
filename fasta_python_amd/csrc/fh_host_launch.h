// fh_host_launch.h -- kernel launchers of libfasta_hip.so: shapes, grids and workspace of every kernel in fh_dense.h / fh_tv.h /
// fh_prox.h / fh_fused.h, the one-pass kernel's shape rule and dispatch table, the co-residency probe, and the three-stage form
// (local launch / sum over row blocks / n-side epilogue) the C ABI in fasta_hip.hip builds its entry points from.
#pragma once

// ------------------------------------------------------------------------------------------------
// kernel launchers
// ------------------------------------------------------------------------------------------------
static ProxP make_prox(fh_ctx* c, double tau) {
  ProxP px;
  px.kind = c->prox_kind;
  px.thr = tau * c->mu;               // `t*self.mu`, examples/sparse_least_squares.py:44
  px.lo = c->lo; px.hi = c->hi;
  px.level = c->dscal + FH_NSCALARS;  // device scalar written by the level search
  px.seq = 0;
  return px;
}

// K-fwd / K-adj: non-temporal loads of A (+10 % on matrices that only stream through) -- except for a matrix of at most 256 MiB, the size of the device's
// last-level cache, which the next launch finds there again if it was loaded with the default policy: 4096^2 K-adj 0.049 against 0.075 ms, 5120^2 0.063 / 0.088,
// K-fwd 0.030 / 0.033; from 8192^2 (512 MiB) on the non-temporal form is the faster one again (profiles/r06_placement.txt, section 11).  The one-pass kernels
// always load non-temporally (plain: +-3 % at these sizes, where their time is mostly fixed cost).
static inline int nt_for(const fh_ctx* c) {
  if (c->nt_loads >= 0) return c->nt_loads;
  return (uint64_t)c->mp * c->ld * (c->f32 ? 4u : 8u) > ((uint64_t)256 << 20);
}

template <int R, int KIND>
static void launch_fwd_rk(fh_ctx* c, const FwdP& p, unsigned grid) {
  if (c->f32) {                                   // float32 storage: non-temporal loads only, R = 4 or 8
    if constexpr (R == 16) k_fwd_dense<8, 1, KIND, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
    else k_fwd_dense<R, 1, KIND, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  }
  else if (nt_for(c)) k_fwd_dense<R, 1, KIND><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
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
  p.px.seq = seq_offer(c);
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
  else if (nt_for(c)) k_adj_dense<CPT, 1><<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
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
  // K-adj keeps its contiguous slabs: dealt cyclically (FH_TUNE_ADJ_CYCLIC = 1) it is 1-14 % faster on some small and mid shapes but 6 % SLOWER at 65536^2 and not
  // less dependent on where the matrix lies (its workgroups are not in lockstep as the one-pass kernel's teams are; profiles/r06_placement.txt)
  p.cyclic = c->adj_cyclic == 1;
  if (p.ncc + CNT_ADJ_CC > (uint32_t)CNT_DIAG) return fail(FH_E_ARG, "too many column chunks (%u)", p.ncc);
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
  p.seq = io.mode == 0 ? seq_offer(c) : 0u;      // (the plain-gradient / sharded forms are followed by further launches)
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
  if (n > 16u * LVL_WG && n <= (uint32_t)LVL_MAXG * LVL_MEPT * LVL_WG) {      // several workgroups, LVL_MEPT values per thread (csrc/fh_prox.h)
    if (!c->lvl_rec) {
      HIP_TRY(hipMalloc((void**)&c->lvl_rec, (size_t)LVL_MAXPASS * LVL_MAXG * 2 * sizeof(double)));
      HIP_TRY(hipMalloc((void**)&c->lvl_cnt, (LVL_MAXPASS + 1) * sizeof(unsigned)));
      HIP_TRY(hipMemsetAsync(c->lvl_cnt, 0, (LVL_MAXPASS + 1) * sizeof(unsigned), c->stream));
    }
    LevelWs ws = {c->lvl_rec, c->lvl_cnt, c->counters + CNT_DIAG};
    const unsigned G = (n + LVL_MEPT * LVL_WG - 1) / (LVL_MEPT * LVL_WG);
    const int hooks = ((c->test_hooks & FH_HOOK_LEVEL_WITHHOLD) ? LVL_HOOK_WITHHOLD : 0) | ((c->test_hooks & FH_HOOK_LEVEL_NO_FALLBACK) ? LVL_HOOK_NO_FALLBACK : 0);
    t_begin(c, FH_K_LEVEL);
    k_level_search_multi<<<dim3(G), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out, ws, hooks);
    t_end(c, FH_K_LEVEL);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  t_begin(c, FH_K_LEVEL);
  if (n <= 4u * 256u) k_level_search<4, 256><<<dim3(1), dim3(256), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 16u * 256u) k_level_search<16, 256><<<dim3(1), dim3(256), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 32u * 256u) k_level_search<32, 256><<<dim3(1), dim3(256), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 16u * LVL_WG) k_level_search<16><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else if (n <= 64u * LVL_WG) k_level_search<64><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  else k_level_search<0><<<dim3(1), dim3(LVL_WG), 0, c->stream>>>(x0, g0, n, tau, radius, out);
  t_end(c, FH_K_LEVEL);
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
// The scheduling word a launch gets.  Rows are dealt cyclically (bit 32) where a team has at least 128 rows: below that the blocked dealing is as fast or
// 1-4 % faster (8192^2, teams of 2 with 64 rows each: 0.112 against 0.114 ms; 4096 x 8192: 0.0705 / 0.0735), from 256 rows per team on the cyclic one
// is 3-4 % faster on a well-placed matrix and up to 14 % on a badly placed one (profiles/r06_placement.txt) -- matrices of that size are small enough for
// their placement not to matter.  A word set through FH_TUNE_FUSED_VARIANT is passed on unchanged.
static int fused_variant_for(const fh_ctx* c, uint32_t rows_per_team) {
  int v = c->fused_variant;
  if (c->fused_variant_auto && rows_per_team < 128u) v &= ~32;
  return v;
}

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
      // A full 8-piece shape of 4 or 8 members (n in (28672, 32768] and (57344, 65536] = BASELINE config 2's
      // width) runs as TWICE the members x 4 pieces with TWO workgroups per CU (fh_fused.h:fused_wpc) and the x slice back in registers.
      // A float32 piece costs four conversions and eight float64 multiply-adds, twice the issue slots per byte of the float64 kernel, and
      // ONE wave per SIMD has nobody to hide its LDS, conversion and hand-off stalls behind; two have each other (x + g1 slices 64
      // registers, four 16-register row buffers: 219 of the 256 registers, no scratch in the loops).  profiles/r05_f32_shapes.txt:
      // 65536^2 2.577 -> 2.49-2.53 ms, 65536 x 32768 1.31-1.34 -> 1.26 ms.  (variant bit 8: the one-per-CU shapes, for A/B)
      // Teams of 1 and 2 (n <= 16384) stay one per CU: there the launch is short and the doubled grid barrier and hand-off cost more than
      // the second wave hides (8192^2: 91 -> 111 us, 16384^2: 212 -> 230 us; 32768^2 equal, 65536 x 32768 and 65536^2 3-5 % faster).
      if (ppt == 8 && team >= 4 && team <= 8 && !(variant & 8) && ncu % (2 * team) == 0) sh = {4, 2 * team, 1, 0, 4};
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
// workgroups per CU of a shape (fh_fused.h:fused_wpc): two for the float32-storage shape of 16 members x <= 4 pieces, else one
static int fused_wpc_of(const FusedShape& sh, int f32) { return (f32 && sh.team >= 8 && sh.ppt == 4 && !sh.xlds) ? 2 : 1; }
static const FusedEntry* fused_lookup(const FusedShape& sh, int f32) {
  for (const FusedEntry& e : kFusedTable)
    if (e.ppt == sh.ppt && e.pipe == sh.pipe && e.team == sh.team && e.xlds == sh.xlds && e.nbo == sh.nbo && e.f32 == f32) return &e;
  return nullptr;
}
// CUs (= workgroups) of the one-pass dense launch.  FH_TUNE_FUSED_CUS caps them, so that several one-pass grids can be co-resident
// on ONE device: two solves side by side, or -- in the tests -- the ranks of a row-sharded run that share a GPU (K ranks x ncu / K CUs).
static int fused_ncu(fh_ctx* c) { return c->fused_cus > 0 ? std::min(c->fused_cus, std::max(1, c->ncu)) : c->ncu; }
static FusedShape fused_shape(fh_ctx* c) {
  if (c->op != OP_DENSE || c->prox_kind == FH_PROX_TVBALL) return FusedShape{0, 0, 0, 0, 0};
  return fused_shape_for(c->n, c->ld, c->f32, c->fused_variant, fused_ncu(c));
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

static const ChainEntry* chain_lookup(const FusedShape& sh, int f32) {
  for (const ChainEntry& e : kChainTable)
    if (e.ppt == sh.ppt && e.pipe == sh.pipe && e.team == sh.team && e.xlds == sh.xlds && e.nbo == sh.nbo && e.f32 == f32) return &e;
  return nullptr;
}
// chain != nullptr: the CHAINED form (k_fused_chain): tau and the solver-state pointers of `io` are placeholders, the launch takes them
// from chain->st; no per-launch event records (the caller brackets the whole chain) and no sequence number (nobody waits for a single launch)
static int launch_fused_dense(fh_ctx* c, double tau, const FusedIO& io, const ChainP* chain = nullptr) {
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
  p.nteams = (uint32_t)(fused_ncu(c) * fused_wpc_of(sh, c->f32) / sh.team);
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
  p.bar = c->counters + CNT_FUSED_BAR; p.gbar = c->gridbar; p.err = c->counters + CNT_FUSED_ERR;
  p.variant = fused_variant_for(c, p.rows_per_team) | ((c->test_hooks & FH_HOOK_WITHHOLD_PARTIAL) ? 64 : 0);      // (bit 64 of FusedP.variant: the kernel's fault-injection switch)
  p.out = scalar_out(c);
  const ChainEntry* chain_entry = chain ? chain_lookup(sh, c->f32) : nullptr;
  if (chain && !chain_entry) return fail(FH_E_STATE, "chained one-pass launch: no instantiation for this shape (teams of 1 / 2 / 4 members, float64)");
  if (!chain) t_begin(c, FH_K_FUSED);
  {
    // signature of everything the slot layout depends on; 0 = "refill" (set after a timed-out launch, see fused_after)
    uint64_t sig = fh_mix((uint64_t)(uintptr_t)c->slotbuf ^ fh_mix(slots_elems * 131 + (uint64_t)sh.team * 7 + p.nteams)) | 1ull;
    if (sig != c->slots_sig) {
      // (a team of one exchanges nothing through the slots, but its grid barrier and error word are the same counters: a launch
      // that timed out in another shape must not leave them armed for it)
      if (sh.team > 1) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->slotbuf, (int)FT_SENTINEL_HI, 2 * slots_elems * 2, c->stream));
      HIP_TRY(hipMemsetAsync(c->counters + CNT_FUSED_BAR, 0, 8 * sizeof(unsigned), c->stream));
      HIP_TRY(hipMemsetAsync(c->gridbar, 0, 2 * GB_WORDS * sizeof(unsigned), c->stream));
      c->slots_sig = sig;
      c->slots_parity = 0;
    }
    p.slots = c->slotbuf + (size_t)c->slots_parity * slots_elems;
    p.slots_next = c->slotbuf + (size_t)(c->slots_parity ^ 1) * slots_elems;
    c->slots_parity ^= 1;
  }
  if (chain) {
    p.px.seq = 0u;
    chain_entry->kernel<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p, *chain);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  p.px.seq = io.mode == 0 ? seq_offer(c) : 0u;
  k_fused_dense_entry->kernel<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(p);
  t_end(c, FH_K_FUSED);
  HIP_TRY(hipGetLastError());
  return 0;
}

// z = A x, g = A^T grad f(z) from ONE read of A when the one-pass kernel pays off (single GPU, fused_pays()):
// identity prox and tau = 0 make xprox = x.  `xhat` and the prox target serve as the launch's scratch outputs.
// ---- co-residency probe ----------------------------------------------------------------------------------------------------
// The one-pass dense kernel needs ONE WORKGROUP ON EVERY compute unit it is launched on -- all the device reports, or the
// FH_TUNE_FUSED_CUS of them -- all at the same time (teams exchange partial sums while they run).  A CU mask, a partition mode or a co-tenant that hides CUs does not change the reported count, and
// the kernel would run into its bounded spins (0.4-0.5 s) before the solver drops it.  k_coresident finds out in ~20 us instead:
// it launches that many workgroups, each claiming more than half of a CU's LDS (so no two can share a CU), which count themselves
// in and wait -- at most 2 ms -- for the count to reach the grid size.  If some of them cannot start until others have ended, the
// count stalls: not co-resident, and fh_fused_supported reports 0 for the dense operator.  Run once per context, on first use.
#define FH_PROBE_LDS (96 * 1024)
__global__ __launch_bounds__(FH_WG, 1) void k_coresident(unsigned* counter, unsigned* failed) {
  __shared__ volatile unsigned claim[FH_PROBE_LDS / 4];
  if (threadIdx.x == 0) {
    claim[blockIdx.x % (FH_PROBE_LDS / 4)] = 1u;                               // volatile round trip: the allocation cannot be optimised away
    const unsigned one = claim[blockIdx.x % (FH_PROBE_LDS / 4)];
    __hip_atomic_fetch_add(counter, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
    bool all = false;
    while (__builtin_amdgcn_s_memrealtime() - t0 < 200000ull) {                // 2 ms
      if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) { all = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!all) __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
static bool run_coresident_probe(fh_ctx* c, unsigned grid) {                    // false on any failure
  unsigned* w = c->counters + CNT_PROBE;
  if (hipSetDevice(c->device) != hipSuccess) return false;
  if (hipMemsetAsync(w, 0, 2 * sizeof(unsigned), c->stream) != hipSuccess) return false;
  k_coresident<<<dim3(grid), dim3(FH_WG), 0, c->stream>>>(w, w + 1);
  unsigned host[2] = {0, 1};
  if (hipGetLastError() != hipSuccess) return false;
  if (hipMemcpyAsync(host, w, sizeof(host), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
  (void)hipMemsetAsync(w, 0, 2 * sizeof(unsigned), c->stream);
  return host[1] == 0 && host[0] == grid;
}
static bool co_resident(fh_ctx* c) {
  if (c->test_hooks & FH_HOOK_PROBE_SAYS_NO) return false;      // test hook (FH_TUNE_TEST_HOOKS, csrc/fh_experimental.h): "this rank's probe said no"
  if (c->coresident >= 0) return c->coresident != 0;
  // (the probe decides, nothing else: no environment variable is consulted.  A second try: a transient co-tenant must not cost this
  // context the one-pass kernel.)
  const unsigned grid = (unsigned)std::max(1, fused_ncu(c));
  c->coresident = (run_coresident_probe(c, grid) || run_coresident_probe(c, grid)) ? 1 : 0;
  return c->coresident != 0;
}
// diagnostic: can `workgroups` whole-CU workgroups run side by side on this context's device?  (ncu: yes on a healthy device;
// ncu + 1: never -- which is how the tests see the probe say "no")
extern "C" int fh_coresident_probe(fh_ctx* c, int workgroups, int* ok) {
  if (!c || !ok || workgroups < 1 || workgroups > 65536) return fail(FH_E_ARG, "fh_coresident_probe: bad argument");
  if (!c->shards.empty()) c = c->shards[0];
  *ok = run_coresident_probe(c, (unsigned)workgroups) ? 1 : 0;
  return 0;
}
static bool plain_pair_fused_ok(fh_ctx* c) { return c->op == OP_DENSE && !row_sharded(c) && c->shards.empty() && fused_ppt(c) && fused_pays(c) && co_resident(c); }
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
  if (c->pending_step) return fail(FH_E_STATE, "a step issued by fh_step_begin is still in flight on this context: call fh_step_end first");
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

// ---- stencil one-pass launchers ---------------------------------------------------------------------------------------------
#ifdef FH_EXPERIMENTAL      // the round-1 one-pass stencil kernels that stream z (FH_TUNE_TV_ZFREE = 0, csrc/fh_experimental.h)
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

#endif

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
    // SMALL images (round 6; the reference's own example is 512 x 512, tv_denoising.py:113-125): a chunk's rows are walked two (four) at a
    // time, each trip a dependent ~1-us round trip to L2 -- with 32-row chunks a 512^2 sweep is 48 workgroups x 18 trips = 23 us for 10 MB.
    // 8-row chunks (192 workgroups x 6 trips) take 16.5 us, 4 and 16 rows 18.5; at 1024^2 the plain sweep gains the same way (36.0 -> 27.9 us),
    // the FISTA sweep (two streams, 4-row trips) does not (31.0 -> 35.2); from 2048^2 on 32 rows are best again (profiles/r06_tv_rows_small.txt).
    const uint64_t pixels = (uint64_t)p.H * p.W;
    const uint32_t min_rows = pixels <= (1u << 18) ? 8u : ((pixels <= (1u << 20) && !accel) ? 8u : 32u);
    const uint32_t slots = (uint32_t)std::max(1, c->ncu) * 5u;
    const uint32_t chunks = std::max(1u, slots / p.strip_groups);
    p.rows_wg = std::min(p.H, std::max(min_rows, (p.H + chunks - 1) / chunks));
  }
  // rows per trip / rotating trip buffers: 2 rows, load-then-consume for the plain sweep; 4 rows x 3 rotating buffers with FISTA
  // (two streams to read): profiles/r03_tune_tv.txt.  Every combination produces the same bits (scripts/probes/tune_tvz.py).
  const int tvu = c->tv_u ? c->tv_u : (accel ? 4 : 2);
  if (accel) { p.p1 = nq(c, c->lq1); p.p0 = nq(c, c->lq0); p.pn = nq(c, c->lqn); p.cprev = c->lc; }
  else { p.p1 = c->X[c->xi]; p.p0 = c->X[c->xi]; p.pn = c->P[c->pc ^ 1]; p.cprev = 0.0; }
  p.b = c->b; p.tau = tau; p.coef = coef; p.restart = restart;
  p.xcd_order = c->tv_xcd != 2;                  // FH_TUNE_TV_XCD: 0 / 1 = on, 2 = off
  p.nchunks = p.strip_groups * ((p.H + p.rows_wg - 1) / p.rows_wg);
  // FH_TUNE_TV_SLOTS: persistent form -- at most that many workgroups per CU, each walking the chunk ids with the grid as its stride
  // (0 = one workgroup per chunk, the round-3 form)
  unsigned grid = p.nchunks;
#ifdef FH_EXPERIMENTAL
  if (c->tv_slots > 0) grid = std::min(grid, (unsigned)std::max(1, c->ncu) * (unsigned)c->tv_slots);
#endif
  FH_TRY(ensure_ws(c, (size_t)grid * 16 * sizeof(double)));
  p.red = c->ws; p.counter = c->counters + CNT_FWD; p.out = scalar_out(c);
  p.arrive = c->gridbar + 2 * GB_WORDS;
  t_begin(c, FH_K_FUSED);
  p.seq = seq_offer(c);
  // tunables -> template parameters.  Non-temporal LOADS lose 12 % here (the halo columns and rows are re-read by the neighbouring
  // waves and workgroups through L2), so this sweep only distinguishes non-temporal (default) and plain STORES (FH_TUNE_TV_NT = 3).
  const int nb = c->tv_pipe ? c->tv_pipe : (accel ? 3 : 1);
  const bool nts = c->tv_nt != 3;       // stores are non-temporal unless FH_TUNE_TV_NT = 3 asks for plain ones (+2-3 %: xprox is not re-read by this launch)
  const bool ident = c->prox_kind != FH_PROX_TVBALL;
  // FH_TUNE_TV_LDS_PAD: bytes of (unused) dynamic LDS per workgroup -- an occupancy limiter for experiments (fewer, fatter streams)
#define TVZ(ID, AC, U, NT, NB) k_tv_onepass<ID, AC, U, NT, NB><<<dim3(grid), dim3(FH_WG), (size_t)c->tv_lds_pad, c->stream>>>(p)
#define TVZ_NB(AC, U, NT) do { if (nb >= 2) TVZ(0, AC, U, NT, 3); else TVZ(0, AC, U, NT, 1); } while (0)
#define TVZ_U(AC, NT) do { if (tvu <= 2) TVZ_NB(AC, 2, NT); else if (tvu == 8) TVZ_NB(AC, 8, NT); else TVZ_NB(AC, 4, NT); } while (0)
#define TVZ_NT(AC) do { if (nts) TVZ_U(AC, 2); else TVZ_U(AC, 0); } while (0)
#ifdef FH_EXPERIMENTAL
  // FH_TUNE_TV_RING: LDS-DMA trip ring (2-row trips; the b pieces need 16-byte aligned rows, i.e. an even width)
  const int ring = (c->tv_ring >= 2 && p.W % 2 == 0 && !ident) ? c->tv_ring : 0;
#define TVZ_RING(AC, R) do { if (nts) k_tv_onepass<0, AC, 2, 2, 1, R><<<dim3(grid), dim3(FH_WG), (size_t)c->tv_lds_pad, c->stream>>>(p); \
                             else k_tv_onepass<0, AC, 2, 0, 1, R><<<dim3(grid), dim3(FH_WG), (size_t)c->tv_lds_pad, c->stream>>>(p); } while (0)
  if (ring) {
    if (accel) { if (ring == 2) TVZ_RING(1, 2); else TVZ_RING(1, 3); }
    else { if (ring == 2) TVZ_RING(0, 2); else TVZ_RING(0, 3); }
  } else
#undef TVZ_RING
#endif
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

