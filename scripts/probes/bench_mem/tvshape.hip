// Round-4 access-shape probe for the TV sweep (traffic of k_tv_onepass + a stand-in for its arithmetic), MI355X.
// What round 3 left open: the strip walkers' traffic alone takes 0.51-0.55 ms, the same bytes in a tile shape 0.436 ms.
// This probe varies, one at a time,
//   LAYOUT 0 : image rows in memory order (row r at r * pitch)                                   -- the product's layout
//   LAYOUT 2 : BAND-INTERLEAVED rows: a chunk (band) k of RB rows keeps its row t at memory row t * B + k (B bands), so the B
//              workgroup rows that sweep their bands in step read B ADJACENT memory rows: one compact window per array
//   staging  : registers (NB rotating trips of U rows, as k_tv_onepass) or a per-wave LDS ring filled by LDS-DMA
//              (global_load_lds_dwordx4, D rows deep: the wave that issues the DMA is the one that reads the slot, so only its own
//              counted vmcnt orders it -- no barrier)
//   OWN      : 60 owned lanes of 64 (2 halo lanes per side, misaligned strips: the product) or 64 (aligned, no halo lanes)
//   ALU      : dependent float64 FMAs per pixel between load and store (0 = traffic only; ~170 stands for the sweep's ~0.35 ms)
//   hipcc --offload-arch=gfx950 -O3 -o tvshape tvshape.hip && ./tvshape [set]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

struct Geo {
  uint32_t H, W, pitch, rows_wg, strip_groups, bands;   // bands = ceil(H / rows_wg)
  int xcd;
  int up;          // 1 = sweep the chunk bottom to top
  uint32_t stagger; // start delay step (x s_sleep 8) by workgroup id
  uint32_t* queue; uint32_t nchunks;   // QUEUE variants
};

__device__ __forceinline__ uint32_t xcd_order(uint32_t b, uint32_t grid, int on) {
  const uint32_t per = grid / 8u;
  if (!on || b >= per * 8u) return b;
  return (b % 8u) * per + b / 8u;
}

// memory row of image row r
template <int LAYOUT>
__device__ __forceinline__ uint32_t mem_row(uint32_t r, const Geo& g) {
  if (LAYOUT == 2) return (r % g.rows_wg) * g.bands + r / g.rows_wg;
  return r;
}

template <int ALU>
__device__ __forceinline__ d2 work(d2 v, double b) {
  double a0 = v.x, a1 = v.y + b;
#pragma unroll
  for (int i = 0; i < ALU / 2; ++i) { a0 = fma(a0, 0.999999, 1e-9); a1 = fma(a1, 1.000001, -1e-9); }
  d2 r; r.x = a0; r.y = a1;
  return r;
}

// ---- register-staged strip walk (the product's shape) -------------------------------------------------------------------
template <int OWN, int U, int NB, int NTS, int LAYOUT, int ALU, int MODE = 0, int NW = 4, int QUEUE = 0>
__global__ __launch_bounds__(64 * NW) void k_strip(const d2* __restrict__ x, const double* __restrict__ b, d2* __restrict__ xp, const Geo g) {
  constexpr int HALO = OWN == 64 ? 0 : 2;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ uint32_t s_next;
  uint32_t wg = xcd_order(blockIdx.x, gridDim.x, g.xcd);
  if (g.stagger) { const uint32_t n = (blockIdx.x / 8u) % 16u * g.stagger; for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8); }
  for (;;) {
  if (QUEUE) {                                  // persistent workgroups pull chunk ids in order (band-major: a compact moving window)
    __syncthreads();
    if (threadIdx.x == 0) s_next = atomicAdd(g.queue, 1u);
    __syncthreads();
    wg = s_next;
    if (wg >= g.nchunks) return;
  }
  const uint32_t sg = wg % g.strip_groups, rc = wg / g.strip_groups;
  const uint32_t i0 = rc * g.rows_wg;
  const uint32_t rows = min(g.rows_wg, g.H - i0);
  const uint32_t first = (sg * NW + wave) * OWN;
  const uint32_t cw = (first + lane + 2u * g.W - HALO) % g.W;
  const uint32_t c = first + lane - HALO;
  const bool own = lane >= (uint32_t)HALO && lane < (uint32_t)(HALO + OWN) && c < g.W;
  const int total = (int)rows + 2 * HALO;
  struct Trip { d2 x[U]; double b[U]; };
  auto row_of = [&](int off) -> uint32_t { int r = (int)i0 + off; if (r < 0) r += (int)g.H; if (r >= (int)g.H) r -= (int)g.H; return (uint32_t)r; };
  auto load = [&](Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int tt = min(t0 + q, total - 1);
      const int s = (g.up ? total - 1 - tt : tt) - HALO;
      const uint64_t pix = (uint64_t)mem_row<LAYOUT>(row_of(s), g) * g.pitch + cw;
      if (!(MODE & 1)) { T.x[q] = x[pix]; T.b[q] = b[pix]; } else { T.x[q] = (d2){(double)pix, 1.0}; T.b[q] = (double)cw; }
    }
    asm volatile("" ::: "memory");
  };
  auto store = [&](const Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int s = (g.up ? total - 1 - (t0 + q) : t0 + q) - HALO;
      const d2 v = work<ALU>(T.x[q], T.b[q]);
      if (MODE & 2) { if (v.x == 1.2345e300) xp[0] = v; }
      else if (t0 + q < total && own && s >= 0 && s < (int)rows) st<NTS>(xp + (uint64_t)mem_row<LAYOUT>(i0 + s, g) * g.pitch + c, v);
    }
    asm volatile("" ::: "memory");
  };
  if constexpr (NB == 3) {
    Trip T0, T1, T2;
    load(T0, 0); load(T1, U);
    for (int t0 = 0; t0 < total; t0 += 3 * U) {
      load(T2, t0 + 2 * U); store(T0, t0);
      load(T0, t0 + 3 * U); store(T1, t0 + U);
      load(T1, t0 + 4 * U); store(T2, t0 + 2 * U);
    }
  } else {
    for (int t0 = 0; t0 < total; t0 += U) { Trip T0; load(T0, t0); store(T0, t0); }
  }
  if (!QUEUE) return;
  }
}

// ---- the same walk with a per-wave LDS ring filled by LDS-DMA --------------------------------------------------------------
// slot = one PAIR of rows: x row a (1 KiB) | x row a+1 (1 KiB) | b rows a, a+1 (2 x 512 B, lanes 0-31 / 32-63 of ONE dwordx4 DMA)
// D pair slots per wave; 3 DMA instructions + 2 stores per pair.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int OWN, int D, int NTS, int LAYOUT, int ALU>
__global__ __launch_bounds__(256) void k_strip_lds(const d2* __restrict__ x, const double* __restrict__ b, d2* __restrict__ xp, const Geo g) {
  constexpr int HALO = OWN == 64 ? 0 : 2;
  constexpr uint32_t SLOT = 3072;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wg = xcd_order(blockIdx.x, gridDim.x, g.xcd);
  const uint32_t sg = wg % g.strip_groups, rc = wg / g.strip_groups;
  const uint32_t i0 = rc * g.rows_wg;
  const uint32_t rows = min(g.rows_wg, g.H - i0);
  const uint32_t first = (sg * 4 + wave) * OWN;
  const uint32_t cw = (first + lane + 2u * g.W - HALO) % g.W;
  const uint32_t cw0 = (first + 2u * g.W - HALO) % g.W;            // column of lane 0
  const uint32_t c = first + lane - HALO;
  const bool own = lane >= (uint32_t)HALO && lane < (uint32_t)(HALO + OWN) && c < g.W;
  const int total = (int)rows + 2 * HALO;
  const int npairs = (total + 1) / 2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const uint32_t ring = __builtin_amdgcn_readfirstlane(lds0 + wave * (D * SLOT));      // LDS byte address of this wave's ring (wave-uniform)
  const bool stores = __builtin_amdgcn_ballot_w64(own) != 0;                            // this wave issues store instructions at all
  auto row_of = [&](int off) -> uint32_t { int r = (int)i0 + off; if (r < 0) r += (int)g.H; if (r >= (int)g.H) r -= (int)g.H; return (uint32_t)r; };
  // strips that wrap around the image edge (cw not monotone over the lanes) are only the first / last ones; the b DMA below assumes
  // lane l's pixel is cw0 + l, which holds whenever the strip does not wrap.  The probe launches W = multiple of 4 * OWN so only the
  // first strip's two halo lanes wrap: they read a neighbouring pixel instead (traffic identical).
  auto issue = [&](int pr) {                                             // DMA pair pr into slot pr % D
    const uint32_t slot = ring + (uint32_t)(pr % D) * SLOT;
    const int ta = min(2 * pr, total - 1), tb = min(2 * pr + 1, total - 1);
    const int sa = (g.up ? total - 1 - ta : ta) - HALO, sb = (g.up ? total - 1 - tb : tb) - HALO;
    const uint64_t ra = (uint64_t)mem_row<LAYOUT>(row_of(sa), g) * g.pitch, rb = (uint64_t)mem_row<LAYOUT>(row_of(sb), g) * g.pitch;
    glds16(x + ra + cw, slot);
    glds16(x + rb + cw, slot + 1024);
    const uint64_t rr = lane < 32 ? ra : rb;
    glds16(b + rr + cw0 + 2 * (lane & 31), slot + 2048);                // 32 lanes x 16 B = 64 pixels of b per row
  };
  for (int pr = 0; pr < D - 1; ++pr) issue(pr);
  for (int pr = 0; pr < npairs; ++pr) {
    issue(pr + D - 1);                                                   // clamped past the chunk: re-reads its last rows
    // younger than pair pr's three DMAs: the 3 DMAs of each of the D-1 later pairs and, in the steady state, the 2 stores of each of
    // the D-1 pairs consumed since (vmcnt counts loads, stores and LDS-DMA together, in issue order).  Waiting for too FEW
    // outstanding operations is always safe, so the store-free cases use the smaller count.
    if (stores && pr >= D && 2 * pr + 1 - HALO < (int)rows && !g.up) wait_vm<3 * (D - 1) + 2 * (D - 1)>(); else wait_vm<3 * (D - 1)>();
    const uint32_t slot = ring + (uint32_t)(pr % D) * SLOT;
    const d2 xa = *reinterpret_cast<const __attribute__((address_space(3))) d2*>((uintptr_t)(slot + lane * 16));
    const d2 xb = *reinterpret_cast<const __attribute__((address_space(3))) d2*>((uintptr_t)(slot + 1024 + lane * 16));
    const double ba = *reinterpret_cast<const __attribute__((address_space(3))) double*>((uintptr_t)(slot + 2048 + lane * 8));
    const double bb = *reinterpret_cast<const __attribute__((address_space(3))) double*>((uintptr_t)(slot + 2560 + lane * 8));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const d2 va = work<ALU>(xa, ba), vb = work<ALU>(xb, bb);
    const int sa = (g.up ? total - 1 - 2 * pr : 2 * pr) - HALO, sb = (g.up ? total - 2 - 2 * pr : 2 * pr + 1) - HALO;
    if (own && sa >= 0 && sa < (int)rows) st<NTS>(xp + (uint64_t)mem_row<LAYOUT>(i0 + sa, g) * g.pitch + c, va);
    if (own && sb >= 0 && sb < (int)rows && 2 * pr + 1 < total) st<NTS>(xp + (uint64_t)mem_row<LAYOUT>(i0 + sb, g) * g.pitch + c, vb);
    asm volatile("" ::: "memory");
  }
  wait_vm<0>();
}

// ---- (12) TILED STORAGE WITH IN-TILE HALOS (round 5: the one layout DESIGN.md 7.3 left untried) ---------------------------------
// Every array is stored as tiles  [row band k][strip group sg][tile row t = 0 .. RB+3][wave w = 0..3][lane 0..63] : tile row t of band
// k is image row k*RB + t - 2 (periodic), lane L of wave w of strip group sg is image column (4 sg + w) * 60 + L - 2 (periodic).  So the
// two halo columns per side AND the two halo rows per side are stored with the tile (64/60 x (RB+4)/RB = 1.086x the bytes), every
// wave-instruction reads ONE whole aligned 1-KiB run (x) / 512-B run (b), a workgroup's reads are one contiguous range of
// (RB + 4) x 4 KiB that it streams front to back, and nothing is fetched from another workgroup's region.  The price is on the store
// side: a pixel in a halo position has up to four homes, so besides its 60 owned lanes (960 B of the 1-KiB run) a wave stores lanes
// 2,3 / 60,61 again into the neighbouring strips' halo lanes (one 4-lane store: 2 x 32 B) and, for the two rows at either end of its band,
// all of that once more into the neighbouring band's tile.
struct TGeo { uint32_t H, W, RB, strip_groups, bands; int xcd; };
template <int U, int NB, int NTS, int ALU>
__global__ __launch_bounds__(256) void k_tile(const d2* __restrict__ x, const double* __restrict__ b, d2* __restrict__ xp, const TGeo g) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wg = xcd_order(blockIdx.x, gridDim.x, g.xcd);
  const uint32_t sg = wg % g.strip_groups, band = wg / g.strip_groups;
  const uint32_t TR = g.RB + 4u;                                        // tile rows
  const uint64_t tile_px = (uint64_t)TR * 256u;                          // pixels per tile (4 waves x 64 lanes per tile row)
  const uint64_t base = ((uint64_t)band * g.strip_groups + sg) * tile_px + wave * 64u + lane;
  const uint32_t nstrips = g.strip_groups * 4u;
  const uint32_t strip = sg * 4u + wave;
  // where the duplicates of this wave's edge lanes live: lanes 2,3 -> lanes 62,63 of the strip to the left; lanes 60,61 -> lanes 0,1 to the right
  const uint32_t lstrip = (strip + nstrips - 1u) % nstrips, rstrip = (strip + 1u) % nstrips;
  const bool own = lane >= 2u && lane < 62u;
  const bool dupl = lane == 2u || lane == 3u, dupr = lane == 60u || lane == 61u;
  const uint32_t dstrip = dupl ? lstrip : rstrip;
  const uint32_t dlane = dupl ? lane + 60u : lane - 60u;
  const uint32_t upb = (band + g.bands - 1u) % g.bands, dnb = (band + 1u) % g.bands;
  auto tile_at = [&](uint32_t bd, uint32_t st, uint32_t t, uint32_t l) -> uint64_t {
    return ((uint64_t)bd * g.strip_groups + st / 4u) * tile_px + (uint64_t)t * 256u + (st % 4u) * 64u + l;
  };
  const int total = (int)TR;
  struct Trip { d2 x[U]; double b[U]; };
  auto load = [&](Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const uint64_t pix = base + (uint64_t)min(t0 + q, total - 1) * 256u;
      T.x[q] = x[pix]; T.b[q] = b[pix];
    }
    asm volatile("" ::: "memory");
  };
  auto store = [&](const Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int t = t0 + q;                                             // tile row; owned rows are t = 2 .. RB+1
      const d2 v = work<ALU>(T.x[q], T.b[q]);
      if (t >= 2 && t < (int)g.RB + 2) {
        if (own) st<NTS>(xp + base + (uint64_t)t * 256u, v);
        if (dupl || dupr) st<NTS>(xp + tile_at(band, dstrip, (uint32_t)t, dlane), v);
        // the first / last two owned rows are also the halo rows RB+2, RB+3 / 0, 1 of the band above / below
        if (t < 4 || t >= (int)g.RB) {
          const uint32_t nb = t < 4 ? upb : dnb;
          const uint32_t nt = t < 4 ? (uint32_t)t + g.RB : (uint32_t)t - g.RB;
          if (own) st<NTS>(xp + tile_at(nb, strip, nt, lane), v);
          if (dupl || dupr) st<NTS>(xp + tile_at(nb, dstrip, nt, dlane), v);
        }
      }
    }
    asm volatile("" ::: "memory");
  };
  if constexpr (NB == 3) {
    Trip T0, T1, T2;
    load(T0, 0); load(T1, U);
    for (int t0 = 0; t0 < total; t0 += 3 * U) {
      load(T2, t0 + 2 * U); store(T0, t0);
      load(T0, t0 + 3 * U); store(T1, t0 + U);
      load(T1, t0 + 4 * U); store(T2, t0 + 2 * U);
    }
  } else {
    for (int t0 = 0; t0 < total; t0 += U) { Trip T0; load(T0, t0); store(T0, t0); }
  }
}

static hipEvent_t e0, e1;
template <typename F> static int run(const char* name, double bytes, F launch) {
  launch(); CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float best = 1e30f, tot = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; tot += ms; if (ms < best) best = ms;
  }
  printf("%-96s best %7.4f ms %6.0f GB/s   mean %7.4f ms\n", name, best, bytes / best / 1e6, tot / 3);
  fflush(stdout);
  return 0;
}

template <typename F> static int run_i(const char* name, double bytes, F launch) {   // launch(i): i = running launch index
  int it = 0;
  launch(it++); launch(it++); CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float best = 1e30f, tot = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(it++); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; tot += ms; if (ms < best) best = ms;
  }
  printf("%-96s best %7.4f ms %6.0f GB/s   mean %7.4f ms\n", name, best, bytes / best / 1e6, tot / 3);
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int set = argc > 1 ? atoi(argv[1]) : 0;
  const uint32_t H = 8192, W = 8192;
  const uint64_t P = (uint64_t)H * W;
  const uint64_t ROWS_ALLOC = H + 512;                 // band-interleaved layouts pad H up to bands * rows_wg
  d2 *x, *xp; double* b;
  CK(hipMalloc(&x, ROWS_ALLOC * (W + 256) * 16)); CK(hipMalloc(&xp, ROWS_ALLOC * (W + 256) * 16)); CK(hipMalloc(&b, ROWS_ALLOC * (W + 256) * 8));
  CK(hipMemset(x, 0, ROWS_ALLOC * (W + 256) * 16)); CK(hipMemset(xp, 0, ROWS_ALLOC * (W + 256) * 16)); CK(hipMemset(b, 0, ROWS_ALLOC * (W + 256) * 8));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  char name[200];
  auto geo = [&](int own, uint32_t rows, uint32_t pitch, int xcd) {
    Geo g; g.H = H; g.W = W; g.pitch = pitch; g.rows_wg = rows; g.strip_groups = ((W + own - 1) / own + 3) / 4; g.bands = (H + rows - 1) / rows; g.xcd = xcd; g.up = 0; g.stagger = 0; g.queue = nullptr; g.nchunks = 0;
    return g;
  };
#define STRIP(OWN, U, NB, NTS, LAYOUT, ALU, ROWS, PITCH, XCD) do { const Geo g = geo(OWN, ROWS, PITCH, XCD); const uint32_t grid = g.strip_groups * g.bands; \
    if ((uint64_t)g.bands * g.rows_wg > ROWS_ALLOC) { printf("skip rows=%d\n", ROWS); break; } \
    snprintf(name, sizeof name, "reg  own=%d U=%d NB=%d st=%s layout=%s alu=%3d rows=%3d pitch=%d xcd=%d grid=%u", OWN, U, NB, NTS ? "nt" : "pl", LAYOUT == 2 ? "bands" : "rows ", ALU, ROWS, PITCH, XCD, grid); \
    if (run(name, 40.0 * P, [&] { k_strip<OWN, U, NB, NTS, LAYOUT, ALU><<<grid, 256>>>(x, b, xp, g); })) return 1; } while (0)
#define LSTRIP(OWN, D, NTS, LAYOUT, ALU, ROWS, PITCH, XCD) do { const Geo g = geo(OWN, ROWS, PITCH, XCD); const uint32_t grid = g.strip_groups * g.bands; \
    if ((uint64_t)g.bands * g.rows_wg > ROWS_ALLOC) { printf("skip rows=%d\n", ROWS); break; } \
    snprintf(name, sizeof name, "lds  own=%d D=%d pairs  st=%s layout=%s alu=%3d rows=%3d pitch=%d xcd=%d grid=%u lds=%d", OWN, D, NTS ? "nt" : "pl", LAYOUT == 2 ? "bands" : "rows ", ALU, ROWS, PITCH, XCD, grid, 4 * D * 3072); \
    if (run(name, 40.0 * P, [&] { k_strip_lds<OWN, D, NTS, LAYOUT, ALU><<<grid, 256, 4 * D * 3072>>>(x, b, xp, g); })) return 1; } while (0)

  if (set == 0 || set == 1) {
    printf("=== (1) layout: rows vs band-interleaved, register staging, traffic only (40*P algorithmic bytes)\n");
    STRIP(60, 2, 1, 1, 0, 0, 228, 8192, 1); STRIP(60, 2, 1, 1, 2, 0, 228, 8192, 1);
    STRIP(60, 2, 3, 1, 0, 0, 228, 8192, 1); STRIP(60, 2, 3, 1, 2, 0, 228, 8192, 1);
    STRIP(60, 2, 1, 1, 0, 0, 228, 8192, 0); STRIP(60, 2, 1, 1, 2, 0, 228, 8192, 0);
    STRIP(60, 2, 3, 1, 0, 0, 228, 8192, 0); STRIP(60, 2, 3, 1, 2, 0, 228, 8192, 0);
    STRIP(60, 4, 3, 1, 2, 0, 228, 8192, 1); STRIP(60, 1, 3, 1, 2, 0, 228, 8192, 1);
    STRIP(60, 2, 3, 1, 2, 0, 128, 8192, 1); STRIP(60, 2, 3, 1, 2, 0, 256, 8192, 1); STRIP(60, 2, 3, 1, 2, 0, 456, 8192, 1); STRIP(60, 2, 3, 1, 2, 0, 64, 8192, 1);
    STRIP(60, 2, 3, 1, 2, 0, 228, 8200, 1); STRIP(60, 2, 3, 1, 2, 0, 228, 8320, 1);
    STRIP(64, 2, 3, 1, 0, 0, 228, 8192, 1); STRIP(64, 2, 3, 1, 2, 0, 228, 8192, 1); STRIP(64, 2, 3, 1, 2, 0, 228, 8320, 1); STRIP(64, 2, 1, 1, 2, 0, 228, 8192, 1);
    STRIP(64, 4, 3, 1, 2, 0, 256, 8192, 1); STRIP(64, 2, 3, 1, 2, 0, 256, 8192, 0);
  }
  if (set == 0 || set == 2) {
    printf("=== (2) the same with a stand-in for the sweep's arithmetic (alu = dependent float64 FMAs per pixel)\n");
    STRIP(60, 2, 1, 1, 0, 170, 228, 8192, 1); STRIP(60, 2, 1, 1, 2, 170, 228, 8192, 1);
    STRIP(60, 2, 3, 1, 0, 170, 228, 8192, 1); STRIP(60, 2, 3, 1, 2, 170, 228, 8192, 1);
    STRIP(60, 2, 1, 1, 0, 120, 228, 8192, 1); STRIP(60, 2, 1, 1, 2, 120, 228, 8192, 1);
    STRIP(60, 2, 3, 1, 2, 120, 228, 8192, 1); STRIP(64, 2, 3, 1, 2, 170, 228, 8192, 1);
  }
  if (set == 0 || set == 3) {
    printf("=== (3) per-wave LDS ring filled by LDS-DMA (D row pairs deep), traffic only and with the arithmetic stand-in\n");
    LSTRIP(60, 2, 1, 0, 0, 228, 8192, 1); LSTRIP(60, 3, 1, 0, 0, 228, 8192, 1); LSTRIP(60, 4, 1, 0, 0, 228, 8192, 1);
    LSTRIP(60, 2, 1, 2, 0, 228, 8192, 1); LSTRIP(60, 3, 1, 2, 0, 228, 8192, 1); LSTRIP(60, 4, 1, 2, 0, 228, 8192, 1);
    LSTRIP(64, 3, 1, 2, 0, 228, 8192, 1); LSTRIP(64, 3, 1, 0, 0, 228, 8192, 1);
    LSTRIP(60, 2, 1, 0, 170, 228, 8192, 1); LSTRIP(60, 3, 1, 0, 170, 228, 8192, 1); LSTRIP(60, 4, 1, 0, 170, 228, 8192, 1);
    LSTRIP(60, 2, 1, 2, 170, 228, 8192, 1); LSTRIP(60, 3, 1, 2, 170, 228, 8192, 1); LSTRIP(60, 4, 1, 2, 170, 228, 8192, 1);
    LSTRIP(64, 3, 1, 2, 170, 228, 8192, 1);
  }
  if (set == 0 || set == 4) {
    printf("=== (4) what the stand-in costs alone: no loads / no stores / neither (register staging, NB = 1)\n");
#define MSTRIP(ALU, MODE) do { const Geo g = geo(60, 228, 8192, 1); const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "reg  own=60 U=2 NB=1 alu=%3d %s", ALU, MODE == 3 ? "no loads, no stores" : MODE == 1 ? "no loads" : MODE == 2 ? "no stores" : "complete"); \
    if (run(name, 40.0 * P, [&] { k_strip<60, 2, 1, 1, 0, ALU, MODE><<<grid, 256>>>(x, b, xp, g); })) return 1; } while (0)
    MSTRIP(170, 3); MSTRIP(170, 1); MSTRIP(170, 2); MSTRIP(170, 0); MSTRIP(120, 3); MSTRIP(240, 3); MSTRIP(240, 0); MSTRIP(0, 1); MSTRIP(0, 2);
  }
  if (set == 0 || set == 5) {
    printf("=== (5) output of launch i = input of launch i+1 (as the solver's buffers rotate); sweep direction fixed or alternating\n");
#define PSTRIP(U, NB, ALU, ALT, XSWAP) do { Geo g = geo(60, 228, 8192, 1); const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "reg  own=60 U=%d NB=%d alu=%3d buffers %s, direction %s", U, NB, ALU, XSWAP ? "rotate" : "fixed ", ALT ? "alternates" : "fixed"); \
    if (run_i(name, 40.0 * P, [&](int i) { g.up = ALT ? (i & 1) : 0; const bool sw = XSWAP && (i & 1); \
        k_strip<60, U, NB, 1, 0, ALU><<<grid, 256>>>(sw ? xp : x, b, sw ? x : xp, g); })) return 1; } while (0)
    PSTRIP(2, 1, 0, 0, 0); PSTRIP(2, 1, 0, 0, 1); PSTRIP(2, 1, 0, 1, 1); PSTRIP(2, 1, 0, 1, 0);
    PSTRIP(2, 3, 0, 0, 1); PSTRIP(2, 3, 0, 1, 1);
    PSTRIP(2, 1, 170, 0, 1); PSTRIP(2, 1, 170, 1, 1); PSTRIP(2, 3, 170, 0, 1); PSTRIP(2, 3, 170, 1, 1);
#define PLSTRIP(D, ALU, ALT) do { Geo g = geo(60, 228, 8192, 1); const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "lds  own=60 D=%d alu=%3d buffers rotate, direction %s", D, ALU, ALT ? "alternates" : "fixed"); \
    if (run_i(name, 40.0 * P, [&](int i) { g.up = ALT ? (i & 1) : 0; const bool sw = (i & 1); \
        k_strip_lds<60, D, 1, 0, ALU><<<grid, 256, 4 * D * 3072>>>(sw ? xp : x, b, sw ? x : xp, g); })) return 1; } while (0)
    PLSTRIP(3, 0, 0); PLSTRIP(3, 0, 1); PLSTRIP(3, 170, 0); PLSTRIP(3, 170, 1);
  }
  if (set == 0 || set == 6) {
    printf("=== (6) LDS ring: chunk rows / occupancy\n");
    LSTRIP(60, 2, 1, 0, 170, 128, 8192, 1); LSTRIP(60, 2, 1, 0, 170, 200, 8192, 1); LSTRIP(60, 3, 1, 0, 170, 285, 8192, 1); LSTRIP(60, 4, 1, 0, 170, 380, 8192, 1);
    LSTRIP(60, 6, 1, 0, 170, 570, 8192, 1); LSTRIP(60, 8, 1, 0, 170, 1140, 8192, 1);
    LSTRIP(60, 2, 1, 0, 0, 128, 8192, 1); LSTRIP(60, 3, 1, 0, 0, 285, 8192, 1); LSTRIP(60, 6, 1, 0, 0, 570, 8192, 1); LSTRIP(60, 8, 1, 0, 0, 1140, 8192, 1);
  }
  if (set == 0 || set == 7) {
    printf("=== (7) traffic only and with the stand-in: waves per workgroup, chunk rows (several rounds of workgroups), staggered starts, persistent workgroups pulling chunks in order\n");
    uint32_t* queue; CK(hipMalloc(&queue, 64)); 
#define WSTRIP(NW, U, NB, ALU, ROWS, STAG) do { Geo g = geo(60, ROWS, 8192, 1); g.strip_groups = ((W + 59) / 60 + NW - 1) / NW; g.stagger = STAG; const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "reg  own=60 waves/wg=%d U=%d NB=%d alu=%3d rows=%3d stagger=%d grid=%u", NW, U, NB, ALU, ROWS, STAG, grid); \
    if (run(name, 40.0 * P, [&] { k_strip<60, U, NB, 1, 0, ALU, 0, NW><<<grid, 64 * NW>>>(x, b, xp, g); })) return 1; } while (0)
#define QSTRIP(NW, U, NB, ALU, ROWS, PERCU, XCD) do { Geo g = geo(60, ROWS, 8192, XCD); g.strip_groups = ((W + 59) / 60 + NW - 1) / NW; g.queue = queue; g.nchunks = g.strip_groups * g.bands; const uint32_t grid = 256 * PERCU; \
    snprintf(name, sizeof name, "reg  own=60 waves/wg=%d U=%d NB=%d alu=%3d rows=%3d QUEUE %d wg/CU, %u chunks", NW, U, NB, ALU, ROWS, PERCU, g.nchunks); \
    if (run(name, 40.0 * P, [&] { hipMemsetAsync(queue, 0, 4, 0); k_strip<60, U, NB, 1, 0, ALU, 0, NW, 1><<<grid, 64 * NW>>>(x, b, xp, g); })) return 1; } while (0)
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 0) {
        WSTRIP(4, 2, 1, 0, 228, 0); WSTRIP(2, 2, 1, 0, 228, 0); WSTRIP(1, 2, 1, 0, 228, 0); WSTRIP(4, 2, 3, 0, 228, 0); WSTRIP(2, 2, 3, 0, 228, 0); WSTRIP(1, 2, 3, 0, 228, 0);
        WSTRIP(4, 2, 1, 0, 32, 0); WSTRIP(4, 2, 1, 0, 64, 0); WSTRIP(4, 2, 1, 0, 128, 0); WSTRIP(4, 2, 3, 0, 32, 0); WSTRIP(4, 2, 3, 0, 64, 0); WSTRIP(1, 2, 3, 0, 64, 0);
        WSTRIP(4, 2, 1, 0, 228, 4); WSTRIP(4, 2, 1, 0, 228, 16); WSTRIP(4, 2, 1, 0, 228, 64); WSTRIP(4, 2, 3, 0, 228, 16);
        QSTRIP(4, 2, 1, 0, 32, 5, 0); QSTRIP(4, 2, 1, 0, 64, 5, 0); QSTRIP(4, 2, 3, 0, 32, 5, 0); QSTRIP(4, 2, 3, 0, 64, 5, 0); QSTRIP(4, 2, 3, 0, 64, 8, 0); QSTRIP(4, 2, 1, 0, 64, 8, 0); QSTRIP(4, 2, 1, 0, 16, 5, 0);
      } else {
        WSTRIP(4, 2, 1, 170, 228, 0); WSTRIP(2, 2, 1, 170, 228, 0); WSTRIP(1, 2, 1, 170, 228, 0); WSTRIP(4, 2, 3, 170, 228, 0); WSTRIP(1, 2, 3, 170, 228, 0);
        WSTRIP(4, 2, 1, 170, 32, 0); WSTRIP(4, 2, 1, 170, 64, 0); WSTRIP(4, 2, 3, 170, 64, 0);
        WSTRIP(4, 2, 1, 170, 228, 16); WSTRIP(4, 2, 3, 170, 228, 16);
        QSTRIP(4, 2, 1, 170, 32, 5, 0); QSTRIP(4, 2, 1, 170, 64, 5, 0); QSTRIP(4, 2, 3, 170, 64, 5, 0);
      }
    }
  }
  if (set == 8) {
    printf("=== (8) as (7) with every launch capped at 5 workgroups per CU (32 KiB of dynamic LDS each, as the product's registers cap it)\n");
    uint32_t* queue; CK(hipMalloc(&queue, 64));
    const int CAP = 32768;
#define WSTRIP8(U, NB, ALU, ROWS) do { Geo g = geo(60, ROWS, 8192, 1); const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "reg  own=60 U=%d NB=%d alu=%3d rows=%3d grid=%u (5 wg/CU cap)", U, NB, ALU, ROWS, grid); \
    if (run(name, 40.0 * P, [&] { k_strip<60, U, NB, 1, 0, ALU, 0, 4><<<grid, 256, CAP>>>(x, b, xp, g); })) return 1; } while (0)
#define QSTRIP8(U, NB, ALU, ROWS, XCD) do { Geo g = geo(60, ROWS, 8192, XCD); g.queue = queue; g.nchunks = g.strip_groups * g.bands; const uint32_t grid = 1280; \
    snprintf(name, sizeof name, "reg  own=60 U=%d NB=%d alu=%3d rows=%3d QUEUE 1280 wgs (5/CU cap), %u chunks", U, NB, ALU, ROWS, g.nchunks); \
    if (run(name, 40.0 * P, [&] { hipMemsetAsync(queue, 0, 4, 0); k_strip<60, U, NB, 1, 0, ALU, 0, 4, 1><<<grid, 256, CAP>>>(x, b, xp, g); })) return 1; } while (0)
    WSTRIP8(2, 1, 0, 228); WSTRIP8(2, 3, 0, 228); WSTRIP8(2, 1, 0, 32); WSTRIP8(2, 1, 0, 64); WSTRIP8(2, 3, 0, 32);
    QSTRIP8(2, 1, 0, 16, 0); QSTRIP8(2, 1, 0, 32, 0); QSTRIP8(2, 1, 0, 64, 0); QSTRIP8(2, 3, 0, 16, 0); QSTRIP8(2, 3, 0, 32, 0); QSTRIP8(2, 1, 0, 8, 0);
    WSTRIP8(2, 1, 170, 228); WSTRIP8(2, 3, 170, 228); WSTRIP8(2, 1, 170, 32); WSTRIP8(2, 1, 170, 64);
    QSTRIP8(2, 1, 170, 16, 0); QSTRIP8(2, 1, 170, 32, 0); QSTRIP8(2, 1, 170, 64, 0); QSTRIP8(2, 3, 170, 32, 0);
    WSTRIP8(2, 1, 240, 228); WSTRIP8(2, 3, 240, 228); QSTRIP8(2, 1, 240, 32, 0); QSTRIP8(2, 1, 240, 64, 0); QSTRIP8(2, 3, 240, 64, 0);
  }
  if (set == 9) {
    printf("=== (9) 5 wg/CU cap, one round of 228-row chunks: small start staggers (x 0.25 us, by (id / 8) %% 16); traffic only and alu=170\n");
    const int CAP = 32768;
#define SSTRIP(U, NB, ALU, STAG) do { Geo g = geo(60, 228, 8192, 1); g.stagger = STAG; const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "reg  own=60 U=%d NB=%d alu=%3d rows=228 stagger=%d (5 wg/CU cap)", U, NB, ALU, STAG); \
    if (run(name, 40.0 * P, [&] { k_strip<60, U, NB, 1, 0, ALU, 0, 4><<<grid, 256, CAP>>>(x, b, xp, g); })) return 1; } while (0)
    SSTRIP(2, 1, 0, 0); SSTRIP(2, 1, 0, 1); SSTRIP(2, 1, 0, 2); SSTRIP(2, 3, 0, 0); SSTRIP(2, 3, 0, 1); SSTRIP(2, 3, 0, 2);
    SSTRIP(2, 1, 170, 0); SSTRIP(2, 1, 170, 1); SSTRIP(2, 1, 170, 2); SSTRIP(2, 3, 170, 0); SSTRIP(2, 3, 170, 1);
    SSTRIP(4, 3, 0, 0); SSTRIP(4, 1, 0, 0); SSTRIP(8, 1, 0, 0); SSTRIP(4, 3, 170, 0); SSTRIP(4, 1, 170, 0);
  }
  if (set == 11) {
    printf("=== (11) the review's shape taken literally: ONE workgroup per CU (1024-row chunks, 280 workgroups) resp. two (512 rows), deep LDS-DMA rings, against the product's 5 per CU\n");
    LSTRIP(60, 8, 1, 0, 0, 1024, 8192, 1); LSTRIP(60, 8, 1, 0, 170, 1024, 8192, 1); LSTRIP(60, 12, 1, 0, 0, 1024, 8192, 1); LSTRIP(60, 12, 1, 0, 170, 1024, 8192, 1);
    LSTRIP(60, 6, 1, 0, 0, 512, 8192, 1); LSTRIP(60, 6, 1, 0, 170, 512, 8192, 1);
    LSTRIP(60, 2, 1, 0, 0, 228, 8192, 1); LSTRIP(60, 2, 1, 0, 170, 228, 8192, 1);
  }
  if (set == 12) {
    printf("=== (12) TILED storage with in-tile halo columns and rows (aligned 1-KiB reads, one contiguous range per workgroup, duplicate stores) against the\n"
           "===      product's row-major strip walk; both capped at 5 workgroups per CU (32 KiB dynamic LDS), same grid of 35 strip groups x bands, 40*P basis\n");
    const int CAP = 32768;
    d2 *tx, *txp; double* tb;
    const uint64_t tpx = (uint64_t)(H / 64 + 1) * 35 * (64 + 4) * 256;      // enough for every RB >= 64 below
    CK(hipMalloc(&tx, tpx * 16)); CK(hipMalloc(&txp, tpx * 16)); CK(hipMalloc(&tb, tpx * 8));
    CK(hipMemset(tx, 0, tpx * 16)); CK(hipMemset(txp, 0, tpx * 16)); CK(hipMemset(tb, 0, tpx * 8));
#define TILE(U, NB, ALU, RB_, CAPB) do { TGeo g; g.H = H; g.W = W; g.RB = RB_; g.strip_groups = 35; g.bands = (H + RB_ - 1) / RB_; g.xcd = 1; \
    const uint32_t grid = g.strip_groups * g.bands; \
    if ((uint64_t)grid * (RB_ + 4) * 256 > tpx) { printf("skip RB=%d\n", RB_); break; } \
    snprintf(name, sizeof name, "tile own=60+4 U=%d NB=%d alu=%3d rows=%3d grid=%u%s", U, NB, ALU, RB_, grid, CAPB ? " (5 wg/CU cap)" : ""); \
    if (run(name, 40.0 * P, [&] { k_tile<U, NB, 1, ALU><<<grid, 256, CAPB>>>(tx, tb, txp, g); })) return 1; } while (0)
#define RSTRIP(U, NB, ALU, ROWS, CAPB) do { Geo g = geo(60, ROWS, 8192, 1); const uint32_t grid = g.strip_groups * g.bands; \
    snprintf(name, sizeof name, "reg  own=60   U=%d NB=%d alu=%3d rows=%3d grid=%u%s", U, NB, ALU, ROWS, grid, CAPB ? " (5 wg/CU cap)" : ""); \
    if (run(name, 40.0 * P, [&] { k_strip<60, U, NB, 1, 0, ALU, 0, 4><<<grid, 256, CAPB>>>(x, b, xp, g); })) return 1; } while (0)
    for (int rep = 0; rep < 2; ++rep) {          // twice: the second pass shows the run-to-run band on this box
      RSTRIP(2, 1, 0, 228, CAP); TILE(2, 1, 0, 228, CAP); RSTRIP(2, 3, 0, 228, CAP); TILE(2, 3, 0, 228, CAP); TILE(4, 1, 0, 228, CAP); TILE(4, 3, 0, 228, CAP);
      RSTRIP(2, 1, 170, 228, CAP); TILE(2, 1, 170, 228, CAP); RSTRIP(2, 3, 170, 228, CAP); TILE(2, 3, 170, 228, CAP); TILE(4, 3, 170, 228, CAP);
    }
    RSTRIP(2, 1, 0, 228, 0); TILE(2, 1, 0, 228, 0); TILE(2, 3, 0, 228, 0);
    TILE(2, 1, 0, 64, CAP); TILE(2, 3, 0, 64, CAP); TILE(2, 1, 0, 128, CAP); TILE(2, 1, 170, 64, CAP); TILE(2, 1, 170, 128, CAP);
    RSTRIP(2, 1, 240, 228, CAP); TILE(2, 1, 240, 228, CAP); TILE(2, 3, 240, 228, CAP);
  }
  return 0;
}
