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
template <int OWN, int U, int NB, int NTS, int LAYOUT, int ALU>
__global__ __launch_bounds__(256) void k_strip(const d2* __restrict__ x, const double* __restrict__ b, d2* __restrict__ xp, const Geo g) {
  constexpr int HALO = OWN == 64 ? 0 : 2;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wg = xcd_order(blockIdx.x, gridDim.x, g.xcd);
  const uint32_t sg = wg % g.strip_groups, rc = wg / g.strip_groups;
  const uint32_t i0 = rc * g.rows_wg;
  const uint32_t rows = min(g.rows_wg, g.H - i0);
  const uint32_t first = (sg * 4 + wave) * OWN;
  const uint32_t cw = (first + lane + 2u * g.W - HALO) % g.W;
  const uint32_t c = first + lane - HALO;
  const bool own = lane >= (uint32_t)HALO && lane < (uint32_t)(HALO + OWN) && c < g.W;
  const int total = (int)rows + 2 * HALO;
  struct Trip { d2 x[U]; double b[U]; };
  auto row_of = [&](int off) -> uint32_t { int r = (int)i0 + off; if (r < 0) r += (int)g.H; if (r >= (int)g.H) r -= (int)g.H; return (uint32_t)r; };
  auto load = [&](Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int s = min(t0 + q, total - 1) - HALO;
      const uint64_t pix = (uint64_t)mem_row<LAYOUT>(row_of(s), g) * g.pitch + cw;
      T.x[q] = x[pix];
      T.b[q] = b[pix];
    }
    asm volatile("" ::: "memory");
  };
  auto store = [&](const Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int s = t0 + q - HALO;
      const d2 v = work<ALU>(T.x[q], T.b[q]);
      if (t0 + q < total && own && s >= 0 && s < (int)rows) st<NTS>(xp + (uint64_t)mem_row<LAYOUT>(i0 + s, g) * g.pitch + c, v);
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
    const int sa = min(2 * pr, total - 1) - HALO, sb = min(2 * pr + 1, total - 1) - HALO;
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
    if (stores && pr >= D && 2 * pr + 1 - HALO < (int)rows) wait_vm<3 * (D - 1) + 2 * (D - 1)>(); else wait_vm<3 * (D - 1)>();
    const uint32_t slot = ring + (uint32_t)(pr % D) * SLOT;
    const d2 xa = *reinterpret_cast<const __attribute__((address_space(3))) d2*>((uintptr_t)(slot + lane * 16));
    const d2 xb = *reinterpret_cast<const __attribute__((address_space(3))) d2*>((uintptr_t)(slot + 1024 + lane * 16));
    const double ba = *reinterpret_cast<const __attribute__((address_space(3))) double*>((uintptr_t)(slot + 2048 + lane * 8));
    const double bb = *reinterpret_cast<const __attribute__((address_space(3))) double*>((uintptr_t)(slot + 2560 + lane * 8));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const d2 va = work<ALU>(xa, ba), vb = work<ALU>(xb, bb);
    const int sa = 2 * pr - HALO, sb = 2 * pr + 1 - HALO;
    if (own && sa >= 0 && sa < (int)rows) st<NTS>(xp + (uint64_t)mem_row<LAYOUT>(i0 + sa, g) * g.pitch + c, va);
    if (own && sb >= 0 && sb < (int)rows && 2 * pr + 1 < total) st<NTS>(xp + (uint64_t)mem_row<LAYOUT>(i0 + sb, g) * g.pitch + c, vb);
    asm volatile("" ::: "memory");
  }
  wait_vm<0>();
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
    Geo g; g.H = H; g.W = W; g.pitch = pitch; g.rows_wg = rows; g.strip_groups = ((W + own - 1) / own + 3) / 4; g.bands = (H + rows - 1) / rows; g.xcd = xcd;
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
  return 0;
}
