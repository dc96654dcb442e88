// Mixed read/write ceiling probe for MI355X, built like the product's read probe (fh_dense.h:k_stream_probe):
// persistent workgroups, NB rotating register buffers of U 16-byte accesses per lane (so (NB-1)*U loads stay in flight
// behind the buffer being stored), every wave-level store covers whole 128-byte lines, loads and stores independently
// non-temporal or plain, 1..8 workgroups per CU.  Two traffic shapes:
//   copy : read 16 B, write 16 B per element                          (the guide's float4-copy figure: 6.29 TB/s)
//   tvmix: read 16 B (x) + 8 B (b), write 16 B (xprox) per pixel      (the z-free TV sweep: 24 R + 16 W = 40 B/pixel)
//   hipcc --offload-arch=gfx950 -O3 -o mixprobe mixprobe.hip && ./mixprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// copy: tiles of U*256 pieces; NB buffers rotate (NB = 2 or 3)
template <int U, int NB, int NTL, int NTS>
__global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ s, d2* __restrict__ d, uint64_t npieces) {
  const uint64_t tile = (uint64_t)U * 256;
  const uint64_t ntiles = npieces / tile;                 // npieces is a multiple of the tile
  d2 b0[U], b1[U], b2[NB == 3 ? U : 1];
  auto load = [&](d2 (&buf)[U], uint64_t t) {
    const uint64_t base = (t < ntiles ? t : ntiles - 1) * tile + threadIdx.x;
#pragma unroll
    for (int j = 0; j < U; ++j) buf[j] = ld<NTL>(s + base + (uint64_t)j * 256);
    asm volatile("" ::: "memory");
  };
  auto store = [&](const d2 (&buf)[U], uint64_t t) {
    if (t < ntiles) {
      const uint64_t base = t * tile + threadIdx.x;
#pragma unroll
      for (int j = 0; j < U; ++j) st<NTS>(d + base + (uint64_t)j * 256, buf[j]);
    }
    asm volatile("" ::: "memory");
  };
  const uint64_t g = gridDim.x;
  uint64_t t = blockIdx.x;
  if constexpr (NB == 3) {
    load(b0, t); load(b1, t + g);
    for (; t < ntiles; t += 3 * g) {
      load(b2, t + 2 * g); store(b0, t);
      load(b0, t + 3 * g); store(b1, t + g);
      load(b1, t + 4 * g); store(b2, t + 2 * g);
    }
  } else {
    load(b0, t);
    for (; t < ntiles; t += 2 * g) {
      load(b1, t + g); store(b0, t);
      load(b0, t + 2 * g); store(b1, t + g);
    }
  }
}

// tvmix: per tile of U*256 pixels: x (16 B/pixel), b (8 B/pixel, read as d2 by half the lanes' worth of accesses), xp (16 B/pixel)
// B8 = 1: b is read as ONE 8-byte value per lane and pixel (U loads of 512 B per wave), the way the sweep reads it, instead of U/2
// 16-byte loads; OWN < 64: only the first OWN lanes of each wave store (the sweep's 60 of 64)
template <int U, int NB, int NTL, int NTS, int B8 = 0, int OWN = 64>
__global__ __launch_bounds__(256) void k_tvmix(const d2* __restrict__ x, const d2* __restrict__ b, d2* __restrict__ xp, uint64_t npix) {
  static_assert(U % 2 == 0, "U even");
  const uint64_t tile = (uint64_t)U * 256;
  const uint64_t ntiles = npix / tile;
  struct Buf { d2 x[U]; d2 b[U / 2]; double b8[B8 ? U : 1]; };
  Buf b0, b1, b2;
  auto load = [&](Buf& buf, uint64_t t) {
    const uint64_t tt = t < ntiles ? t : ntiles - 1;
    const uint64_t base = tt * tile + threadIdx.x, bbase = tt * (tile / 2) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < U; ++j) buf.x[j] = ld<NTL>(x + base + (uint64_t)j * 256);
    if (B8) {
#pragma unroll
      for (int j = 0; j < U; ++j) buf.b8[j] = reinterpret_cast<const double*>(b)[base + (uint64_t)j * 256];
    } else {
#pragma unroll
      for (int j = 0; j < U / 2; ++j) buf.b[j] = ld<NTL>(b + bbase + (uint64_t)j * 256);
    }
    asm volatile("" ::: "memory");
  };
  auto store = [&](const Buf& buf, uint64_t t) {
    if (t < ntiles && (threadIdx.x & 63) < OWN) {
      const uint64_t base = t * tile + threadIdx.x;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        d2 v = buf.x[j];
        if (B8) v.x += buf.b8[j]; else { v.x += buf.b[j / 2].x; v.y += buf.b[j / 2].y; }
        st<NTS>(xp + base + (uint64_t)j * 256, v);
      }
    }
    asm volatile("" ::: "memory");
  };
  const uint64_t g = gridDim.x;
  uint64_t t = blockIdx.x;
  if constexpr (NB == 3) {
    load(b0, t); load(b1, t + g);
    for (; t < ntiles; t += 3 * g) {
      load(b2, t + 2 * g); store(b0, t);
      load(b0, t + 3 * g); store(b1, t + g);
      load(b1, t + 4 * g); store(b2, t + 2 * g);
    }
  } else {
    load(b0, t);
    for (; t < ntiles; t += 2 * g) {
      load(b1, t + g); store(b0, t);
      load(b0, t + 2 * g); store(b1, t + g);
    }
  }
}

// pure write in the same shape (what the store side alone sustains)
template <int U, int NTS>
__global__ __launch_bounds__(256) void k_write(d2* __restrict__ d, uint64_t npieces) {
  const uint64_t tile = (uint64_t)U * 256;
  const uint64_t ntiles = npieces / tile;
  const d2 v = {1.0, 2.0};
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t base = t * tile + threadIdx.x;
#pragma unroll
    for (int j = 0; j < U; ++j) st<NTS>(d + base + (uint64_t)j * 256, v);
  }
}
// pure read, same shape as the product's probe
template <int U, int NTL>
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ s, uint64_t npieces, double* sink) {
  const uint64_t tile = (uint64_t)U * 256;
  const uint64_t ntiles = npieces / tile;
  d2 b0[U], b1[U], b2[U];
  double acc = 0, acc2 = 0;
  auto load = [&](d2 (&buf)[U], uint64_t t) {
    const uint64_t base = (t < ntiles ? t : ntiles - 1) * tile + threadIdx.x;
#pragma unroll
    for (int j = 0; j < U; ++j) buf[j] = ld<NTL>(s + base + (uint64_t)j * 256);
    asm volatile("" ::: "memory");
  };
  auto eat = [&](const d2 (&buf)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) { acc += buf[j].x; acc2 += buf[j].y; }
    asm volatile("" ::: "memory");
  };
  const uint64_t g = gridDim.x;
  uint64_t t = blockIdx.x;
  load(b0, t); load(b1, t + g);
  for (; t < ntiles; t += 3 * g) {
    load(b2, t + 2 * g); eat(b0);
    load(b0, t + 3 * g); eat(b1);
    load(b1, t + 4 * g); eat(b2);
  }
  if (acc + acc2 == 1.2345e300) sink[0] = acc;
}

// the plain form a "float4 copy" benchmark usually takes: one 16-byte element per thread (E = 1) or E per thread, transient workgroups
template <int E, int NTL, int NTS>
__global__ __launch_bounds__(256) void k_copy_simple(const d2* __restrict__ s, d2* __restrict__ d, uint64_t n) {
  const uint64_t base = (uint64_t)blockIdx.x * (256 * E) + threadIdx.x;
  d2 v[E];
#pragma unroll
  for (int j = 0; j < E; ++j) v[j] = ld<NTL>(s + base + (uint64_t)j * 256);
#pragma unroll
  for (int j = 0; j < E; ++j) st<NTS>(d + base + (uint64_t)j * 256, v[j]);
}

// The TV sweep's own access SHAPE without its arithmetic: a workgroup = 4 waves side by side, each walking down `rows` image rows
// of a 64-lane column strip (row pitch W pixels: 16 B/pixel for x and xprox, 8 B/pixel for b); OWN = 64: aligned strips, no halo;
// OWN = 60: lanes 2..61 own, the strip starts 2 pixels left of its first owned column (misaligned, neighbours overlap by 4 lanes)
// and HALO extra rows are read above/below the chunk.  Trips of U rows, NB rotating trip buffers (NB = 1: load a trip, store it).
// ORDER = 0: consecutive workgroups are horizontal neighbours (the product's order); 1: vertical neighbours.
// PANEL = 1: the image is stored as column panels of PW = NW*OWN pixels (pixel (row, col) at (col / PW) * H * PW + row * PW + col % PW), one
// workgroup per panel and row chunk: a workgroup then streams ONE contiguous region (rows x PW pixels); only its 2 + 2 halo columns live in the
// neighbouring panels.
template <int OWN, int U, int NB, int NTS, int ORDER, int NW = 4, int PANEL = 0>
__global__ __launch_bounds__(64 * NW) void k_strip(const d2* __restrict__ x, const double* __restrict__ b, d2* __restrict__ xp,
                                                   uint32_t H, uint32_t W, uint32_t pitch, uint32_t rows_wg, uint32_t strip_groups) {
  constexpr int HALO = OWN == 64 ? 0 : 2;                  // columns per side; rows: 2 above, 2 below
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nrc = (H + rows_wg - 1) / rows_wg;
  const uint32_t sg = ORDER ? blockIdx.x / nrc : blockIdx.x % strip_groups, rc = ORDER ? blockIdx.x % nrc : blockIdx.x / strip_groups;
  const uint32_t i0 = rc * rows_wg;
  const uint32_t rows = min(rows_wg, H - i0);
  const uint32_t first = (sg * NW + wave) * OWN;
  const uint32_t cw = (first + lane + 2u * W - HALO) % W;
  const uint32_t c = first + lane - HALO;
  const bool own = lane >= (uint32_t)HALO && lane < (uint32_t)(HALO + OWN) && c < W;
  const int total = (int)rows + 2 * HALO;
  struct Trip { d2 x[U]; double b[U]; };
  auto row_of = [&](int off) -> uint32_t { int r = (int)i0 + off; if (r < 0) r += (int)H; if (r >= (int)H) r -= (int)H; return (uint32_t)r; };
  constexpr uint32_t PW = NW * OWN;
  auto addr = [&](uint32_t row, uint32_t col) -> uint64_t {
    if (PANEL) return (uint64_t)(col / PW) * H * PW + (uint64_t)row * PW + col % PW;
    return (uint64_t)row * pitch + col;
  };
  auto load = [&](Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int s = min(t0 + q, total - 1) - HALO;
      const uint64_t pix = addr(row_of(s), cw);
      T.x[q] = x[pix];
      T.b[q] = b[pix];
    }
    asm volatile("" ::: "memory");
  };
  auto store = [&](const Trip& T, int t0) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int s = t0 + q - HALO;
      if (t0 + q < total && own && s >= 0 && s < (int)rows) { d2 v = T.x[q]; v.x += T.b[q]; st<NTS>(xp + addr(i0 + s, c), v); }
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
  float best = 1e30f, tot = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; tot += ms; if (ms < best) best = ms;
  }
  printf("%-58s best %7.4f ms %6.0f GB/s   mean %7.4f ms %6.0f GB/s\n", name, best, bytes / best / 1e6, tot / 3, bytes / (tot / 3) / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  const uint64_t P = (uint64_t)8192 * 8192;        // pixels / 16-byte pieces (1 GiB per pair plane)
  d2 *x, *xp, *b; double* sink;
  CK(hipMalloc(&x, P * 16)); CK(hipMalloc(&xp, P * 16)); CK(hipMalloc(&b, P * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(x, 0, P * 16)); CK(hipMemset(xp, 0, P * 16)); CK(hipMemset(b, 0, P * 8));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  char name[160];
#define COPY(U, NB, NTL, NTS, G) do { snprintf(name, sizeof name, "copy  U=%-2d NB=%d ld=%s st=%s grid=%d", U, NB, NTL ? "nt" : "pl", NTS ? "nt" : "pl", G); \
    if (run(name, 32.0 * P, [&] { k_copy<U, NB, NTL, NTS><<<G, 256>>>(x, xp, P); })) return 1; } while (0)
#define MIX(U, NB, NTL, NTS, G) do { snprintf(name, sizeof name, "tvmix U=%-2d NB=%d ld=%s st=%s grid=%d", U, NB, NTL ? "nt" : "pl", NTS ? "nt" : "pl", G); \
    if (run(name, 40.0 * P, [&] { k_tvmix<U, NB, NTL, NTS><<<G, 256>>>(x, b, xp, P); })) return 1; } while (0)
  {
    printf("=== tile-shaped tvmix, one change at a time towards the sweep's accesses (grid 256 and 2048)\n");
#define MIXV(U, B8, OWN, G) do { snprintf(name, sizeof name, "tvmix U=%d NB=3 ld=nt st=nt b=%s stores=%d/64 lanes grid=%d", U, B8 ? "8B/lane" : "16B", OWN, G); \
    if (run(name, (24.0 + 16.0 * OWN / 64.0) * P, [&] { k_tvmix<U, 3, 1, 1, B8, OWN><<<G, 256>>>(x, b, xp, P); })) return 1; } while (0)
    for (int G : {256, 2048}) { MIXV(4, 0, 64, G); MIXV(4, 1, 64, G); MIXV(4, 0, 60, G); MIXV(4, 1, 60, G); MIXV(2, 0, 64, G); MIXV(2, 1, 64, G); MIXV(2, 1, 60, G); }
  }
  {
    const uint32_t H = 8192, W = 8192;
    printf("=== TV sweep access shape, no arithmetic (40*P algorithmic bytes); pitch = row stride in pixels\n");
    d2 *xq, *xpq; double* bq;                       // padded-pitch copies (pitch up to W + 256)
    CK(hipMalloc(&xq, (uint64_t)H * (W + 256) * 16)); CK(hipMalloc(&xpq, (uint64_t)H * (W + 256) * 16)); CK(hipMalloc(&bq, (uint64_t)H * (W + 256) * 8));
    CK(hipMemset(xq, 0, (uint64_t)H * (W + 256) * 16)); CK(hipMemset(xpq, 0, (uint64_t)H * (W + 256) * 16)); CK(hipMemset(bq, 0, (uint64_t)H * (W + 256) * 8));
#define STRIP(OWN, U, NB, NTS, ORDER, NW, ROWS, PITCH) do { const uint32_t sgs = ((W + OWN - 1) / OWN + NW - 1) / NW; const uint32_t g = sgs * ((H + ROWS - 1) / ROWS); \
    snprintf(name, sizeof name, "strip own=%d U=%d NB=%d st=%s %s waves=%d rows=%d pitch=%d grid=%u", OWN, U, NB, NTS ? "nt" : "pl", ORDER ? "vert" : "horz", NW, ROWS, PITCH, g); \
    if (run(name, 40.0 * P, [&] { k_strip<OWN, U, NB, NTS, ORDER, NW><<<g, 64 * NW>>>(xq, bq, xpq, H, W, PITCH, ROWS, sgs); })) return 1; } while (0)
#define PSTRIP(OWN, U, NB, NTS, NW, ROWS, PANEL) do { const uint32_t sgs = ((W + OWN - 1) / OWN + NW - 1) / NW; const uint32_t g = sgs * ((H + ROWS - 1) / ROWS); \
    if (PANEL && (uint64_t)sgs * NW * OWN > W + 256) { printf("skipped: panels of %d pixels overrun the arrays\n", NW * OWN); break; } \
    snprintf(name, sizeof name, "strip own=%d U=%d NB=%d st=%s waves=%d rows=%d layout=%s grid=%u", OWN, U, NB, NTS ? "nt" : "pl", NW, ROWS, PANEL ? "panels" : "rows", g); \
    if (run(name, 40.0 * P, [&] { k_strip<OWN, U, NB, NTS, 0, NW, PANEL><<<g, 64 * NW>>>(xq, bq, xpq, H, W, W, ROWS, sgs); })) return 1; } while (0)
    PSTRIP(60, 2, 3, 1, 4, 128, 0); PSTRIP(64, 2, 3, 1, 4, 128, 0); PSTRIP(60, 2, 3, 1, 4, 32, 0);
    // aligned 64-lane strips in PANELS of 256 pixels (32 panels x 4 KiB rows: every workgroup streams one contiguous region, no halo at all)
    PSTRIP(64, 2, 3, 1, 4, 128, 1); PSTRIP(64, 4, 3, 1, 4, 128, 1); PSTRIP(64, 4, 3, 1, 4, 256, 1); PSTRIP(64, 4, 3, 1, 4, 32, 1); PSTRIP(64, 4, 3, 1, 4, 1024, 1);
    PSTRIP(64, 2, 1, 1, 4, 128, 1); PSTRIP(64, 4, 3, 0, 4, 128, 1);
    CK(hipFree(xq)); CK(hipFree(xpq)); CK(hipFree(bq));
  }
  return 0;
  printf("=== transient workgroups, E 16-byte elements per thread\n");
#define SIMPLE(E, NTL, NTS) do { snprintf(name, sizeof name, "copy simple E=%d ld=%s st=%s grid=%llu", E, NTL ? "nt" : "pl", NTS ? "nt" : "pl", (unsigned long long)(P / (256 * E))); \
    if (run(name, 32.0 * P, [&] { k_copy_simple<E, NTL, NTS><<<(unsigned)(P / (256 * E)), 256>>>(x, xp, P); })) return 1; } while (0)
  SIMPLE(1, 0, 0); SIMPLE(1, 1, 1); SIMPLE(1, 1, 0); SIMPLE(4, 0, 0); SIMPLE(4, 1, 1); SIMPLE(4, 1, 0); SIMPLE(8, 0, 0); SIMPLE(8, 1, 0); SIMPLE(16, 1, 0);
  for (int G : {256, 1024}) {
    printf("=== grid %d (%d workgroup(s) per CU)\n", G, G / 256);
    snprintf(name, sizeof name, "read  U=16 ld=nt grid=%d", G);
    run(name, 16.0 * P, [&] { k_read<16, 1><<<G, 256>>>(x, P, (double*)sink); });
    snprintf(name, sizeof name, "write U=8  st=pl grid=%d", G);
    run(name, 16.0 * P, [&] { k_write<8, 0><<<G, 256>>>(xp, P); });
    snprintf(name, sizeof name, "write U=8  st=nt grid=%d", G);
    run(name, 16.0 * P, [&] { k_write<8, 1><<<G, 256>>>(xp, P); });
    COPY(16, 3, 1, 0, G); COPY(16, 3, 1, 1, G); COPY(16, 3, 0, 0, G); COPY(16, 3, 0, 1, G);
    COPY(8, 3, 1, 0, G);  COPY(8, 3, 1, 1, G);  COPY(16, 2, 1, 0, G); COPY(16, 2, 1, 1, G);
    COPY(4, 3, 1, 0, G);  COPY(4, 3, 1, 1, G);
    MIX(16, 3, 1, 0, G); MIX(16, 3, 1, 1, G); MIX(16, 3, 0, 0, G);
    MIX(8, 3, 1, 0, G);  MIX(8, 3, 1, 1, G);  MIX(8, 2, 1, 0, G);
    MIX(4, 3, 1, 0, G);  MIX(4, 3, 1, 1, G);
  }
  return 0;
}
