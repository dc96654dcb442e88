// Standalone HBM probe for MI355X: read / write / copy rates for the access shapes the FBS kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT> __global__ __launch_bounds__(256) void k_read(const d2* p, uint64_t n, double* sink) {
  d2 acc = {0, 0};
  const uint64_t stride = (uint64_t)gridDim.x * 256;
#pragma unroll 8
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
  if (acc.x + acc.y == 1.2345e300) sink[0] = acc.x;
}
template <int NT> __global__ __launch_bounds__(256) void k_write(d2* p, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  const d2 v = {1.0, 2.0};
#pragma unroll 8
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
template <int NT> __global__ __launch_bounds__(256) void k_copy(const d2* s, d2* d, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
#pragma unroll 8
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const d2 v = NT ? __builtin_nontemporal_load(s + i) : s[i];
    if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v;
  }
}
// image-style: each wave walks down `rows` rows of its own 1 KiB-wide column strip (row pitch = pitch d2's)
template <int NT, int WR> __global__ __launch_bounds__(256) void k_strip(const d2* s, d2* d, uint32_t H, uint32_t pitch, uint32_t rows) {
  const uint32_t strips = pitch / 64;
  const uint32_t groups = strips / 4;
  const uint32_t sg = blockIdx.x % groups, rc = blockIdx.x / groups;
  const uint32_t col = (sg * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
  d2 acc = {0, 0};
#pragma unroll 8
  for (uint32_t r = 0; r < rows; ++r) {
    const uint64_t i = (uint64_t)(rc * rows + r) * pitch + col;
    const d2 v = NT ? __builtin_nontemporal_load(s + i) : s[i];
    if (WR) { if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v; } else acc += v;
  }
  if (!WR && acc.x + acc.y == 1.2345e300) d[0] = acc;
}

// TV-K-fwd-like stream mix per pixel: read 16 B + 8 B + 8 B, write 16 B + 8 B, in image strips.
// PIX2 = 0: one pixel per lane (8-byte accesses for the scalar planes); PIX2 = 1: two pixels per lane so that
// every access is 16 B (pixels L and L+64 of a 128-column strip for the pair planes, pixels 2L,2L+1 for the scalars).
template <int PIX2, int NT> __global__ __launch_bounds__(256) void k_tvmix(const d2* x, const double* z, const double* b, d2* xp, double* zn,
                                                                            uint32_t W, uint32_t rows) {
  const uint32_t cols_per_wave = PIX2 ? 128 : 64;
  const uint32_t groups = W / (4 * cols_per_wave);
  const uint32_t sg = blockIdx.x % groups, rc = blockIdx.x / groups;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c0 = (sg * 4 + wave) * cols_per_wave;
#pragma unroll 4
  for (uint32_t r = 0; r < rows; ++r) {
    const uint64_t row = (uint64_t)(rc * rows + r) * W;
    if (!PIX2) {
      const uint64_t i = row + c0 + lane;
      const d2 v = NT ? __builtin_nontemporal_load(x + i) : x[i];
      const double a = NT ? __builtin_nontemporal_load(z + i) : z[i];
      const double bb = NT ? __builtin_nontemporal_load(b + i) : b[i];
      d2 o = v; o.x += a; o.y -= bb;
      xp[i] = o; zn[i] = a + bb + v.x;
    } else {
      const uint64_t i0 = row + c0 + lane, i1 = i0 + 64;
      const uint64_t j = (row + c0) / 2 + lane;                 // scalar planes as d2: pixels c0+2L, c0+2L+1
      const d2 v0 = NT ? __builtin_nontemporal_load(x + i0) : x[i0];
      const d2 v1 = NT ? __builtin_nontemporal_load(x + i1) : x[i1];
      const d2 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(z) + j) : reinterpret_cast<const d2*>(z)[j];
      const d2 bb = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(b) + j) : reinterpret_cast<const d2*>(b)[j];
      d2 o0 = v0, o1 = v1; o0.x += a.x; o1.y -= bb.y;
      xp[i0] = o0; xp[i1] = o1;
      d2 zo; zo.x = a.x + bb.x + v0.x; zo.y = a.y + bb.y + v1.x;
      reinterpret_cast<d2*>(zn)[j] = zo;
    }
  }
}

int main() {
  const uint64_t bytes = 8ull << 30;          // 8 GiB per buffer
  const uint64_t n = bytes / 16;
  d2 *a, *b; double* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double gb, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms, gb / ms * 1e3 / 1e9 * 1e0);
  };
  const double GB = (double)bytes;
  for (int grid : {2048, 8192}) {
    printf("--- grid %d\n", grid);
    run("read  plain", GB, [&] { k_read<0><<<grid, 256>>>(a, n, sink); });
    run("read  nt", GB, [&] { k_read<1><<<grid, 256>>>(a, n, sink); });
    run("write plain", GB, [&] { k_write<0><<<grid, 256>>>(b, n); });
    run("write nt", GB, [&] { k_write<1><<<grid, 256>>>(b, n); });
    run("copy  plain (bytes = read+write)", 2 * GB, [&] { k_copy<0><<<grid, 256>>>(a, b, n); });
    run("copy  nt    (bytes = read+write)", 2 * GB, [&] { k_copy<1><<<grid, 256>>>(a, b, n); });
  }
  // image 8192 x (8192 px * 16 B): pitch = 8192 d2; H = 65536 rows in 8 GiB
  const uint32_t pitch = 8192, H = (uint32_t)(n / pitch);
  for (uint32_t rows : {16u, 64u, 256u}) {
    const uint32_t grid = (pitch / 256) * (H / rows);
    printf("--- strips, %u rows per workgroup, grid %u\n", rows, grid);
    run("strip read  plain", GB, [&] { k_strip<0, 0><<<grid, 256>>>(a, b, H, pitch, rows); });
    run("strip read  nt", GB, [&] { k_strip<1, 0><<<grid, 256>>>(a, b, H, pitch, rows); });
    run("strip copy  plain (read+write)", 2 * GB, [&] { k_strip<0, 1><<<grid, 256>>>(a, b, H, pitch, rows); });
    run("strip copy  nt    (read+write)", 2 * GB, [&] { k_strip<1, 1><<<grid, 256>>>(a, b, H, pitch, rows); });
  }
  {
    // TV mix on an 8192 x 8192 image: x,xp = 1 GiB each (d2 per pixel), z,b,zn = 0.5 GiB each
    const uint32_t W = 8192, Himg = 8192;
    const uint64_t P = (uint64_t)W * Himg;
    d2* x = a; d2* xp = a + P;                       // inside the 8 GiB buffer a
    double* z = (double*)b; double* bb = (double*)b + P; double* zn = (double*)b + 2 * P;
    const double bytes56 = 56.0 * P;
    for (uint32_t rows : {16u, 32u, 64u}) {
      printf("--- TV-like mix (56 B/pixel), %u rows per workgroup\n", rows);
      run("tvmix 1 px/lane plain", bytes56, [&] { k_tvmix<0, 0><<<(W / 256) * (Himg / rows), 256>>>(x, z, bb, xp, zn, W, rows); });
      run("tvmix 1 px/lane nt-loads", bytes56, [&] { k_tvmix<0, 1><<<(W / 256) * (Himg / rows), 256>>>(x, z, bb, xp, zn, W, rows); });
      run("tvmix 2 px/lane plain (all 16 B)", bytes56, [&] { k_tvmix<1, 0><<<(W / 512) * (Himg / rows), 256>>>(x, z, bb, xp, zn, W, rows); });
      run("tvmix 2 px/lane nt-loads (all 16 B)", bytes56, [&] { k_tvmix<1, 1><<<(W / 512) * (Himg / rows), 256>>>(x, z, bb, xp, zn, W, rows); });
    }
  }
  return 0;
}
