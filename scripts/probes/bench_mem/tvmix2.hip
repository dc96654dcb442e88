// Traffic-only model of the z-free stencil sweep (read x 16 B + b 8 B, write xprox 16 B per pixel; no arithmetic, no halos):
// what does this read/write mix sustain on MI355X, with one or two pixels per lane and various rows per workgroup?
//   hipcc --offload-arch=gfx950 -O3 -o tvmix2 tvmix2.hip && ./tvmix2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PIX2, int U> __global__ __launch_bounds__(256) void k_mix(const d2* x, const double* b, d2* xp, uint32_t W, uint32_t rows) {
  const uint32_t cols = PIX2 ? 128 : 64;
  const uint32_t groups = W / (4 * cols);
  const uint32_t sg = blockIdx.x % groups, rc = blockIdx.x / groups;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c0 = (sg * 4 + wave) * cols;
  for (uint32_t r0 = 0; r0 < rows; r0 += U) {
    d2 xv[U][PIX2 ? 2 : 1]; d2 bv2[U]; double bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t row = (uint64_t)(rc * rows + r0 + u) * W;
      if (PIX2) {
        xv[u][0] = x[row + c0 + 2 * lane]; xv[u][1] = x[row + c0 + 2 * lane + 1];
        bv2[u] = reinterpret_cast<const d2*>(b)[(row + c0) / 2 + lane];
      } else { xv[u][0] = x[row + c0 + lane]; bv[u] = b[row + c0 + lane]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t row = (uint64_t)(rc * rows + r0 + u) * W;
      if (PIX2) {
        d2 a = xv[u][0], c = xv[u][1]; a.x += bv2[u].x; c.y += bv2[u].y;
        xp[row + c0 + 2 * lane] = a; xp[row + c0 + 2 * lane + 1] = c;
      } else { d2 a = xv[u][0]; a.x += bv[u]; xp[row + c0 + lane] = a; }
    }
  }
}
// pure copy / read references
__global__ __launch_bounds__(256) void k_copy(const d2* s, d2* d, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}

int main() {
  const uint32_t H = 8192, W = 8192; const uint64_t P = (uint64_t)H * W;
  d2 *x, *xp; double* b;
  CK(hipMalloc(&x, P * 16)); CK(hipMalloc(&xp, P * 16)); CK(hipMalloc(&b, P * 8));
  CK(hipMemset(x, 0, P * 16)); CK(hipMemset(xp, 0, P * 16)); CK(hipMemset(b, 0, P * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s %7.4f ms  %6.0f GB/s\n", name, ms, bytes / ms / 1e6); return 0;
  };
  const double bytes = 40.0 * P;
  for (uint32_t rows : {32u, 64u, 128u, 256u}) {
    printf("--- rows per workgroup %u\n", rows);
    run("1 px/lane, 2 rows in flight", bytes, [&] { k_mix<0, 2><<<(W / 256) * (H / rows), 256>>>(x, b, xp, W, rows); });
    run("1 px/lane, 4 rows in flight", bytes, [&] { k_mix<0, 4><<<(W / 256) * (H / rows), 256>>>(x, b, xp, W, rows); });
    run("1 px/lane, 8 rows in flight", bytes, [&] { k_mix<0, 8><<<(W / 256) * (H / rows), 256>>>(x, b, xp, W, rows); });
    run("2 px/lane, 2 rows in flight", bytes, [&] { k_mix<1, 2><<<(W / 512) * (H / rows), 256>>>(x, b, xp, W, rows); });
    run("2 px/lane, 4 rows in flight", bytes, [&] { k_mix<1, 4><<<(W / 512) * (H / rows), 256>>>(x, b, xp, W, rows); });
  }
  run("copy 1 GiB -> 1 GiB (grid 2048)", 32.0 * P, [&] { k_copy<<<2048, 256>>>(x, xp, P); });
  return 0;
}
