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
  return 0;
}
