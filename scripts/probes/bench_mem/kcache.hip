// kcache.hip -- is the scalar cache invalidated between kernel launches?  Kernel R reads a buffer with scalar loads
// (constant address space, no glc); between two launches of R the buffer is rewritten (by a fill kernel, by hipMemcpy
// and by hipMemsetD32Async).  Any launch that still returns the old value saw a stale scalar-cache (or L2) line.
//   hipcc --offload-arch=gfx950 -O3 -o kcache kcache.hip && ./kcache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(unsigned* buf, unsigned n, unsigned v) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = v;
}
// every workgroup reads the whole buffer with wave-uniform scalar loads and counts entries != expect
__global__ void k_sread(const unsigned* buf, unsigned n, unsigned expect, unsigned* bad) {
  const auto* cb = (const __attribute__((address_space(4))) unsigned*)(uintptr_t)buf;
  unsigned wrong = 0;
  for (unsigned i = 0; i < n; i += 16) {           // one 64-byte line per step
    const unsigned v = cb[i];
    wrong += v != expect;
  }
  if (threadIdx.x == 0 && wrong) atomicAdd(bad, wrong);
}

int main() {
  const unsigned n = 2048;                         // 8 KiB: fits every scalar cache
  unsigned *buf, *bad;
  CHECK(hipMalloc(&buf, n * 4)); CHECK(hipMalloc(&bad, 4));
  std::vector<unsigned> host(n);
  const char* how[3] = {"fill kernel", "hipMemcpy H2D", "hipMemsetD32Async"};
  for (int mode = 0; mode < 3; ++mode) {
    unsigned total_bad = 0;
    for (unsigned round = 1; round <= 50; ++round) {
      const unsigned v = mode * 1000 + round;
      if (mode == 0) k_fill<<<64, 256>>>(buf, n, v);
      else if (mode == 1) { for (auto& h : host) h = v; CHECK(hipMemcpy(buf, host.data(), n * 4, hipMemcpyHostToDevice)); }
      else CHECK(hipMemsetD32Async((hipDeviceptr_t)buf, (int)v, n, 0));
      CHECK(hipMemsetAsync(bad, 0, 4, 0));
      k_sread<<<1024, 64>>>(buf, n, v, bad);
      unsigned b = 0;
      CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
      total_bad += b;
    }
    printf("buffer rewritten by %-18s: %u stale scalar reads in 50 rounds x 1024 workgroups x %u lines\n", how[mode], total_bad, n / 16);
  }
  // the same without any host synchronisation between the launches (one stream, back-to-back dispatches)
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(hipMemsetAsync(bad, 0, 4, 0));
    for (unsigned round = 1; round <= 200; ++round) {
      const unsigned v = 5000 + mode * 1000 + round;
      if (mode == 0) k_fill<<<64, 256>>>(buf, n, v);
      else CHECK(hipMemsetD32Async((hipDeviceptr_t)buf, (int)v, n, 0));
      k_sread<<<1024, 64>>>(buf, n, v, bad);
    }
    unsigned b = 0;
    CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
    printf("back-to-back, rewritten by %-18s: %u stale scalar reads in 200 rounds\n", mode == 0 ? "fill kernel" : "hipMemsetD32Async", b);
  }
  return 0;
}
