// gridbar.hip -- what a grid barrier of co-resident workgroups costs on MI355X, by form (round 5: the persistent FBS loop of csrc/fh_run.h
// passes two per iteration, 12-13 us of a 47-us iteration at 4096^2).  All forms are generation barriers on counters that only grow; every
// workgroup's thread 0 arrives and polls, the other threads wait at __syncthreads.
//   flat      one counter: G atomic adds on ONE address, G pollers on it                       (what fh_run.h shipped first)
//   two-level NG group counters (workgroups b % NG share one: with NG = 8 that is one per XCD) + one release word: an arrival is an atomic
//             add on its group's counter; the last arriver of a group adds to the top counter; the last of those publishes the generation
//             in a release word that everybody polls with plain sc1 loads (no atomics on the polled line)
//   poll via  vector sc1 load | scalar s_load glc
//   hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_smem(const unsigned* p) {
  unsigned r;
  const unsigned long long a = (unsigned long long)p;
  const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xFFFFFFFFu));
  asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(u) : "memory");
  return r;
}

// mode 0: flat, vector poll; 1: flat, scalar poll; 2: two-level, vector poll; 3: two-level, scalar poll
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, int iters, int ng, int sleep, unsigned long long* ticks) {
  const unsigned G = gridDim.x, b = blockIdx.x;
  unsigned* flat = ctr;                    // [0]
  unsigned* grp = ctr + 64;                // [g * 32]: one 128-byte line per group counter
  unsigned* top = ctr + 64 + 32 * 64;      // top counter
  unsigned* rel = top + 32;                // release word
  const unsigned g = b % (unsigned)ng;
  const unsigned gsize = (G - g + (unsigned)ng - 1u) / (unsigned)ng;     // workgroups in my group
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE < 2) {
        __hip_atomic_fetch_add(flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)it * G;
        while ((MODE == 0 ? ld_sc1(flat) : ld_smem(flat)) < target) { if (sleep) __builtin_amdgcn_s_sleep(1); }
      } else {
        const unsigned old = __hip_atomic_fetch_add(grp + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned)it * gsize) {                                   // last of my group in this generation
          const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (o2 + 1u == (unsigned)it * (unsigned)ng) __hip_atomic_store(rel, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while ((MODE == 2 ? ld_sc1(rel) : ld_smem(rel)) < (unsigned)it) { if (sleep) __builtin_amdgcn_s_sleep(1); }
      }
    }
    __syncthreads();
  }
  if (b == 0 && threadIdx.x == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  unsigned* ctr; unsigned long long* ticks;
  CHECK(hipMalloc(&ctr, 65536)); CHECK(hipMalloc(&ticks, 8));
  const int iters = 2000;
  printf("grid barrier of G co-resident 256-thread workgroups, us per barrier (%d barriers back to back, 100 MHz clock of workgroup 0), %d CUs\n", iters, ncu);
  for (int G : {ncu, ncu / 2, 2 * ncu}) {
    for (int sleep : {1, 0}) {
      for (int mode = 0; mode < 4; ++mode) {
        for (int ng : {8, 16, 32}) {
          if (mode < 2 && ng != 8) continue;
          CHECK(hipMemset(ctr, 0, 65536));
          void (*k)(unsigned*, int, int, int, unsigned long long*) = mode == 0 ? k_bar<0> : mode == 1 ? k_bar<1> : mode == 2 ? k_bar<2> : k_bar<3>;
          k<<<G, 256>>>(ctr, iters, ng, sleep, ticks);
          CHECK(hipDeviceSynchronize());
          unsigned long long t; CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
          printf("G=%4d  %-28s %s  %7.3f us\n", G, mode == 0 ? "flat, vector sc1 poll" : mode == 1 ? "flat, scalar glc poll" : mode == 2 ? "two-level, vector sc1 poll" : "two-level, scalar glc poll",
                 mode < 2 ? "      " : (ng == 8 ? "ng= 8 " : ng == 16 ? "ng=16 " : "ng=32 "), t * 0.01 / iters);
          fflush(stdout);
          (void)sleep;
        }
      }
      printf("   (s_sleep(1) between polls: %s)\n", sleep ? "yes, above" : "no, above");
    }
  }
  return 0;
}
