// handoff.hip -- cross-workgroup hand-off latency probe (MI355X).  Two workgroups ping-pong a counter through two
// 8-byte mailboxes; each mode pairs a store flavour with a poll flavour.  Reports the round trip / 2 per mode and
// the dependent-load latency of each load flavour on an L2-resident line.
//   hipcc --offload-arch=gfx950 -O3 -o handoff handoff.hip && ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { LD_SC1 = 0, LD_SC0SC1 = 1, LD_INV_PLAIN = 2, LD_SC0 = 3, LD_NT_SC1 = 4, LD_SMEM_GLC = 5, LD_SMEM_INV = 6 };
enum { ST_SC1 = 0, ST_SC0SC1 = 1, ST_PLAIN_WB = 2, ST_PLAIN = 3, ST_SC0 = 4, ST_ATOMIC = 5 };

template <int LD>
__device__ __forceinline__ unsigned long long poll_load(const unsigned long long* p) {
  unsigned long long v;
  if (LD == LD_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == LD_SC0SC1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == LD_INV_PLAIN) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == LD_SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == LD_NT_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == LD_SMEM_GLC) asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  if (LD == LD_SMEM_INV) asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
template <int ST>
__device__ __forceinline__ void post_store(unsigned long long* p, unsigned long long v) {
  if (ST == ST_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (ST == ST_SC0SC1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  if (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (ST == ST_SC0) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if (ST == ST_ATOMIC) asm volatile("global_atomic_swap_x2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (ST == ST_PLAIN_WB) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

// block `a` and block `b` ping-pong `rounds` times; every other block exits at once
template <int LD, int ST>
__global__ void k_pingpong(unsigned long long* box, int a, int b, int rounds, unsigned long long* ticks, unsigned* xcc) {
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (me < 0) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[me] = id;
  unsigned long long* mine = box + 8 * me;        // separate 64-byte lines
  unsigned long long* theirs = box + 8 * (1 - me);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool lost = false;                               // a bounded spin ran out: stop (both sides then run out quickly)
  for (int r = 1; r <= rounds && !lost; ++r) {
    long spin = 0;
    if (me == 0) {
      post_store<ST>(theirs, (unsigned long long)r);
      while (poll_load<LD>(mine) < (unsigned long long)r && ++spin < 200000) {}
    } else {
      while (poll_load<LD>(mine) < (unsigned long long)r && ++spin < 200000) {}
      post_store<ST>(theirs, (unsigned long long)r);
    }
    lost = spin >= 200000;
  }
  ticks[me] = lost ? 0ull : __builtin_amdgcn_s_memrealtime() - t0;   // 100 MHz; 0 = the mode never saw the other side
}

template <int LD>
__global__ void k_chase(const unsigned long long* buf, int iters, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long idx = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) idx = poll_load<LD>(buf + idx);
  out[0] = __builtin_amdgcn_s_memrealtime() - t0;
  out[1] = idx;
}

// one wave: write-through store to a fresh line, wait for the ack and 2 us more, then time a device-scope load of it
// and a second one right after: is a line that was just written through still resident in L2?
template <int ST>
__global__ void k_readback(unsigned long long* lines, int n, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long first = 0, second = 0, sink = 0;
  for (int i = 0; i < n; ++i) {
    unsigned long long* p = lines + 8 * i;
    post_store<ST>(p, (unsigned long long)i + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - w0 < 200) {}
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    sink += poll_load<LD_SC1>(p);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink += poll_load<LD_SC1>(p);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    first += t1 - t0; second += t2 - t1;
  }
  out[0] = first; out[1] = second; out[2] = sink;
}
template <int ST>
static void readback(const char* name, unsigned long long* lines, unsigned long long* out) {
  const int n = 512;
  CHECK(hipMemset(lines, 0, n * 64));
  k_readback<ST><<<1, 64>>>(lines, n, out);
  CHECK(hipDeviceSynchronize());
  unsigned long long o[3];
  CHECK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
  printf("sc1 load 2 us after %-22s first %7.1f ticks, again %7.1f ticks (s_memtime)\n", name, (double)o[0] / n, (double)o[1] / n);
}

template <int LD, int ST>
static void run(const char* name, unsigned long long* box, unsigned long long* ticks, unsigned* xcc, int a, int b) {
  const int rounds = 2000;
  CHECK(hipMemset(box, 0, 1024));
  k_pingpong<LD, ST><<<256, 64>>>(box, a, b, rounds, ticks, xcc);
  CHECK(hipDeviceSynchronize());
  unsigned long long t[2]; unsigned x[2];
  CHECK(hipMemcpy(t, ticks, sizeof t, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(x, xcc, sizeof x, hipMemcpyDeviceToHost));
  printf("%-34s blocks %3d/%3d (xcc %u/%u): one-way hand-off %7.1f ns\n", name, a, b, x[0], x[1], t[0] * 10.0 / rounds / 2);
}

template <int LD>
static void chase(const char* name, unsigned long long* buf, unsigned long long* out) {
  k_chase<LD><<<1, 64>>>(buf, 20000, out);
  CHECK(hipDeviceSynchronize());
  unsigned long long o[2];
  CHECK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
  printf("dependent load %-20s %7.1f ns\n", name, o[0] * 10.0 / 20000);
}

int main() {
  unsigned long long *box, *ticks, *buf, *out; unsigned* xcc;
  CHECK(hipMalloc(&box, 1024)); CHECK(hipMalloc(&ticks, 64)); CHECK(hipMalloc(&xcc, 64)); CHECK(hipMalloc(&out, 64));
  const int N = 4096;                                  // 32 KiB chase ring, stride 17 lines
  CHECK(hipMalloc(&buf, N * 8));
  unsigned long long* h = (unsigned long long*)malloc(N * 8);
  for (int i = 0; i < N; ++i) h[i] = (i + 17 * 8) % N;
  CHECK(hipMemcpy(buf, h, N * 8, hipMemcpyHostToDevice));
  chase<LD_SC1>("sc1", buf, out);
  chase<LD_SC0SC1>("sc0 sc1", buf, out);
  chase<LD_NT_SC1>("sc1 nt", buf, out);
  chase<LD_SC0>("sc0", buf, out);
  chase<LD_INV_PLAIN>("buffer_inv sc1 + plain", buf, out);
  chase<LD_SMEM_GLC>("s_load glc", buf, out);
  chase<LD_SMEM_INV>("s_dcache_inv + s_load", buf, out);
  {
    unsigned long long* lines;
    CHECK(hipMalloc(&lines, 512 * 64));
    readback<ST_SC1>("store sc1", lines, out);
    readback<ST_PLAIN>("plain store", lines, out);
    readback<ST_ATOMIC>("atomic swap sc1", lines, out);
  }
  for (int pass = 0; pass < 2; ++pass) {
    const int a = 0, b = pass == 0 ? 8 : 1;            // block 8: same XCD as block 0 under round-robin dispatch; block 1: next XCD
    printf("--- %s\n", pass == 0 ? "same XCD (blocks 0 and 8)" : "different XCDs (blocks 0 and 1)");
    run<LD_SC1, ST_SC1>("store sc1 / poll sc1", box, ticks, xcc, a, b);
    run<LD_SC0SC1, ST_SC0SC1>("store sc0 sc1 / poll sc0 sc1", box, ticks, xcc, a, b);
    run<LD_NT_SC1, ST_SC1>("store sc1 / poll sc1 nt", box, ticks, xcc, a, b);
    run<LD_INV_PLAIN, ST_SC1>("store sc1 / inv + plain poll", box, ticks, xcc, a, b);
    run<LD_INV_PLAIN, ST_PLAIN_WB>("store + wbl2 / inv + plain poll", box, ticks, xcc, a, b);
    run<LD_SC0, ST_SC1>("store sc1 / poll sc0", box, ticks, xcc, a, b);
    run<LD_SMEM_GLC, ST_SC1>("store sc1 / poll s_load glc", box, ticks, xcc, a, b);
    run<LD_SMEM_INV, ST_SC1>("store sc1 / s_dcache_inv + s_load", box, ticks, xcc, a, b);
    run<LD_SMEM_GLC, ST_PLAIN>("store plain / poll s_load glc", box, ticks, xcc, a, b);
    run<LD_SC1, ST_PLAIN>("store plain / poll sc1", box, ticks, xcc, a, b);
    run<LD_SC1, ST_SC0>("store sc0 / poll sc1", box, ticks, xcc, a, b);
    run<LD_SC1, ST_ATOMIC>("atomic swap sc1 / poll sc1", box, ticks, xcc, a, b);
    run<LD_INV_PLAIN, ST_PLAIN>("store plain / inv + plain poll", box, ticks, xcc, a, b);
  }
  return 0;
}
