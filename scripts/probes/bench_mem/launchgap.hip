// launchgap.hip -- what it costs to get from the end of one kernel to the start of the next on MI355X, by who decides in between (round 6:
// an FBS iteration at 8192^2 is a 113-us kernel plus ~10 us of "launch, wait, read 16 scalars, launch again"; profiles/r06_launchgap.txt).
//   A  launch + hipStreamSynchronize per kernel                         (what fh_step / fh_iterate do: the host decides between launches)
//   B  launch + spin on a host-mapped word the kernel writes last       (host decides, but does not wait for the completion signal)
//   C  N launches back to back, ONE synchronisation                     (nobody decides in between: the floor of a chain of dependent launches)
//   D  the same N launches as one hipGraph launch
// each with an "empty" kernel (256 workgroups that do nothing) and with a kernel that spins ~100 us (s_memrealtime), so that the host has time
// to run ahead in C / D.      hipcc --offload-arch=gfx950 -O3 -o launchgap launchgap.hip && ./launchgap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_work(unsigned long long ticks, volatile unsigned* flag, unsigned seq, unsigned* arrive) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    // the last workgroup to get here publishes the sequence number (as a finaliser would publish its scalars)
    const unsigned t = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1u) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (flag) __hip_atomic_store((unsigned*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  CHECK(hipSetDeviceFlags(hipDeviceScheduleSpin));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* flag; unsigned* flag_dev; unsigned* arrive;
  CHECK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
  CHECK(hipHostGetDevicePointer((void**)&flag_dev, flag, 0));
  CHECK(hipMalloc((void**)&arrive, 64));
  CHECK(hipMemset(arrive, 0, 64));
  const int N = 400, G = 256;
  for (unsigned long long ticks : {0ull, 10000ull}) {            // 0 and 100 us of the 100 MHz clock
    const double body = ticks * 0.01;
    for (int w = 0; w < 20; ++w) k_work<<<G, 256, 0, s>>>(ticks, nullptr, 0, arrive);
    CHECK(hipStreamSynchronize(s));
    // A
    double t0 = now_us();
    for (int i = 0; i < N; ++i) { k_work<<<G, 256, 0, s>>>(ticks, nullptr, 0, arrive); CHECK(hipStreamSynchronize(s)); }
    const double a = (now_us() - t0) / N;
    // B
    *flag = 0;
    t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      k_work<<<G, 256, 0, s>>>(ticks, flag_dev, (unsigned)i, arrive);
      while (*(volatile unsigned*)flag != (unsigned)i) { }
    }
    const double b = (now_us() - t0) / N;
    CHECK(hipStreamSynchronize(s));
    // C
    t0 = now_us();
    for (int i = 0; i < N; ++i) k_work<<<G, 256, 0, s>>>(ticks, nullptr, 0, arrive);
    const double c_issue = (now_us() - t0) / N;
    CHECK(hipStreamSynchronize(s));
    const double c = (now_us() - t0) / N;
    // D
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) k_work<<<G, 256, 0, s>>>(ticks, nullptr, 0, arrive);
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, s)); CHECK(hipStreamSynchronize(s));
    t0 = now_us();
    CHECK(hipGraphLaunch(exec, s)); CHECK(hipStreamSynchronize(s));
    const double d = (now_us() - t0) / N;
    CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    printf("kernel body %6.1f us, %d workgroups:  A launch+sync %7.2f us | B launch+poll mapped word %7.2f us | C back-to-back %7.2f us (host issue %5.2f us) | D hipGraph %7.2f us   [per kernel; minus the body: %5.2f / %5.2f / %5.2f / %5.2f]\n",
           body, G, a, b, c, c_issue, d, a - body, b - body, c - body, d - body);
  }
  return 0;
}
