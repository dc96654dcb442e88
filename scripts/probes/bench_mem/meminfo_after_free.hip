// meminfo_after_free.hip -- can a process SEE that the driver is still clearing device memory somebody freed (this process, or one that has just exited)?
// (round 6: a matrix allocated while that clearing runs is mapped in small pieces and the row-strided kernels run 6-13 % slow on it for its whole life,
// profiles/r06_alloc_settle.txt; the library's wait covers only its own frees.)
//   ./meminfo_after_free hold  GiB     allocate + fill GiB, exit without freeing               (the process "before")
//   ./meminfo_after_free watch SEC     print hipMemGetInfo's free bytes and sysfs mem_info_vram_used of every card whenever one of them changes
//   ./meminfo_after_free both  GiB     allocate + fill, hipFree, then watch 4 s                (the same inside one process)
// hipcc --offload-arch=gfx950 -O2 -o meminfo_after_free meminfo_after_free.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <glob.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::vector<std::string> cards() {
  std::vector<std::string> v; glob_t g;
  if (glob("/sys/class/drm/card*/device/mem_info_vram_used", 0, nullptr, &g) == 0) { for (size_t i = 0; i < g.gl_pathc; ++i) v.push_back(g.gl_pathv[i]); globfree(&g); }
  return v;
}
static long long read_ll(const std::string& p) { FILE* f = fopen(p.c_str(), "r"); if (!f) return -1; long long x = -1; if (fscanf(f, "%lld", &x) != 1) x = -1; fclose(f); return x; }
static void watch(double seconds) {
  auto cs = cards();
  std::vector<long long> last(cs.size(), -2); size_t last_free = 0;
  const double t0 = now_s();
  printf("watching %zu sysfs cards + hipMemGetInfo for %.1f s\n", cs.size(), seconds);
  while (now_s() - t0 < seconds) {
    size_t fr, tot; CHECK(hipMemGetInfo(&fr, &tot));
    bool changed = fr != last_free; last_free = fr;
    std::string line;
    for (size_t i = 0; i < cs.size(); ++i) { long long u = read_ll(cs[i]); if (u != last[i]) { changed = true; last[i] = u; } }
    if (changed) {
      printf("  t=%6.3f s  hipMemGetInfo free %8.3f GiB |", now_s() - t0, fr / 1073741824.0);
      for (size_t i = 0; i < cs.size(); ++i) if (last[i] > (1ll << 30)) printf(" card[%zu] used %8.3f GiB", i, last[i] / 1073741824.0);
      printf("\n"); fflush(stdout);
    }
    struct timespec ts = {0, 20 * 1000 * 1000}; nanosleep(&ts, nullptr);
  }
}
int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "watch";
  const double arg = argc > 2 ? atof(argv[2]) : 3.0;
  if (mode == "watch") { watch(arg); return 0; }
  const size_t bytes = (size_t)(arg * 1073741824.0);
  void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 1, bytes)); CHECK(hipDeviceSynchronize());
  size_t fr, tot; CHECK(hipMemGetInfo(&fr, &tot));
  printf("%s: %.0f GiB resident, hipMemGetInfo free %.3f of %.3f GiB\n", mode.c_str(), arg, fr / 1073741824.0, tot / 1073741824.0); fflush(stdout);
  if (mode == "hold") return 0;                       // exit with the block allocated
  const double t0 = now_s(); CHECK(hipFree(p)); printf("hipFree took %.3f s\n", now_s() - t0);
  watch(4.0);
  return 0;
}
