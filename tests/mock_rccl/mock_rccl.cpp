// mock_rccl.cpp -- TEST INFRASTRUCTURE, never shipped and never loaded by the product unless a test points FASTA_RCCL_LIB at it.
//
// A stand-in for librccl that lets SEVERAL PROCESSES ON ONE GPU form a communicator (real RCCL refuses two ranks on one device:
// "invalid usage"), so that the product's one-process-per-GPU row sharding -- fh_comm_unique_id / fh_comm_init / one
// ncclAllReduce(n + 3) per iteration / the lock-step timeout verdict -- can run with 2+ real ranks on a one-GPU box.
// Exactly the entry points csrc/fh_host_ctx.h:rccl_load binds.  The all-reduce stages through POSIX shared memory:
// stream sync -> D2H of this rank's buffer into its slot -> barrier -> every rank sums the slots IN RANK ORDER -> barrier -> H2D.
// (Synchronous on the host, which is a stronger ordering than the stream-ordered collective it replaces.)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <vector>

typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclInvalidUsage = 5, ncclSystemError = 2 };
static const size_t kMaxDoubles = (size_t)1 << 21;         // 16 MiB per rank slot
static const int kMaxRanks = 8;

struct Shared {
  volatile unsigned arrive[2];                              // sense-reversing barrier: arrivals per phase
  volatile unsigned phase;
  double slot[kMaxRanks][kMaxDoubles];
};
// communicators made by ncclCommInitAll live in ONE process (one host thread drives them all): their all-reduce can only run once every
// rank's call has been issued, i.e. at ncclGroupEnd
struct Local { int nranks; std::vector<std::vector<double>> slot; };
struct Comm { Shared* sh; int nranks, rank; char name[64]; std::vector<double> tmp; unsigned local_phase; Local* local = nullptr; };
struct Pending { const void* send; void* recv; size_t count; Comm* comm; hipStream_t stream; };
static thread_local int g_depth = 0;
static thread_local std::vector<Pending> g_pending;

static void barrier(Comm* c) {
  Shared* s = c->sh;
  const unsigned ph = c->local_phase & 1u;
  const unsigned n = __sync_add_and_fetch(&s->arrive[ph], 1u);
  if (n == (unsigned)c->nranks) { s->arrive[ph] = 0; __sync_synchronize(); s->phase = c->local_phase + 1; }
  else {
    const time_t t0 = time(nullptr);
    while (s->phase != c->local_phase + 1) {
      if (time(nullptr) - t0 > 60) { fprintf(stderr, "mock_rccl: rank %d waited 60 s at a barrier -- a peer died?\n", c->rank); abort(); }
      usleep(20);
    }
  }
  c->local_phase += 1;
}

extern "C" int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/mock_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
  return ncclSuccess;
}
extern "C" int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidUsage;
  Comm* c = new Comm();
  c->nranks = nranks; c->rank = rank; c->local_phase = 0;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return ncclSystemError;
  } else {
    for (int tries = 0; tries < 3000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(10000); }
    if (fd < 0) return ncclSystemError;
    for (int tries = 0; tries < 3000; ++tries) { off_t len = lseek(fd, 0, SEEK_END); if (len >= (off_t)sizeof(Shared)) break; usleep(10000); }
  }
  c->sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED) return ncclSystemError;
  barrier(c);                                               // everybody has mapped it
  if (rank == 0) shm_unlink(c->name);                       // the mapping lives on; nothing is left behind in /dev/shm
  *comm = c;
  return ncclSuccess;
}
extern "C" int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
  (void)devlist;                                            // (any device list, repeated ids included)
  if (ndev < 1 || ndev > 64) return ncclInvalidUsage;
  Local* g = new Local();
  g->nranks = ndev; g->slot.resize((size_t)ndev);
  for (int r = 0; r < ndev; ++r) {
    Comm* c = new Comm();
    c->sh = nullptr; c->nranks = ndev; c->rank = r; c->local_phase = 0; c->local = g; c->name[0] = 0;
    comms[r] = c;
  }
  return ncclSuccess;
}
extern "C" int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclSuccess;
  if (c->local) { if (c->rank == 0) delete c->local; }      // (the library destroys its shards in reverse order: rank 0 goes last and frees the group)
  else munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}
static int flush_local() {
  // every rank of an in-process group has issued its call: D2H all, sum in rank order, H2D all
  std::vector<Pending> ops;
  ops.swap(g_pending);
  while (!ops.empty()) {
    Local* g = ops[0].comm->local;
    const size_t count = ops[0].count;
    std::vector<Pending> mine, rest;
    for (const Pending& p : ops) ((p.comm->local == g && p.count == count && (int)mine.size() < g->nranks) ? mine : rest).push_back(p);
    if ((int)mine.size() != g->nranks) return ncclInvalidUsage;           // a rank of the group is missing from this ncclGroupEnd
    for (const Pending& p : mine) {
      if (hipStreamSynchronize(p.stream) != hipSuccess) return ncclSystemError;
      g->slot[(size_t)p.comm->rank].resize(count);
      if (hipMemcpy(g->slot[(size_t)p.comm->rank].data(), p.send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
    }
    std::vector<double> sum(count);
    for (size_t i = 0; i < count; ++i) { double acc = g->slot[0][i]; for (int r = 1; r < g->nranks; ++r) acc += g->slot[(size_t)r][i]; sum[i] = acc; }
    for (const Pending& p : mine)
      if (hipMemcpy(p.recv, sum.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
    ops.swap(rest);
  }
  return ncclSuccess;
}
extern "C" int ncclCommCount(const void* comm, int* count) { *count = ((const Comm*)comm)->nranks; return ncclSuccess; }
extern "C" int ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (datatype != 8 || op != 0 || count > kMaxDoubles) return ncclInvalidUsage;      // ncclDouble, ncclSum
  if (c->local) {                                           // in-process group: runs when all of its ranks have called
    g_pending.push_back(Pending{sendbuff, recvbuff, count, c, stream});
    return g_depth == 0 ? flush_local() : ncclSuccess;      // (outside a group only a group of one can complete)
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclSystemError;
  if (hipMemcpy((void*)c->sh->slot[c->rank], sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
  barrier(c);
  c->tmp.assign(count, 0.0);
  for (size_t i = 0; i < count; ++i) {
    double acc = c->sh->slot[0][i];
    for (int r = 1; r < c->nranks; ++r) acc += c->sh->slot[r][i];
    c->tmp[i] = acc;
  }
  barrier(c);                                               // nobody overwrites a slot before everybody has read it
  if (hipMemcpy(recvbuff, c->tmp.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
  return ncclSuccess;
}
extern "C" int ncclGroupStart() { g_depth += 1; return ncclSuccess; }
extern "C" int ncclGroupEnd() { g_depth -= 1; return (g_depth == 0 && !g_pending.empty()) ? flush_local() : ncclSuccess; }
extern "C" const char* ncclGetErrorString(int r) { return r == ncclSuccess ? "success" : (r == ncclInvalidUsage ? "invalid usage (mock)" : "system error (mock)"); }
