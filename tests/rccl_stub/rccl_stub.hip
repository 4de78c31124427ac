// TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl.so.1 that lets ONE GPU rehearse the RCCL branch of
// bn254_mgpu.hip (gather(): ncclGroupStart / per-device ncclAllGather in place + ncclAllReduce / ncclGroupEnd issued from one
// thread over ncclCommInitAll communicators) with G > 1 ranks — the pool gives the builder no multi-GPU box, and RCCL itself
// refuses two ranks on one device.  Found through LD_LIBRARY_PATH by tests/test_mgpu_rccl_stub.py only; the library under test
// loads it with the same dlopen("librccl.so.1") it uses for the real thing.
//
// What it implements is the CONTRACT of the seven calls the layer uses, strictly enough that a misuse fails here as it would
// there: every rank of a communicator must post the same collective, with the same count, inside one group; a collective is
// ordered behind whatever is already on each rank's stream, and every rank's stream leaves it only when all ranks have
// (data movement by device-to-device copies on the destination's stream, events for the ordering).  What it relaxes, and
// announces by exporting bn254_rccl_stub_shared_devices: several ranks may name the same device.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

namespace {
struct World;
struct Comm { World* world; int rank, device; };
struct Op { int kind; const void* send; void* recv; size_t count; int dtype; int redop; Comm* comm; hipStream_t stream; };
struct World {
  int n;
  std::vector<Comm*> comms;
  std::vector<std::vector<Op>> pending;     // per rank, in posting order
  unsigned long long* tmp = nullptr;         // all-reduce staging: [rank][n] words
};
std::mutex g_m;
int g_depth = 0;
bool g_poisoned = false;
std::vector<World*> g_touched;
struct Stats { int allgather, allreduce, inplace, groups, max_ranks_in_group, failed; } g_stats;

size_t dtype_size(int t) {
  switch (t) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; case 6: return 2; case 7: return 4; case 8: return 8; default: return 0; }
}
__global__ void k_sum_u64(const unsigned long long* part, int n, size_t count, unsigned long long* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  unsigned long long s = 0;
  for (int r = 0; r < n; ++r) s += part[(size_t)r * count + i];
  out[i] = s;
}
#define ST(expr) do { if ((expr) != hipSuccess) return ncclUnhandledCudaError; } while (0)

ncclResult_t run_collective(World* w, int j) {
  const int n = w->n;
  const Op& first = w->pending[0][j];
  for (int r = 0; r < n; ++r) {
    const Op& o = w->pending[r][j];
    if (o.kind != first.kind || o.count != first.count || o.dtype != first.dtype || o.redop != first.redop) return ncclInvalidArgument;
  }
  const size_t bytes = first.count * dtype_size(first.dtype);
  if (!bytes) return ncclInvalidArgument;
  std::vector<hipEvent_t> ready(n), done(n);
  for (int r = 0; r < n; ++r) {
    const Op& o = w->pending[r][j];
    ST(hipSetDevice(o.comm->device));
    ST(hipEventCreateWithFlags(&ready[r], hipEventDisableTiming));
    ST(hipEventCreateWithFlags(&done[r], hipEventDisableTiming));
    ST(hipEventRecord(ready[r], o.stream));
  }
  if (first.kind == 0) {                               // all-gather: rank r's recv[s * bytes ..] <- rank s's send
    for (int r = 0; r < n; ++r) {
      const Op& d = w->pending[r][j];
      ST(hipSetDevice(d.comm->device));
      for (int s = 0; s < n; ++s) {
        const Op& src = w->pending[s][j];
        uint8_t* to = (uint8_t*)d.recv + (size_t)s * bytes;
        if (s == r) {
          if ((const void*)to == src.send) { ++g_stats.inplace; continue; }
        } else {
          ST(hipStreamWaitEvent(d.stream, ready[s], 0));
        }
        ST(hipMemcpyAsync(to, src.send, bytes, hipMemcpyDeviceToDevice, d.stream));
      }
    }
    ++g_stats.allgather;
  } else {                                             // all-reduce (sum of 64-bit words)
    if (first.redop != 0 || dtype_size(first.dtype) != 8) return ncclInvalidArgument;
    if (!w->tmp) { ST(hipSetDevice(w->comms[0]->device)); ST(hipMalloc((void**)&w->tmp, (size_t)n * n * 4096)); }
    if (bytes > 4096) return ncclInvalidArgument;
    for (int r = 0; r < n; ++r) {
      const Op& d = w->pending[r][j];
      ST(hipSetDevice(d.comm->device));
      unsigned long long* mine = w->tmp + (size_t)r * n * 512;
      for (int s = 0; s < n; ++s) {
        if (s != r) ST(hipStreamWaitEvent(d.stream, ready[s], 0));
        ST(hipMemcpyAsync(mine + (size_t)s * first.count, w->pending[s][j].send, bytes, hipMemcpyDeviceToDevice, d.stream));
      }
      k_sum_u64<<<(unsigned)((first.count + 63) / 64), 64, 0, d.stream>>>(mine, n, first.count, (unsigned long long*)d.recv);
      ST(hipGetLastError());
    }
    ++g_stats.allreduce;
  }
  // every rank leaves the collective only when all have: a source's buffers may be rewritten as soon as its stream goes on
  for (int r = 0; r < n; ++r) { ST(hipSetDevice(w->comms[r]->device)); ST(hipEventRecord(done[r], w->pending[r][j].stream)); }
  for (int r = 0; r < n; ++r) {
    ST(hipSetDevice(w->comms[r]->device));
    for (int s = 0; s < n; ++s) if (s != r) ST(hipStreamWaitEvent(w->pending[r][j].stream, done[s], 0));
  }
  for (int r = 0; r < n; ++r) { (void)hipEventDestroy(ready[r]); (void)hipEventDestroy(done[r]); }   // released when they complete
  return ncclSuccess;
}
ncclResult_t flush() {
  ncclResult_t rc = ncclSuccess;
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  for (World* w : g_touched) {
    size_t depth = w->pending[0].size();
    int ranks = 0;
    for (int r = 0; r < w->n; ++r) { if (w->pending[r].size() != depth) rc = ncclInvalidUsage; if (!w->pending[r].empty()) ++ranks; }
    if (ranks > g_stats.max_ranks_in_group) g_stats.max_ranks_in_group = ranks;
    if (rc == ncclSuccess && !g_poisoned)
      for (size_t j = 0; j < depth && rc == ncclSuccess; ++j) rc = run_collective(w, (int)j);
    for (int r = 0; r < w->n; ++r) w->pending[r].clear();
  }
  g_touched.clear();
  (void)hipSetDevice(dev0);
  if (g_poisoned) { g_poisoned = false; return ncclInternalError; }
  ++g_stats.groups;
  return rc;
}
ncclResult_t post(const Op& o) {
  std::lock_guard<std::mutex> lk(g_m);
  const char* f = getenv("BN254_RCCL_STUB_FAIL_RANK");
  if (f && *f && atoi(f) == o.comm->rank) { g_poisoned = g_depth > 0; ++g_stats.failed; return ncclInternalError; }
  World* w = o.comm->world;
  w->pending[o.comm->rank].push_back(o);
  bool seen = false;
  for (World* t : g_touched) seen = seen || t == w;
  if (!seen) g_touched.push_back(w);
  if (g_depth == 0) return flush();                    // outside a group: legal for a one-rank communicator only (flush checks)
  return ncclSuccess;
}
}  // namespace

extern "C" {
int bn254_rccl_stub_shared_devices = 1;
void bn254_rccl_stub_stats(int* out6) {
  std::lock_guard<std::mutex> lk(g_m);
  out6[0] = g_stats.allgather; out6[1] = g_stats.allreduce; out6[2] = g_stats.inplace; out6[3] = g_stats.groups;
  out6[4] = g_stats.max_ranks_in_group; out6[5] = g_stats.failed;
}
ncclResult_t ncclCommInitAll(Comm** comms, int n, const int* devs) {
  if (!comms || n < 1) return ncclInvalidArgument;
  World* w = new World();
  w->n = n; w->pending.resize(n);
  for (int r = 0; r < n; ++r) { Comm* c = new Comm{w, r, devs ? devs[r] : r}; w->comms.push_back(c); comms[r] = c; }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(Comm* c) {
  if (!c) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_m);
  World* w = c->world;
  w->comms[c->rank] = nullptr;
  delete c;
  bool any = false;
  for (Comm* x : w->comms) any = any || x;
  if (!any) { if (w->tmp) (void)hipFree(w->tmp); delete w; }
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> lk(g_m); ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  std::lock_guard<std::mutex> lk(g_m);
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  return flush();
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, Comm* comm, hipStream_t s) {
  if (!send || !recv || !comm) return ncclInvalidArgument;
  return post(Op{0, send, recv, count, dt, 0, comm, s});
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, Comm* comm, hipStream_t s) {
  if (!send || !recv || !comm) return ncclInvalidArgument;
  return post(Op{1, send, recv, count, dt, op, comm, s});
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (stub)";
    case ncclInternalError: return "internal error (stub: BN254_RCCL_STUB_FAIL_RANK)";
    case ncclInvalidArgument: return "invalid argument (stub)";
    case ncclInvalidUsage: return "invalid usage (stub: the ranks of a communicator posted different collectives in one group)";
    default: return "error (stub)";
  }
}
}  // extern "C"
