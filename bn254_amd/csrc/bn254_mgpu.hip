// bn254_mgpu.hip — the multi-GPU layer of libbn254hip.so (include/bn254_hip.h, section "Multi-GPU").
//
// ONE process drives the G devices of a node: per device one bn254_ctx, one stream and one host worker thread that is
// started at creation and parked on a condition variable between calls.  A batch is cut into G contiguous shards of
// S = ceil(n / G) items; tuples share no state (/root/reference/src/ecdsa.rs:49-64), so the devices never exchange
// anything on the data path.  The one exchange is the gather of the status bytes (and an 8-byte sum for the pairing
// workload): RCCL's C API over xGMI — ncclAllGather IN PLACE on the caller's G*S-byte buffers (device g's shard sits at
// offset g*S of its own buffer, so "position in the gathered buffer" == "global item index") — or, for handles that list
// a device twice (RCCL refuses two ranks on one device; the single-GPU test rigs do exactly that), peer copies pulled
// by every destination on its own stream.  Everything here is written on top of the library's own public single-GPU
// entry points: this file adds no arithmetic, only the split, the threads and the collective.
//
// librccl.so.1 is loaded with dlopen at the first call that needs it: the single-GPU user never pays for it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

#include "../../include/bn254_hip.h"

#define MG_HIP(expr)                                       \
  do {                                                     \
    hipError_t e_ = (expr);                                \
    if (e_ != hipSuccess) return -(int)e_;                 \
  } while (0)

namespace {

// ---- RCCL through dlopen ------------------------------------------------------------------------------------------------
// The handful of RCCL names this file uses, declared here so that the library builds without the RCCL development headers
// (the values are those of rccl.h / nccl.h, a stable ABI: ncclSuccess = 0, ncclUint8 = 1, ncclUint64 = 5, ncclSum = 0).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint8 = 1, ncclUint64 = 5;
constexpr ncclRedOp_t ncclSum = 0;
struct Rccl {
  void* so;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  const char* (*GetErrorString)(ncclResult_t);
  bool shared_devices_ok;       // the loaded library accepts several ranks on ONE device: only the test stub does (tests/rccl_stub), it says
                                // so by exporting bn254_rccl_stub_shared_devices; RCCL itself refuses, and then so does this layer
};

struct Dev;
typedef int (*JobFn)(bn254_mgpu*, Dev*, void*);

struct Dev {
  int index, device;
  bn254_ctx* ctx;
  hipStream_t stream;
  hipEvent_t ev_done, ev_pulled, ev_end, t0, t1, t2;
  ncclComm_t comm;
  uint64_t* off_tmp;            // host: the shard's message offsets rebased to its first byte
  size_t off_cap;
  unsigned long long* d_sum;    // device: [0] this shard's Gt checksum, [1 .. G] the partials pulled from every device (copy mode)
  bool end_armed;               // ev_end was recorded behind the last *_device call's gather (what destroy waits for: the handle keeps no
                                // handle of a caller's stream, which the caller may have destroyed since)
  float host_ms;                // host entry points: the worker's wall clock
  bool timed;                   // t0/t1/t2 were recorded by the last call
  // worker thread
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  JobFn fn;
  void* arg;
  int state;                    // 0 idle, 1 job posted, 2 job done, 3 quit
  int rc;
  bool started;
};

}  // namespace

struct bn254_mgpu {
  int G;
  Dev* dev;
  bool distinct;                // no device listed twice
  int gather_opt;               // BN254_MGPU_OPT_GATHER
  int timing;
  bool comm_ready;
  Rccl rccl;
  char err[256];
};

namespace {

// ---- kernels (the only device code of this file) ------------------------------------------------------------------------
// additive checksum over the little-endian 64-bit words of a shard's Gt bytes: grid-stride, wave reduction, one atomic per wave
__global__ void __launch_bounds__(256) k_mg_checksum(const unsigned long long* w, size_t n_words, unsigned long long* out) {
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) s += w[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
__global__ void k_mg_sum_partials(const unsigned long long* part, int G, unsigned long long* out) {
  unsigned long long s = 0;
  for (int g = 0; g < G; ++g) s += part[g];
  *out = s;
}

// ---- workers ---------------------------------------------------------------------------------------------------------------
void worker_main(bn254_mgpu* mg, Dev* d) {
  (void)hipSetDevice(d->device);
  for (;;) {
    std::unique_lock<std::mutex> lk(d->m);
    d->cv.wait(lk, [d] { return d->state == 1 || d->state == 3; });
    if (d->state == 3) return;
    JobFn fn = d->fn;
    void* arg = d->arg;
    lk.unlock();
    const int rc = fn(mg, d, arg);
    lk.lock();
    d->rc = rc;
    d->state = 2;
    lk.unlock();
    d->cv.notify_all();
  }
}
// run fn(mg, dev, arg) once per device — on the workers, or on the calling thread for a single device (no hand-off on
// the G = 1 path: its cost against the single-GPU entry point is one function call) — and return the first failure
int run_all(bn254_mgpu* mg, JobFn fn, void* arg) {
  if (mg->G == 1) return fn(mg, &mg->dev[0], arg);
  for (int g = 0; g < mg->G; ++g) {
    Dev* d = &mg->dev[g];
    {
      std::lock_guard<std::mutex> lk(d->m);
      d->fn = fn; d->arg = arg; d->state = 1;
    }
    d->cv.notify_all();
  }
  int rc = 0;
  for (int g = 0; g < mg->G; ++g) {
    Dev* d = &mg->dev[g];
    std::unique_lock<std::mutex> lk(d->m);
    d->cv.wait(lk, [d] { return d->state == 2; });
    d->state = 0;
    if (d->rc && !rc) rc = d->rc;
  }
  return rc;
}

inline size_t shard_len(const bn254_mgpu* mg, size_t n) { return (n + (size_t)mg->G - 1) / (size_t)mg->G; }
inline void shard_range(const bn254_mgpu* mg, size_t n, int g, size_t& lo, size_t& hi) {
  const size_t S = shard_len(mg, n);
  lo = (size_t)g * S < n ? (size_t)g * S : n;
  hi = lo + S < n ? lo + S : n;
}
inline hipStream_t stream_of(Dev* d, void* const* streams) {
  return (streams && streams[d->index]) ? (hipStream_t)streams[d->index] : d->stream;
}

// The whole offsets array is checked ONCE on the calling thread before any shard starts (as the single-GPU entry points do): a
// shard-local check would let shard 0 follow a span that only a later shard's slice shows to be malformed.
inline bool offsets_ok(const uint64_t* off, size_t n) {
  for (size_t i = 0; i < n; ++i) if (off[i] > off[i + 1]) return false;
  return true;
}
// the caller's current HIP device is restored on every exit path of an entry point (the layer calls hipSetDevice on the caller's thread)
struct DeviceGuard {
  int dev;
  bool ok;
  DeviceGuard() : dev(0), ok(hipGetDevice(&dev) == hipSuccess) {}
  ~DeviceGuard() { if (ok) (void)hipSetDevice(dev); }
};

int rebased_offsets(Dev* d, const uint64_t* off, size_t lo, size_t hi) {
  const size_t need = hi - lo + 1;
  if (need > d->off_cap) {
    uint64_t* p = (uint64_t*)realloc(d->off_tmp, need * sizeof(uint64_t));
    if (!p) return BN254_E_NO_MEMORY;
    d->off_tmp = p; d->off_cap = need;
  }
  const uint64_t base = off[lo];
  for (size_t i = 0; i + 1 < need; ++i) if (off[lo + i] > off[lo + i + 1]) return BN254_E_BAD_ARGUMENT;   // backstop: the entry points have checked the whole array
  for (size_t i = 0; i < need; ++i) d->off_tmp[i] = off[lo + i] - base;
  return 0;
}

// ---- RCCL ------------------------------------------------------------------------------------------------------------------
int rccl_fail(bn254_mgpu* mg, const char* what, ncclResult_t r) {
  snprintf(mg->err, sizeof mg->err, "%s: %s", what, mg->rccl.GetErrorString ? mg->rccl.GetErrorString(r) : "RCCL error");
  return BN254_E_RCCL;
}
#define MG_NCCL(what, expr)                                   \
  do {                                                        \
    ncclResult_t r_ = (expr);                                 \
    if (r_ != ncclSuccess) return rccl_fail(mg, what, r_);    \
  } while (0)

int rccl_load(bn254_mgpu* mg) {
  Rccl& R = mg->rccl;
  if (R.so) return 0;
  void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!so) { snprintf(mg->err, sizeof mg->err, "dlopen(librccl.so.1): %s", dlerror()); return BN254_E_RCCL; }
  R.CommInitAll = (decltype(R.CommInitAll))dlsym(so, "ncclCommInitAll");
  R.CommDestroy = (decltype(R.CommDestroy))dlsym(so, "ncclCommDestroy");
  R.AllGather = (decltype(R.AllGather))dlsym(so, "ncclAllGather");
  R.AllReduce = (decltype(R.AllReduce))dlsym(so, "ncclAllReduce");
  R.GroupStart = (decltype(R.GroupStart))dlsym(so, "ncclGroupStart");
  R.GroupEnd = (decltype(R.GroupEnd))dlsym(so, "ncclGroupEnd");
  R.GetErrorString = (decltype(R.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!R.CommInitAll || !R.CommDestroy || !R.AllGather || !R.AllReduce || !R.GroupStart || !R.GroupEnd || !R.GetErrorString) {
    snprintf(mg->err, sizeof mg->err, "librccl.so.1 lacks one of ncclCommInitAll / ncclAllGather / ncclAllReduce / ncclGroup*");
    dlclose(so);
    return BN254_E_RCCL;
  }
  R.shared_devices_ok = dlsym(so, "bn254_rccl_stub_shared_devices") != nullptr;
  R.so = so;
  return 0;
}
// A one-device handle has nothing to gather: it takes the (empty) copy branch and never loads librccl — unless the caller asks for
// the collective explicitly (BN254_MGPU_OPT_GATHER = 1: the one-rank rehearsal of the RCCL calls).
bool use_rccl(const bn254_mgpu* mg) { return mg->gather_opt == 1 || (mg->gather_opt == 0 && mg->distinct && mg->G > 1); }

int comm_init(bn254_mgpu* mg) {
  if (mg->comm_ready) return 0;
  int rc = rccl_load(mg);
  if (rc) return rc;
  if (!mg->distinct && !mg->rccl.shared_devices_ok) { snprintf(mg->err, sizeof mg->err, "RCCL needs distinct devices: this handle lists one twice"); return BN254_E_RCCL; }
  ncclComm_t comms[64];
  int devs[64];
  for (int g = 0; g < mg->G; ++g) devs[g] = mg->dev[g].device;
  MG_NCCL("ncclCommInitAll", mg->rccl.CommInitAll(comms, mg->G, devs));
  for (int g = 0; g < mg->G; ++g) mg->dev[g].comm = comms[g];
  mg->comm_ready = true;
  return 0;
}

// The gather that closes a *_device call.  Every device g has enqueued its shard on its stream and recorded ev_done / t1 there.
// all[g] = its G*S-byte buffer with its own shard at offset g*S; sum != nullptr: the 8-byte checksum all-reduce rides along.
int gather(bn254_mgpu* mg, uint8_t* const* all, size_t S, void* const* streams, uint64_t* const* d_checksum) {
  const int G = mg->G;
  if (use_rccl(mg)) {
    int rc = comm_init(mg);
    if (rc) return rc;
    MG_NCCL("ncclGroupStart", mg->rccl.GroupStart());
    for (int g = 0; g < G; ++g) {
      Dev* d = &mg->dev[g];
      hipStream_t s = stream_of(d, streams);
      ncclResult_t r = mg->rccl.AllGather(all[g] + (size_t)g * S, all[g], S, ncclUint8, d->comm, s);
      if (r == ncclSuccess && d_checksum) r = mg->rccl.AllReduce(d->d_sum, d_checksum[g], 1, ncclUint64, ncclSum, d->comm, s);
      if (r != ncclSuccess) { (void)mg->rccl.GroupEnd(); return rccl_fail(mg, "ncclAllGather / ncclAllReduce", r); }
    }
    MG_NCCL("ncclGroupEnd", mg->rccl.GroupEnd());
  } else {
    // peer copies: destination h waits for every source's ev_done on ITS stream and pulls the G - 1 foreign shards; then every
    // stream waits for every destination's pulls — the call ends as a join across the devices, like a collective, so that the
    // next call may overwrite any shard
    for (int h = 0; h < G; ++h) {
      Dev* dh = &mg->dev[h];
      hipStream_t sh = stream_of(dh, streams);
      MG_HIP(hipSetDevice(dh->device));
      for (int g = 0; g < G; ++g) {
        if (g == h) continue;
        Dev* dg = &mg->dev[g];
        MG_HIP(hipStreamWaitEvent(sh, dg->ev_done, 0));
        MG_HIP(hipMemcpyPeerAsync(all[h] + (size_t)g * S, dh->device, all[g] + (size_t)g * S, dg->device, S, sh));
        if (d_checksum) MG_HIP(hipMemcpyPeerAsync(dh->d_sum + 1 + g, dh->device, dg->d_sum, dg->device, 8, sh));
      }
      if (d_checksum) {
        MG_HIP(hipMemcpyAsync(dh->d_sum + 1 + h, dh->d_sum, 8, hipMemcpyDeviceToDevice, sh));
        k_mg_sum_partials<<<1, 1, 0, sh>>>(dh->d_sum + 1, G, (unsigned long long*)d_checksum[h]);
        MG_HIP(hipGetLastError());
      }
      MG_HIP(hipEventRecord(dh->ev_pulled, sh));
    }
    for (int g = 0; g < G; ++g) {
      Dev* dg = &mg->dev[g];
      hipStream_t sg = stream_of(dg, streams);
      MG_HIP(hipSetDevice(dg->device));
      for (int h = 0; h < G; ++h)
        if (h != g) MG_HIP(hipStreamWaitEvent(sg, mg->dev[h].ev_pulled, 0));
    }
  }
  for (int g = 0; g < G; ++g) {
    Dev* d = &mg->dev[g];
    MG_HIP(hipSetDevice(d->device));
    if (mg->timing) MG_HIP(hipEventRecord(d->t2, stream_of(d, streams)));
    MG_HIP(hipEventRecord(d->ev_end, stream_of(d, streams)));
    d->end_armed = true;
  }
  return 0;
}

// ---- jobs ------------------------------------------------------------------------------------------------------------------
struct HostTimer {
  Dev* d;
  std::chrono::steady_clock::time_point t;
  explicit HostTimer(Dev* dev) : d(dev), t(std::chrono::steady_clock::now()) { d->timed = false; }
  ~HostTimer() { d->host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count(); }
};

struct VerifyHostArgs { const uint8_t* msgs; const uint64_t* off; const uint8_t* sigs; const uint8_t* pks; size_t n; uint32_t flags; uint8_t* status; };
int job_verify_host(bn254_mgpu* mg, Dev* d, void* p) {
  const VerifyHostArgs& a = *(const VerifyHostArgs*)p;
  HostTimer timer(d);
  size_t lo, hi;
  shard_range(mg, a.n, d->index, lo, hi);
  if (lo == hi) return 0;
  int rc = rebased_offsets(d, a.off, lo, hi);
  if (rc) return rc;
  return bn254_batch_verify(d->ctx, a.msgs ? a.msgs + a.off[lo] : nullptr, d->off_tmp, a.sigs + 64 * lo, a.pks + 128 * lo, hi - lo, a.flags,
                            a.status + lo);
}

struct HashHostArgs { const uint8_t* msgs; const uint64_t* off; size_t n; uint8_t* points; uint8_t* status; uint8_t* tries; };
int job_hash_host(bn254_mgpu* mg, Dev* d, void* p) {
  const HashHostArgs& a = *(const HashHostArgs*)p;
  HostTimer timer(d);
  size_t lo, hi;
  shard_range(mg, a.n, d->index, lo, hi);
  if (lo == hi) return 0;
  int rc = rebased_offsets(d, a.off, lo, hi);
  if (rc) return rc;
  return bn254_batch_hash_to_g1(d->ctx, a.msgs ? a.msgs + a.off[lo] : nullptr, d->off_tmp, hi - lo, a.points + 64 * lo, a.status + lo,
                                a.tries ? a.tries + lo : nullptr);
}

struct CompressedHostArgs { const uint8_t* msgs; const uint64_t* off; const uint8_t* sigs33; const uint8_t* pks65; size_t n; uint8_t* status; };
int job_verify_compressed_host(bn254_mgpu* mg, Dev* d, void* p) {
  const CompressedHostArgs& a = *(const CompressedHostArgs*)p;
  HostTimer timer(d);
  size_t lo, hi;
  shard_range(mg, a.n, d->index, lo, hi);
  if (lo == hi) return 0;
  int rc = rebased_offsets(d, a.off, lo, hi);
  if (rc) return rc;
  return bn254_batch_verify_compressed(d->ctx, a.msgs ? a.msgs + a.off[lo] : nullptr, d->off_tmp, a.sigs33 + 33 * lo, a.pks65 + 65 * lo, hi - lo, a.status + lo);
}

struct RegisterKeysArgs { const uint8_t* pks; size_t n_keys; uint32_t flags; uint8_t* key_status; };
int job_register_keys(bn254_mgpu*, Dev* d, void* p) {                   // the whole key set on every device; entry 0 reports the statuses
  const RegisterKeysArgs& a = *(const RegisterKeysArgs*)p;
  return bn254_ctx_register_keys(d->ctx, a.pks, a.n_keys, a.flags, d->index == 0 ? a.key_status : nullptr);
}
struct KeyedHostArgs { const uint8_t* msgs; const uint64_t* off; const uint8_t* sigs; const uint32_t* key_idx; size_t n; uint32_t flags; uint8_t* status; };
int job_verify_keyed_host(bn254_mgpu* mg, Dev* d, void* p) {
  const KeyedHostArgs& a = *(const KeyedHostArgs*)p;
  HostTimer timer(d);
  size_t lo, hi;
  shard_range(mg, a.n, d->index, lo, hi);
  if (lo == hi) return 0;
  int rc = rebased_offsets(d, a.off, lo, hi);
  if (rc) return rc;
  return bn254_batch_verify_keyed(d->ctx, a.msgs ? a.msgs + a.off[lo] : nullptr, d->off_tmp, a.sigs + 64 * lo, a.key_idx + lo, hi - lo, a.flags, a.status + lo);
}

// aggregate verify: the TUPLES are sharded; messages and pools are what every tuple may name, so every device gets all of them (and builds
// its own subset tables).  A shard's tuple_off slice is rebased to its first signer entry.
struct AggregateHostArgs { const uint8_t* msgs; const uint64_t* msg_off; size_t n_msgs; const uint8_t* pk_pool; size_t n_signers; const uint8_t* sig_pool;
                           const uint32_t* tuple_msg; const uint64_t* tuple_off; const uint32_t* signer_idx; size_t n; uint32_t flags; uint8_t* status; };
int job_aggregate_host(bn254_mgpu* mg, Dev* d, void* p) {
  const AggregateHostArgs& a = *(const AggregateHostArgs*)p;
  HostTimer timer(d);
  size_t lo, hi;
  shard_range(mg, a.n, d->index, lo, hi);
  if (lo == hi) return 0;
  int rc = rebased_offsets(d, a.tuple_off, lo, hi);
  if (rc) return rc;
  return bn254_batch_aggregate_verify(d->ctx, a.msgs, a.msg_off, a.n_msgs, a.pk_pool, a.n_signers, a.sig_pool, a.tuple_msg + lo, d->off_tmp,
                                      a.signer_idx + a.tuple_off[lo], hi - lo, a.flags, a.status + lo);
}

struct PairingHostArgs { const uint8_t* g1; const uint8_t* g2; size_t n, k; uint32_t flags; uint8_t* gt; uint8_t* status; uint64_t* partial; };
int job_pairing_host(bn254_mgpu* mg, Dev* d, void* p) {
  const PairingHostArgs& a = *(const PairingHostArgs*)p;
  HostTimer timer(d);
  size_t lo, hi;
  shard_range(mg, a.n, d->index, lo, hi);
  if (a.partial) a.partial[d->index] = 0;
  if (lo == hi) return 0;
  int rc = bn254_batch_pairing(d->ctx, a.g1 + 64 * a.k * lo, a.g2 + 128 * a.k * lo, hi - lo, a.k, a.flags, a.gt + 384 * lo,
                               a.status ? a.status + lo : nullptr);
  if (rc) return rc;
  if (a.partial) {                                   // the bytes are in host memory already: every worker sums its own shard
    uint64_t s = 0;
    const uint8_t* b = a.gt + 384 * lo;
    for (size_t i = 0, words = (hi - lo) * 48; i < words; ++i) { uint64_t w; memcpy(&w, b + 8 * i, 8); s += w; }   // little-endian hosts only (x86-64)
    a.partial[d->index] = s;
  }
  return 0;
}

struct VerifyDevArgs { const uint8_t* const* msgs; const uint64_t* const* off; const uint8_t* const* sigs; const uint8_t* const* pks; size_t n;
                       uint32_t flags; uint8_t* const* all; void* const* streams; };
int job_verify_dev(bn254_mgpu* mg, Dev* d, void* p) {
  const VerifyDevArgs& a = *(const VerifyDevArgs*)p;
  const int g = d->index;
  size_t lo, hi;
  shard_range(mg, a.n, g, lo, hi);
  const size_t S = shard_len(mg, a.n);
  hipStream_t s = stream_of(d, a.streams);
  MG_HIP(hipSetDevice(d->device));
  d->timed = mg->timing != 0;
  if (mg->timing) MG_HIP(hipEventRecord(d->t0, s));
  if (hi > lo) {
    int rc = bn254_batch_verify_device(d->ctx, a.msgs[g], a.off[g], a.sigs[g], a.pks[g], hi - lo, a.flags, a.all[g] + (size_t)g * S, s);
    if (rc) return rc;
  }
  if (mg->timing) MG_HIP(hipEventRecord(d->t1, s));
  MG_HIP(hipEventRecord(d->ev_done, s));
  return 0;
}

struct PairingDevArgs { const uint8_t* const* g1; const uint8_t* const* g2; size_t n, k; uint32_t flags; uint8_t* const* gt; uint8_t* const* all;
                        bool checksum; void* const* streams; };
int job_pairing_dev(bn254_mgpu* mg, Dev* d, void* p) {
  const PairingDevArgs& a = *(const PairingDevArgs*)p;
  const int g = d->index;
  size_t lo, hi;
  shard_range(mg, a.n, g, lo, hi);
  const size_t S = shard_len(mg, a.n);
  hipStream_t s = stream_of(d, a.streams);
  MG_HIP(hipSetDevice(d->device));
  d->timed = mg->timing != 0;
  if (mg->timing) MG_HIP(hipEventRecord(d->t0, s));
  if (a.checksum) MG_HIP(hipMemsetAsync(d->d_sum, 0, 8, s));
  if (hi > lo) {
    int rc = bn254_batch_pairing_device(d->ctx, a.g1[g], a.g2[g], hi - lo, a.k, a.flags, a.gt ? a.gt[g] : nullptr, a.all[g] + (size_t)g * S, s);
    if (rc) return rc;
    if (a.checksum) {
      const size_t words = (hi - lo) * 48;
      size_t blocks = (words + 1023) / 1024;               // four words per lane at least; 2048 workgroups cover the 256 CUs eight times
      if (blocks > 2048) blocks = 2048;
      k_mg_checksum<<<(unsigned)blocks, 256, 0, s>>>((const unsigned long long*)a.gt[g], words, d->d_sum);
      MG_HIP(hipGetLastError());
    }
  }
  if (mg->timing) MG_HIP(hipEventRecord(d->t1, s));
  MG_HIP(hipEventRecord(d->ev_done, s));
  return 0;
}

struct ReserveArgs { size_t per_dev; };
int job_reserve(bn254_mgpu*, Dev* d, void* p) { return bn254_ctx_reserve(d->ctx, ((const ReserveArgs*)p)->per_dev); }

int job_sync(bn254_mgpu*, Dev* d, void*) {
  MG_HIP(hipSetDevice(d->device));
  MG_HIP(hipStreamSynchronize(d->stream));
  return bn254_ctx_synchronize(d->ctx);
}

void destroy_dev(Dev* d) {
  if (d->started) {
    { std::lock_guard<std::mutex> lk(d->m); d->state = 3; }
    d->cv.notify_all();
    if (d->th.joinable()) d->th.join();
  }
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->ctx) bn254_ctx_destroy(d->ctx);
  if (d->d_sum) (void)hipFree(d->d_sum);
  hipEvent_t* evs[6] = {&d->ev_done, &d->ev_pulled, &d->ev_end, &d->t0, &d->t1, &d->t2};
  for (hipEvent_t* e : evs) if (*e) (void)hipEventDestroy(*e);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  free(d->off_tmp);
}

}  // namespace

extern "C" {

int bn254_mgpu_create(const int* devices, int n_dev, bn254_mgpu** out) {
  if (!out || !devices || n_dev < 1 || n_dev > 64) return BN254_E_BAD_ARGUMENT;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BN254_E_NO_DEVICE;   // no CPU fallback, by design
  for (int g = 0; g < n_dev; ++g) if (devices[g] < 0 || devices[g] >= count) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  bn254_mgpu* mg = new (std::nothrow) bn254_mgpu();
  if (!mg) return BN254_E_NO_MEMORY;
  mg->G = n_dev;
  mg->dev = new (std::nothrow) Dev[n_dev]();
  if (!mg->dev) { delete mg; return BN254_E_NO_MEMORY; }
  mg->distinct = true;
  for (int g = 0; g < n_dev; ++g)
    for (int h = 0; h < g; ++h) if (devices[g] == devices[h]) mg->distinct = false;
  int rc = 0;
  for (int g = 0; g < n_dev && !rc; ++g) {
    Dev* d = &mg->dev[g];
    d->index = g; d->device = devices[g];
    hipError_t e = hipSetDevice(d->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_pulled, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_end, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&d->t0);
    if (e == hipSuccess) e = hipEventCreate(&d->t1);
    if (e == hipSuccess) e = hipEventCreate(&d->t2);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_sum, sizeof(unsigned long long) * (size_t)(n_dev + 1));
    if (e != hipSuccess) { rc = -(int)e; break; }
    rc = bn254_ctx_create(d->device, &d->ctx);
    if (rc) break;
    if (n_dev > 1) {
      try {
        d->th = std::thread(worker_main, mg, d);
        d->started = true;
      } catch (...) { rc = BN254_E_NO_MEMORY; }
    }
  }
  if (!rc && mg->distinct && n_dev > 1) {
    // peer access for the copy mode and for RCCL's own transports; a pair that cannot be mapped falls back to staged copies
    for (int g = 0; g < n_dev; ++g) {
      (void)hipSetDevice(devices[g]);
      for (int h = 0; h < n_dev; ++h) {
        int can = 0;
        if (h != g && hipDeviceCanAccessPeer(&can, devices[g], devices[h]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[h], 0);
      }
    }
    (void)hipGetLastError();                          // hipErrorPeerAccessAlreadyEnabled is not a failure
  }
  if (rc) {
    for (int g = 0; g < n_dev; ++g) destroy_dev(&mg->dev[g]);
    delete[] mg->dev;
    delete mg;
    return rc;
  }
  *out = mg;
  return 0;
}

void bn254_mgpu_destroy(bn254_mgpu* mg) {
  if (!mg) return;
  DeviceGuard guard;
  for (int g = 0; g < mg->G; ++g) {                   // nothing of a collective may still be in flight when its communicator goes
    Dev* d = &mg->dev[g];
    (void)hipSetDevice(d->device);
    if (d->end_armed) (void)hipEventSynchronize(d->ev_end);
    (void)hipStreamSynchronize(d->stream);
  }
  if (mg->comm_ready)
    for (int g = 0; g < mg->G; ++g) if (mg->dev[g].comm) (void)mg->rccl.CommDestroy(mg->dev[g].comm);
  for (int g = 0; g < mg->G; ++g) destroy_dev(&mg->dev[g]);
  // librccl stays loaded: unloading a library that owns threads and device state is not safe
  delete[] mg->dev;
  delete mg;
}

int bn254_mgpu_device_count(const bn254_mgpu* mg) { return mg ? mg->G : 0; }
bn254_ctx* bn254_mgpu_ctx(bn254_mgpu* mg, int g) { return (mg && g >= 0 && g < mg->G) ? mg->dev[g].ctx : nullptr; }
size_t bn254_mgpu_shard_len(const bn254_mgpu* mg, size_t n) { return mg ? shard_len(mg, n) : 0; }
size_t bn254_mgpu_gathered_len(const bn254_mgpu* mg, size_t n) { return mg ? shard_len(mg, n) * (size_t)mg->G : 0; }
int bn254_mgpu_shard_range(const bn254_mgpu* mg, size_t n, int g, size_t* lo, size_t* hi) {
  if (!mg || g < 0 || g >= mg->G || !lo || !hi) return BN254_E_BAD_ARGUMENT;
  shard_range(mg, n, g, *lo, *hi);
  return 0;
}
const char* bn254_mgpu_last_error(const bn254_mgpu* mg) { return mg ? mg->err : ""; }

int bn254_mgpu_set_option(bn254_mgpu* mg, int option, int value) {
  if (!mg) return BN254_E_BAD_ARGUMENT;
  if (option == BN254_MGPU_OPT_GATHER) {
    if (value < 0 || value > 2) return BN254_E_BAD_ARGUMENT;
    if (value == 1 && !mg->distinct && (rccl_load(mg) != 0 || !mg->rccl.shared_devices_ok)) return BN254_E_BAD_ARGUMENT;   // RCCL refuses two ranks on one device
    mg->gather_opt = value;
    return 0;
  }
  if (option == BN254_MGPU_OPT_TIMING) { mg->timing = value != 0; return 0; }
  return BN254_E_BAD_ARGUMENT;
}

int bn254_mgpu_reserve(bn254_mgpu* mg, size_t n_total, int init_collectives) {
  if (!mg) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  ReserveArgs a = {shard_len(mg, n_total)};
  int rc = run_all(mg, job_reserve, &a);
  if (rc) return rc;
  if (init_collectives && use_rccl(mg)) return comm_init(mg);
  return 0;
}

int bn254_mgpu_synchronize(bn254_mgpu* mg) {
  if (!mg) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  return run_all(mg, job_sync, nullptr);
}

int bn254_mgpu_last_timing(bn254_mgpu* mg, float* compute_ms, float* collective_ms) {
  if (!mg || !mg->timing || !compute_ms || !collective_ms) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  for (int g = 0; g < mg->G; ++g) {
    Dev* d = &mg->dev[g];
    if (!d->timed) { compute_ms[g] = d->host_ms; collective_ms[g] = 0.0f; continue; }
    MG_HIP(hipSetDevice(d->device));
    MG_HIP(hipEventSynchronize(d->t2));
    MG_HIP(hipEventElapsedTime(&compute_ms[g], d->t0, d->t1));
    MG_HIP(hipEventElapsedTime(&collective_ms[g], d->t1, d->t2));
  }
  return 0;
}

int bn254_mgpu_batch_verify(bn254_mgpu* mg, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint8_t* pks, size_t n,
                            uint32_t flags, uint8_t* status) {
  if (!mg || (n && (!off || !sigs || !pks || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;      // the WHOLE array, before any shard starts
  if (off[n] && !msgs) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  VerifyHostArgs a = {msgs, off, sigs, pks, n, flags, status};
  return run_all(mg, job_verify_host, &a);
}

int bn254_mgpu_batch_hash_to_g1(bn254_mgpu* mg, const uint8_t* msgs, const uint64_t* off, size_t n, uint8_t* points, uint8_t* status,
                                uint8_t* tries) {
  if (!mg || (n && (!off || !points || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;      // the WHOLE array, before any shard starts
  if (off[n] && !msgs) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  HashHostArgs a = {msgs, off, n, points, status, tries};
  return run_all(mg, job_hash_host, &a);
}

int bn254_mgpu_batch_verify_compressed(bn254_mgpu* mg, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs33, const uint8_t* pks65, size_t n,
                                       uint8_t* status) {
  if (!mg || (n && (!off || !sigs33 || !pks65 || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;      // the WHOLE array, before any shard starts
  if (off[n] && !msgs) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  CompressedHostArgs a = {msgs, off, sigs33, pks65, n, status};
  return run_all(mg, job_verify_compressed_host, &a);
}

int bn254_mgpu_register_keys(bn254_mgpu* mg, const uint8_t* pks, size_t n_keys, uint32_t flags, uint8_t* key_status) {
  if (!mg || (n_keys && !pks)) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  RegisterKeysArgs a = {pks, n_keys, flags, key_status};
  return run_all(mg, job_register_keys, &a);
}

int bn254_mgpu_batch_verify_keyed(bn254_mgpu* mg, const uint8_t* msgs, const uint64_t* off, const uint8_t* sigs, const uint32_t* key_idx, size_t n,
                                  uint32_t flags, uint8_t* status) {
  if (!mg || (n && (!off || !sigs || !key_idx || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  if (!offsets_ok(off, n)) return BN254_E_BAD_ARGUMENT;      // the WHOLE array, before any shard starts
  if (off[n] && !msgs) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  KeyedHostArgs a = {msgs, off, sigs, key_idx, n, flags, status};
  return run_all(mg, job_verify_keyed_host, &a);
}

int bn254_mgpu_batch_aggregate_verify(bn254_mgpu* mg, const uint8_t* msgs, const uint64_t* msg_off, size_t n_msgs, const uint8_t* pk_pool, size_t n_signers,
                                      const uint8_t* sig_pool, const uint32_t* tuple_msg, const uint64_t* tuple_off, const uint32_t* signer_idx, size_t n,
                                      uint32_t flags, uint8_t* status) {
  if (!mg || !n_msgs || !n_signers || (n && (!msg_off || !pk_pool || !sig_pool || !tuple_msg || !tuple_off || !signer_idx || !status))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  // as the single-GPU entry point (bn254_group.hip): both offset arrays non-decreasing, or nothing is touched.  Every shard's signer slice
  // signer_idx + tuple_off[lo] of length tuple_off[hi] - tuple_off[lo] then lies inside the caller's signer_idx[0 .. tuple_off[n]).
  if (!offsets_ok(tuple_off, n) || !offsets_ok(msg_off, n_msgs)) return BN254_E_BAD_ARGUMENT;
  if (msg_off[n_msgs] && !msgs) return BN254_E_BAD_ARGUMENT;
  DeviceGuard guard;
  mg->err[0] = 0;
  AggregateHostArgs a = {msgs, msg_off, n_msgs, pk_pool, n_signers, sig_pool, tuple_msg, tuple_off, signer_idx, n, flags, status};
  return run_all(mg, job_aggregate_host, &a);
}

int bn254_mgpu_batch_pairing(bn254_mgpu* mg, const uint8_t* g1, const uint8_t* g2, size_t n, size_t k, uint32_t flags, uint8_t* gt,
                             uint8_t* status, uint64_t* checksum) {
  if (!mg || k == 0 || (n && (!g1 || !g2 || !gt))) return BN254_E_BAD_ARGUMENT;
  if (checksum) *checksum = 0;
  if (n == 0) return 0;
  DeviceGuard guard;
  mg->err[0] = 0;
  uint64_t partial[64];
  PairingHostArgs a = {g1, g2, n, k, flags, gt, status, checksum ? partial : nullptr};
  int rc = run_all(mg, job_pairing_host, &a);
  if (rc) return rc;
  if (checksum) { uint64_t s = 0; for (int g = 0; g < mg->G; ++g) s += partial[g]; *checksum = s; }
  return 0;
}

int bn254_mgpu_batch_verify_device(bn254_mgpu* mg, const uint8_t* const* d_msgs, const uint64_t* const* d_off, const uint8_t* const* d_sigs,
                                   const uint8_t* const* d_pks, size_t n, uint32_t flags, uint8_t* const* d_status_all, void* const* streams) {
  if (!mg || (n && (!d_msgs || !d_off || !d_sigs || !d_pks || !d_status_all))) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  for (int g = 0; g < mg->G; ++g) {
    size_t lo, hi;
    shard_range(mg, n, g, lo, hi);
    if (!d_status_all[g] || (hi > lo && (!d_msgs[g] || !d_off[g] || !d_sigs[g] || !d_pks[g]))) return BN254_E_BAD_ARGUMENT;
  }
  DeviceGuard guard;
  mg->err[0] = 0;
  if (use_rccl(mg)) { int rc = comm_init(mg); if (rc) return rc; }     // before anything is enqueued: a failure leaves nothing in flight
  VerifyDevArgs a = {d_msgs, d_off, d_sigs, d_pks, n, flags, d_status_all, streams};
  int rc = run_all(mg, job_verify_dev, &a);
  if (rc) return rc;
  return gather(mg, d_status_all, shard_len(mg, n), streams, nullptr);
}

int bn254_mgpu_batch_pairing_device(bn254_mgpu* mg, const uint8_t* const* d_g1, const uint8_t* const* d_g2, size_t n, size_t k, uint32_t flags,
                                    uint8_t* const* d_gt, uint8_t* const* d_status_all, uint64_t* const* d_checksum, void* const* streams) {
  if (!mg || k == 0 || (n && (!d_g1 || !d_g2 || !d_status_all)) || (d_checksum && !d_gt)) return BN254_E_BAD_ARGUMENT;
  if (n == 0) return 0;
  for (int g = 0; g < mg->G; ++g) {
    size_t lo, hi;
    shard_range(mg, n, g, lo, hi);
    if (!d_status_all[g] || (d_checksum && !d_checksum[g]) || (hi > lo && (!d_g1[g] || !d_g2[g] || (d_gt && !d_gt[g])))) return BN254_E_BAD_ARGUMENT;
    if (d_gt && d_gt[g] && ((uintptr_t)d_gt[g] & 7u)) return BN254_E_MISALIGNED;
  }
  DeviceGuard guard;
  mg->err[0] = 0;
  if (use_rccl(mg)) { int rc = comm_init(mg); if (rc) return rc; }
  PairingDevArgs a = {d_g1, d_g2, n, k, flags, d_gt, d_status_all, d_checksum != nullptr, streams};
  int rc = run_all(mg, job_pairing_dev, &a);
  if (rc) return rc;
  return gather(mg, d_status_all, shard_len(mg, n), streams, d_checksum);
}

}  // extern "C"
