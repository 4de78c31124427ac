// Host side shared by the translation units that implement the C ABI (include/bn254_hip.h): the context, its buffers, and the launchers of
// the kernels more than one unit needs.  bn254_hip.hip owns the definitions; everything here is internal to libbn254hip.so (hidden).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/bn254_hip.h"

#define BN_HIDDEN __attribute__((visibility("hidden")))

// aggregate verify: which of the context's pool buffers hold valid tables, and for which pools (bn254_group.hip: agg_build_tables)
struct AggTables { int valid; size_t n_msgs, n_signers, n_groups, groups4, built_for; int wide2, wide1; };

struct bn254_ctx {
  int device;
  hipStream_t stream;
  Ws ws;
  // staging buffers for the host-pointer entry points (device memory, grown on demand)
  uint8_t* stage[8];
  size_t stage_cap[8];
  int profiling;
  int split_miller;  // A/B knob: one pairing per lane (k_miller_verify_split) instead of the fused 2-pair loop
  Pool pool[8];       // aggregate verify: pk pool, sig pool, H(m) pool, subset sums of the pk pool and of the signature pool, and their widened
                      // forms (16 keys / 8 signatures per entry) for the largest batches (grown on demand)
  size_t pool_fp[8];  // coordinates per entry: 4, 2, 2, 4, 2, 4, 2, 2 ([7]: the pair table the 4-signer signature tables are built from)
  int agg_wide_min_tuples;    // aggregate verify: the widened tables from this many tuples on (0 = never)
  int agg_subset_min_tuples;  // aggregate verify: tabulate subset sums of the pk pool for batches of at least this many tuples (0 = never)
  int agg_sort_by_msg;        // aggregate verify: bucket the tuples by message before the aggregation kernel (default 1; A/B and test knob)
  int pair_lanes;    // verify: Miller loop + final exponentiation on lane pairs (bn254_pair.hip); default on
  int rand_min_batch;      // randomised verify: batches below this size run the exact kernels (default RAND_MIN_BATCH_DEFAULT)
  int rand_items_per_lane; // randomised verify: 0 = by batch size, 1 or 2 forced (A/B and tests)
  int hash_max_tries; // test knob: counters tried before HashToPointError (0 = the reference's 255)
  int trio_wave_roles; // octet layout: the Miller loop's four lane pairs as the four waves of a workgroup (k_miller_verify_quad) instead of one wave
  int hash_direct_width; // small batches: counters tried at once with the square root itself (k_hash_direct); 0 = rounds only
  int trio_max_batch; // verify / check_public_keys batches up to this size run in the octet layout (bn254_trio.hip); 0 = never
  int lm_max_batch;    // ... and up to this size their Miller loop runs as the lane machine (bn254_lmiller.hip); 0 = never
  int nonet_wide;      // ... on eighteen lane pairs (one verify per wave) while the batch is at most one verify per SIMD (BN254_OPT_NONET_WIDE)
  int nonet_max_batch; // ... and up to this size their final exponentiation runs on nine lane pairs per verify (bn254_nonet.hip); 0 = never
  hipEvent_t ev[5];
  int ev_valid;
  int ev_hash_first;   // the recorded intervals are hash, decode, ... (host-pointer verify) instead of decode, hash, ...
  hipStream_t copy_stream;   // host-pointer verify: signatures and keys cross PCIe here while the hash rounds run on `stream`
  uint8_t* pin;              // ... through this PINNED host buffer (hipHostMalloc, grown on demand): BN254_OPT_PINNED_STAGING
  size_t pin_cap;
  int pinned_staging;        // 0 = hipMemcpyAsync straight from the caller's (pageable) buffers
  hipEvent_t copy_done;
  uint64_t msgs_len_next;    // bn254_ctx_expect_msgs_len: size of the d_msgs buffer of the NEXT call that hashes messages
  int msgs_len_declared;
  uint64_t msgs_len_call;    // ... as taken by the entry point now running (MsgsLenScope); UINT64_MAX = not declared
  int entry_depth;           // the host-pointer entry points call their *_device forms: only the outermost one takes the declaration
  int32_t* key_lines;        // keyed verify: registered keys (bn254_ctx_register_keys), see KeyTable in bn254_ws.h
  int32_t* key_xy;           // ... and their affine coordinates (4 x 9 words per key) for the small-batch route
  uint8_t* key_st;
  uint8_t* key_inf;
  size_t n_keys, key_cap;
  hipEvent_t last_done;      // recorded on the CALLER's stream when a *_device call on such a stream returns (CallDone below): what ctx_quiesce
  bool last_done_armed;      // waits for.  The context keeps no handle of a stream it does not own — the caller may destroy its stream any time.
  Pool g2_comb;              // fixed-base table of the G2 generator for key derivation (bn254_group.hip: g2_comb_build), built at the first keygen call
  int g2_comb_ready;
  Pool g1_comb;              // ... and of the G1 generator (PublicKeyG1::from_private_key)
  int g1_comb_ready;
  int g2_fixed_base;         // BN254_OPT_G2_FIXED_BASE (developer option, default 1): key derivation through the comb table; 0 = the 256-step ladder
  AggTables reg_pools;       // bn254_ctx_register_pools: the tables of the registered pools (valid until the next registration or raw-pool call)
  int max_chunk;             // BN254_OPT_MAX_CHUNK: verify-shaped batches above this size are processed in slices (0 = only when the workspace would not fit)
  int assume_free_mb;        // test knob (BN254_OPT_ASSUME_FREE_MB): the automatic rule prices the workspace against this much free memory instead of hipMemGetInfo
  bool fits_w8, fits_quad, fits_trio;   // the device can hold a workgroup of the small-batch kernels (LDS), asked at creation
};

struct ScopedEvents {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t create() {
    hipError_t e = hipEventCreate(&e0);
    return e == hipSuccess ? hipEventCreate(&e1) : e;
  }
  ~ScopedEvents() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
  ScopedEvents() = default;
  ScopedEvents(const ScopedEvents&) = delete;
  ScopedEvents& operator=(const ScopedEvents&) = delete;
};
// the context's thresholds as the routing table's limits (bn254_ws.h: bn_route) — every size-dependent choice of layout goes through here
static inline BnRouteLimits route_limits(const bn254_ctx* c) {
  BnRouteLimits L;
  L.small_max = c->pair_lanes && c->trio_max_batch > 0 ? (size_t)c->trio_max_batch : 0;
  L.lm_max = c->lm_max_batch > 0 ? (size_t)c->lm_max_batch : 0;
  L.nonet_max = c->nonet_max_batch > 0 ? (size_t)c->nonet_max_batch : 0;
  L.nonet_wide_max = c->nonet_wide ? (size_t)NONET_WIDE_MAX_BATCH : 0;
  return L;
}
static inline BnRoute route_for(const bn254_ctx* c, size_t n) { return bn_route(route_limits(c), n); }
// the decode-time helpers of the smallest batches (G2 subgroup ladder on the lane machine's level tables; the pairing API's small-batch
// kernels) follow the lane machine's own threshold, whatever the size of the small-batch family
static inline bool route_lane_machine_helpers(const bn254_ctx* c, size_t n) { return c->pair_lanes && c->lm_max_batch > 0 && n <= (size_t)c->lm_max_batch; }
static inline unsigned grid_for(size_t n) { return (unsigned)((n + BN_WAVE - 1) / BN_WAVE); }

// before a buffer of the context is freed or rewritten: wait for the context's own streams and for the END of its last *_device call on a
// caller's stream (an event the context owns) — not for the whole device (other contexts, other streams and a stream capture running
// elsewhere in the process are left alone)
BN_HIDDEN int ctx_quiesce(bn254_ctx* c);
// Opened by every *_device entry point once it knows its stream: on every exit path (errors included — kernels may have been enqueued)
// the context's own event is recorded behind whatever the call put on a caller's stream.  Nothing is recorded for the context's own stream
// (ctx_quiesce synchronises that one directly), so the default path pays nothing.
struct CallDone {
  bn254_ctx* c;
  hipStream_t s;
  CallDone(bn254_ctx* ctx, hipStream_t stream) : c(ctx), s(stream) {}
  ~CallDone() {
    if (s == c->stream) return;
    if (hipEventRecord(c->last_done, s) == hipSuccess) c->last_done_armed = true;
    else (void)hipGetLastError();                    // e.g. a stream under capture: nothing of this call can outlive the capture's owner
  }
  CallDone(const CallDone&) = delete;
  CallDone& operator=(const CallDone&) = delete;
};
BN_HIDDEN int ws_reserve(bn254_ctx* c, size_t n);
// Oversized batches: the slice length the verify-shaped entry points cut a batch of n items into, or 0 = one piece.  BN254_OPT_MAX_CHUNK when
// set; otherwise only when the workspace of the whole batch (WS_BYTES_PER_ITEM each) would not fit what the device has free (+ what the
// context's present workspace would give back): then the largest multiple of 65 536 items that fits in 80 % of it.
BN_HIDDEN size_t ws_chunk_for(bn254_ctx* c, size_t n);
BN_HIDDEN int stage_reserve(bn254_ctx* c, int slot, size_t bytes);
BN_HIDDEN int stage_in(bn254_ctx* c, int slot, const void* host, size_t bytes);
BN_HIDDEN int stage_out(bn254_ctx* c, int slot, void* host, size_t bytes);
BN_HIDDEN int pool_reserve(bn254_ctx* c, int which, size_t n_fp, size_t entries);
BN_HIDDEN int pool_reserve_one(bn254_ctx* c, Pool* p, size_t n_fp, size_t entries);
static inline bool misaligned(const void* p) { return ((uintptr_t)p & 3u) != 0; }
// host-pointer entry points: an offsets array (n + 1 entries) must be non-decreasing — a kernel computes lengths as
// off[i+1] - off[i], and a wrapped length walks far outside the staged buffer.  O(n) on memory the host already has.
// (The *_device variants cannot look: there the kernels check every span themselves, include/bn254_hip.h.)
static inline bool offsets_ok(const uint64_t* off, size_t n) {
  for (size_t i = 0; i < n; ++i) if (off[i] > off[i + 1]) return false;
  return true;
}
// bn254_ctx_expect_msgs_len is consumed by the NEXT entry point that hashes messages — whatever that call goes on to do: every such
// entry point opens with a MsgsLenScope, which takes the declaration and clears it before any argument check, staging step or
// allocation can return early (a declaration left armed would bound-check an unrelated later call against the wrong length).
struct MsgsLenScope {
  bn254_ctx* c;
  explicit MsgsLenScope(bn254_ctx* ctx) : c(ctx) {
    if (c && c->entry_depth++ == 0) {
      c->msgs_len_call = c->msgs_len_declared ? c->msgs_len_next : UINT64_MAX;
      c->msgs_len_declared = 0;
    }
  }
  ~MsgsLenScope() { if (c) --c->entry_depth; }
  MsgsLenScope(const MsgsLenScope&) = delete;
  MsgsLenScope& operator=(const MsgsLenScope&) = delete;
};

// Enqueue the hash-to-G1 rounds for n messages; points land in planes (px, px+1), statuses in BY_ST_HASH.

#define PROF_MARK(idx) do { if (c->profiling) HIP_TRY(hipEventRecord(c->ev[idx], s)); } while (0)

// launchers of kernels that live in bn254_hip.hip and are used by other units too (a kernel is launched from the unit that defines it)
BN_HIDDEN int launch_decode_g1(bn254_ctx* c, hipStream_t s, const uint8_t* d_pts, size_t n, uint32_t flags, int px, int inf_plane, int accumulate);
BN_HIDDEN int launch_decode_g2(bn254_ctx* c, hipStream_t s, const uint8_t* d_pts, size_t n, uint32_t flags, int accumulate);
BN_HIDDEN int launch_hash_rounds(bn254_ctx* c, hipStream_t s, const uint8_t* d_msgs, const uint64_t* d_off, size_t n, int px, int inf_plane,
                                 uint8_t* d_tries, int mark_finish = -1);
BN_HIDDEN int launch_small_final_exp(bn254_ctx* c, hipStream_t s, size_t n, int use_hash, uint8_t* d_status);
BN_HIDDEN int launch_pair_or_trio(bn254_ctx* c, hipStream_t s, size_t n, int use_hash, uint8_t* d_status, int mode, bool mark);
// one lane per item: k_miller_verify (map / count: a device-side queue of items, or null) and k_final_exp (the arguments of the kernel)
BN_HIDDEN int launch_miller_verify_lane(bn254_ctx* c, hipStream_t s, size_t n, const uint32_t* map, const uint32_t* count);
BN_HIDDEN int launch_final_exp_lane(bn254_ctx* c, hipStream_t s, size_t n, size_t k, size_t item_stride, size_t pair_stride, int use_hash, uint8_t* gt_out,
                                    uint8_t* status_out, int raw_only, size_t base, const uint32_t* map, const uint32_t* count);
BN_HIDDEN int launch_encode_g1(bn254_ctx* c, hipStream_t s, size_t n, int px, int inf_plane, uint8_t* out, uint8_t* status_out);
