/*
 * bn254_hip.h — C ABI of libbn254hip.so, the MI355X (gfx950) batch BN254 aggregate-signature
 * verifier.  This is the drop-in boundary: plain pointers and sizes, no HIP/torch types.
 *
 * The reference (sedaprotocol/bn254, a pure-Rust crate) has no FFI; its only boundary is the
 * Rust API re-exported at /root/reference/src/lib.rs:60-63.  Each entry point below states the
 * reference function whose per-item semantics it reproduces; INTEGRATION.md shows the Rust
 * `extern "C"` block + `ECDSA::batch_verify` shim a maintainer would add.
 *
 * Byte formats (the reference's *uncompressed* encodings):
 *   G1 point  : 64 bytes  x || y                       big-endian   (src/utils.rs:182-194)
 *   G2 point  : 128 bytes x.re || x.im || y.re || y.im big-endian   (src/utils.rs:161-179)
 *   scalar    : 32 bytes big-endian
 *   Gt        : 384 bytes, 12 x BE32 in tower order (Fq12 = Fq6[w]/(w^2-v), Fq6 = Fq2[v]/(v^3-xi)):
 *               a0.re a0.im a1.re a1.im a2.re a2.im b0.re ... b2.im  (build-defined: the reference
 *               never serialises Gt, src/lib.rs:60-63)
 *   identity  : all-zero bytes (the reference's typed API can hold it but to_uncompressed cannot
 *               encode it — PointInJacobian, src/utils.rs:163,184)
 *   messages  : one concatenated byte buffer + n+1 offsets (msg i = msgs[off[i] .. off[i+1]))
 *
 * Per-item status byte: 0 = Ok, otherwise 1 + the index of the reference's Error variant
 * (src/error.rs:6-29):  1 HashToPointError, 2 IndexOutOfBounds, 3 InvalidEncoding,
 * 4 InvalidGroupPoint, 5 InvalidLength, 6 NotMemberError, 7 ToAffineConversion, 8 PointInJacobian,
 * 9 VerificationFailed, 10 SerializationError, 11 HexDecodeFailed.
 *
 * Return value of every call: 0 on success (bad *items* only set their status byte),
 * -(hipError_t) for a HIP runtime failure, BN254_E_* for bad arguments.
 *
 * Ownership/threading: the caller owns every buffer; the library keeps no pointer after a
 * host-pointer call returns.  A bn254_ctx is used by one thread at a time; distinct contexts
 * (distinct devices) are fully concurrent.  There is NO CPU fallback: every entry point runs
 * HIP kernels on the context's device and fails if that is impossible.
 *
 * *_device variants take DEVICE pointers (4-byte aligned, resident in HBM), enqueue on `stream`
 * (a hipStream_t passed as void*) and do not synchronise.  stream = NULL means the context's OWN stream, which is
 * created non-blocking: it has NO implicit ordering with the legacy null stream or any other stream, so a caller
 * that fills the inputs or reads the outputs on another stream must pass that stream (or order the two with events /
 * bn254_ctx_synchronize).  A context carries ONE call in flight: its workspace in HBM is shared by all its calls,
 * so a second *_device call may be enqueued only on the same stream as the first (stream order then keeps them
 * apart) — for concurrent calls on several streams create one context per stream.
 * Offsets arrays (n + 1 entries) must be non-decreasing: the host-pointer entry points check it and return
 * BN254_E_BAD_ARGUMENT.  The *_device variants cannot read their arrays on the host; the kernels check every pair
 * themselves: a message whose offsets are reversed — or, when the caller has declared the size of the message buffer
 * with bn254_ctx_expect_msgs_len, run past it — is never dereferenced and its item reports 5 (InvalidLength); the other
 * items of the batch are unaffected.
 */
#ifndef BN254_HIP_H
#define BN254_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bn254_ctx bn254_ctx;

#define BN254_FLAG_G2_SUBGROUP_CHECK 1u /* decode G2 inputs with the order-r check AffineG2::new performs */
#define BN254_FLAG_REJECT_IDENTITY 2u   /* treat all-zero encodings as InvalidGroupPoint (from_uncompressed behaviour) */
#define BN254_FLAG_RAND64 0x100u        /* bn254_batch_verify_randomized: 64-bit instead of 128-bit random scalars */
#define BN254_FLAG_RAND_GLV 0x200u      /* ... : r_i = k1 + k2*lambda mod r with k1, k2 the two 64-bit halves of the 128 random bits
                                           (lambda = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd, the eigenvalue of
                                           (x, y) -> (beta x, y) on G1): still 2^128 distinct multipliers, 30 % cheaper to apply */

#define BN254_E_BAD_ARGUMENT (-10001)
#define BN254_E_MISALIGNED (-10002)
#define BN254_E_NO_DEVICE (-10003)
#define BN254_E_RCCL (-10004)      /* multi-GPU layer: librccl.so.1 could not be loaded, or an RCCL call failed (bn254_mgpu_last_error has the text) */
#define BN254_E_NO_MEMORY (-10005) /* host allocation or thread creation failed */

/* status codes */
#define BN254_OK 0
#define BN254_ERR_HASH_TO_POINT 1
#define BN254_ERR_INDEX_OUT_OF_BOUNDS 2 /* keyed verify: key_idx >= n_keys; aggregate verify: signer / message index out of range */
#define BN254_ERR_INVALID_ENCODING 3
#define BN254_ERR_INVALID_GROUP_POINT 4
#define BN254_ERR_INVALID_LENGTH 5
#define BN254_ERR_NOT_MEMBER 6
#define BN254_ERR_TO_AFFINE_CONVERSION 7 /* never produced by the library (host mirrors: the reference's vocabulary, src/error.rs:6-29) */
#define BN254_ERR_POINT_IN_JACOBIAN 8
#define BN254_ERR_VERIFICATION_FAILED 9
#define BN254_ERR_SERIALIZATION 10       /* host mirrors only */
#define BN254_ERR_HEX_DECODE_FAILED 11   /* host mirrors only */

const char *bn254_version(void);

/* opaque context: device id, stream, workspace in HBM (grown on demand, reused across calls) */
int bn254_ctx_create(int hip_device, bn254_ctx **out);
void bn254_ctx_destroy(bn254_ctx *ctx);
/* pre-size the HBM workspace for batches of up to n items / n*k pairs (optional; avoids a
 * hipMalloc inside a later *_device call) */
int bn254_ctx_reserve(bn254_ctx *ctx, size_t n_items);
/* the same for the HOST-pointer bn254_batch_verify: workspace plus the staging buffers in HBM for n_items tuples whose messages total
 * msg_bytes, so that the steady state allocates nothing.  (Whenever a call does have to grow a buffer it first waits for the context's
 * own streams and for the stream of the context's last *_device call — not for the whole device.) */
int bn254_ctx_reserve_host(bn254_ctx *ctx, size_t n_items, size_t msg_bytes);
int bn254_ctx_synchronize(bn254_ctx *ctx);
/* Declares the size in bytes of the d_msgs buffer of the NEXT call on this context that hashes messages (verify, verify_compressed,
 * verify_randomized, hash_to_g1, sign, aggregate_verify and their *_device forms); the declaration is consumed by that call.
 * With it every message span is bounds-checked on the device (offsets non-decreasing and <= msgs_len), without it only
 * reversed offset pairs can be detected.  No counterpart in the reference: a Rust slice (src/ecdsa.rs:49) carries its length. */
int bn254_ctx_expect_msgs_len(bn254_ctx *ctx, uint64_t msgs_len);

/* status[i] = what ECDSA::verify(msg_i, sig_i, pk_i) returns (src/ecdsa.rs:49-64):
 * e(H(m), pk) * e(sig, -G2::one()) == 1.  Decoding errors of sig (first) or pk are reported with
 * the code from_uncompressed would give (src/utils.rs:107-127). */
int bn254_batch_verify(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off /* n+1 */, const uint8_t *sigs /* n*64 */,
                       const uint8_t *pks /* n*128 */, size_t n, uint32_t flags, uint8_t *status /* n */);
int bn254_batch_verify_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, const uint8_t *d_sigs,
                              const uint8_t *d_pks, size_t n, uint32_t flags, uint8_t *d_status, void *stream);

/* bn254_batch_verify from the COMPRESSED encodings callers store (serde of the reference: src/serde.rs:39, :54):
 * sigs n*33 B = 0x02/0x03 || x (src/utils.rs:84-104; G1::from_compressed, src/types.rs:233-237), pks n*65 B =
 * 0x0a/0x0b || BE64(x.im*q + x.re) (src/utils.rs:130-158; G2::from_compressed, src/types.rs:91-93, which checks the
 * order-r subgroup).  status[i] = the error the reference's from_compressed would give for the signature, else for
 * the public key (3 InvalidEncoding, 6 NotMemberError), else what verify gives.  No alignment requirement on the
 * 33- / 65-byte arrays. */
int bn254_batch_verify_compressed(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs33,
                                  const uint8_t *pks65, size_t n, uint8_t *status);
int bn254_batch_verify_compressed_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                         const uint8_t *d_sigs33, const uint8_t *d_pks65, size_t n, uint8_t *d_status,
                                         void *stream);

/* Keyed verify — the same check for public keys REGISTERED with the context beforehand (a validator set).
 * The reference validates a PublicKey once, at construction (PublicKey::from_uncompressed, src/types.rs:96-99 ->
 * src/utils.rs:107-116), and every ECDSA::verify (src/ecdsa.rs:49-64) then repeats the key-dependent half of the Miller
 * loop; here registration also tabulates that half (the 87 line functions of the key, 12.5 KB per key in HBM) and a keyed
 * verify reads it back instead of recomputing it: about a quarter of the Miller loop's field products disappear.
 *
 * bn254_ctx_register_keys replaces the context's key set with `n_keys` uncompressed G2 points (n_keys * 128 bytes, host
 * memory).  key_status[j] (may be NULL) = what PublicKey::from_uncompressed reports for key j: 0, 6 (a coordinate >= q) or 4
 * (not on the curve / not in the order-r subgroup — the subgroup check ALWAYS runs here, as in AffineG2::new; flags: only
 * BN254_FLAG_REJECT_IDENTITY is looked at).  An all-zero key is the identity (its pair contributes 1).  The call
 * waits for the context's own streams and for the stream of its last *_device call before it touches the tables — a keyed
 * verify enqueued earlier has finished reading them (a context carries one call in flight) — and returns with the new set in
 * place; the same holds whenever a call has to grow the context's workspace or staging buffers (presize with bn254_ctx_reserve /
 * bn254_ctx_reserve_host to keep that out of the steady state).  Other contexts and streams are not waited for.
 *
 * bn254_batch_verify_keyed[_device]: as bn254_batch_verify with key_idx[i] (uint32) in place of the i-th public key.
 * status[i] = the signature's decode error, else 2 (IndexOutOfBounds) if key_idx[i] >= n_keys, else the key's registration
 * status, else what verify gives.  Same result bytes as bn254_batch_verify(flags | BN254_FLAG_G2_SUBGROUP_CHECK) on the
 * expanded keys.  Batches of up to BN254_OPT_LM_MAX_BATCH tuples run the lane machine's keyed form on the line tables (latency: no twist
 * point to walk), batches of up to BN254_OPT_TRIO_MAX_BATCH the small-batch kernels on the expanded keys; the line tables serve the larger
 * ones too (throughput). */
int bn254_ctx_register_keys(bn254_ctx *ctx, const uint8_t *pks /* n_keys*128 */, size_t n_keys, uint32_t flags, uint8_t *key_status /* n_keys or NULL */);
int bn254_batch_verify_keyed(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off /* n+1 */, const uint8_t *sigs /* n*64 */,
                             const uint32_t *key_idx /* n */, size_t n, uint32_t flags, uint8_t *status /* n */);
int bn254_batch_verify_keyed_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, const uint8_t *d_sigs,
                                    const uint32_t *d_key_idx, size_t n, uint32_t flags, uint8_t *d_status, void *stream);

/* Keyed randomised batch verification — OPT-IN, probabilistic, for REGISTERED keys.  Items that share a public key share the G2
 * argument of their pairings, so 64 of them are checked by ONE product  e(sum r_i H(m_i), pk) * e(sum r_i sig_i, -G2) == 1  (two
 * table-driven Miller loops and one final exponentiation per 64 items; per item the two scalar multiplications by r_i).  Items are
 * grouped by key on the device; r_i as in bn254_batch_verify_randomized (first 16 / 8 bytes of SHA-256(seed32 || le64(i)), same flags
 * BN254_FLAG_RAND64 / BN254_FLAG_RAND_GLV); the items of a failing group are re-checked one by one with the exact keyed kernels.
 * Same inputs and status bytes as bn254_batch_verify_keyed: a non-zero status is always the exact one, a zero is wrong with
 * probability <= 2^-128 (2^-64) per group for a fresh secret seed.  Batches below BN254_OPT_RAND_MIN_BATCH, and contexts without
 * registered keys, take the exact keyed path.  No counterpart in the reference (src/ecdsa.rs:49-64 verifies one tuple at a time). */
int bn254_batch_verify_keyed_randomized(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs,
                                        const uint32_t *key_idx, size_t n, uint32_t flags, const uint8_t *seed32, uint8_t *status);
int bn254_batch_verify_keyed_randomized_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, const uint8_t *d_sigs,
                                               const uint32_t *d_key_idx, size_t n, uint32_t flags, const uint8_t *seed32,
                                               uint8_t *d_status, void *stream);

/* Randomised batch verification — OPT-IN, probabilistic.  No counterpart in the
 * reference, which verifies one tuple at a time (src/ecdsa.rs:49-64); same inputs and status bytes as
 * bn254_batch_verify.  Items are taken 64 at a time; with r_i = the first 16 bytes (BN254_FLAG_RAND64: 8) of
 * SHA-256(seed32 || le64(i)) read little-endian (0 -> 1; BN254_FLAG_RAND_GLV: see the flag), a group passes iff
 *     prod_i e(r_i * H(m_i), pk_i) * e(sum_i r_i * sig_i, -G2::one()) == 1      (over its items that decode and hash)
 * i.e. 64 + 1 Miller loops and ONE final exponentiation per 64 verifies.  Items of a passing group get status 0
 * (or their decode / hash error); every item of a failing group is re-verified exactly (the kernels of
 * bn254_batch_verify), so a non-zero status is always exact.  A zero status is wrong with probability <= 2^-128
 * (2^-64) per group PROVIDED seed32 is fresh, unpredictable to whoever produced the signatures, and the public
 * keys are in the order-r subgroup (validated earlier, or pass BN254_FLAG_G2_SUBGROUP_CHECK).  group_ok (optional,
 * ceil(n/64) bytes): 1 = the group's combined check passed.  The combined check has a fixed latency of one
 * Miller loop + one final exponentiation on n/64 lanes: it pays off from ~100 k items per call on an MI355X; smaller
 * batches are routed to the exact kernels (BN254_OPT_RAND_MIN_BATCH), see DESIGN.md.  Host variant synchronises; device variant only enqueues. */
int bn254_batch_verify_randomized(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs,
                                  const uint8_t *pks, size_t n, uint32_t flags, const uint8_t *seed32, uint8_t *status,
                                  uint8_t *group_ok);
int bn254_batch_verify_randomized_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                         const uint8_t *d_sigs, const uint8_t *d_pks, size_t n, uint32_t flags,
                                         const uint8_t *seed32 /* host memory */, uint8_t *d_status, uint8_t *d_group_ok,
                                         void *stream);

/* points[i] = hash_to_try_and_increment(msg_i) (src/hash.rs:29-63), uncompressed; status 1 =
 * HashToPointError; tries[i] (optional, may be NULL) = number of counters consumed (1..255). */
int bn254_batch_hash_to_g1(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, size_t n, uint8_t *points /* n*64 */,
                           uint8_t *status /* n */, uint8_t *tries /* n or NULL */);
int bn254_batch_hash_to_g1_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t n, uint8_t *d_points,
                                  uint8_t *d_status, uint8_t *d_tries, void *stream);

/* status[i] = (bn::pairing_batch(&[(g1[i*k+j], g2[i*k+j]); k]) == Gt::one()) ? 0 : 9 — the kernel
 * shared by ECDSA::verify and check_public_keys (src/ecdsa.rs:57-63, :86-92). */
int bn254_batch_pairing_check(bn254_ctx *ctx, const uint8_t *g1 /* n*k*64 */, const uint8_t *g2 /* n*k*128 */, size_t n, size_t k,
                              uint32_t flags, uint8_t *status /* n */);
/* gt[i] = prod_j e(g1[i*k+j], g2[i*k+j]) = Miller product ^ ((q^12-1)/r), 384 bytes each.  Batches that cannot fill the chip (n*k <=
 * BN254_OPT_LM_MAX_BATCH pairs, n <= 1024 items) take the small-batch kernels of a verify — the lane machine with the fixed pair skipped, the
 * exact final exponentiation on eighteen lane pairs per item: one pairing in 1.1 ms instead of 5.7; same bytes. */
int bn254_batch_pairing(bn254_ctx *ctx, const uint8_t *g1, const uint8_t *g2, size_t n, size_t k, uint32_t flags,
                        uint8_t *gt /* n*384 */, uint8_t *status /* n */);
int bn254_batch_pairing_device(bn254_ctx *ctx, const uint8_t *d_g1, const uint8_t *d_g2, size_t n, size_t k, uint32_t flags,
                               uint8_t *d_gt /* n*384 or NULL */, uint8_t *d_status, void *stream);

/* status[i] = check_public_keys(pk_g2[i], pk_g1[i]) (src/ecdsa.rs:78-93) */
int bn254_batch_check_public_keys(bn254_ctx *ctx, const uint8_t *pk_g2 /* n*128 */, const uint8_t *pk_g1 /* n*64 */, size_t n,
                                  uint32_t flags, uint8_t *status);

/* group operations: aggregation = `Add for Signature/PublicKey` (src/types.rs:126-132, :264-270),
 * sign / key derivation = G1*Fr, G2*Fr (src/ecdsa.rs:31, src/types.rs:86, :156).
 * reduce_scalar != 0: scalars are first reduced mod r like Fr::from_slice; 0: used as 256-bit integers.
 * p == NULL in the two _mul entry points multiplies the group's GENERATOR — key derivation, PublicKeyG1 / PublicKey::from_private_key
 * (src/types.rs:155-157, :85-87): a fixed base, served from a comb table of the generator's multiples built once per context (65 additions on a
 * lane pair instead of a 256-step ladder on one lane; window entries found by constant-time scans: the scalar is a private key).
 * With explicit G1 points, and in bn254_batch_sign, the multiplication runs a joint 128-step ladder over the curve's endomorphism
 * (k mod r = k1 + k2 lambda with 128-bit halves; every point of the G1 curve has order r, so a raw 256-bit scalar acts mod r — the same
 * point as the 256-step ladder gives); window entries by scans, complete additions. */
int bn254_batch_g1_add(bn254_ctx *ctx, const uint8_t *a /* n*64 */, const uint8_t *b /* n*64 */, size_t n, uint8_t *out, uint8_t *status);
int bn254_batch_g2_add(bn254_ctx *ctx, const uint8_t *a /* n*128 */, const uint8_t *b /* n*128 */, size_t n, uint8_t *out, uint8_t *status);
int bn254_batch_g1_mul(bn254_ctx *ctx, const uint8_t *p /* n*64 */, const uint8_t *scalars /* n*32 */, size_t n, int reduce_scalar,
                       uint8_t *out, uint8_t *status);
int bn254_batch_g2_mul(bn254_ctx *ctx, const uint8_t *p /* n*128 */, const uint8_t *scalars /* n*32 */, size_t n, int reduce_scalar,
                       uint8_t *out, uint8_t *status);
int bn254_batch_g1_mul_device(bn254_ctx *ctx, const uint8_t *d_p, const uint8_t *d_scalars, size_t n, int reduce_scalar,
                              uint8_t *d_out, uint8_t *d_status, void *stream);
int bn254_batch_g2_mul_device(bn254_ctx *ctx, const uint8_t *d_p, const uint8_t *d_scalars, size_t n, int reduce_scalar,
                              uint8_t *d_out, uint8_t *d_status, void *stream);
/* sigs[i] = ECDSA::sign(msg_i, sk_i) = H(msg_i) * sk_i (src/ecdsa.rs:26-35), uncompressed */
int bn254_batch_sign(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sks /* n*32 */, size_t n,
                     uint8_t *sigs /* n*64 */, uint8_t *status);
int bn254_batch_sign_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, const uint8_t *d_sks, size_t n,
                            uint8_t *d_sigs, uint8_t *d_status, void *stream);
/* segmented aggregation: out[i] = sum of `counts[i]` consecutive points starting at points[first[i]]
 * (unit-scalar "MSM" of config 3: aggregate public key / aggregate signature) */
int bn254_batch_g1_sum(bn254_ctx *ctx, const uint8_t *points, const uint64_t *seg_off /* n+1 */, size_t n, uint8_t *out /* n*64 */, uint8_t *status);
int bn254_batch_g2_sum(bn254_ctx *ctx, const uint8_t *points, const uint64_t *seg_off /* n+1 */, size_t n, uint8_t *out /* n*128 */, uint8_t *status);

/* aggregate verify over shared pools (BASELINE config 3): tuple i names a message tuple_msg[i] and a
 * list of signers signer_idx[tuple_off[i] .. tuple_off[i+1]); status[i] = ECDSA::verify(msg, sum of the
 * listed signers' signatures on that message, sum of their public keys) — aggregation is plain point
 * addition (src/types.rs:126-132, :264-270) and only meaningful for one common message (src/lib.rs:34-38).
 * sig_pool[(m * n_signers + s) * 64]: signature of signer s on message m; pk_pool[s * 128].
 * An out-of-range signer index or message index (tuple_msg[i] >= n_msgs) gives status 2 (IndexOutOfBounds), as does
 * a decreasing tuple_off pair in the _device variant; an undecodable pool entry gives its decode status to every
 * tuple that uses it.  d_signer_idx must hold at least d_tuple_off[n] entries. */
int bn254_batch_aggregate_verify(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off /* n_msgs+1 */, size_t n_msgs,
                                 const uint8_t *pk_pool /* n_signers*128 */, size_t n_signers, const uint8_t *sig_pool /* n_msgs*n_signers*64 */,
                                 const uint32_t *tuple_msg /* n */, const uint64_t *tuple_off /* n+1 */, const uint32_t *signer_idx, size_t n,
                                 uint32_t flags, uint8_t *status /* n */);
int bn254_batch_aggregate_verify_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t n_msgs, const uint8_t *d_pk_pool,
                                        size_t n_signers, const uint8_t *d_sig_pool, const uint32_t *d_tuple_msg, const uint64_t *d_tuple_off,
                                        const uint32_t *d_signer_idx, size_t n, uint32_t flags, uint8_t *d_status, void *stream);
/* REGISTERED POOLS — the aggregate counterpart of bn254_ctx_register_keys, for a caller whose pools are fixed (a validator set and the
 * messages it has signed) while tuples keep arriving: everything that depends on the pools alone — decoding, H(m) of every message, the
 * subset-sum tables of both pools (8 / 16 keys and 4 / 8 signatures per entry; ~8 ms per call for 1 024 x 1 024 pools) — is done ONCE.
 * expect_tuples = the batch size the tables are chosen for (the thresholds BN254_OPT_AGG_SUBSET_MIN_TUPLES / _AGG_WIDE_MIN_TUPLES are applied
 * to it; 0 = no tables).  bn254_batch_aggregate_verify_registered[_device] then takes only the tuples: same status bytes as
 * bn254_batch_aggregate_verify on the same pools (pool-entry decode statuses included; flags as given at registration).  The tables live in
 * the context until the next registration OR the next bn254_batch_aggregate_verify* call with raw pools (which reuses the same buffers):
 * after either, the registered call returns BN254_E_BAD_ARGUMENT until pools are registered again.  Reference: the sums are
 * `Add for PublicKey / Signature` (src/types.rs:126-132, :264-270), the check ECDSA::verify (src/ecdsa.rs:49-64). */
int bn254_ctx_register_pools(bn254_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off /* n_msgs+1 */, size_t n_msgs, const uint8_t *pk_pool /* n_signers*128 */,
                             size_t n_signers, const uint8_t *sig_pool /* n_msgs*n_signers*64 */, uint32_t flags, size_t expect_tuples);
int bn254_ctx_register_pools_device(bn254_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, size_t n_msgs, const uint8_t *d_pk_pool, size_t n_signers,
                                    const uint8_t *d_sig_pool, uint32_t flags, size_t expect_tuples, void *stream);
int bn254_batch_aggregate_verify_registered(bn254_ctx *ctx, const uint32_t *tuple_msg /* n */, const uint64_t *tuple_off /* n+1 */, const uint32_t *signer_idx,
                                            size_t n, uint8_t *status /* n */);
int bn254_batch_aggregate_verify_registered_device(bn254_ctx *ctx, const uint32_t *d_tuple_msg, const uint64_t *d_tuple_off, const uint32_t *d_signer_idx,
                                                   size_t n, uint8_t *d_status, void *stream);

/* compressed wire formats (src/utils.rs:84-104, :130-158): out = uncompressed point, status as
 * bn::G1::from_compressed / bn::G2::from_compressed report through src/types.rs:91-93, :233-237, checked in the order
 * those decoders work (an input with several faults reports the first):
 *   G1: x >= q -> 6 NotMemberError; no square root -> 6; prefix byte not 0x02 / 0x03 -> 3 InvalidEncoding.
 *   G2: x.im >= q (the U512 does not split into two field elements) -> 6 NotMemberError; no square root -> 6; sign
 *       byte not 0x0a / 0x0b -> 3; not in the order-r subgroup -> 6.
 * (The x.im >= q code is UNPINNED: no reference vector exists and zeropool-bn is not vendored.  Upstream's
 * Fq2::from_slice is recalled as mapping a missing `divrem` quotient through `ok_or(FieldError::NotMember)`, which
 * src/error.rs:44-51 turns into NotMemberError; library versions before 0.6 returned 3 here.) */
int bn254_batch_g1_decompress(bn254_ctx *ctx, const uint8_t *in /* n*33 */, size_t n, uint8_t *out /* n*64 */, uint8_t *status);
int bn254_batch_g2_decompress(bn254_ctx *ctx, const uint8_t *in /* n*65 */, size_t n, uint8_t *out /* n*128 */, uint8_t *status);

/* timing of the most recent batch_verify*(…) on this context, from HIP events recorded on the
 * launch stream around each kernel: ms[0] decode, ms[1] hash-to-G1, ms[2] Miller loop,
 * ms[3] final exponentiation.  Synchronises the stream.  Requires bn254_ctx_set_profiling(ctx, 1). */
int bn254_ctx_set_profiling(bn254_ctx *ctx, int enabled);
/* Options a CALLER may want to touch: the batch-size thresholds of the routing table (which kernel layout serves which batch size — ONE
 * table, bn254_amd/csrc/bn254_ws.h: bn_route; defaults measured on an MI355X), the thresholds of the aggregate / randomised paths, and the
 * staging mode of the host-pointer verify.  Every setting returns the same status bytes; they trade latency against throughput.
 * (The A/B layouts of earlier rounds — one lane per verify, one pairing per lane, four wave roles, lane groups — and the test / measurement
 * knobs are developer options: section BN254_DEV_HOOKS at the end of this header.) */
#define BN254_OPT_TRIO_MAX_BATCH 6 /* verify / check_public_keys batches of up to this many items run in the OCTET layout (eight lanes per item: the
                                     three Fq6 products of every Fq12 operation in three lane pairs) — fewer instructions per lane, i.e. lower
                                     latency when the batch cannot fill the chip anyway; same status bytes.  Default 16384 (two passes of one wave
                                     on each of the 1024 SIMDs: 2.3 ms for 1 verify, 3.3 ms for 8192, 6.3 ms for 16384, against 6.4 / 7.1 /
                                     7.9 ms on lane pairs); 0 = never */
#define BN254_OPT_NONET_MAX_BATCH 13 /* small batches: up to this many items the final exponentiation runs on NINE lane pairs per item (18 lanes,
                                      three items per wave): the nine squarings of a cyclotomic squaring at once, the 18 products of an Fq12
                                      multiplication in two rounds; same status bytes.  0 = never (octet layout) */
#define BN254_OPT_LM_MAX_BATCH 15 /* small batches: up to this many items the Miller loop runs as the LANE MACHINE (nine lane pairs in each of four
                                   waves per item: every product of a dependency level in its own lane pair; twist-point formulas rearranged
                                   for depth; bn254_batch_verify_keyed: its keyed form on the registered keys' line tables); same status
                                   bytes.  0 = never (wave roles / octet layout) */
#define BN254_OPT_RAND_MIN_BATCH 5 /* randomised verify: batches with fewer items run the exact kernels instead (same statuses; group_ok = no item of the
                                     group failed the pairing check).  Default 131072, the measured break-even on an MI355X; 0 = always randomised */
#define BN254_OPT_AGG_SUBSET_MIN_TUPLES 9 /* aggregate verify: from this many tuples on (default 4096) the sums of all subsets of every 8 consecutive
                                            keys of the pool are tabulated once per call and a tuple adds one table entry per group instead of one
                                            key per signer (pools of up to 2048 signers, lists longer than n_signers / 8); 0 = never.  Same statuses. */
#define BN254_OPT_AGG_WIDE_MIN_TUPLES 14 /* aggregate verify: from this many tuples on (default 262144) the subset-sum tables are WIDENED once more —
                                          keys: the sums of all subsets of every 16 consecutive signers (n_signers / 16 x 65536 entries, 671 MB
                                          for 1024 signers), signatures per message: of every 8 — by one batched affine addition per entry, so
                                          that a tuple adds half as many entries; 0 = never.  Same statuses. */
#define BN254_OPT_PINNED_STAGING 12 /* bn254_batch_verify (host pointers), batches of >= 8192: T = 1..16 threads copy the caller's (pageable)
                                     buffers through a pinned staging buffer of the context in 1 MB pieces, each piece's DMA enqueued as soon
                                     as it is in place; 0 = hipMemcpyAsync straight from the caller's buffers (the runtime stages them) */
#define BN254_OPT_MAX_CHUNK 17 /* *_device and host entry points of verify / verify_compressed / verify_keyed: a batch of more than this many items
                                is processed in slices of this size (the 792 B/item workspace of a slice is reused; statuses land at the items'
                                own positions, so the result is that of one call).  0 (default) = automatic: slice only when the workspace of
                                the whole batch does not fit the device's free memory */
int bn254_ctx_set_option(bn254_ctx *ctx, int option, int value);
/* per-kernel times of the last verify-shaped call with profiling on (HIP events on the call's stream):
 * ms[0] decode, ms[1] hash-to-G1, ms[2] Miller loop, ms[3] final exponentiation.  The host-pointer bn254_batch_verify runs
 * the hash first and its ms[1] includes the transfer of the messages.  Other *_device calls reuse the slots: pairing ms[1] = 0;
 * hash_to_g1 ms[0] = the filter rounds (SHA-256 + Jacobi symbol per tested counter), ms[1] = the square roots (one per message), ms[2] = encoding the
 * points, ms[3] = 0; aggregate_verify ms[0] = the pools
 * (decoding, hashing the messages, the subset-sum table), ms[1] = the aggregation kernel. */
int bn254_ctx_last_kernel_ms(bn254_ctx *ctx, float ms[4]);
/* with BN254_OPT_CLOCK_PROBE on: achieved shader clock in MHz of the lane-pair Miller kernels [0], final exponentiations [1] and probe
 * kernels (bn254_probe_issue_rate, bn254_probe_leaf_floor) [2] launched on this context SINCE THE PREVIOUS CALL of this function (or
 * since the option was set) — the counters accumulate over launches and every call reads and clears them, so a caller brackets exactly
 * the launches it wants the clock of (bench.py: the timed steps).  0 = no such kernel ran in between.  Synchronises the device. */
int bn254_ctx_last_clocks(bn254_ctx *ctx, double sclk_mhz[3]);

/* =====================================================================================================================
 * Multi-GPU: the batch sharded over the GPUs of one node, ONE process (bn254_mgpu.hip).
 *
 * Every tuple of a batch is independent (/root/reference/src/ecdsa.rs:49-64 shares no state between calls), so a batch of n
 * items splits into G contiguous shards of S = ceil(n / G) items (the last ones shorter or empty): shard g = items
 * [g*S, min((g+1)*S, n)).  A bn254_mgpu owns, per entry of `devices`, one bn254_ctx, one stream and one host worker thread
 * (started once at creation; no thread is created per call, nothing is ever exec'ed).  The ONLY exchange between devices is
 * the gather of the per-item status bytes (and, for the pairing entry point, an 8-byte sum): RCCL's C API over xGMI
 * (ncclAllGather / ncclAllReduce on ncclCommInitAll communicators; librccl.so.1 is loaded with dlopen at the first call that
 * needs it, so single-GPU users of the library never load it).  A device may be listed more than once — several contexts on
 * one GPU, which is how the single-GPU test rigs run this code; RCCL refuses two ranks on one device, so such a handle gathers
 * with peer copies (hipMemcpyPeerAsync, pulled by every destination on its own stream) instead; BN254_MGPU_OPT_GATHER selects
 * either explicitly.  This is the layer /root/reference/src/lib.rs:60-63 would grow an `ECDSA::batch_verify(&Gpus, ...)` on
 * (INTEGRATION.md).
 *
 * Host-pointer entry points (bn254_mgpu_batch_verify, _batch_pairing, _batch_hash_to_g1): the caller passes the WHOLE batch;
 * every worker stages its shard to its device, runs the single-GPU entry point of the same name and copies the results straight
 * into the caller's slices — no collective.  They return when all shards are done; the return value is the first non-zero
 * return code of any shard (all shards are always run to the end).
 *
 * *_device entry points: the caller passes, per device g, DEVICE pointers to shard g's inputs (resident on devices[g], offsets
 * relative to that shard's own message buffer) and a status buffer d_status_all[g] of bn254_mgpu_gathered_len(mg, n) = G*S
 * bytes; device g writes its shard's statuses at offset g*S of ITS buffer and the gather (in place) leaves every device with
 * all n status bytes at d_status_all[g][0 .. n).  streams[g] (hipStream_t as void*; the array or an entry may be NULL = the
 * handle's own stream of that device) carries the kernels and the collective of device g; the calls only enqueue.
 * A handle carries one call in flight, like a context. */
typedef struct bn254_mgpu bn254_mgpu;

int bn254_mgpu_create(const int *devices, int n_dev /* 1..64 */, bn254_mgpu **out);
void bn254_mgpu_destroy(bn254_mgpu *mg);
int bn254_mgpu_device_count(const bn254_mgpu *mg);
/* the context of entry g (options, bn254_ctx_reserve, bn254_ctx_register_keys ... per device); owned by the handle */
bn254_ctx *bn254_mgpu_ctx(bn254_mgpu *mg, int g);
/* S = ceil(n / G); [lo, hi) of shard g; G * S */
size_t bn254_mgpu_shard_len(const bn254_mgpu *mg, size_t n);
int bn254_mgpu_shard_range(const bn254_mgpu *mg, size_t n, int g, size_t *lo, size_t *hi);
size_t bn254_mgpu_gathered_len(const bn254_mgpu *mg, size_t n);
/* presize every context for batches of n_total items over all devices (bn254_ctx_reserve(ceil(n_total / G)) each) and, when
 * init_collectives != 0, create the RCCL communicators now instead of inside the first *_device call */
int bn254_mgpu_reserve(bn254_mgpu *mg, size_t n_total, int init_collectives);
int bn254_mgpu_synchronize(bn254_mgpu *mg); /* waits for the handle's own streams */
#define BN254_MGPU_OPT_GATHER 1 /* 0 (default) = RCCL when the handle has two or more DISTINCT devices, else peer copies (a one-device handle has
                                   nothing to gather and never loads librccl); 1 = RCCL (an error for duplicate devices — RCCL refuses two
                                   ranks on one device; with one device: the one-rank rehearsal of the RCCL calls); 2 = peer copies */
#define BN254_MGPU_OPT_TIMING 2 /* 1 = record per device the time of its shard's compute and of the collective (bn254_mgpu_last_timing) */
int bn254_mgpu_set_option(bn254_mgpu *mg, int option, int value);
/* per device g of the last call, in ms: compute_ms[g] = its shard's kernels (device entry points: HIP events on its stream; host
 * entry points: the worker's wall clock around staging + kernels + copy-back), collective_ms[g] = the gather / all-reduce as
 * seen on its stream (0 for host entry points).  Synchronises the streams.  Needs BN254_MGPU_OPT_TIMING. */
int bn254_mgpu_last_timing(bn254_mgpu *mg, float *compute_ms /* G */, float *collective_ms /* G */);
/* text of the last RCCL / loader failure on this handle ("" if none); valid until the next call on the handle */
const char *bn254_mgpu_last_error(const bn254_mgpu *mg);

/* status[i] = ECDSA::verify(msg_i, sig_i, pk_i) (src/ecdsa.rs:49-64) for the whole batch, sharded over the devices */
int bn254_mgpu_batch_verify(bn254_mgpu *mg, const uint8_t *msgs, const uint64_t *msg_off /* n+1 */, const uint8_t *sigs /* n*64 */,
                            const uint8_t *pks /* n*128 */, size_t n, uint32_t flags, uint8_t *status /* n */);
int bn254_mgpu_batch_verify_device(bn254_mgpu *mg, const uint8_t *const *d_msgs, const uint64_t *const *d_msg_off,
                                   const uint8_t *const *d_sigs, const uint8_t *const *d_pks, size_t n /* whole batch */, uint32_t flags,
                                   uint8_t *const *d_status_all /* G x gathered_len */, void *const *streams /* G or NULL */);
/* the other verify-shaped host entry points, sharded the same way (same arguments and status bytes as their single-GPU namesakes):
 * from the compressed encodings; with REGISTERED keys (bn254_mgpu_register_keys puts the whole key set on every device; key_status as
 * bn254_ctx_register_keys); the aggregate verify of BASELINE configs[2] — the TUPLES are sharded, messages and pools go to every device,
 * which builds its own subset tables. */
int bn254_mgpu_batch_verify_compressed(bn254_mgpu *mg, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs33, const uint8_t *pks65,
                                       size_t n, uint8_t *status);
int bn254_mgpu_register_keys(bn254_mgpu *mg, const uint8_t *pks /* n_keys*128 */, size_t n_keys, uint32_t flags, uint8_t *key_status /* or NULL */);
int bn254_mgpu_batch_verify_keyed(bn254_mgpu *mg, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint32_t *key_idx, size_t n,
                                  uint32_t flags, uint8_t *status);
int bn254_mgpu_batch_aggregate_verify(bn254_mgpu *mg, const uint8_t *msgs, const uint64_t *msg_off, size_t n_msgs, const uint8_t *pk_pool,
                                      size_t n_signers, const uint8_t *sig_pool, const uint32_t *tuple_msg, const uint64_t *tuple_off,
                                      const uint32_t *signer_idx, size_t n, uint32_t flags, uint8_t *status);
/* gt[i], status[i] as bn254_batch_pairing (bn::pairing_batch, src/ecdsa.rs:57); *checksum (optional) = the sum mod 2^64 of all
 * little-endian 64-bit words of the n*384 Gt bytes — BASELINE configs[3]'s cross-shard check.  Device form: d_gt[g] = shard g's
 * n_g*384 bytes; d_checksum[g] (the array or NULL) = 8 bytes on every device receiving the all-reduced (ncclAllReduce, sum,
 * uint64) checksum. */
int bn254_mgpu_batch_pairing(bn254_mgpu *mg, const uint8_t *g1 /* n*k*64 */, const uint8_t *g2 /* n*k*128 */, size_t n, size_t k,
                             uint32_t flags, uint8_t *gt /* n*384 */, uint8_t *status /* n or NULL */, uint64_t *checksum /* or NULL */);
int bn254_mgpu_batch_pairing_device(bn254_mgpu *mg, const uint8_t *const *d_g1, const uint8_t *const *d_g2, size_t n, size_t k,
                                    uint32_t flags, uint8_t *const *d_gt, uint8_t *const *d_status_all, uint64_t *const *d_checksum,
                                    void *const *streams);
/* points[i] = hash_to_try_and_increment(msg_i) (src/hash.rs:29-63) for the whole batch, sharded over the devices */
int bn254_mgpu_batch_hash_to_g1(bn254_mgpu *mg, const uint8_t *msgs, const uint64_t *msg_off, size_t n, uint8_t *points /* n*64 */,
                                uint8_t *status /* n */, uint8_t *tries /* n or NULL */);

/* =====================================================================================================================
 * BN254_DEV_HOOKS — developer hooks, NOT part of the drop-in ABI (bn254_devhooks.hip).  No reference function corresponds to any of
 * them; a binding of the reference's API never needs them.  bn254_debug_* let the parity tests compare every layer of
 * the HIP arithmetic with the oracle; bn254_probe_* are the measurement kernels behind bench.py's roofline figures.  Define
 * BN254_NO_DEV_HOOKS before including this header to hide the declarations.
 * ===================================================================================================================== */
#ifndef BN254_NO_DEV_HOOKS
/* developer options of bn254_ctx_set_option: A/B layouts kept for measurements and for the randomised path's re-check queue, test and
 * measurement knobs.  Same status bytes with every setting; none of them is something a caller of the drop-in ABI should set. */
#define BN254_OPT_SPLIT_MILLER 1 /* 1 = the two Miller loops of a verify in two lanes of different waves (one pairing per lane, one-lane layout only;
                                 default 0) */
#define BN254_OPT_PAIR_LANES 4 /* verify: Miller loop + final exponentiation on lane pairs, two waves per SIMD (default 1); 0 = one lane per verify */
#define BN254_OPT_RAND_ITEMS_PER_LANE 3 /* randomised verify: items per lane in the Miller kernel; 0 = by batch size (default), 1, 2 */
#define BN254_OPT_TRIO_WAVE_ROLES 8 /* octet path, Miller loop: the lane pairs of a verify as WAVES of a workgroup, each with its own instruction
                                     stream (twist point / line product / the halves of f), values exchanged through LDS between barriers:
                                     2 (default) = eight waves per 32 verifies (every Fq6 product split over two waves), 1 = four waves,
                                     0 = four lane pairs of one wave (every pair runs all the linear work).  Same status bytes. */
#define BN254_OPT_HASH_DIRECT_WIDTH 7 /* hash-to-G1 of batches of up to 4096 messages: this many counters of every message are tried at once, in
                                       lanes of one wave, with the square root itself (latency 0.17 ms instead of 0.25); a power of two <= 32,
                                       default 32; 0 = always the filter rounds.  Same points and try counts either way. */
#define BN254_OPT_NONET_WIDE 16 /* ... and, while the batch is at most one item per SIMD (1 024), on EIGHTEEN lane pairs, one item per wave: the 18
                                 products of a multiplication in one round (default 1; 0 = nine lane pairs at every size) */
#define BN254_OPT_AGG_SORT_BY_MSG 11 /* aggregate verify, batches that use the per-message signature tables: bucket the tuples by message on the
                                      device (counting sort into an index map) so that a workgroup of the aggregation kernel gathers from ONE
                                      message's table; statuses land at the tuples' own indices either way.  Default 1; 0 = the caller's order */
#define BN254_OPT_CLOCK_PROBE 10 /* measurement: 1 = the lane-pair Miller / final-exponentiation kernels and the issue probe record, per workgroup,
                                  shader-clock cycles (s_memtime) and constant-rate ticks (s_memrealtime) between entry and exit, read back by
                                  bn254_ctx_last_clocks: the clock the chip actually sustains under this load (power-limited parts run below
                                  their nominal 2.4 GHz).  Costs two scalar clock reads per workgroup; default 0 */
#define BN254_OPT_HASH_MAX_TRIES 2 /* test knob: counters tried before HashToPointError; 0 = 255 as in src/hash.rs:40 */
#define BN254_OPT_G2_FIXED_BASE 19 /* key derivation (bn254_batch_g2_mul with points = NULL: sk * G2::one()): 1 (default) = 65 additions from a table of the
                                   generator's multiples, built once per context, on a lane pair; 0 = the general 256-step ladder on one lane.  Same bytes. */
#define BN254_OPT_ASSUME_FREE_MB 18 /* test knob for the automatic slicing rule (BN254_OPT_MAX_CHUNK = 0): price the workspace of a batch against this many MB
                                    of free device memory instead of what hipMemGetInfo reports; 0 = ask the runtime */
/* the routing table of this context as it stands (defaults + options): rows (max_n[i], miller[i], fe[i]) in ascending order of max_n, the last
 * row max_n = UINT64_MAX; miller: 0 lane machine, 1 wave roles, 2 lane pairs; fe: 0 eighteen lane pairs, 1 nine lane pairs, 2 octets, 3 lane
 * pairs.  Returns the number of rows (<= cap) or a negative error.  The parity tests generate every boundary +-1 from it. */
int bn254_debug_route_table(bn254_ctx *ctx, uint64_t *max_n, int *miller, int *fe, int cap);
/* test hooks: element-wise field/tower operations on byte-encoded operands, used by the parity
 * tests to compare each layer of the HIP arithmetic with the oracle.
 *   op: 0 mul, 1 add, 2 sub, 3 inverse(a), 4 square(a), 5 sqrt(a) (status 6 if none)   [Fq, 32 B]
 *   fp12 op: 0 mul, 1 square, 2 inverse, 3 conj, 4 frobenius^1, 5 ^2, 6 ^3, 7 cyclotomic square,
 *            8 final exponentiation                                                    [384 B] */
int bn254_debug_fp_op(bn254_ctx *ctx, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, uint8_t *status);
int bn254_debug_fp12_op(bn254_ctx *ctx, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
/* the final exponentiation of ECDSA::verify / bn::pairing_batch (src/ecdsa.rs:57-59) on caller-supplied LIMB vectors — n x 12 coefficients
 * (Gt order) x 9 int32 limbs, value = sum limb_k 2^(29 k) in Montgomery form (R = 2^261) — in the layout named: 0 one lane per item, exact
 * exponent, gt = canonical Gt bytes | 1 lane pairs, exact, gt | 2 lane pairs, the == one chain | 3 lane octets (straight-line chains below
 * 128 items, accumulator machine from 128 on) | 4 nine lane pairs per item | 5 one lane, the == one chain | 6 eighteen lane pairs per item (one per wave).  status[i] = 0 (the value is one) or 9.
 * Exists so that the parity tests can hand every layout NON-CANONICAL representatives with extreme balanced digits — what the interval
 * tracker's contract for a Miller value allows (limbs 0..7 in [-2^28, 2^28], |value| <= 0.5215 q) but no byte decoder produces. */
int bn254_debug_final_exp_limbs(bn254_ctx *ctx, int layout, const int32_t *limbs /* n*108 */, size_t n, uint8_t *gt /* n*384, layouts 0 / 1, or NULL */,
                                uint8_t *status /* n */);
/* what ONE pass of the try loop of hash_to_try_and_increment does with a chosen 256-bit digest value h (32 B
 * big-endian each) instead of SHA-256(msg || ctr): the h >= 5q rule (src/hash.rs:49-51), mod_u256's strict '>'
 * (src/utils.rs:27-37) and G1::from_compressed(0x02 || x) (src/utils.rs:56-63).  status 0: out = the point; 1: the
 * loop would move to the next counter (out = zeros).  Exists because h = k*q has no known preimage. */
int bn254_debug_hash_candidate(bn254_ctx *ctx, const uint8_t *h /* n*32 */, size_t n, uint8_t *out /* n*64 */, uint8_t *status);
/* un-exponentiated Miller-loop value of each single pair (debugging aid) */
int bn254_debug_miller_loop(bn254_ctx *ctx, const uint8_t *g1, const uint8_t *g2, size_t n, uint8_t *f /* n*384 */);

/* calibration probe for the roofline figures of bench.py: sustained wave-instructions per second of one VALU
 * instruction (op 0 v_mad_u64_u32, 1 v_add_u32, 2 v_mul_lo_u32; 16 independent chains) with waves_per_simd (1..8)
 * waves on every SIMD of the device; n_simd (optional) = SIMD count.  Synchronises the context's stream. */
int bn254_probe_issue_rate(bn254_ctx *ctx, int op, int waves_per_simd, double *wave_inst_per_s, int *n_simd);
/* measurement: duration in ms of a kernel that runs ONLY the field-product calls of one verify's Miller loop (mode 0: 3 219 dual
 * products, 435 squarings, 348 scalings per lane) or final exponentiation (mode 1: 945 dual products, 1 701 squarings) for n lane pairs —
 * no tower additions, carries, twist point or LDS traffic — on the launch shape of those kernels: a floor for any arrangement of the
 * code around the product leaves.  n <= the size of the workspace.  With BN254_OPT_CLOCK_PROBE its clock lands in slot [2].
 * Modes 2 / 3: 3 219 dual products with the leaf inlined into the loop / called — what the calling convention costs per product.
 * Modes 4 / 5: the product counts of modes 0 / 1 as FOUR independent chains per lane (in modes 0 / 1 every product waits for its
 * predecessor, which a lone wave per SIMD cannot hide): the floor to quote is the smaller of the two.  Modes 6 / 7: controls for mode 4 —
 * the same loop with one chain / two chains. */
int bn254_probe_leaf_floor(bn254_ctx *ctx, size_t n, int mode, float *ms);
/* measurement: duration in ms of the final exponentiation's accumulator machine (the interpreter of the lane-pair kernel) running a
 * caller-supplied program of n_steps (opcode, argument) byte pairs — 1 LOAD slot, 2 STORE slot, 3 CSQR, 4 MUL slot, 5 CONJ, 6 FROB 1..3,
 * 7 INV; slots 0..9 — for n lane pairs on the values the last verify left in the workspace.  Programs of one operation kind give the cost
 * of that operation in place (bench.py: roofline.final_exp_split); results are not meaningful values. */
int bn254_probe_fe_program(bn254_ctx *ctx, size_t n, const uint8_t *prog, size_t n_steps, float *ms);
#endif /* BN254_NO_DEV_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* BN254_HIP_H */
