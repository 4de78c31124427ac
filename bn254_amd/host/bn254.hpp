// bn254.hpp — C++ host-side mirror of the reference's public API (/root/reference/src/lib.rs:60-63)
// over the C ABI of libbn254hip.so (include/bn254_hip.h).  Header-only; link with -lbn254hip.
//
//   bn254::ECDSA::sign / verify / batch_verify      /root/reference/src/ecdsa.rs:26-35, :49-64 (+ new batch entry)
//   bn254::check_public_keys                        /root/reference/src/ecdsa.rs:78-93
//   bn254::PrivateKey / PublicKey / PublicKeyG1 / Signature   /root/reference/src/types.rs:13,81,151,222
//   bn254::Error                                    /root/reference/src/error.rs:6-29
//
// Points are held as the reference's uncompressed encodings (identity = all-zero bytes); every
// group operation runs on the GPU.  A failed verification throws Error{VerificationFailed}, like
// the reference returns Err(Error::VerificationFailed) (not Ok(false)).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bn254_hip.h"

namespace bn254 {

enum class ErrorKind : uint8_t {   // the status codes of include/bn254_hip.h = 1 + the variant's index in /root/reference/src/error.rs:6-29
  HashToPointError = BN254_ERR_HASH_TO_POINT, IndexOutOfBounds = BN254_ERR_INDEX_OUT_OF_BOUNDS, InvalidEncoding = BN254_ERR_INVALID_ENCODING,
  InvalidGroupPoint = BN254_ERR_INVALID_GROUP_POINT, InvalidLength = BN254_ERR_INVALID_LENGTH, NotMemberError = BN254_ERR_NOT_MEMBER,
  ToAffineConversion = BN254_ERR_TO_AFFINE_CONVERSION, PointInJacobian = BN254_ERR_POINT_IN_JACOBIAN,
  VerificationFailed = BN254_ERR_VERIFICATION_FAILED, SerializationError = BN254_ERR_SERIALIZATION, HexDecodeFailed = BN254_ERR_HEX_DECODE_FAILED
};
struct Error : std::runtime_error {
  ErrorKind kind;
  explicit Error(ErrorKind k) : std::runtime_error("bn254 error " + std::to_string((int)k)), kind(k) {}
};
struct NativeError : std::runtime_error {
  int rc;
  NativeError(const char* fn, int r) : std::runtime_error(std::string(fn) + " failed, rc=" + std::to_string(r)), rc(r) {}
};
inline void check_rc(const char* fn, int rc) { if (rc != 0) throw NativeError(fn, rc); }
inline void check_status(uint8_t s) { if (s != 0) throw Error((ErrorKind)s); }

class Engine {   // one per GPU; not thread-safe (one thread at a time per context)
 public:
  explicit Engine(int device = 0) { check_rc("bn254_ctx_create", bn254_ctx_create(device, &ctx_)); }
  ~Engine() { bn254_ctx_destroy(ctx_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  bn254_ctx* raw() const { return ctx_; }
  static Engine& default_engine() { static Engine e(0); return e; }
 private:
  bn254_ctx* ctx_ = nullptr;
};

// All the GPUs of a node behind one handle (include/bn254_hip.h, section "Multi-GPU"): the batch is cut into contiguous shards, one per
// device entry, each verified by that device's own context from its own parked worker thread; ECDSA::batch_verify(gpus, ...) below.
class Gpus {
 public:
  explicit Gpus(const std::vector<int>& devices) { check_rc("bn254_mgpu_create", bn254_mgpu_create(devices.data(), (int)devices.size(), &mg_)); }
  ~Gpus() { bn254_mgpu_destroy(mg_); }
  Gpus(const Gpus&) = delete;
  Gpus& operator=(const Gpus&) = delete;
  bn254_mgpu* raw() const { return mg_; }
  int count() const { return bn254_mgpu_device_count(mg_); }
 private:
  bn254_mgpu* mg_ = nullptr;
};

struct PrivateKey {   // PrivateKey(Fr): 32-byte big-endian scalar, reduced mod r on use
  std::array<uint8_t, 32> bytes{};
  static PrivateKey try_from(const uint8_t* data, size_t len) {
    if (len != 32) throw Error(ErrorKind::InvalidLength);          // types_test.rs:29-46
    PrivateKey k; std::memcpy(k.bytes.data(), data, 32); return k;
  }
};

template <size_t N> struct PointBytes {
  std::array<uint8_t, N> raw{};
  bool is_identity() const { for (auto b : raw) if (b) return false; return true; }
  const std::array<uint8_t, N>& to_uncompressed() const {
    if (is_identity()) throw Error(ErrorKind::PointInJacobian);    // utils.rs:163,184
    return raw;
  }
};

struct Signature : PointBytes<64> {
  static Signature from_uncompressed(const uint8_t* data, size_t len, Engine& e = Engine::default_engine()) {
    if (len != 64) throw Error(ErrorKind::InvalidLength);          // utils.rs:120
    Signature s; std::memcpy(s.raw.data(), data, 64);
    if (s.is_identity()) throw Error(ErrorKind::InvalidGroupPoint);
    uint8_t zero[64] = {0}, out[64], st = 0;
    check_rc("bn254_batch_g1_add", bn254_batch_g1_add(e.raw(), s.raw.data(), zero, 1, out, &st));
    check_status(st);
    return s;
  }
  static Signature from_compressed(const uint8_t* data, size_t len, Engine& e = Engine::default_engine()) {   // types.rs:233-237
    if (len != 33) throw Error(ErrorKind::InvalidEncoding);
    Signature s; uint8_t st = 0;
    check_rc("bn254_batch_g1_decompress", bn254_batch_g1_decompress(e.raw(), data, 1, s.raw.data(), &st));
    check_status(st);
    return s;
  }
  std::array<uint8_t, 33> to_compressed() const {                  // utils.rs:84-104
    if (is_identity()) throw Error(ErrorKind::PointInJacobian);
    std::array<uint8_t, 33> o{};
    o[0] = (raw[63] & 1) ? 3 : 2;
    std::memcpy(o.data() + 1, raw.data(), 32);
    return o;
  }
  Signature operator+(const Signature& o) const {                  // types.rs:264-270
    Signature r; uint8_t st = 0;
    check_rc("bn254_batch_g1_add", bn254_batch_g1_add(Engine::default_engine().raw(), raw.data(), o.raw.data(), 1, r.raw.data(), &st));
    check_status(st);
    return r;
  }
};
struct PublicKeyG1 : PointBytes<64> {
  static PublicKeyG1 from_private_key(const PrivateKey& k, Engine& e = Engine::default_engine()) {   // types.rs:155-157
    PublicKeyG1 r; uint8_t st = 0;       // points = nullptr: G1::one(), the fixed-base table of the generator
    check_rc("bn254_batch_g1_mul", bn254_batch_g1_mul(e.raw(), nullptr, k.bytes.data(), 1, 1, r.raw.data(), &st));
    check_status(st);
    return r;
  }
};
struct PublicKey : PointBytes<128> {
  static PublicKey from_private_key(const PrivateKey& k, Engine& e = Engine::default_engine()) {     // types.rs:85-87
    PublicKey r; uint8_t st = 0;
    check_rc("bn254_batch_g2_mul", bn254_batch_g2_mul(e.raw(), nullptr, k.bytes.data(), 1, 1, r.raw.data(), &st));
    check_status(st);
    return r;
  }
  static PublicKey from_compressed(const uint8_t* data, size_t len, Engine& e = Engine::default_engine()) {   // types.rs:91-93
    if (len != 65) throw Error(ErrorKind::InvalidEncoding);
    PublicKey r; uint8_t st = 0;
    check_rc("bn254_batch_g2_decompress", bn254_batch_g2_decompress(e.raw(), data, 1, r.raw.data(), &st));
    check_status(st);
    return r;
  }
  PublicKey operator+(const PublicKey& o) const {                  // types.rs:126-132
    PublicKey r; uint8_t st = 0;
    check_rc("bn254_batch_g2_add", bn254_batch_g2_add(Engine::default_engine().raw(), raw.data(), o.raw.data(), 1, r.raw.data(), &st));
    check_status(st);
    return r;
  }
  bool operator==(const PublicKey& o) const { return raw == o.raw; }
};

struct ECDSA {
  static Signature sign(const std::vector<uint8_t>& message, const PrivateKey& k, Engine& e = Engine::default_engine()) {
    uint64_t off[2] = {0, message.size()};
    Signature s; uint8_t st = 0;
    check_rc("bn254_batch_sign", bn254_batch_sign(e.raw(), message.data(), off, k.bytes.data(), 1, s.raw.data(), &st));
    check_status(st);
    return s;
  }
  static void verify(const std::vector<uint8_t>& message, const Signature& s, const PublicKey& pk, Engine& e = Engine::default_engine()) {
    uint64_t off[2] = {0, message.size()};
    uint8_t st = 0;
    check_rc("bn254_batch_verify", bn254_batch_verify(e.raw(), message.data(), off, s.raw.data(), pk.raw.data(), 1, 0, &st));
    check_status(st);
  }
  // result[i] == 0 iff verify(messages[i], signatures[i], public_keys[i]) succeeds, else the ErrorKind it would throw
  static std::vector<uint8_t> batch_verify(const std::vector<std::vector<uint8_t>>& messages, const std::vector<Signature>& signatures,
                                           const std::vector<PublicKey>& public_keys, Engine& e = Engine::default_engine()) {
    size_t n = messages.size();
    if (signatures.size() != n || public_keys.size() != n) throw Error(ErrorKind::InvalidLength);
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<uint8_t> msgs, sigs(n * 64), pks(n * 128), status(n, 0);
    for (size_t i = 0; i < n; ++i) {
      off[i] = msgs.size();
      msgs.insert(msgs.end(), messages[i].begin(), messages[i].end());
      std::memcpy(&sigs[64 * i], signatures[i].raw.data(), 64);
      std::memcpy(&pks[128 * i], public_keys[i].raw.data(), 128);
    }
    off[n] = msgs.size();
    check_rc("bn254_batch_verify", bn254_batch_verify(e.raw(), msgs.data(), off.data(), sigs.data(), pks.data(), n, 0, status.data()));
    return status;
  }
  // the same over all the GPUs of a node: shard g of the batch on device entry g, statuses straight into result's slices
  static std::vector<uint8_t> batch_verify(Gpus& gpus, const std::vector<std::vector<uint8_t>>& messages, const std::vector<Signature>& signatures,
                                           const std::vector<PublicKey>& public_keys) {
    size_t n = messages.size();
    if (signatures.size() != n || public_keys.size() != n) throw Error(ErrorKind::InvalidLength);
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<uint8_t> msgs, sigs(n * 64), pks(n * 128), status(n, 0);
    for (size_t i = 0; i < n; ++i) {
      off[i] = msgs.size();
      msgs.insert(msgs.end(), messages[i].begin(), messages[i].end());
      std::memcpy(&sigs[64 * i], signatures[i].raw.data(), 64);
      std::memcpy(&pks[128 * i], public_keys[i].raw.data(), 128);
    }
    off[n] = msgs.size();
    check_rc("bn254_mgpu_batch_verify", bn254_mgpu_batch_verify(gpus.raw(), msgs.data(), off.data(), sigs.data(), pks.data(), n, 0, status.data()));
    return status;
  }
  // Keyed verify: a validator set registered once (PublicKey::from_uncompressed per key, types.rs:96-99, plus the key's Miller-loop
  // lines tabulated in HBM), then tuples name their key by index.  register_keys: result[j] == 0 or the ErrorKind of key j.
  static std::vector<uint8_t> register_keys(const std::vector<PublicKey>& keys, Engine& e = Engine::default_engine()) {
    std::vector<uint8_t> pks(keys.size() * 128), status(keys.size(), 0);
    for (size_t j = 0; j < keys.size(); ++j) std::memcpy(&pks[128 * j], keys[j].raw.data(), 128);
    check_rc("bn254_ctx_register_keys", bn254_ctx_register_keys(e.raw(), pks.data(), keys.size(), 0, status.data()));
    return status;
  }
  // result[i] == 0 iff verify(messages[i], signatures[i], registered[key_indices[i]]) succeeds; 2 (IndexOutOfBounds) outside the set
  static std::vector<uint8_t> batch_verify_keyed(const std::vector<std::vector<uint8_t>>& messages, const std::vector<Signature>& signatures,
                                                 const std::vector<uint32_t>& key_indices, Engine& e = Engine::default_engine()) {
    size_t n = messages.size();
    if (signatures.size() != n || key_indices.size() != n) throw Error(ErrorKind::InvalidLength);
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<uint8_t> msgs, sigs(n * 64), status(n, 0);
    for (size_t i = 0; i < n; ++i) {
      off[i] = msgs.size();
      msgs.insert(msgs.end(), messages[i].begin(), messages[i].end());
      std::memcpy(&sigs[64 * i], signatures[i].raw.data(), 64);
    }
    off[n] = msgs.size();
    check_rc("bn254_batch_verify_keyed", bn254_batch_verify_keyed(e.raw(), msgs.data(), off.data(), sigs.data(), key_indices.data(), n, 0, status.data()));
    return status;
  }
  // opt-in randomised mode (include/bn254_hip.h: bn254_batch_verify_randomized): same result shape; non-zero entries
  // are exact, a zero is wrong with probability <= 2^-128 per group of 64 for a fresh secret 32-byte seed
  static std::vector<uint8_t> batch_verify_randomized(const std::vector<std::vector<uint8_t>>& messages, const std::vector<Signature>& signatures,
                                                      const std::vector<PublicKey>& public_keys, const std::array<uint8_t, 32>& seed,
                                                      Engine& e = Engine::default_engine()) {
    size_t n = messages.size();
    if (signatures.size() != n || public_keys.size() != n) throw Error(ErrorKind::InvalidLength);
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<uint8_t> msgs, sigs(n * 64), pks(n * 128), status(n, 0);
    for (size_t i = 0; i < n; ++i) {
      off[i] = msgs.size();
      msgs.insert(msgs.end(), messages[i].begin(), messages[i].end());
      std::memcpy(&sigs[64 * i], signatures[i].raw.data(), 64);
      std::memcpy(&pks[128 * i], public_keys[i].raw.data(), 128);
    }
    off[n] = msgs.size();
    check_rc("bn254_batch_verify_randomized",
             bn254_batch_verify_randomized(e.raw(), msgs.data(), off.data(), sigs.data(), pks.data(), n, 0, seed.data(), status.data(), nullptr));
    return status;
  }
};

inline void check_public_keys(const PublicKey& pk_g2, const PublicKeyG1& pk_g1, Engine& e = Engine::default_engine()) {
  uint8_t st = 0;
  check_rc("bn254_batch_check_public_keys", bn254_batch_check_public_keys(e.raw(), pk_g2.raw.data(), pk_g1.raw.data(), 1, 0, &st));
  check_status(st);
}

}  // namespace bn254
