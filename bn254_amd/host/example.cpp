// The reference's example scenario (/root/reference/examples/bn254.rs:3-34) through the C++ host API:
// two signers, same message, aggregate signature and public key, verify.  Needs an MI355X.
//   g++ -std=c++17 example.cpp -L.. -lbn254hip -Wl,-rpath,'$ORIGIN/..' -o example
#include <cstdio>
#include "bn254.hpp"

static std::array<uint8_t, 32> unhex(const char* s) {
  std::array<uint8_t, 32> o{};
  for (int i = 0; i < 32; ++i) { unsigned v; sscanf(s + 2 * i, "%2x", &v); o[i] = (uint8_t)v; }
  return o;
}
int main() {
  try {
    bn254::PrivateKey k1, k2;
    k1.bytes = unhex("c9afa9d845ba75166b5c215767b1d6934e50c3db36e89b127b8a622b120f6721");
    k2.bytes = unhex("a55e93edb1350916bf5beea1b13d8f198ef410033445bcb645b65be5432722f1");
    auto pk1 = bn254::PublicKey::from_private_key(k1), pk2 = bn254::PublicKey::from_private_key(k2);
    std::vector<uint8_t> msg = {'s', 'a', 'm', 'p', 'l', 'e'};
    auto s1 = bn254::ECDSA::sign(msg, k1), s2 = bn254::ECDSA::sign(msg, k2);
    bn254::ECDSA::verify(msg, s1 + s2, pk1 + pk2);
    printf("Successful aggregate signature verification\n");
    try { bn254::ECDSA::verify(msg, s1, pk2); printf("ERROR: wrong key accepted\n"); return 2; }
    catch (const bn254::Error& e) { if (e.kind != bn254::ErrorKind::VerificationFailed) return 3; }
    // the same through the keyed verify: the two keys and their aggregate registered once, tuples name them by index
    auto kst = bn254::ECDSA::register_keys({pk1, pk2, pk1 + pk2});
    auto st = bn254::ECDSA::batch_verify_keyed({msg, msg, msg, msg, msg}, {s1, s2, s1 + s2, s1, s2}, {0, 1, 2, 1, 7});
    if (kst != std::vector<uint8_t>{0, 0, 0} || st != std::vector<uint8_t>{0, 0, 0, 9, 2}) { printf("ERROR: keyed verify\n"); return 4; }
    // and over "all the GPUs of the node" — here two contexts on device 0, the batch cut in two shards (three tuples and two)
    bn254::Gpus gpus({0, 0});
    auto mst = bn254::ECDSA::batch_verify(gpus, {msg, msg, msg, msg, msg}, {s1, s2, s1 + s2, s1, s2}, {pk1, pk2, pk1 + pk2, pk2, pk1});
    if (gpus.count() != 2 || mst != std::vector<uint8_t>{0, 0, 0, 9, 9}) { printf("ERROR: multi-GPU verify\n"); return 5; }
    return 0;
  } catch (const std::exception& e) { printf("failed: %s\n", e.what()); return 1; }
}
