"""bn254_amd — MI355X-native batch BN254 aggregate-signature verifier.

The hot path of sedaprotocol/bn254 (`ECDSA::verify` = try-and-increment hash-to-G1 + a
2-pair optimal-ate pairing check) as hand-written HIP kernels for gfx950 behind a C ABI
(include/bn254_hip.h), with a host-side mirror of the reference's API (bn254_amd.api).
"""
from .api import (ECDSA, Error, ErrorKind, PrivateKey, PublicKey, PublicKeyG1, Signature, check_public_keys,  # noqa: F401
                  format_pairing_check_uncompressed_values, format_pairing_check_values)
from .engine import Engine, MultiEngine, NativeError, default_engine  # noqa: F401
