"""Host-side mirror of the reference's public API (/root/reference/src/lib.rs:60-63) on top of the
C ABI: same names, argument meaning and error behaviour, plus the batch entry point of the
north star.  Rust is not available in this image, so this mirror is Python (tests/bench) and
bn254_amd/host/bn254.hpp (C++); INTEGRATION.md holds the Rust shim a maintainer would add.

    ECDSA.sign(message, private_key) -> Signature               src/ecdsa.rs:26-35
    ECDSA.verify(message, signature, public_key) -> None/raise   src/ecdsa.rs:49-64
    ECDSA.batch_verify(messages, signatures, public_keys) -> [None | Error, ...]      (new)
    ECDSA.batch_verify_randomized(messages, signatures, public_keys, seed) -> same    (new, opt-in, probabilistic)
    check_public_keys(public_key_g2, public_key_g1)              src/ecdsa.rs:78-93
    PrivateKey / PublicKey / PublicKeyG1 / Signature             src/types.rs:13,81,151,222

All group arithmetic runs on the GPU through libbn254hip.so (no CPU fallback).  Points are
held as the reference's uncompressed byte encodings (identity = all-zero bytes).
"""
import enum

from . import engine as _engine

_Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


class ErrorKind(enum.IntEnum):
    """1 + index of the variant in /root/reference/src/error.rs:6-29 (0 = Ok)."""
    HashToPointError = 1
    IndexOutOfBounds = 2
    InvalidEncoding = 3
    InvalidGroupPoint = 4
    InvalidLength = 5
    NotMemberError = 6
    ToAffineConversion = 7
    PointInJacobian = 8
    VerificationFailed = 9
    SerializationError = 10
    HexDecodeFailed = 11


class Error(Exception):
    def __init__(self, kind):
        self.kind = ErrorKind(kind)
        super().__init__(self.kind.name)

    def __eq__(self, other):
        return isinstance(other, Error) and other.kind == self.kind

    def __hash__(self):
        return hash(self.kind)


def _raise(status):
    if status:
        raise Error(status)


def _eng():
    return _engine.default_engine()


def _neg_fq_bytes(b):
    v = int.from_bytes(b, "big")
    return (0 if v == 0 else _Q - v).to_bytes(32, "big")


class PrivateKey:
    """PrivateKey(Fr), /root/reference/src/types.rs:13-77."""

    def __init__(self, value):
        self.value = value % _R

    @classmethod
    def try_from(cls, data):
        if isinstance(data, str):
            try:
                data = bytes.fromhex(data)
            except ValueError:
                raise Error(ErrorKind.HexDecodeFailed)
        if len(data) != 32:                          # Fr::from_slice -> InvalidLength (types_test.rs:29-46)
            raise Error(ErrorKind.InvalidLength)
        return cls(int.from_bytes(data, "big"))      # values >= r are reduced (examples/bn254.rs:7-12)

    @classmethod
    def random(cls, rng):
        """rng: object with .randbytes(n) (e.g. random.Random) — Fr::random."""
        return cls(int.from_bytes(rng.randbytes(64), "big"))

    def to_bytes(self):
        return self.value.to_bytes(32, "big")

    def to_hex(self):
        return self.to_bytes().hex()

    def __eq__(self, other):
        return isinstance(other, PrivateKey) and other.value == self.value


class _G1Point:
    SIZE = 64

    def __init__(self, raw):
        assert len(raw) == 64
        self.raw = bytes(raw)

    @classmethod
    def from_uncompressed(cls, data):
        """utils.rs:119-127: length, coordinates < q, on curve."""
        data = bytes(data)
        if len(data) != 64:
            raise Error(ErrorKind.InvalidLength)
        _, st = _eng().batch_g1_add(data, bytes(64), 1)     # decode + (P + O) on the device
        _raise(st[0] if data != bytes(64) else ErrorKind.InvalidGroupPoint)
        return cls(data)

    @classmethod
    def from_compressed(cls, data):
        """bn::G1::from_compressed (types.rs:233-237): 0x02/0x03 || x."""
        data = bytes(data)
        if len(data) != 33:
            raise Error(ErrorKind.InvalidEncoding)
        out, st = _eng().batch_g1_decompress(data, 1)
        _raise(st[0])
        return cls(out)

    def to_uncompressed(self):
        if self.raw == bytes(64):
            raise Error(ErrorKind.PointInJacobian)          # utils.rs:184
        return self.raw

    def to_compressed(self):
        """utils.rs:84-104: 0x02 (y even) / 0x03 (y odd) || x."""
        if self.raw == bytes(64):
            raise Error(ErrorKind.PointInJacobian)
        return bytes([3 if self.raw[63] & 1 else 2]) + self.raw[:32]

    def __add__(self, other):
        out, st = _eng().batch_g1_add(self.raw, other.raw, 1)
        _raise(st[0])
        return type(self)(out)

    def __neg__(self):
        if self.raw == bytes(64):
            return type(self)(self.raw)
        return type(self)(self.raw[:32] + _neg_fq_bytes(self.raw[32:]))

    def __sub__(self, other):
        return self + (-other)


class Signature(_G1Point):
    """Signature(G1), /root/reference/src/types.rs:222-286 (no PartialEq in the reference)."""


class PublicKeyG1(_G1Point):
    """PublicKeyG1(G1), /root/reference/src/types.rs:151-218."""

    @classmethod
    def from_private_key(cls, private_key):
        out, st = _eng().batch_g1_mul(None, private_key.to_bytes(), 1, reduce_scalar=True)      # None = G1::one(): the fixed-base table
        _raise(st[0])
        return cls(out)


class PublicKey:
    """PublicKey(G2), /root/reference/src/types.rs:81-148."""
    SIZE = 128

    def __init__(self, raw):
        assert len(raw) == 128
        self.raw = bytes(raw)

    @classmethod
    def from_private_key(cls, private_key):
        out, st = _eng().batch_g2_mul(None, private_key.to_bytes(), 1, reduce_scalar=True)
        _raise(st[0])
        return cls(out)

    @classmethod
    def from_uncompressed(cls, data):
        """utils.rs:107-116: length, coordinates < q, on curve and in the order-r subgroup."""
        data = bytes(data)
        if len(data) != 128:
            raise Error(ErrorKind.InvalidLength)
        if data == bytes(128):
            raise Error(ErrorKind.InvalidGroupPoint)
        # decode with the subgroup check on the device: e(G1, pk) pairing-check path validates it
        st = _eng().batch_pairing_check((1).to_bytes(32, "big") + (2).to_bytes(32, "big"), data, 1, 1,
                                        flags=_engine.FLAG_G2_SUBGROUP_CHECK)
        if st[0] not in (0, ErrorKind.VerificationFailed):
            raise Error(st[0])
        return cls(data)

    @classmethod
    def from_compressed(cls, data):
        """bn::G2::from_compressed (types.rs:91-93): 0x0a/0x0b || BE64(x.im*q + x.re); subgroup-checked."""
        data = bytes(data)
        if len(data) != 65:
            raise Error(ErrorKind.InvalidEncoding)
        out, st = _eng().batch_g2_decompress(data, 1)
        _raise(st[0])
        return cls(out)

    def to_uncompressed(self):
        if self.raw == bytes(128):
            raise Error(ErrorKind.PointInJacobian)          # utils.rs:163
        return self.raw

    def to_compressed(self):
        """utils.rs:130-158: sign || BE64(x.im*q + x.re), sign 0x0b iff u512(y) > u512(-y)."""
        if self.raw == bytes(128):
            raise Error(ErrorKind.PointInJacobian)
        w = [int.from_bytes(self.raw[i:i + 32], "big") for i in range(0, 128, 32)]
        y = w[3] * _Q + w[2]
        yn = ((-w[3]) % _Q) * _Q + ((-w[2]) % _Q)
        return bytes([0x0B if y > yn else 0x0A]) + (w[1] * _Q + w[0]).to_bytes(64, "big")

    def __add__(self, other):
        out, st = _eng().batch_g2_add(self.raw, other.raw, 1)
        _raise(st[0])
        return PublicKey(out)

    def __neg__(self):
        if self.raw == bytes(128):
            return PublicKey(self.raw)
        return PublicKey(self.raw[:64] + _neg_fq_bytes(self.raw[64:96]) + _neg_fq_bytes(self.raw[96:]))

    def __sub__(self, other):
        return self + (-other)

    def __eq__(self, other):
        return isinstance(other, PublicKey) and other.raw == self.raw

    def __hash__(self):
        return hash(self.raw)


class ECDSA:
    """BLS-style aggregate signatures on BN254 (the reference calls the struct ECDSA, src/ecdsa.rs:12-13)."""

    @staticmethod
    def sign(message, private_key):
        sigs, st = _eng().batch_sign([bytes(message)], private_key.to_bytes())
        _raise(st[0])
        return Signature(sigs)

    @staticmethod
    def verify(message, signature, public_key):
        """Returns None on success, raises Error(VerificationFailed / HashToPointError) otherwise."""
        st = _eng().batch_verify([bytes(message)], signature.raw, public_key.raw)
        _raise(st[0])

    @staticmethod
    def batch_verify(messages, signatures, public_keys, engine=None):
        """result[i] is None iff ECDSA.verify(messages[i], signatures[i], public_keys[i]) succeeds,
        else the Error it would raise."""
        n = len(messages)
        if not (len(signatures) == n and len(public_keys) == n):
            raise Error(ErrorKind.InvalidLength)
        eng = engine or _eng()
        st = eng.batch_verify([bytes(m) for m in messages], b"".join(s.raw for s in signatures), b"".join(p.raw for p in public_keys))
        return [None if s == 0 else Error(s) for s in st]


    @staticmethod
    def register_keys(public_keys, engine=None):
        """Register a validator set with the engine for `batch_verify_keyed` (replaces the previous set): result[j] is None,
        or the Error PublicKey.from_uncompressed would raise for key j (types.rs:96-99; the subgroup check always runs)."""
        eng = engine or _eng()
        st = eng.register_keys(b"".join(p.raw for p in public_keys))
        return [None if s == 0 else Error(s) for s in st]

    @staticmethod
    def batch_verify_keyed(messages, signatures, key_indices, engine=None):
        """batch_verify with public_keys[i] named by its index in the registered set: result[i] is None iff
        ECDSA.verify(messages[i], signatures[i], registered[key_indices[i]]) succeeds, Error(IndexOutOfBounds) for an index
        outside the set, else the Error verify would raise."""
        n = len(messages)
        if not (len(signatures) == n and len(key_indices) == n):
            raise Error(ErrorKind.InvalidLength)
        eng = engine or _eng()
        st = eng.batch_verify_keyed([bytes(m) for m in messages], b"".join(s.raw for s in signatures), [int(k) for k in key_indices])
        return [None if s == 0 else Error(s) for s in st]

    @staticmethod
    def batch_verify_keyed_randomized(messages, signatures, key_indices, seed=None, engine=None, rand64=False):
        """batch_verify_keyed through the combined check of items that share a key (64 per pairing product; include/bn254_hip.h:
        bn254_batch_verify_keyed_randomized).  Errors are exact; a None is wrong with probability <= 2^-128 per group."""
        import os
        n = len(messages)
        if not (len(signatures) == n and len(key_indices) == n):
            raise Error(ErrorKind.InvalidLength)
        eng = engine or _eng()
        st = eng.batch_verify_keyed_randomized([bytes(m) for m in messages], b"".join(s.raw for s in signatures), [int(k) for k in key_indices],
                                               seed if seed is not None else os.urandom(32), flags=_engine.FLAG_RAND64 if rand64 else 0)
        return [None if s == 0 else Error(s) for s in st]

    @staticmethod
    def batch_verify_compressed(messages, signatures33, public_keys65, engine=None):
        """batch_verify straight from the compressed wire encodings (33-byte signatures, 65-byte public keys):
        result[i] is None, or the Error that Signature/PublicKey.from_compressed or verify would raise."""
        n = len(messages)
        if not (len(signatures33) == n and len(public_keys65) == n):
            raise Error(ErrorKind.InvalidLength)
        if any(len(s) != 33 for s in signatures33) or any(len(p) != 65 for p in public_keys65):
            raise Error(ErrorKind.InvalidEncoding)
        eng = engine or _eng()
        st = eng.batch_verify_compressed([bytes(m) for m in messages], b"".join(bytes(s) for s in signatures33),
                                         b"".join(bytes(p) for p in public_keys65))
        return [None if s == 0 else Error(s) for s in st]

    @staticmethod
    def batch_verify_randomized(messages, signatures, public_keys, seed=None, engine=None, rand64=False):
        """Same result shape as batch_verify through the randomised combined check (64 items per pairing
        product, include/bn254_hip.h: bn254_batch_verify_randomized).  Errors are always exact; a None is wrong
        with probability <= 2^-128 per group for a fresh secret `seed` (32 bytes; default os.urandom)."""
        import os
        n = len(messages)
        if not (len(signatures) == n and len(public_keys) == n):
            raise Error(ErrorKind.InvalidLength)
        eng = engine or _eng()
        st, _ = eng.batch_verify_randomized([bytes(m) for m in messages], b"".join(s.raw for s in signatures),
                                            b"".join(p.raw for p in public_keys), seed if seed is not None else os.urandom(32),
                                            flags=_engine.FLAG_RAND64 if rand64 else 0)
        return [None if s == 0 else Error(s) for s in st]


def check_public_keys(public_key_g2, public_key_g1):
    """/root/reference/src/ecdsa.rs:78-93."""
    st = _eng().batch_check_public_keys(public_key_g2.raw, public_key_g1.raw, 1)
    _raise(st[0])


def _le_chunks(data):
    """each 32-byte big-endian chunk byte-reversed: zeropool-bn's Borsh (little-endian) affine coordinates"""
    return b"".join(data[i:i + 32][::-1] for i in range(0, len(data), 32))


_NEG_G2_ONE = None


def _neg_g2_one():
    global _NEG_G2_ONE
    if _NEG_G2_ONE is None:
        g2 = PublicKey.from_private_key(PrivateKey(1))
        _NEG_G2_ONE = (-g2).raw
    return _NEG_G2_ONE


def format_pairing_check_uncompressed_values(message, signature, public_key):
    """/root/reference/src/utils.rs:216-239: the two (G1, G2) tuples of the verification equation
    e(H(m), pk) * e(sig, -G2::one()) as 64-/128-byte little-endian buffers for an on-chain alt_bn128
    pairing precompile.  `signature` (64 B) and `public_key` (128 B) are the uncompressed big-endian
    encodings; like the reference this does NOT validate them (it only re-orders bytes) and raises
    IndexError-like InvalidLength on short input where the reference would panic."""
    signature, public_key = bytes(signature), bytes(public_key)
    if len(signature) < 64 or len(public_key) < 128:
        raise Error(ErrorKind.InvalidLength)
    pts, st, _ = _eng().batch_hash_to_g1([bytes(message)])
    _raise(st[0])
    return [(_le_chunks(pts), _le_chunks(public_key[:128])), (_le_chunks(signature[:64]), _le_chunks(_neg_g2_one()))]


def format_pairing_check_values(message, signature, public_key):
    """/root/reference/src/utils.rs:197-214: the same from COMPRESSED signature (33 B) and public key (65 B);
    both are decoded (and thereby validated) first."""
    sig = Signature.from_compressed(signature)
    pk = PublicKey.from_compressed(public_key)
    return format_pairing_check_uncompressed_values(message, sig.raw, pk.raw)
