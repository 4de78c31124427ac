"""ctypes binding of oracle/libbn254_oracle.so (the C restatement of the reference path).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from bn254_amd/.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbn254_oracle.so")

FLAG_G2_SUBGROUP_CHECK = 1
FLAG_REJECT_IDENTITY = 2


def build(force=False):
    src = os.path.join(_HERE, "bn254_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbn254_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p, u64p = ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)
        L.bn254o_fp_mul_count.restype = ctypes.c_uint64
        L.bn254o_hash_to_g1.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.POINTER(ctypes.c_int)]
        L.bn254o_verify.argtypes = [u8p, ctypes.c_size_t, u8p, u8p, ctypes.c_uint32]
        L.bn254o_check_public_keys.argtypes = [u8p, u8p, ctypes.c_uint32]
        L.bn254o_batch_verify.argtypes = [u8p, u64p, u8p, u8p, ctypes.c_size_t, ctypes.c_uint32, u8p, ctypes.c_int]
        L.bn254o_batch_verify.restype = ctypes.c_uint64
        L.bn254o_batch_verify_randomized.argtypes = [u8p, u64p, u8p, u8p, ctypes.c_size_t, ctypes.c_uint32, u8p, u8p, u8p]
        L.bn254o_pairing_check.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_uint32]
        L.bn254o_pairing.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_uint32, u8p]
        L.bn254o_miller_loop.argtypes = [u8p, u8p, ctypes.c_size_t, u8p]
        L.bn254o_batch_pairing.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, u8p, u8p, ctypes.c_int]
        L.bn254o_hash_candidate.argtypes = [u8p, u8p]
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.bn254o_batch_aggregate_verify.argtypes = [u8p, u64p, ctypes.c_size_t, u8p, ctypes.c_size_t, u8p, u32p, u64p, u32p, ctypes.c_size_t,
                                                    ctypes.c_uint32, u8p, ctypes.c_int]
        for name in ("bn254o_g1_add", "bn254o_g1_mul", "bn254o_g2_add", "bn254o_g2_mul"):
            getattr(L, name).argtypes = [u8p, u8p, u8p]
        L.bn254o_g1_validate.argtypes = [u8p, ctypes.c_uint32]
        L.bn254o_g2_validate.argtypes = [u8p, ctypes.c_uint32]
        L.bn254o_sign.argtypes = [u8p, ctypes.c_size_t, u8p, u8p]
        L.bn254o_public_key_g2.argtypes = [u8p, u8p]
        L.bn254o_public_key_g1.argtypes = [u8p, u8p]
        L.bn254o_g1_compress.argtypes = [u8p, u8p]
        L.bn254o_g1_decompress.argtypes = [u8p, u8p]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle status %d" % code)
        self.code = code


def _buf(n):
    return ctypes.create_string_buffer(n)


def hash_to_g1(msg):
    out, tries = _buf(64), ctypes.c_int(0)
    st = lib().bn254o_hash_to_g1(bytes(msg), len(msg), out, ctypes.byref(tries))
    return st, out.raw, tries.value


def verify(msg, sig64, pk128, flags=FLAG_G2_SUBGROUP_CHECK):
    return lib().bn254o_verify(bytes(msg), len(msg), bytes(sig64), bytes(pk128), flags)


def check_public_keys(pk_g2, pk_g1, flags=FLAG_G2_SUBGROUP_CHECK):
    return lib().bn254o_check_public_keys(bytes(pk_g2), bytes(pk_g1), flags)


def batch_verify(msgs, sigs, pks, flags=FLAG_G2_SUBGROUP_CHECK, nthreads=1):
    """msgs: list of bytes; sigs/pks: concatenated bytes. returns (status bytes, fp_mul count)"""
    n = len(msgs)
    offs = (ctypes.c_uint64 * (n + 1))()
    pos = 0
    for i, m in enumerate(msgs):
        offs[i] = pos
        pos += len(m)
    offs[n] = pos
    status = _buf(max(n, 1))
    cnt = lib().bn254o_batch_verify(b"".join(msgs), offs, bytes(sigs), bytes(pks), n, flags, status, nthreads)
    return status.raw[:n], cnt


FLAG_RAND64 = 0x100
FLAG_RAND_GLV = 0x200


def batch_verify_randomized(msgs, sigs, pks, seed32, flags=FLAG_G2_SUBGROUP_CHECK):
    """randomised batch verification in groups of 64 -> (status bytes, group_ok bytes)"""
    n = len(msgs)
    offs = (ctypes.c_uint64 * (n + 1))()
    pos = 0
    for i, m in enumerate(msgs):
        offs[i] = pos
        pos += len(m)
    offs[n] = pos
    status = _buf(max(n, 1))
    groups = _buf(max((n + 63) // 64, 1))
    assert len(seed32) == 32
    lib().bn254o_batch_verify_randomized(b"".join(msgs), offs, bytes(sigs), bytes(pks), n, flags, bytes(seed32), status, groups)
    return status.raw[:n], groups.raw[:(n + 63) // 64]


def pairing_check(g1s, g2s, k, flags=0):
    return lib().bn254o_pairing_check(bytes(g1s), bytes(g2s), k, flags)


def pairing(g1s, g2s, k=1, flags=0):
    out = _buf(384)
    st = lib().bn254o_pairing(bytes(g1s), bytes(g2s), k, flags, out)
    if st:
        raise OracleError(st)
    return out.raw


def batch_pairing(g1s, g2s, n, k=1, flags=0, nthreads=1, want_gt=True):
    """n independent products of k pairings -> (canonical Gt bytes n*384 or None, status bytes)"""
    assert len(g1s) == n * k * 64 and len(g2s) == n * k * 128
    gt = _buf(max(n, 1) * 384) if want_gt else None
    status = _buf(max(n, 1))
    st = lib().bn254o_batch_pairing(bytes(g1s), bytes(g2s), n, k, flags, gt, status, nthreads)
    if st:
        raise OracleError(st)
    return (gt.raw[:n * 384] if want_gt else None), status.raw[:n]


def batch_aggregate_verify(messages, pk_pool, sig_pool, tuple_msg, tuple_off, signer_idx, flags=0, nthreads=1):
    """aggregate verification over shared pools (config 2); tuple_off has n+1 entries into signer_idx -> status bytes"""
    n, n_msgs = len(tuple_msg), len(messages)
    n_signers = len(pk_pool) // 128
    assert len(sig_pool) == n_msgs * n_signers * 64 and len(tuple_off) == n + 1
    moff = (ctypes.c_uint64 * (n_msgs + 1))()
    pos = 0
    for i, m in enumerate(messages):
        moff[i] = pos
        pos += len(m)
    moff[n_msgs] = pos
    import numpy as np
    tm = np.ascontiguousarray(np.asarray(tuple_msg, dtype=np.uint32).reshape(-1))
    to = np.ascontiguousarray(np.asarray(tuple_off, dtype=np.uint64).reshape(-1))
    si = np.ascontiguousarray(np.concatenate([np.asarray(signer_idx, dtype=np.uint32).reshape(-1), np.zeros(1, dtype=np.uint32)]))
    u32p, u64p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)
    status = _buf(max(n, 1))
    lib().bn254o_batch_aggregate_verify(b"".join(messages), moff, n_msgs, bytes(pk_pool), n_signers, bytes(sig_pool), tm.ctypes.data_as(u32p),
                                        to.ctypes.data_as(u64p), si.ctypes.data_as(u32p), n, flags, status, nthreads)
    return status.raw[:n]


def hash_candidate(h32):
    """what one pass of the try loop does with the digest value h: (1, point64) or (0, zeros)"""
    out = _buf(64)
    ok = lib().bn254o_hash_candidate(bytes(h32), out)
    return ok, out.raw


def miller_loop(g1s, g2s, k=1):
    out = _buf(384)
    st = lib().bn254o_miller_loop(bytes(g1s), bytes(g2s), k, out)
    if st:
        raise OracleError(st)
    return out.raw


def _binop(name, a, b, n):
    out = _buf(n)
    st = getattr(lib(), name)(bytes(a), bytes(b), out)
    if st:
        raise OracleError(st)
    return out.raw


def g1_add(a, b):
    return _binop("bn254o_g1_add", a, b, 64)


def g1_mul(p, scalar32):
    return _binop("bn254o_g1_mul", p, scalar32, 64)


def g2_add(a, b):
    return _binop("bn254o_g2_add", a, b, 128)


def g2_mul(p, scalar32):
    return _binop("bn254o_g2_mul", p, scalar32, 128)


def g1_generator():
    out = _buf(64)
    lib().bn254o_g1_generator(out)
    return out.raw


def g2_generator():
    out = _buf(128)
    lib().bn254o_g2_generator(out)
    return out.raw


def g1_validate(p, flags=0):
    return lib().bn254o_g1_validate(bytes(p), flags)


def g2_validate(p, flags=FLAG_G2_SUBGROUP_CHECK):
    return lib().bn254o_g2_validate(bytes(p), flags)


def sign(msg, sk32):
    out = _buf(64)
    st = lib().bn254o_sign(bytes(msg), len(msg), bytes(sk32), out)
    if st:
        raise OracleError(st)
    return out.raw


def public_key_g2(sk32):
    out = _buf(128)
    lib().bn254o_public_key_g2(bytes(sk32), out)
    return out.raw


def public_key_g1(sk32):
    out = _buf(64)
    lib().bn254o_public_key_g1(bytes(sk32), out)
    return out.raw


def g1_compress(p64):
    out = _buf(33)
    st = lib().bn254o_g1_compress(bytes(p64), out)
    if st:
        raise OracleError(st)
    return out.raw


def g1_decompress(c33):
    out = _buf(64)
    st = lib().bn254o_g1_decompress(bytes(c33), out)
    if st:
        raise OracleError(st)
    return out.raw


def fp_mul_count_reset():
    lib().bn254o_fp_mul_count_reset()


def fp_mul_count():
    return lib().bn254o_fp_mul_count()
