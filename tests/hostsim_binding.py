"""ctypes binding of tests/hostsim/libhostsim.so (device headers compiled for the host; test-only)."""
import ctypes
import os
import subprocess

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
_lib = None
_built = False


def build_all():
    """all host builds of the device headers the CPU tests load, compiled side by side once per test process (the tracker
    and sanitizer builds take a minute each; one after the other they were most of the suite's run time)"""
    global _built
    if not _built:
        subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
        _built = True


def lib():
    global _lib
    if _lib is None:
        build_all()
        L = ctypes.CDLL(os.path.join(_HERE, "libhostsim.so"))
        cp = ctypes.c_char_p
        L.hs_hash_to_g1.argtypes = [cp, ctypes.c_uint64, cp, ctypes.POINTER(ctypes.c_int)]
        L.hs_verify.argtypes = [cp, ctypes.c_uint64, cp, cp, ctypes.c_uint32]
        L.hs_pairing.argtypes = [cp, cp, ctypes.c_uint64, ctypes.c_uint32, cp, ctypes.c_int]
        L.hs_check_public_keys.argtypes = [cp, cp, ctypes.c_uint32]
        for f in ("hs_g1_add", "hs_g2_add"):
            getattr(L, f).argtypes = [cp, cp, cp]
        for f in ("hs_g1_mul", "hs_g2_mul"):
            getattr(L, f).argtypes = [cp, cp, ctypes.c_int, cp]
        L.hs_sign.argtypes = [cp, ctypes.c_uint64, cp, cp]
        L.hs_fp_op.argtypes = [ctypes.c_int, cp, cp, cp]
        L.hs_g1_decompress.argtypes = [cp, cp]
        L.hs_g1_msum.argtypes = [cp, ctypes.c_uint64, cp]
        L.hs_g2_msum.argtypes = [cp, ctypes.c_uint64, cp]
        L.hs_g2_decompress.argtypes = [cp, cp]
        L.hs_g1_mul_u128.argtypes = [cp, cp, cp]
        L.hs_verify_randomized.argtypes = [cp, ctypes.POINTER(ctypes.c_uint64), cp, cp, ctypes.c_uint64, ctypes.c_uint32, cp, cp, cp]
        _lib = L
    return _lib


def _b(n):
    return ctypes.create_string_buffer(n)


def hash_to_g1(msg):
    o, t = _b(64), ctypes.c_int(0)
    st = lib().hs_hash_to_g1(bytes(msg), len(msg), o, ctypes.byref(t))
    return st, o.raw, t.value


def verify(msg, sig, pk, flags=1):
    return lib().hs_verify(bytes(msg), len(msg), bytes(sig), bytes(pk), flags)


def pairing(g1s, g2s, k=1, flags=0, raw=False):
    o = _b(384)
    st = lib().hs_pairing(bytes(g1s), bytes(g2s), k, flags, o, 1 if raw else 0)
    return st, o.raw


def check_public_keys(pk2, pk1, flags=1):
    return lib().hs_check_public_keys(bytes(pk2), bytes(pk1), flags)


def g1_add(a, b):
    o = _b(64); st = lib().hs_g1_add(bytes(a), bytes(b), o); return st, o.raw


def g2_add(a, b):
    o = _b(128); st = lib().hs_g2_add(bytes(a), bytes(b), o); return st, o.raw


def g1_mul(p, k, reduce=False):
    o = _b(64); st = lib().hs_g1_mul(bytes(p), bytes(k), int(reduce), o); return st, o.raw


def g1_mul_plain_ladder(p, k, reduce=False):
    """the 256-step ladder the G1 kernels used before round 6 (hs_g1_mul: reduce | 2)"""
    o = _b(64); st = lib().hs_g1_mul(bytes(p), bytes(k), int(reduce) | 2, o); return st, o.raw


def glv_decompose(k):
    """(k1, k2) of the device's GLV decomposition of k mod r"""
    L = lib()
    L.hs_glv_decompose.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    a, b = _b(16), _b(16)
    neg = L.hs_glv_decompose(int(k).to_bytes(32, "big"), a, b)
    return int.from_bytes(a.raw, "little"), int.from_bytes(b.raw, "little") * (-1 if neg else 1)


def g2_mul(p, k, reduce=False):
    o = _b(128); st = lib().hs_g2_mul(None if p is None else bytes(p), bytes(k), int(reduce), o); return st, o.raw


def sign(msg, sk):
    o = _b(64); st = lib().hs_sign(bytes(msg), len(msg), bytes(sk), o); return st, o.raw


def fp_op(op, a, b=None):
    o = _b(32); st = lib().hs_fp_op(op, bytes(a), None if b is None else bytes(b), o); return st, o.raw


def g1_decompress(c33):
    o = _b(64); st = lib().hs_g1_decompress(bytes(c33), o); return st, o.raw


def g2_decompress(c65):
    o = _b(128); st = lib().hs_g2_decompress(bytes(c65), o); return st, o.raw


def g1_msum(pts):
    o = _b(64); st = lib().hs_g1_msum(b"".join(pts), len(pts), o); return st, o.raw


def g2_msum(pts):
    o = _b(128); st = lib().hs_g2_msum(b"".join(pts), len(pts), o); return st, o.raw


def g1_mul_u128(p, k):
    o = _b(64)
    st = lib().hs_g1_mul_u128(bytes(p), int(k).to_bytes(16, "little"), o)
    return o.raw, st


def verify_randomized(msgs, sigs, pks, seed32, flags=0):
    n = len(msgs)
    off = (ctypes.c_uint64 * (n + 1))()
    pos = 0
    for i, m in enumerate(msgs):
        off[i] = pos
        pos += len(m)
    off[n] = pos
    st, gr = _b(max(n, 1)), _b(max((n + 63) // 64, 1))
    lib().hs_verify_randomized(b"".join(msgs), off, bytes(sigs), bytes(pks), n, flags, bytes(seed32), st, gr)
    return st.raw[:n], gr.raw[:(n + 63) // 64]
