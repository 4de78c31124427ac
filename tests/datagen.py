"""Deterministic synthetic inputs: everything derives from
D(tag, i) = SHA256(tag || le64(i)); no RNG state.  Valid signatures are produced ON THE GPU by
the product's own batch_sign / batch_g2_mul kernels (the oracle is only ever the checker)."""
import hashlib

R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
KEY_POOL = 256


def D(tag, i):
    return hashlib.sha256(tag.encode() + i.to_bytes(8, "little")).digest()


def sk_bytes(j):
    return ((int.from_bytes(D("bn254/sk", j), "big") % (R_ORDER - 1)) + 1).to_bytes(32, "big")


def make_verify_batch(eng, n, corrupt_every=64, tag="bn254/msg2", pool=KEY_POOL):
    """config-2 shaped batch: m_i = D(tag,i) (32 B), sig_i = sk_{i mod K} * H(m_i); every item with
    i % corrupt_every == corrupt_every-1 gets the previous item's signature (expected status 9)."""
    pool = min(pool, n)
    sks = [sk_bytes(j) for j in range(pool)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), pool, reduce_scalar=True)
    assert st == bytes(pool)
    msgs = [D(tag, i) for i in range(n)]
    sigs, st = eng.batch_sign(msgs, b"".join(sks[i % pool] for i in range(n)))
    assert st == bytes(n)
    sigs = bytearray(sigs)
    expected = bytearray(n)
    if corrupt_every:
        good = bytes(sigs)
        for i in range(corrupt_every - 1, n, corrupt_every):
            sigs[64 * i:64 * i + 64] = good[64 * (i - 1):64 * i]
            expected[i] = 9
    pks = b"".join(pk_pool[128 * (i % pool):128 * (i % pool) + 128] for i in range(n))
    return msgs, bytes(sigs), pks, bytes(expected)
