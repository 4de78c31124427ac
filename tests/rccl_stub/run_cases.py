"""Runs in a process of its own (tests/test_mgpu_rccl_stub.py), with tests/rccl_stub first on LD_LIBRARY_PATH and WITHOUT torch:
the RCCL branch of bn254_mgpu.hip (gather(): one ncclGroupStart/End around an in-place ncclAllGather — and for the pairing form an
ncclAllReduce — per device, all issued from the calling thread) with FOUR ranks on the box's one GPU, through the stand-in librccl.so.1.
Every device's gathered buffer is compared with the oracle's statuses; the all-reduced Gt checksum with the host's sum.
Prints one JSON line."""
import ctypes
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bn254_amd                                       # noqa: E402
from bn254_amd.engine import MGPU_GATHER_RCCL, MGPU_OPT_GATHER, MGPU_OPT_TIMING, NativeError   # noqa: E402
from oracle import c_oracle as c                       # noqa: E402
from tests.datagen import make_verify_batch            # noqa: E402
from tests.hip_ctypes import DevBuf, Stream, current_device, device_synchronize   # noqa: E402

assert "torch" not in sys.modules
G = 4
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def stub():
    lib = ctypes.CDLL("librccl.so.1")
    with open("/proc/self/maps") as f:
        maps = f.read()
    mine = [ln for ln in maps.splitlines() if "librccl" in ln]
    assert mine and all("tests/rccl_stub" in ln for ln in mine), mine[:3]       # the stand-in, and only the stand-in
    return lib


def stats(lib):
    a = (ctypes.c_int * 6)()
    lib.bn254_rccl_stub_stats(a)
    return dict(zip(("allgather", "allreduce", "inplace", "groups", "max_ranks_in_group", "failed"), a))


def faulty_batch(eng, n, seed):
    rnd = random.Random(seed)
    msgs, sigs, pks, _ = make_verify_batch(eng, n, corrupt_every=5, tag="bn254/stub%d" % seed)
    msgs = [m[:rnd.randrange(0, 33)] if rnd.randrange(4) == 0 else m + bytes(rnd.randrange(0, 70)) for m in msgs]
    sigs, pks = bytearray(sigs), bytearray(pks)
    for i in range(n):
        kind = rnd.randrange(12)
        if kind == 0:
            sigs[64 * i + rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            pks[128 * i + rnd.randrange(128)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            sigs[64 * i:64 * i + 64] = bytes(64)
    return msgs, bytes(sigs), bytes(pks)


def shards(mg, msgs, sigs, pks, n):
    keep, d_msgs, d_off, d_sigs, d_pks = [], [], [], [], []
    for g in range(mg.n_dev):
        lo, hi = mg.shard_range(n, g)
        blob = b"".join(msgs[lo:hi])
        offs = [0]
        for m in msgs[lo:hi]:
            offs.append(offs[-1] + len(m))
        t = [DevBuf(len(blob), data=blob), DevBuf(8 * len(offs), data=b"".join(o.to_bytes(8, "little") for o in offs)),
             DevBuf(64 * (hi - lo), data=sigs[64 * lo:64 * hi]), DevBuf(128 * (hi - lo), data=pks[128 * lo:128 * hi])]
        keep.append(t)
        for lst, x in zip((d_msgs, d_off, d_sigs, d_pks), t):
            lst.append(x.ptr)
    return keep, d_msgs, d_off, d_sigs, d_pks


def main():
    lib = stub()
    eng = bn254_amd.Engine(0)
    dev_before = current_device()
    mg = bn254_amd.MultiEngine([0] * G)
    out = {"cases": []}
    mg.set_option(MGPU_OPT_GATHER, MGPU_GATHER_RCCL)    # accepted for a shared device ONLY because the loaded library is the stand-in
    mg.set_option(MGPU_OPT_TIMING, 1)
    mg.reserve(6000, init_collectives=True)
    # ---- verify: in-place all-gather, own streams and the caller's, two different batches back to back (a stale or misplaced gather shows)
    for n, flags, own in ((3, 0, True), (6, 1, False), (1001, 3, True), (4100, 0, False), (4099, 0, True)):
        msgs, sigs, pks = faulty_batch(eng, n, 900 + n)
        want, _ = c.batch_verify(msgs, sigs, pks, flags=flags, nthreads=8)
        keep, d_msgs, d_off, d_sigs, d_pks = shards(mg, msgs, sigs, pks, n)
        L = mg.gathered_len(n)
        alls = [DevBuf(L, fill=0xEE) for _ in range(G)]
        streams = None if own else [Stream() for _ in range(G)]
        before = stats(lib)
        mg.batch_verify_device(d_msgs, d_off, d_sigs, d_pks, n, [a.ptr for a in alls], flags=flags, streams=None if own else [s.handle for s in streams])
        mg.synchronize()
        if streams:
            for s in streams:
                s.synchronize()
        after = stats(lib)
        assert after["allgather"] == before["allgather"] + 1 and after["groups"] == before["groups"] + 1, (before, after)
        assert after["inplace"] == before["inplace"] + G and after["max_ranks_in_group"] == G, (before, after)
        for g in range(G):
            got = alls[g].download(n)
            bad = [i for i in range(n) if got[i] != want[i]]
            assert not bad, (n, flags, g, bad[:5], [(got[i], want[i]) for i in bad[:5]])
        comp, coll = mg.last_timing()
        assert len(comp) == G and all(x >= 0 for x in comp + coll)
        out["cases"].append({"n": n, "flags": flags, "own_streams": own, "statuses_seen": sorted(set(want))})
        for t in keep:
            for b in t:
                b.free()
        for a in alls:
            a.free()
        if streams:
            for s in streams:
                s.destroy()
    # ---- pairing: all-gather + 8-byte all-reduce of the Gt checksum in ONE group per device; twice (no partial sum survives a call)
    n = 1030
    g1, g2 = c.g1_generator(), c.g2_generator()
    pool_p = [c.g1_mul(g1, (int.from_bytes(hashlib.sha256(b"sp%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(16)]
    pool_q = [c.g2_mul(g2, (int.from_bytes(hashlib.sha256(b"sq%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(16)]
    ps = [pool_p[(7 * i + 3) % 16] for i in range(n)]
    qs = [pool_q[(5 * i + i // 16) % 16] for i in range(n)]
    want_gt, st_eng = c.batch_pairing(b"".join(ps), b"".join(qs), n, nthreads=8)      # the oracle's canonical Gt bytes and statuses
    keep, d_p, d_q, d_gt, d_all, d_cs = [], [], [], [], [], []
    for g in range(G):
        lo, hi = mg.shard_range(n, g)
        t = [DevBuf(64 * (hi - lo), data=b"".join(ps[lo:hi])), DevBuf(128 * (hi - lo), data=b"".join(qs[lo:hi])), DevBuf(384 * (hi - lo), fill=0),
             DevBuf(mg.gathered_len(n), fill=0xEE), DevBuf(8, fill=0)]
        keep.append(t)
        for lst, x in zip((d_p, d_q, d_gt, d_all, d_cs), t):
            lst.append(x.ptr)
    before = stats(lib)
    for _ in range(2):
        mg.batch_pairing_device(d_p, d_q, n, 1, d_gt, d_all, d_cs)
    mg.synchronize()
    after = stats(lib)
    assert after["allgather"] == before["allgather"] + 2 and after["allreduce"] == before["allreduce"] + 2 and after["groups"] == before["groups"] + 2
    want_cs = sum(int.from_bytes(want_gt[8 * i:8 * i + 8], "little") for i in range(len(want_gt) // 8)) & 0xFFFFFFFFFFFFFFFF
    for g in range(G):
        lo, hi = mg.shard_range(n, g)
        assert keep[g][2].download() == want_gt[384 * lo:384 * hi]
        assert keep[g][3].download(n) == st_eng
        assert int.from_bytes(keep[g][4].download(8), "little") == want_cs, (g, hex(want_cs))
    out["pairing"] = {"n": n, "checksum": hex(want_cs)}
    # ---- a failing rank: the call reports BN254_E_RCCL with the library's text, nothing hangs, and the handle works afterwards
    n = 50
    msgs, sigs, pks = faulty_batch(eng, n, 77)
    want, _ = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=4)
    keep2, d_msgs, d_off, d_sigs, d_pks = shards(mg, msgs, sigs, pks, n)
    alls = [DevBuf(mg.gathered_len(n), fill=0xEE) for _ in range(G)]
    os.environ["BN254_RCCL_STUB_FAIL_RANK"] = "2"
    try:
        mg.batch_verify_device(d_msgs, d_off, d_sigs, d_pks, n, [a.ptr for a in alls])
        raise AssertionError("the failing rank went unnoticed")
    except NativeError as e:
        assert e.rc == -10004 and "ncclAllGather" in str(e), (e.rc, str(e))
        out["failing_rank"] = str(e)
    finally:
        del os.environ["BN254_RCCL_STUB_FAIL_RANK"]
    mg.synchronize()
    device_synchronize()
    mg.batch_verify_device(d_msgs, d_off, d_sigs, d_pks, n, [a.ptr for a in alls])
    mg.synchronize()
    for g in range(G):
        assert alls[g].download(n) == want
    # ---- the peer-copy mode on the same handle gives the same bytes; the caller's current device is what it was
    mg.set_option(MGPU_OPT_GATHER, 2)
    alls2 = [DevBuf(mg.gathered_len(n), fill=0xEE) for _ in range(G)]
    before = stats(lib)
    mg.batch_verify_device(d_msgs, d_off, d_sigs, d_pks, n, [a.ptr for a in alls2])
    mg.synchronize()
    assert stats(lib)["allgather"] == before["allgather"]
    for g in range(G):
        assert alls2[g].download(n) == want
    assert current_device() == dev_before
    out["stats"] = stats(lib)
    mg.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
