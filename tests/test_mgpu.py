"""The multi-GPU layer behind the C ABI (include/bn254_hip.h: bn254_mgpu_*; bn254_amd/csrc/bn254_mgpu.hip): one process, one
context + stream + parked worker thread per device, contiguous shards, ONE gather of the status bytes.

CPU: the shard arithmetic, and the refusal to exist without a device.  GPU (-m gpu): four contexts on the box's one GPU (device
list [0, 0, 0, 0] — peer-copy gather, since RCCL refuses two ranks on one device) and the RCCL path with ONE device in the
communicator, against the oracle's status bytes on ragged batches with faults of every class.
Per-tuple semantics: /root/reference/src/ecdsa.rs:49-64; API home: /root/reference/src/lib.rs:60-63."""
import ctypes
import random

import pytest

Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


# ---- CPU ------------------------------------------------------------------------------------------------------------------
def test_shard_arithmetic():
    from bn254_amd.engine import shard_range
    from bn254_amd.sharding import shard_range as torch_side
    for G in (1, 2, 3, 4, 7, 8):
        for n in (0, 1, 2, G - 1, G, G + 1, 63, 64, 65, 65536, 65537, 4194304 + 5):
            if n < 0:
                continue
            S = (n + G - 1) // G
            covered = []
            for g in range(G):
                lo, hi = shard_range(n, g, G)
                assert (lo, hi) == torch_side(n, g, G)          # the Python/torch path of bench.py cuts the same way
                assert 0 <= lo <= hi <= n and hi - lo <= S
                assert lo == min(n, g * S)                        # position in the gathered buffer == global index
                covered.extend(range(lo, hi) if n < 1000 else [])
            if n < 1000:
                assert covered == list(range(n))
            assert sum(shard_range(n, g, G)[1] - shard_range(n, g, G)[0] for g in range(G)) == n


def test_mgpu_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bn254_amd import _native
    _native.build()
    lib = _native.load()
    h = ctypes.c_void_p()
    arr = (ctypes.c_int * 2)(0, 1)
    assert lib.bn254_mgpu_create(arr, 2, ctypes.byref(h)) == -10003          # BN254_E_NO_DEVICE: no CPU fallback
    assert lib.bn254_mgpu_create(arr, 0, ctypes.byref(h)) == -10001
    assert lib.bn254_mgpu_device_count(None) == 0 and lib.bn254_mgpu_shard_len(None, 5) == 0
    with pytest.raises(Exception):
        import bn254_amd
        bn254_amd.MultiEngine([0, 1])


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c():
    from oracle import c_oracle
    return c_oracle


@pytest.fixture(scope="module")
def eng():
    import bn254_amd
    return bn254_amd.Engine(0)


@pytest.fixture(scope="module")
def mg4():
    import bn254_amd
    m = bn254_amd.MultiEngine([0, 0, 0, 0])
    yield m
    m.close()


def _faulty_batch(eng, derived, n, seed):
    """n tuples, about half of them broken in one of the ways the reference's decoders and verify distinguish"""
    from tests.datagen import make_verify_batch
    rnd = random.Random(seed)
    msgs, sigs, pks, _ = make_verify_batch(eng, n, corrupt_every=7, tag="bn254/mgpu%d" % seed)
    msgs = [m[:rnd.randrange(0, 33)] if rnd.randrange(4) == 0 else m + bytes(rnd.randrange(0, 70)) for m in msgs]   # ragged, some empty
    sigs, pks = bytearray(sigs), bytearray(pks)
    off_sub = bytes.fromhex(derived["g2_not_in_subgroup"])
    for i in range(n):
        kind = rnd.randrange(16)
        s, p = memoryview(sigs)[64 * i:64 * i + 64], memoryview(pks)[128 * i:128 * i + 128]
        if kind == 0:
            s[rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            p[rnd.randrange(128)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            s[:32] = (Q + rnd.randrange(1000)).to_bytes(32, "big")
        elif kind == 3:
            j = 32 * rnd.randrange(4)
            p[j:j + 32] = (Q + rnd.randrange(1 << 200)).to_bytes(32, "big")
        elif kind == 4:
            s[:] = bytes(64)
        elif kind == 5:
            p[:] = bytes(128)
        elif kind == 6:
            p[:] = off_sub
    return msgs, bytes(sigs), bytes(pks)


@pytest.mark.gpu
def test_mgpu_host_verify_four_contexts_on_one_gpu_vs_oracle(mg4, eng, c, derived):
    """bn254_mgpu_batch_verify: whole batch in, every shard's statuses straight into the caller's slice — n < G, n = 0, n not
    divisible by G, sizes around the small-batch thresholds of a shard, faults of every class; with and without the flags"""
    assert mg4.n_dev == 4 and mg4.shard_len(10) == 3 and mg4.gathered_len(10) == 12 and mg4.shard_range(10, 3) == (9, 10)
    assert mg4.shard_range(2, 3) == (2, 2)
    assert mg4.batch_verify([], b"", b"") == b""
    seen = set()
    for n, flags in ((1, 0), (2, 3), (3, 0), (5, 1), (64, 0), (257, 3), (1023, 0), (4099, 1), (13001, 0)):
        msgs, sigs, pks = _faulty_batch(eng, derived, n, 100 + n)
        got = mg4.batch_verify(msgs, sigs, pks, flags=flags)
        want, _ = c.batch_verify(msgs, sigs, pks, flags=flags, nthreads=8)
        bad = [i for i in range(n) if got[i] != want[i]]
        assert not bad, (n, flags, bad[:5], [(got[i], want[i]) for i in bad[:5]])
        seen |= set(got)
    assert seen >= {0, 4, 6, 9}
    # one shard == the single-GPU entry point on the same bytes
    msgs, sigs, pks = _faulty_batch(eng, derived, 777, 5)
    assert mg4.batch_verify(msgs, sigs, pks) == eng.batch_verify(msgs, sigs, pks)


@pytest.mark.gpu
def test_mgpu_host_argument_errors(mg4, eng, derived):
    from bn254_amd.engine import NativeError, pack_messages
    lib = mg4._lib
    msgs, sigs, pks = _faulty_batch(eng, derived, 9, 1)
    blob, off = pack_messages(msgs)
    st = ctypes.create_string_buffer(9)
    assert lib.bn254_mgpu_batch_verify(mg4._h, blob, off, sigs, pks, 9, 0, None) == -10001
    off[4], off[5] = off[5], off[4] + 0                      # a reversed pair inside shard 1: refused, nothing followed
    if off[4] > off[5]:
        assert lib.bn254_mgpu_batch_verify(mg4._h, blob, off, sigs, pks, 9, 0, st) == -10001
    with pytest.raises(NativeError):
        mg4.set_option(1, 1)                                 # RCCL forced on a handle that lists a device twice
    with pytest.raises(NativeError):
        mg4.set_option(99, 0)


@pytest.mark.gpu
def test_mgpu_hash_and_pairing_vs_oracle(mg4, eng, c):
    import hashlib
    rnd = random.Random(9)
    msgs = [bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 90))) for _ in range(1501)]
    pts, st, tries = mg4.batch_hash_to_g1(msgs)
    p1, s1, t1 = eng.batch_hash_to_g1(msgs)
    assert (pts, st, tries) == (p1, s1, t1)
    for i in range(0, 1501, 97):
        ost, opt, otries = c.hash_to_g1(msgs[i])
        assert (st[i], pts[64 * i:64 * i + 64], tries[i]) == (ost, opt, otries), i
    # pairings: Gt bytes and the additive checksum over all shards
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    n, k = 203, 2
    g1, g2 = c.g1_generator(), c.g2_generator()
    ps = [c.g1_mul(g1, (int.from_bytes(hashlib.sha256(b"mp%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(n * k)]
    qs = [c.g2_mul(g2, (int.from_bytes(hashlib.sha256(b"mq%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(n * k)]
    gt, st, cs = mg4.batch_pairing(b"".join(ps), b"".join(qs), n, k)
    gt1, st1 = eng.batch_pairing(b"".join(ps), b"".join(qs), n, k)
    assert gt == gt1 and st == st1
    want, want_st = c.batch_pairing(b"".join(ps), b"".join(qs), n, k, nthreads=8)
    assert gt == want and st == want_st
    words = [int.from_bytes(gt[8 * i:8 * i + 8], "little") for i in range(len(gt) // 8)]
    assert cs == sum(words) & 0xFFFFFFFFFFFFFFFF


@pytest.mark.gpu
def test_mgpu_compressed_keyed_and_aggregate_forms_equal_single_gpu(mg4, eng, c, derived):
    """the other verify-shaped host entry points through the multi-GPU layer give the single-GPU entry point's bytes (which the rest of
    the suite pins on the oracle), on ragged batches with faults: compressed encodings, registered keys (the set on every device, an index
    out of range), the aggregate verify of configs[2] (tuples sharded, pools on every device; duplicates, an out-of-range signer)"""
    from bn254_amd.api import PublicKey, Signature            # compression is byte logic in the host mirror
    rnd = random.Random(31)
    n = 203
    msgs, sigs, pks = _faulty_batch(eng, derived, n, 900)
    # compressed: re-encode what decodes, keep a malformed encoding for what does not
    s33, p65 = b"", b""
    for i in range(n):
        try:
            s33 += Signature.from_uncompressed(sigs[64 * i:64 * i + 64]).to_compressed()
        except Exception:
            s33 += b"\x02" + sigs[64 * i:64 * i + 32]
        try:
            p65 += PublicKey.from_uncompressed(pks[128 * i:128 * i + 128]).to_compressed()
        except Exception:
            p65 += b"\x0a" + pks[128 * i:128 * i + 64]
    assert mg4.batch_verify_compressed(msgs, s33, p65) == eng.batch_verify_compressed(msgs, s33, p65)
    # keyed
    keys = b"".join(pks[128 * i:128 * i + 128] for i in range(0, 40))
    kst = mg4.register_keys(keys)
    assert kst == eng.register_keys(keys)
    idx = [rnd.randrange(0, 44) for _ in range(n)]               # 40..43: out of range -> 2
    assert mg4.batch_verify_keyed(msgs, sigs, idx) == eng.batch_verify_keyed(msgs, sigs, idx)
    # aggregate
    from tests.datagen import sk_bytes
    M, S = 3, 21
    amsgs = [b"mg-agg-%d" % m for m in range(M)]
    sks = [sk_bytes(1200 + s) for s in range(S)]
    pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    sig_pool, _ = eng.batch_sign([amsgs[m] for m in range(M) for _ in range(S)], b"".join(sks * M))
    tuples = []
    for i in range(157):
        lst = rnd.sample(range(S), rnd.randrange(0, S + 1))
        if i % 9 == 2 and lst:
            lst.append(lst[0])
        if i % 23 == 5:
            lst.append(S + 3)
        tuples.append((rnd.randrange(M + (1 if i % 31 == 7 else 0)), lst))
    got = mg4.batch_aggregate_verify(amsgs, pk_pool, sig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
    want = eng.batch_aggregate_verify(amsgs, pk_pool, sig_pool, [t[0] for t in tuples], [t[1] for t in tuples])
    off, flat = [0], []
    for _, lst in tuples:
        flat += lst
        off.append(len(flat))
    assert got == want == c.batch_aggregate_verify(amsgs, pk_pool, sig_pool, [t[0] for t in tuples], off, flat, nthreads=8)
    assert {0, 2} <= set(got)


def _device_shards(torch, mg, msgs, sigs, pks, n, dev):
    """per entry g: its shard's buffers resident on the device, offsets relative to the shard's own message buffer"""
    G = mg.n_dev
    keep, d_msgs, d_off, d_sigs, d_pks = [], [], [], [], []
    for g in range(G):
        lo, hi = mg.shard_range(n, g)
        blob = b"".join(msgs[lo:hi])
        offs = [0]
        for m in msgs[lo:hi]:
            offs.append(offs[-1] + len(m))
        t = [torch.frombuffer(bytearray(blob or b"\0"), dtype=torch.uint8).to(dev),
             torch.tensor(offs, dtype=torch.int64, device=dev),
             torch.frombuffer(bytearray(sigs[64 * lo:64 * hi] or b"\0"), dtype=torch.uint8).to(dev),
             torch.frombuffer(bytearray(pks[128 * lo:128 * hi] or b"\0"), dtype=torch.uint8).to(dev)]
        keep.append(t)
        d_msgs.append(t[0].data_ptr()); d_off.append(t[1].data_ptr()); d_sigs.append(t[2].data_ptr()); d_pks.append(t[3].data_ptr())
    return keep, d_msgs, d_off, d_sigs, d_pks


@pytest.mark.gpu
def test_mgpu_device_verify_gather_by_peer_copies_vs_oracle(mg4, eng, c, derived):
    """bn254_mgpu_batch_verify_device on [0, 0, 0, 0]: after the call EVERY entry's buffer holds all n statuses, equal to the
    oracle's; own streams and caller streams; two calls back to back on alternating inputs (a stale or misplaced gather shows)"""
    import torch
    from bn254_amd.engine import MGPU_OPT_TIMING
    dev = torch.device("cuda", 0)
    mg4.set_option(MGPU_OPT_TIMING, 1)
    for n, flags, own_streams in ((1, 0, True), (3, 0, False), (6, 1, True), (1001, 3, False), (4100, 0, True)):
        msgs, sigs, pks = _faulty_batch(eng, derived, n, 300 + n)
        want, _ = c.batch_verify(msgs, sigs, pks, flags=flags, nthreads=8)
        keep, d_msgs, d_off, d_sigs, d_pks = _device_shards(torch, mg4, msgs, sigs, pks, n, dev)
        L = mg4.gathered_len(n)
        alls = [torch.full((L,), 0xEE, dtype=torch.uint8, device=dev) for _ in range(4)]
        streams = None if own_streams else [torch.cuda.Stream(device=dev) for _ in range(4)]
        torch.cuda.synchronize()
        mg4.batch_verify_device(d_msgs, d_off, d_sigs, d_pks, n, [a.data_ptr() for a in alls], flags=flags,
                                streams=None if own_streams else [s.cuda_stream for s in streams])
        mg4.synchronize()
        torch.cuda.synchronize()
        for g in range(4):
            got = bytes(alls[g][:n].cpu().numpy())
            bad = [i for i in range(n) if got[i] != want[i]]
            assert not bad, (n, flags, g, bad[:5], [(got[i], want[i]) for i in bad[:5]])
        comp, coll = mg4.last_timing()
        assert len(comp) == 4 and all(x >= 0 for x in comp + coll)
    mg4.set_option(MGPU_OPT_TIMING, 0)


@pytest.mark.gpu
def test_mgpu_rccl_path_with_one_device_in_the_communicator(eng, c, derived):
    """device list [0]: distinct devices -> the gather is ncclAllGather on an ncclCommInitAll communicator (RCCL's C API, loaded
    with dlopen), the checksum an ncclAllReduce; statuses = the oracle's, Gt checksum = the host's sum"""
    import torch
    import bn254_amd
    from bn254_amd.engine import MGPU_GATHER_RCCL, MGPU_OPT_GATHER, MGPU_OPT_TIMING
    dev = torch.device("cuda", 0)
    mg = bn254_amd.MultiEngine([0])
    try:
        mg.set_option(MGPU_OPT_GATHER, MGPU_GATHER_RCCL)
        mg.set_option(MGPU_OPT_TIMING, 1)
        mg.reserve(5000, init_collectives=True)
        with open("/proc/self/maps") as f:
            assert "librccl" in f.read()
        for n in (5, 4097):
            msgs, sigs, pks = _faulty_batch(eng, derived, n, 500 + n)
            want, _ = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=8)
            keep, d_msgs, d_off, d_sigs, d_pks = _device_shards(torch, mg, msgs, sigs, pks, n, dev)
            out = torch.full((mg.gathered_len(n),), 0xEE, dtype=torch.uint8, device=dev)
            s = torch.cuda.Stream(device=dev)
            torch.cuda.synchronize()
            mg.batch_verify_device(d_msgs, d_off, d_sigs, d_pks, n, [out.data_ptr()], streams=[s.cuda_stream])
            s.synchronize()
            assert bytes(out[:n].cpu().numpy()) == want
            comp, coll = mg.last_timing()
            assert comp[0] > 0 and coll[0] >= 0
        # pairing + checksum all-reduce
        import hashlib
        R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
        n = 130
        g1, g2 = c.g1_generator(), c.g2_generator()
        ps = b"".join(c.g1_mul(g1, (int.from_bytes(hashlib.sha256(b"rp%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(n))
        qs = b"".join(c.g2_mul(g2, (int.from_bytes(hashlib.sha256(b"rq%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(n))
        d_p = torch.frombuffer(bytearray(ps), dtype=torch.uint8).to(dev)
        d_q = torch.frombuffer(bytearray(qs), dtype=torch.uint8).to(dev)
        d_gt = torch.zeros(n * 384, dtype=torch.uint8, device=dev)
        d_st = torch.zeros(mg.gathered_len(n), dtype=torch.uint8, device=dev)
        d_cs = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        mg.batch_pairing_device([d_p.data_ptr()], [d_q.data_ptr()], n, 1, [d_gt.data_ptr()], [d_st.data_ptr()], [d_cs.data_ptr()])
        mg.synchronize()
        gt = bytes(d_gt.cpu().numpy())
        want_gt, _ = eng.batch_pairing(ps, qs, n, 1)
        assert gt == want_gt
        assert int(d_cs.item()) & 0xFFFFFFFFFFFFFFFF == sum(int.from_bytes(gt[8 * i:8 * i + 8], "little") for i in range(len(gt) // 8)) & 0xFFFFFFFFFFFFFFFF
    finally:
        mg.close()


@pytest.mark.gpu
def test_mgpu_device_pairing_checksum_by_peer_copies(mg4, eng, c):
    import hashlib
    import torch
    dev = torch.device("cuda", 0)
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    n = 1030
    g1, g2 = c.g1_generator(), c.g2_generator()
    pool_p = [c.g1_mul(g1, (int.from_bytes(hashlib.sha256(b"cp%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(16)]
    pool_q = [c.g2_mul(g2, (int.from_bytes(hashlib.sha256(b"cq%d" % i).digest(), "big") % R).to_bytes(32, "big")) for i in range(16)]
    ps = [pool_p[(7 * i + 3) % 16] for i in range(n)]
    qs = [pool_q[(5 * i + i // 16) % 16] for i in range(n)]
    want_gt, want_st = eng.batch_pairing(b"".join(ps), b"".join(qs), n, 1)
    keep, d_p, d_q, d_gt, d_all, d_cs = [], [], [], [], [], []
    for g in range(4):
        lo, hi = mg4.shard_range(n, g)
        t = [torch.frombuffer(bytearray(b"".join(ps[lo:hi])), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(b"".join(qs[lo:hi])), dtype=torch.uint8).to(dev),
             torch.zeros((hi - lo) * 384, dtype=torch.uint8, device=dev), torch.full((mg4.gathered_len(n),), 0xEE, dtype=torch.uint8, device=dev),
             torch.zeros(1, dtype=torch.int64, device=dev)]
        keep.append(t)
        for lst, x in zip((d_p, d_q, d_gt, d_all, d_cs), t):
            lst.append(x.data_ptr())
    torch.cuda.synchronize()
    for _ in range(2):                                   # twice: the second call must not see the first call's partial sums
        mg4.batch_pairing_device(d_p, d_q, n, 1, d_gt, d_all, d_cs)
    mg4.synchronize()
    torch.cuda.synchronize()
    want_cs = sum(int.from_bytes(want_gt[8 * i:8 * i + 8], "little") for i in range(len(want_gt) // 8)) & 0xFFFFFFFFFFFFFFFF
    for g in range(4):
        lo, hi = mg4.shard_range(n, g)
        assert bytes(keep[g][2].cpu().numpy()) == want_gt[384 * lo:384 * hi]
        assert bytes(keep[g][3][:n].cpu().numpy()) == want_st
        assert int(keep[g][4].item()) & 0xFFFFFFFFFFFFFFFF == want_cs


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_verify_mgpu_command_line():
    """`bench.py --workload verify-mgpu` — ONE process, the library's own split — on three contexts of the box's GPU (peer-copy gather)
    and with one device entry (RCCL gather): the JSON line of the contract, every device's gathered buffer checked against the pattern
    of every shard, per-device scaling detail, the cost of the layer against the single-GPU entry point"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra, G in ((["--gpus", "3", "--mgpu-devices", "0,0,0"], 3), (["--gpus", "1", "--mgpu-gather", "rccl"], 1)):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "verify-mgpu", "--steps", "3", "--warmup", "1", "--batch", "4096"] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=840)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        r = json.loads(lines[0])
        assert r["n_gpus"] == G and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["unit"] == "pairings/s"
        assert r["config"]["bit_exact_vs_expected"] is True and r["config"]["batch_per_gpu"] == 4096
        assert r["config"]["status_vectors_checked"] == 2 * G * G               # 2 checks x every device's buffer x every shard in it
        assert abs(r["value"] - 2 * G * 4096 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-6
        d = r["scaling_detail"]
        assert len(d["compute_ms_per_step"]["per_device"]) == G and len(d["collective_ms_per_step"]["per_device"]) == G
        assert all(x > 0 for x in d["compute_ms_per_step"]["per_device"]) and all(x >= 0 for x in d["collective_ms_per_step"]["per_device"])
        assert ("ncclAllGather" in r["config"]["collective"]) == (G == 1)
        assert r["single_gpu_direct"]["ms_per_step"] > 0 and "pairings_per_s" in r["host_pointers"]
        assert r["roofline"]["bound"] == "valu" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["kernel_ms"]["miller_loop"] > 0
        assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["cores"] >= 1
