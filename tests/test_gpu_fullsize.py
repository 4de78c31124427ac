"""BASELINE.json configs 1 - 4 (0-based: verify, aggregate verify, pairings, hash) at their full per-GPU sizes, each
with an ORACLE comparison at that size — all 65 536 verify statuses; 2 048 aggregate tuples of the 1 Mi run at
1 024 signers; 4 096 Gt values + their additive checksum of the 512 Ki pairings; 2 001 hashes of the 16 Mi — plus
size-independent properties over the whole batch, and a 45-second slice of the randomised differential soak.  Inputs
live in HBM (torch tensors) and go through the *_device entry points of the C ABI.  Run on the MI355X box: -m gpu."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu

Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


@pytest.fixture(scope="module")
def env():
    import torch
    import bn254_amd
    from oracle import c_oracle
    eng = bn254_amd.Engine(0)
    return torch, eng, c_oracle, torch.device("cuda", 0)


def _dev(torch, dev, data, dtype=None):
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    return t if dtype is None else t.view(dtype)


def _cores():
    import os
    return max(1, len(os.sched_getaffinity(0)))


def test_config1_all_65536_statuses_vs_oracle(env):
    """configs[1] at its own size: every one of the 65 536 status bytes equals the oracle's, on a batch that mixes valid
    tuples, wrong-message signatures, undecodable points, coordinates >= q, identities and off-curve keys"""
    import random
    torch, eng, c, dev = env
    from tests.datagen import make_verify_batch
    n = 65536
    msgs, sigs, pks, _ = make_verify_batch(eng, n)
    sigs, pks = bytearray(sigs), bytearray(pks)
    rnd = random.Random(65536)
    for i in rnd.sample(range(n), 2048):
        kind = rnd.randrange(8)
        if kind == 0:
            sigs[64 * i + rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            pks[128 * i + rnd.randrange(128)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            sigs[64 * i:64 * i + 32] = (Q + rnd.randrange(1000)).to_bytes(32, "big")
        elif kind == 3:
            j = 128 * i + 32 * rnd.randrange(4)
            pks[j:j + 32] = (Q + rnd.randrange(1 << 200)).to_bytes(32, "big")
        elif kind == 4:
            sigs[64 * i:64 * i + 64] = bytes(64)
        elif kind == 5:
            pks[128 * i:128 * i + 128] = bytes(128)
        elif kind == 6:
            pks[128 * i:128 * i + 128] = pks[128 * (i ^ 1):128 * (i ^ 1) + 128]      # a valid key of another signer
        else:
            sigs[64 * i:64 * i + 64] = rnd.randbytes(64)
    sigs, pks = bytes(sigs), bytes(pks)
    d_msgs, d_sigs, d_pks = _dev(torch, dev, b"".join(msgs)), _dev(torch, dev, sigs), _dev(torch, dev, pks)
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        eng.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_st.data_ptr(), flags=0,
                                stream=stream.cuda_stream)
    stream.synchronize()
    got = bytes(d_st.cpu().numpy())
    want, _ = c.batch_verify(msgs, sigs, pks, flags=0, nthreads=_cores())
    assert got == want
    hist = {b: got.count(b) for b in set(got)}
    assert hist.get(0, 0) > 60000 and hist.get(9, 0) > 1000 and len(hist) >= 4, hist


def test_soak_slice_45s(env):
    """a slice of tests/soak_gpu.py (random sizes, lengths, mutations; 5 modes x 2 flag settings vs the oracle)"""
    from types import SimpleNamespace
    from tests import soak_gpu
    res = soak_gpu.soak(SimpleNamespace(seconds=45.0, seed=20261003))
    assert res["mismatches"] == 0 and res["tuples"] > 1000 and len(res["status_histogram"]) >= 4, res


def test_config5_hash_16m(env):
    """16 M messages -> G1: every message gets a point, try counts follow the geometric law
    (mean 1/0.4726 = 2.116), and a subsample is bit-exact against the oracle."""
    torch, eng, c, dev = env
    n = 1 << 24
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    d_msgs = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, device=dev, generator=g)
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    d_pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    d_tries = torch.zeros(n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eng.batch_hash_to_g1_device(d_msgs.data_ptr(), d_off.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), d_tries.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert int(d_st.max()) == 0
    tries = d_tries.to(torch.float64)
    assert abs(float(tries.mean()) - 2.1160) < 0.003
    frac1 = float((d_tries == 1).double().mean())
    assert abs(frac1 - 0.4726) < 0.001                      # P(first counter works) = (5q/2^256)/2
    idx = torch.cat([torch.arange(0, 64), torch.randint(0, n, (1936,), generator=torch.Generator().manual_seed(1)),
                     d_tries.argmax().cpu().reshape(1)])
    msgs = d_msgs.view(n, 32)[idx.to(dev)].cpu().numpy()
    pts = d_pts.view(n, 64)[idx.to(dev)].cpu().numpy()
    tr = d_tries[idx.to(dev)].cpu().numpy()
    for k in range(len(idx)):
        assert (0, pts[k].tobytes(), int(tr[k])) == c.hash_to_g1(msgs[k].tobytes())


def test_config4_pairings_512k(env):
    """512 Ki pairing products per GPU (the 1/8 shard of config 4): e(P,Q) * e(-P,Q) == 1 for every item,
    e(P,Q)^2 != 1, canonical Gt bytes of a subsample equal the oracle's, and Gt(i) depends only on (P_i,Q_i)."""
    import numpy as np
    torch, eng, c, dev = env
    n, pool = 1 << 19, 512
    sc = [hashlib.sha256(b"cfg4-%d" % j).digest() for j in range(2 * pool)]
    g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    P, st = eng.batch_g1_mul(g1 * pool, b"".join(sc[:pool]), pool, reduce_scalar=True)
    Qs, st2 = eng.batch_g2_mul(None, b"".join(sc[pool:]), pool, reduce_scalar=True)
    assert st == bytes(pool) and st2 == bytes(pool)
    Pn = np.frombuffer(P, dtype=np.uint8).reshape(pool, 64)
    Qn = np.frombuffer(Qs, dtype=np.uint8).reshape(pool, 128)
    negP = np.stack([np.frombuffer(P[64 * j:64 * j + 32] + (Q - int.from_bytes(P[64 * j + 32:64 * j + 64], "big")).to_bytes(32, "big"), dtype=np.uint8)
                     for j in range(pool)])
    i = np.arange(n)
    pi, qi = i % pool, (i // pool + 5 * i) % pool            # (pi, qi) distinct for i < pool^2 = 262 144: two copies of each pair
    # k = 2: (P, Q), (-P, Q) -> 1
    g1s = np.stack([Pn[pi], negP[pi]], axis=1).reshape(-1)
    g2s = np.stack([Qn[qi], Qn[qi]], axis=1).reshape(-1)
    d_g1, d_g2 = torch.from_numpy(g1s.copy()).to(dev), torch.from_numpy(g2s.copy()).to(dev)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eng.batch_pairing_device(d_g1.data_ptr(), d_g2.data_ptr(), n, 2, None, d_st.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert int(d_st.max()) == 0
    # k = 1 with Gt output
    d_g1 = torch.from_numpy(Pn[pi].reshape(-1).copy()).to(dev)
    d_g2 = torch.from_numpy(Qn[qi].reshape(-1).copy()).to(dev)
    d_gt = torch.empty(n * 384, dtype=torch.uint8, device=dev)
    eng.batch_pairing_device(d_g1.data_ptr(), d_g2.data_ptr(), n, 1, d_gt.data_ptr(), d_st.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert int(d_st.min()) == 9 and int(d_st.max()) == 9           # non-degenerate: never one
    gt = d_gt.view(n, 384)
    for k in list(range(16)) + [n - 1, n // 2 + 17]:
        assert gt[k].cpu().numpy().tobytes() == c.pairing(Pn[pi[k]].tobytes(), Qn[qi[k]].tobytes())
    # 64 Ki Gt values of DISTINCT pairs spread over the shard vs the oracle (">= 64 Ki subsample per
    # GPU"), byte for byte, and the additive 64-bit checksum over them (the quantity bench.py all-reduces) computed on the
    # device bytes and on the oracle's
    from bn254_amd.sharding import gt_checksum
    sel = np.unique(np.concatenate([np.arange(0, pool * pool, 4), np.random.default_rng(4).integers(0, n, 1024)]))
    sel = np.concatenate([sel[:65535], [n - 1]])
    assert len(sel) == 65536 and len(np.unique(pi[sel[:-1]] * pool + qi[sel[:-1]])) >= 65000
    want_gt, want_st = c.batch_pairing(Pn[pi[sel]].tobytes(), Qn[qi[sel]].tobytes(), len(sel), 1, nthreads=_cores())
    got_sel = gt[torch.from_numpy(sel).to(dev)].contiguous()
    assert got_sel.cpu().numpy().tobytes() == want_gt and want_st == bytes([9]) * len(sel)
    assert int(gt_checksum(got_sel.view(-1)).item()) & (2**64 - 1) == int(np.frombuffer(want_gt, dtype="<u8").sum(dtype=np.uint64))
    # items with the same (P,Q) indices produce identical bytes: period pool^2
    assert bool((gt[:pool * pool] == gt[pool * pool:]).all())


def test_config3_aggregate_1m(env):
    """1 M aggregate verifies, 1024 signers, ~512 per tuple (random subsets), 1024 messages: all valid
    tuples verify; tuples whose signer list is paired with the wrong message do not."""
    torch, eng, c, dev = env
    from tests.datagen import D, sk_bytes
    M = S = 1024
    n = 1 << 20
    sks = [sk_bytes(j) for j in range(S)]
    msgs = [D("bn254/msg3", m) for m in range(M)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
    assert st == bytes(S)
    sig_pool, st = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks) * M)
    assert st == bytes(M * S)
    # poison a few pool entries (signer s's signature on message m replaced by signer s+1's): every tuple of message m
    # that lists s must fail, everything else verifies — so the 1 Mi run has a non-trivial expected status vector
    sig_pool = bytearray(sig_pool)
    poisoned = [(37 * k % M, 91 * k % (S - 1)) for k in range(1, 9)]
    for m, sg in poisoned:
        sig_pool[64 * (m * S + sg):64 * (m * S + sg) + 64] = sig_pool[64 * (m * S + sg + 1):64 * (m * S + sg + 1) + 64]
    sig_pool = bytes(sig_pool)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    tuple_msg = torch.randint(0, M, (n,), dtype=torch.int32, device=dev, generator=g)
    offs = [0]
    chunks = []
    step = 1 << 17
    for lo in range(0, n, step):                       # build the CSR signer lists chunk-wise (bool matrix 128 MB a time)
        bits = torch.rand((step, S), device=dev, generator=g) < 0.5
        nz = bits.nonzero()
        chunks.append(nz[:, 1].to(torch.int32))
        offs.append(offs[-1] + int(bits.sum()))
        cnt = bits.sum(dim=1)
        chunks_counts = cnt if lo == 0 else torch.cat([chunks_counts, cnt])   # noqa: F821
    signer_idx = torch.cat(chunks)
    tuple_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    tuple_off[1:] = torch.cumsum(chunks_counts.to(torch.int64), 0)
    assert int(tuple_off[-1]) == signer_idx.numel()
    d_msgs = _dev(torch, dev, b"".join(msgs))
    d_moff = torch.arange(0, 32 * (M + 1), 32, dtype=torch.int64, device=dev)
    d_pk, d_sig = _dev(torch, dev, pk_pool), _dev(torch, dev, sig_pool)
    d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eng.batch_aggregate_verify_device(d_msgs.data_ptr(), d_moff.data_ptr(), M, d_pk.data_ptr(), S, d_sig.data_ptr(), tuple_msg.data_ptr(),
                                      tuple_off.data_ptr(), signer_idx.data_ptr(), n, d_st.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    # expected statuses from the construction: 9 iff the tuple's message is poisoned at a signer the tuple lists
    tm = tuple_msg.cpu().numpy()
    st_all = d_st.cpu().numpy()
    off_h = tuple_off.cpu().numpy()
    idx_h = signer_idx.cpu().numpy()
    import numpy as np
    bad_of = {}
    for m, sg in poisoned:
        bad_of.setdefault(m, []).append(sg)
    expect = np.zeros(n, dtype=np.uint8)
    for i in np.nonzero(np.isin(tm, list(bad_of)))[0]:
        lst = idx_h[off_h[i]:off_h[i + 1]]
        if np.isin(bad_of[int(tm[i])], lst).any():
            expect[i] = 9
    assert 1000 < int(expect.sum()) // 9 < 20000
    assert (st_all == expect).all()
    # ORACLE at this size (1 024 signers): 2 048 tuples of this very run — 1 024 random ones and 1 024 of the failing ones —
    # re-derived from the pools with the oracle's own point additions and verify
    rng = np.random.default_rng(3)
    sel = np.concatenate([rng.integers(0, n, 1024), rng.choice(np.nonzero(expect)[0], 1024, replace=False)])
    lists = [idx_h[off_h[i]:off_h[i + 1]] for i in sel]
    t_off = np.concatenate([[0], np.cumsum([len(x) for x in lists])])
    want = c.batch_aggregate_verify(msgs, pk_pool, sig_pool, tm[sel], t_off, np.concatenate(lists), flags=0, nthreads=_cores())
    assert bytes(st_all[sel]) == want and want.count(9) >= 1024 and want.count(0) >= 900
    # signatures of message m presented for message m+1: every tuple must fail
    d_sig_shift = torch.roll(d_sig.view(M, S * 64), 1, dims=0).contiguous().view(-1)
    k = 1 << 14
    eng.batch_aggregate_verify_device(d_msgs.data_ptr(), d_moff.data_ptr(), M, d_pk.data_ptr(), S, d_sig_shift.data_ptr(), tuple_msg.data_ptr(),
                                      tuple_off.data_ptr(), signer_idx.data_ptr(), k, d_st.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert int(d_st[:k].min()) == 9 and int(d_st[:k].max()) == 9


def test_two_contexts_two_threads_two_streams(env):
    """include/bn254_hip.h: "distinct contexts are fully concurrent" / "for concurrent calls on several streams create one
    context per stream".  Two contexts, each driven from its own Python thread (ctypes releases the GIL for the call) on its own
    stream, each verifying a DIFFERENT 20 000-tuple batch several times while the other runs; every status vector equals the
    oracle's for that batch.  (20 000 > the small-batch threshold: the lane-pair kernels, workspaces of both contexts live.)"""
    import threading
    import bn254_amd
    torch, eng, c, dev = env
    from tests.datagen import make_verify_batch
    n, reps = 20000, 4
    batches = []
    for t, (tag, every) in enumerate((("bn254/conc-a", 7), ("bn254/conc-b", 11))):
        msgs, sigs, pks, expected = make_verify_batch(eng, n, corrupt_every=every, tag=tag)
        sigs = bytearray(sigs)
        sigs[64 * (100 + t):64 * (100 + t) + 64] = bytes(64)                  # an identity signature, at a different place in each batch
        sigs[64 * (200 + t) + 63] ^= 1                                        # an off-curve one
        want, _ = c.batch_verify(msgs, bytes(sigs), pks, flags=0, nthreads=_cores())
        assert want != bytes(n) and want.count(9) >= n // every - 2
        batches.append((msgs, bytes(sigs), pks, want))
    assert batches[0][3] != batches[1][3]
    engines = [bn254_amd.Engine(0), bn254_amd.Engine(0)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    bufs = []
    for (msgs, sigs, pks, _), e in zip(batches, engines):
        e.reserve(n)
        bufs.append((_dev(torch, dev, b"".join(msgs)), torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev), _dev(torch, dev, sigs),
                     _dev(torch, dev, pks), [torch.full((n,), 255, dtype=torch.uint8, device=dev) for _ in range(reps)]))
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(2)

    def worker(t):
        try:
            d_msgs, d_off, d_sigs, d_pks, outs = bufs[t]
            start.wait()
            for r in range(reps):
                engines[t].batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, outs[r].data_ptr(),
                                               flags=0, stream=streams[t].cuda_stream)
            streams[t].synchronize()
        except Exception as exc:                                              # surfaced below: an exception in a thread is otherwise lost
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
        assert not th.is_alive()
    assert not errors, errors
    torch.cuda.synchronize()
    for t in range(2):
        for r in range(reps):
            got = bytes(bufs[t][4][r].cpu().numpy())
            assert got == batches[t][3], (t, r, [(i, got[i], batches[t][3][i]) for i in range(n) if got[i] != batches[t][3][i]][:5])
    # host-pointer entry points of the two contexts from the two threads at once (each context stages through its own buffers)
    res = [None, None]

    def host_worker(t):
        try:
            start.wait()
            msgs, sigs, pks, _ = batches[t]
            res[t] = engines[t].batch_verify(msgs, sigs, pks, flags=0)
        except Exception as exc:
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=host_worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    assert res[0] == batches[0][3] and res[1] == batches[1][3]
