#!/usr/bin/env python3
"""bench.py — BN254 pairings/s of the batch-verify hot path on N MI355X (driver contract).

A "step" = one pass of ECDSA::verify over one batch of 65 536 (message, signature, public key)
tuples per GPU (BASELINE.json configs[1]), inputs already resident in HBM, through the C ABI
(bn254_batch_verify_device): decode -> SHA-256 try-and-increment hash-to-G1 -> 2-pair Miller loop
-> final exponentiation -> status byte.  One verify = 2 pairings (two Miller loops, one final
exponentiation), so pairings/s = 2 x verifies/s.

Multi-GPU: one process per GPU (torchrun), each rank verifies its own 65 536-tuple shard
(weak scaling, no data-path collective) and the per-item status bytes are all-gathered over
RCCL/xGMI inside every timed step (the "final boolean gather" of the north star).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the roofline + cpu_baseline objects).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 65536                      # tuples per GPU per step (configs[1])
CORRUPT_EVERY = 64                 # every 64th signature is wrong -> expected status 9 there
# Algorithmic work per verify, counted by instrumenting the device arithmetic source compiled for
# the host (tests/test_workcount.py keeps these in sync): Montgomery products per kernel stage.
FP_MUL_DECODE = 19
FP_MUL_HASH_FILTER = 4             # per tested counter: x -> Montgomery, x^3 + 3, back to an integer for the Jacobi symbol
FP_MUL_HASH_FINISH = 311           # once per message: the square-root exponentiation of the winning counter + checks
FP_MUL_MILLER = 11138
FP_MUL_FINAL_EXP = 7449           # width-4 window exponentiations by u; incl. 12 canonicalisations for the == 1 test
MAC32_PER_FP_MUL = 136             # ALGORITHMIC unit: an 8x32-bit Montgomery product = 2*8*8 + 8 MAC32.
MUL_INSTR_PER_FP_MUL = 210         # what the kernels actually issue per product with 10x27-bit limbs: 200 v_mad_*64 + 10 v_mul_lo
# VALU roofline: v_mad_u64_u32 issues once per 4 cycles per SIMD (half the 2-cycle full rate):
# 256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 4 = 39.3 T MAC32/s.  The committed microbenchmark
# (profiles/r01_valu_rates_microbench.jsonl) sustains 29.9 T/s of that.
PEAK_MAC32_THEORETICAL = 256 * 4 * 64 * 2.4e9 / 4
PEAK_MAC32_MEASURED = 2.99e13
HBM_PEAK_GBPS = 8000.0
BYTES_PER_VERIFY_IO = 32 + 8 + 64 + 128 + 1   # message + offset + sig + pk + status


def D(tag, i):
    return hashlib.sha256(tag.encode() + i.to_bytes(8, "little")).digest()


def effective_cores():
    """threads worth starting for the CPU baseline: the scheduler affinity, capped by a cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.999)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.999)))
        except Exception:
            pass
    return n


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/pmc_latest.json, produced by tests/pmc_profile.sh + tests/pmc_to_json.py): FETCH_SIZE and
    WRITE_SIZE are KiB counts collected in separate passes; on gfx950 FETCH_SIZE under-counts wide coalesced
    reads by 2x (MI355X_MICROARCH.md §HBM), so reads are doubled.  None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"][kernel]
        return {"bytes_per_launch": 2.0 * 1024.0 * k["FETCH_SIZE"] + 1024.0 * k["WRITE_SIZE"], "fetch_kib_raw": k["FETCH_SIZE"],
                "write_kib_raw": k["WRITE_SIZE"], "batch": k.get("batch"), "source": "profiles/pmc_latest.json"}
    except Exception:
        return None


def measured_valu_issue(kernel):
    """VALU issue-slot utilisation of `kernel` from the same PMC summary: a 64-wide VALU instruction occupies its
    16-lane SIMD for 4 cycles, so utilisation = SQ_INSTS_VALU (wave instructions per launch) x 4 cycles /
    (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.  None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"][kernel]
        cycles = k["GRBM_GUI_ACTIVE"] / 8.0
        return {"valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"], "per_wave": k["SQ_INSTS_VALU"] / k["SQ_WAVES"],
                "kernel_cycles": cycles, "utilisation": 4.0 * k["SQ_INSTS_VALU"] / (256 * 4 * cycles),
                "note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); source profiles/pmc_latest.json"}
    except Exception:
        return None


def other_workloads(args, torch, eng, dev):
    """informational timings of configs 3-5 and of the host-buffer (PCIe-inclusive) verify; one JSON line"""
    import numpy as np
    from tests.datagen import KEY_POOL, make_verify_batch, sk_bytes
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def dev_bytes(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    out = {"workload": args.workload, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "data": "synthetic"}
    if args.workload == "verify-host":
        n = args.batch
        msgs, sigs, pks, expected = make_verify_batch(eng, n)
        assert eng.batch_verify(msgs, sigs, pks) == expected
        packed = bn254_pack(msgs)
        st = __import__("ctypes").create_string_buffer(n)
        lib, h = eng._lib, eng._h
        dt = timed(lambda: lib.bn254_batch_verify(h, packed[0], packed[1], sigs, pks, n, 0, st), args.steps, args.warmup)
        assert st.raw == expected
        out.update(metric="BN254 pairings/sec (batch verify, host buffers: H2D + kernels + D2H + sync)", value=2 * n / dt, unit="pairings/s",
                   ms_per_step=1e3 * dt, batch=n)
    elif args.workload == "verify-compressed":
        # configs[1] tuples given as the compressed wire encodings (33-byte signatures, 65-byte public keys), device resident
        from bn254_amd import PublicKey, Signature
        n = args.batch
        msgs, sigs, pks, expected = make_verify_batch(eng, n)
        cache = {}

        def pkc(b):
            if b not in cache:
                cache[b] = PublicKey(b).to_compressed()
            return cache[b]
        d_msgs = dev_bytes(b"".join(msgs))
        d_sc = dev_bytes(b"".join(Signature(sigs[64 * i:64 * i + 64]).to_compressed() for i in range(n)))
        d_pc = dev_bytes(b"".join(pkc(pks[128 * i:128 * i + 128]) for i in range(n)))
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        eng.reserve(n)
        lib, h = eng._lib, eng._h
        dt = timed(lambda: lib.bn254_batch_verify_compressed_device(h, d_msgs.data_ptr(), d_off.data_ptr(), d_sc.data_ptr(), d_pc.data_ptr(), n,
                                                                    d_st.data_ptr(), stream), args.steps, args.warmup)
        assert bytes(d_st.cpu().numpy()) == expected
        out.update(metric="BN254 pairings/sec (batch verify from compressed encodings: square roots + subgroup test on decode)", value=2 * n / dt,
                   unit="pairings/s", ms_per_step=1e3 * dt, batch=n)
    elif args.workload == "verify-randomized":
        # opt-in randomised batch verification against the exact path on the same inputs;
        # generated in chunks so that the message list stays small on the host
        n = args.batch if args.batch != BATCH else 1 << 20
        seed = hashlib.sha256(b"bench-seed").digest()
        eng.set_option(5, 0)                                  # BN254_OPT_RAND_MIN_BATCH: time the randomised kernels at every size
        chunk = 1 << 16
        parts = [make_verify_batch(eng, min(chunk, n - lo), corrupt_every=0, tag="bn254/msgR%d" % lo) for lo in range(0, n, chunk)]
        msgs = b"".join(b"".join(p[0]) for p in parts)
        d_msgs, d_sigs, d_pks = dev_bytes(msgs), dev_bytes(b"".join(p[1] for p in parts)), dev_bytes(b"".join(p[2] for p in parts))
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        d_gr = torch.zeros((n + 63) // 64, dtype=torch.uint8, device=dev)
        eng.reserve(n + n // 64 + 512)
        ptrs = (d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n)
        res = {}
        for name, flags in (("rand128", 0), ("rand128_glv", 0x200), ("rand64", 0x100)):
            dt = timed(lambda: eng.batch_verify_randomized_device(*ptrs, seed, d_st.data_ptr(), d_gr.data_ptr(), flags=flags, stream=stream),
                       args.steps, args.warmup)
            assert int(d_st.max()) == 0 and int(d_gr.min()) == 1
            res[name] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        dt = timed(lambda: eng.batch_verify_device(*ptrs, d_st.data_ptr(), flags=0, stream=stream), args.steps, args.warmup)
        assert int(d_st.max()) == 0
        res["exact"] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        # worst case: one corrupted item in every group of 64 -> every group fails and is re-verified exactly
        sig_view = d_sigs.view(n, 64)
        saved = sig_view[63::64].clone()
        sig_view[63::64] = sig_view[62::64]
        dt = timed(lambda: eng.batch_verify_randomized_device(*ptrs, seed, d_st.data_ptr(), d_gr.data_ptr(), stream=stream), args.steps, args.warmup)
        assert int(d_gr.max()) == 0 and int((d_st != 0).sum()) == n // 64 and int(d_st.view(-1)[63::64].min()) == 9
        sig_view[63::64] = saved
        res["rand128_every_group_fails"] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        out.update(metric="BN254 verifies/sec, randomised batch verification (groups of 64) vs exact, all-valid batch", unit="verifies/s",
                   value=res["rand128"]["verifies_per_s"], ms_per_step=res["rand128"]["ms_per_step"], batch=n, modes=res,
                   speedup_vs_exact=res["rand128"]["verifies_per_s"] / res["exact"]["verifies_per_s"])
    elif args.workload == "pairing":
        n = args.batch if args.batch != BATCH else 1 << 19          # config 4: 4 Mi pairings over 8 GPUs
        pool = 512
        sc = [hashlib.sha256(b"cfg4-%d" % j).digest() for j in range(2 * pool)]
        g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
        P, _ = eng.batch_g1_mul(g1 * pool, b"".join(sc[:pool]), pool, reduce_scalar=True)
        Qs, _ = eng.batch_g2_mul(None, b"".join(sc[pool:]), pool, reduce_scalar=True)
        i = np.arange(n)
        d_g1 = torch.from_numpy(np.frombuffer(P, dtype=np.uint8).reshape(pool, 64)[(i * 7 + 3) % pool].reshape(-1).copy()).to(dev)
        d_g2 = torch.from_numpy(np.frombuffer(Qs, dtype=np.uint8).reshape(pool, 128)[(i * 13 + 5) % pool].reshape(-1).copy()).to(dev)
        d_gt = torch.empty(n * 384, dtype=torch.uint8, device=dev)
        d_st = torch.empty(n, dtype=torch.uint8, device=dev)
        eng.reserve(n)
        dt = timed(lambda: eng.batch_pairing_device(d_g1.data_ptr(), d_g2.data_ptr(), n, 1, d_gt.data_ptr(), d_st.data_ptr(), stream=stream),
                   args.steps, args.warmup)
        out.update(metric="BN254 pairings/sec (independent pairings, canonical Gt out)", value=n / dt, unit="pairings/s", ms_per_step=1e3 * dt, batch=n)
    elif args.workload == "hash":
        n = args.batch if args.batch != BATCH else 1 << 24          # config 5: 16 Mi messages
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        d_msgs = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, device=dev, generator=g)
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_st = torch.empty(n, dtype=torch.uint8, device=dev)
        eng.reserve(n)
        dt = timed(lambda: eng.batch_hash_to_g1_device(d_msgs.data_ptr(), d_off.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), None, stream=stream),
                   args.steps, args.warmup)
        assert int(d_st.max()) == 0
        out.update(metric="hash_to_try_and_increment messages/sec", value=n / dt, unit="messages/s", ms_per_step=1e3 * dt, batch=n)
    else:
        n = args.batch if args.batch != BATCH else 1 << 20          # config 3: 1 Mi tuples, 1024 signers
        M = S = 1024
        sks = [sk_bytes(j) for j in range(S)]
        msgs = [D("bn254/msg3", m) for m in range(M)]
        pk_pool, _ = eng.batch_g2_mul(None, b"".join(sks), S, reduce_scalar=True)
        sig_pool, _ = eng.batch_sign([msgs[m] for m in range(M) for _ in range(S)], b"".join(sks) * M)
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        tuple_msg = torch.randint(0, M, (n,), dtype=torch.int32, device=dev, generator=g)
        counts, chunks = [], []
        for lo in range(0, n, 1 << 17):
            bits = torch.rand((min(1 << 17, n - lo), S), device=dev, generator=g) < 0.5
            chunks.append(bits.nonzero()[:, 1].to(torch.int32))
            counts.append(bits.sum(dim=1))
        signer_idx = torch.cat(chunks)
        tuple_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        tuple_off[1:] = torch.cumsum(torch.cat(counts).to(torch.int64), 0)
        d_msgs, d_pk, d_sig = dev_bytes(b"".join(msgs)), dev_bytes(pk_pool), dev_bytes(sig_pool)
        d_moff = torch.arange(0, 32 * (M + 1), 32, dtype=torch.int64, device=dev)
        d_st = torch.empty(n, dtype=torch.uint8, device=dev)
        dt = timed(lambda: eng.batch_aggregate_verify_device(d_msgs.data_ptr(), d_moff.data_ptr(), M, d_pk.data_ptr(), S, d_sig.data_ptr(),
                                                             tuple_msg.data_ptr(), tuple_off.data_ptr(), signer_idx.data_ptr(), n, d_st.data_ptr(),
                                                             stream=stream), args.steps, args.warmup)
        assert int(d_st.max()) == 0
        out.update(metric="aggregate verifies/sec (1024 signers, ~512 per tuple, pools decoded per step)", value=n / dt, unit="verifies/s",
                   ms_per_step=1e3 * dt, batch=n, mean_signers_per_tuple=float(signer_idx.numel()) / n)
    print(json.dumps(out))


def bn254_pack(msgs):
    from bn254_amd.engine import pack_messages
    return pack_messages(msgs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH, help="tuples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pair-lanes", action="store_true", help="one lane per verify instead of lane pairs (A/B)")
    ap.add_argument("--split-miller", action="store_true", help="one pairing per lane instead of the fused 2-pair Miller loop (A/B)")
    ap.add_argument("--workload", default="verify", choices=["verify", "verify-host", "verify-compressed", "verify-randomized", "pairing", "hash", "aggregate"],
                    help="verify = the headline (configs[1]); the others time configs 4, 5, 3 or the host-buffer entry point "
                         "(single GPU, informational — see DESIGN.md §4b)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test knobs (not used by the driver): run several ranks on ONE GPU over gloo to exercise the rank logic
    backend = os.environ.get("BN254_BENCH_BACKEND", "nccl")
    if os.environ.get("BN254_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (bn254_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import bn254_amd
    from bn254_amd.sharding import gather_status
    from tests.datagen import KEY_POOL, sk_bytes

    eng = bn254_amd.Engine(local_rank)
    if args.workload != "verify":
        return other_workloads(args, torch, eng, dev)
    n = args.batch
    eng.reserve(2 * n)
    if args.split_miller:
        eng.set_option(1, 1)
    if args.no_pair_lanes:
        eng.set_option(4, 0)


    # ---- synthetic inputs, generated on the GPU by the product's own sign / keygen kernels -------
    base = rank * n                                          # each rank owns a distinct shard
    msgs = [D("bn254/msg2", base + i) for i in range(n)]
    pool = min(KEY_POOL, n)
    sks = [sk_bytes(j) for j in range(pool)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), pool, reduce_scalar=True)
    assert st == bytes(pool)
    sigs, st = eng.batch_sign(msgs, b"".join(sks[(base + i) % pool] for i in range(n)))
    assert st == bytes(n)
    sigs = bytearray(sigs)
    expected = bytearray(n)
    good = bytes(sigs)
    for i in range(CORRUPT_EVERY - 1, n, CORRUPT_EVERY):
        sigs[64 * i:64 * i + 64] = good[64 * (i - 1):64 * i]
        expected[i] = 9
    pks = b"".join(pk_pool[128 * ((base + i) % pool):128 * ((base + i) % pool) + 128] for i in range(n))

    def to_dev(b, dtype=torch.uint8):
        return torch.frombuffer(bytearray(b), dtype=dtype).to(dev)

    d_msgs = to_dev(b"".join(msgs))
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    d_sigs = to_dev(bytes(sigs))
    d_pks = to_dev(pks)
    d_status = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_all = torch.zeros(n * world, dtype=torch.uint8, device=dev) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_status.data_ptr(),
                                flags=0, stream=stream)
        if world > 1:
            if backend == "nccl":
                gather_status(d_status, out=d_all)           # RCCL all-gather over xGMI: the only collective
            else:
                d_all.copy_(gather_status(d_status.cpu()))   # gloo test mode: staged through the host

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # parity gate before any timing is accepted: device statuses == expected pattern
    got = bytes(d_status.cpu().numpy())
    assert got == bytes(expected), "GPU status bytes differ from the expected pattern"

    eng.set_profiling(True)
    kernel_ms = {"decode": 0.0, "hash_to_g1": 0.0, "miller_loop": 0.0, "final_exp": 0.0}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms = eng.last_kernel_ms()                            # HIP events on the launch stream
        for k in kernel_ms:
            kernel_ms[k] += ms[k]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        assert bytes(d_all[rank * n:(rank + 1) * n].cpu().numpy()) == bytes(expected)

    verifies = n * world * args.steps
    verifies_per_s = verifies / elapsed
    result = {
        "metric": "BN254 pairings/sec (batch verify)",
        "value": 2.0 * verifies_per_s,
        "unit": "pairings/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "configs[1]: batch of 65536 independent e(H(m),pk)*e(sig,-G2)==1 verifies per GPU "
                               "(32-byte messages, 1/64 corrupted), 2 pairings per verify",
                   "batch_per_gpu": n, "verifies_per_s": verifies_per_s, "bit_exact_vs_expected": True},
    }

    if rank == 0:
        k_avg = {k: v / args.steps for k, v in kernel_ms.items()}
        dom = max(("miller_loop", "final_exp"), key=lambda k: k_avg[k])
        pair = not (args.no_pair_lanes or args.split_miller)
        kname = {("miller_loop", True): "k_miller_verify_pair", ("miller_loop", False): "k_miller_verify",
                 ("final_exp", True): "k_final_exp_pair", ("final_exp", False): "k_final_exp"}[(dom, pair)]
        fp_mul = FP_MUL_MILLER if dom == "miller_loop" else FP_MUL_FINAL_EXP
        mac_per_launch = fp_mul * MAC32_PER_FP_MUL * n
        achieved = mac_per_launch / (k_avg[dom] * 1e-3) / 1e12
        io_bytes = BYTES_PER_VERIFY_IO * n
        result["roofline"] = {
            "bound": "valu",                       # integer multiply issue (v_mad_u64_u32); not HBM, not MFMA
            "kernel": kname,
            "layout": "one verify per lane pair (Fq2 coefficients in adjacent lanes), two waves per SIMD" if pair else "one verify per lane",
            "achieved": achieved, "peak": PEAK_MAC32_THEORETICAL / 1e12, "unit": "TMAC32/s",
            "frac": achieved / (PEAK_MAC32_THEORETICAL / 1e12),
            "peak_measured_microbench": PEAK_MAC32_MEASURED / 1e12,
            "frac_of_measured_peak": achieved / (PEAK_MAC32_MEASURED / 1e12),
            # register-resident chains of the same product routines (profiles/r01_fp_mul_chain_ceiling.jsonl):
            # 8.16e10 Fq products/s with one wave per SIMD, 1.37e11 when the multiplier pipe is saturated
            "frac_of_occupancy1_product_ceiling": None if pair else (fp_mul * n / (k_avg[dom] * 1e-3)) / 8.16e10,   # one-lane layout only
            "frac_of_saturated_product_rate": (fp_mul * n / (k_avg[dom] * 1e-3)) / 1.37e11,
            "traffic": (measured_traffic(kname) or {}).get("bytes_per_launch"),   # HBM bytes per launch (PMC), private-segment traffic
            "traffic_detail": measured_traffic(kname),
            "valu_issue": measured_valu_issue(kname),
            "multiplier_issue_frac": (fp_mul * MUL_INSTR_PER_FP_MUL * n / (k_avg[dom] * 1e-3)) / PEAK_MAC32_THEORETICAL,
            "kernel_ms": k_avg,
            "mac32_per_verify": {"miller_loop": FP_MUL_MILLER * MAC32_PER_FP_MUL, "final_exp": FP_MUL_FINAL_EXP * MAC32_PER_FP_MUL,
                                 "hash_to_g1_mean": (FP_MUL_HASH_FILTER * 2.12 + FP_MUL_HASH_FINISH) * MAC32_PER_FP_MUL, "decode": FP_MUL_DECODE * MAC32_PER_FP_MUL},
            "hbm": {"algorithmic_bytes_per_step": io_bytes, "achieved_GBps": io_bytes / (1e-3 * 1e3 * elapsed / args.steps) / 1e9,
                    "peak_GBps": HBM_PEAK_GBPS, "note": "evidence that the path is not memory-bound"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c_oracle
            cores = effective_cores()
            sample = min(n, 8192)                            # ~11 CPU-seconds of work in total (1.4 ms per verify)
            t1 = time.perf_counter()
            st_cpu, _ = c_oracle.batch_verify(msgs[:sample], bytes(sigs[:64 * sample]), pks[:128 * sample], flags=0, nthreads=cores)
            dt_all = time.perf_counter() - t1
            assert st_cpu == bytes(expected[:sample]), "oracle disagrees with the expected pattern"
            one = min(sample, 256)
            t1 = time.perf_counter()
            c_oracle.batch_verify(msgs[:one], bytes(sigs[:64 * one]), pks[:128 * one], flags=0, nthreads=1)
            dt_one = time.perf_counter() - t1
            result["cpu_baseline"] = {
                "value": 2.0 * sample / dt_all, "unit": "pairings/s", "cores": cores, "kind": "port",
                "sample": "first %d tuples of the same batch, oracle/bn254_oracle.c (C restatement of the reference path, "
                          "4x64-bit Montgomery limbs, pthreads, gcc -O2)" % sample,
                "single_thread_value": 2.0 * one / dt_one,
            }
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
