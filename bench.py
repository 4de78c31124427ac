#!/usr/bin/env python3
"""bench.py — BN254 pairings/s of the batch-verify hot path on N MI355X (driver contract).

A "step" = one pass of ECDSA::verify over one batch of 65 536 (message, signature, public key)
tuples per GPU (BASELINE.json configs[1]), inputs already resident in HBM, through the C ABI
(bn254_batch_verify_device): decode -> SHA-256 try-and-increment hash-to-G1 -> 2-pair Miller loop
-> final exponentiation -> status byte.  One verify = 2 pairings (two Miller loops, one final
exponentiation), so pairings/s = 2 x verifies/s.

Multi-GPU: one process per GPU, each rank works on its own shard (weak scaling, no data-path
collective) and the per-item status bytes are all-gathered over RCCL/xGMI inside every timed step
(the "final boolean gather" of the north star).  `--workload pairing` is BASELINE configs[3]
(independent pairings with canonical Gt output, 512 Ki per GPU): besides the status gather, the additive
64-bit checksum over all Gt words is all-reduced (8 bytes) in every step.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(torch.distributed.run, rendezvous on 127.0.0.1) from a parent that never touches the GPU, relays rank 0's
JSON line and exits non-zero if any rank failed.  Under an external torchrun it is a rank.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the roofline + cpu_baseline objects).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 65536                      # tuples per GPU per step (configs[1])
PAIRING_BATCH = 1 << 19            # pairings per GPU per step (configs[3]: 4 Mi over 8 GPUs)
CORRUPT_EVERY = 64                 # one signature in 64 is wrong -> expected status 9 there
# Algorithmic work per verify, counted by instrumenting the device arithmetic source compiled for
# the host (tests/test_workcount.py keeps these in sync): Montgomery products per kernel stage.
FP_MUL_DECODE = 16
FP_MUL_HASH_FILTER = 4             # per tested counter: x -> Montgomery, x^3 + 3, back to an integer for the Jacobi symbol
FP_MUL_HASH_FINISH = 310           # once per message: the square-root exponentiation of the winning counter + checks
FP_MUL_MILLER = 11138
FP_MUL_FINAL_EXP = 6339            # status-only chain (Fuentes-Castaneda hard part, exponentiations by u over the digits {1, 15, 19}); the == 1 test by weak reductions (round 6: no products); the one Fq inversion is by division steps (~2.3 k multiply-adds, not counted as products)
FP_MUL_MILLER_SINGLE = 8419        # one variable pair (configs[3] pairing workload), same instrumentation
FP_MUL_FINAL_EXP_EXACT = 6603      # the exact final exponentiation (canonical Gt) of the pairing workload
MAC32_PER_FP_MUL = 136             # ALGORITHMIC unit: an 8x32-bit Montgomery product = 2*8*8 + 8 MAC32.
# What the pair-layout kernels actually issue per lane (tests/test_workcount.py: hp_lane_counts): multiply-add
# instructions per dual product / single product / square of the device's limb representation.
LIMBS = 9
MADS_DUAL, MADS_SINGLE = 3 * LIMBS * LIMBS, 2 * LIMBS * LIMBS
MUL_LO_PER_PRODUCT = LIMBS
# VALU roofline: v_mad_u64_u32 issues once per 4 cycles per SIMD (half the 2-cycle full rate):
# 256 CU x 4 SIMD x 64 lanes x 2.4 GHz / 4 = 39.3 T MAC32/s.  The rate the multiplier actually sustains is measured
# in this process (bn254_probe_issue_rate) and reported next to it.
PEAK_MAC32_THEORETICAL = 256 * 4 * 64 * 2.4e9 / 4
HBM_PEAK_GBPS = 8000.0
BYTES_PER_VERIFY_IO = 32 + 8 + 64 + 128 + 1   # message + offset + sig + pk + status
BYTES_PER_PAIRING_IO = 64 + 128 + 384 + 1     # G1 + G2 in, Gt + status out


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


PMC_WORKLOAD = "verify"            # which section of profiles/pmc_latest.json the roofline helpers read (set by main from --workload)


def _pmc(kernel):
    """the committed rocprofv3 PMC summary of this same command for `kernel` (profiles/pmc_latest.json, produced by
    tests/pmc_profile.sh + tests/pmc_to_json.py; separate --pmc passes; one section per bench workload), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        return (d["kernels"] if PMC_WORKLOAD == "verify" else d["workloads"][PMC_WORKLOAD]["kernels"])[kernel]
    except Exception:
        return None


def lib_sha16():
    import bn254_amd._native as nat
    try:
        return hashlib.sha256(open(nat.LIB_PATH, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def pmc_as_of():
    """which binary the committed counter summary (profiles/pmc_latest.json) was measured on, and whether that is the library
    loaded now: `traffic` and the instruction counts of a roofline object are static numbers from that file"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
    except Exception:
        return None
    cur = lib_sha16()
    return {"file": "profiles/pmc_latest.json", "lib_sha256_16": d.get("lib_sha256_16"), "current_lib_sha256_16": cur,
            "stale": d.get("lib_sha256_16") != cur, "note": "traffic / instruction counts are read from the file, not measured in this run"}


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel`: FETCH_SIZE and WRITE_SIZE are KiB counts.  On gfx950 FETCH_SIZE
    under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM); whether that holds for the dword scratch
    reloads that make up this traffic is unproven, so both readings are reported and `traffic` is the corrected one."""
    k = _pmc(kernel)
    if not k or "FETCH_SIZE" not in k:
        return None
    return {"bytes_per_launch": 2.0 * 1024.0 * k["FETCH_SIZE"] + 1024.0 * k["WRITE_SIZE"],
            "bytes_per_launch_uncorrected": 1024.0 * k["FETCH_SIZE"] + 1024.0 * k["WRITE_SIZE"],
            "fetch_kib_raw": k["FETCH_SIZE"], "write_kib_raw": k["WRITE_SIZE"], "batch": k.get("batch"), "source": "profiles/pmc_latest.json"}


def measured_valu_issue(kernel, lane_products, probe, kernel_seconds):
    """Cost-weighted VALU issue utilisation of `kernel`: SQ_INSTS_VALU wave instructions (PMC summary) split into the
    multiplier class (v_mad_*64 + v_mul_lo: counted exactly from the per-lane product counts of the host instrumentation
    x the instructions per product) and everything else, each priced at the SECONDS per wave instruction and SIMD the
    issue-rate probe of this run measured for its class (two waves per SIMD, like the kernel), against the kernel's own
    duration in this run:  utilisation = (N_mul x t_mul + N_other x t_other) x waves / (SIMDs x kernel seconds).
    Both sides are wall-clock, so the chip's clock under load cancels; a value near (or slightly above) 1 says the
    kernel issues its instruction mix as fast as the separate single-instruction probes do — it is issue-bound."""
    k = _pmc(kernel)
    if not k or "SQ_INSTS_VALU" not in k or not probe or not lane_products:
        return None
    per_wave = k["SQ_INSTS_VALU"] / k["SQ_WAVES"]
    n_mul = lane_products["dual"] * (MADS_DUAL + MUL_LO_PER_PRODUCT) + lane_products["single"] * (MADS_SINGLE + MUL_LO_PER_PRODUCT)
    n_other = max(per_wave - n_mul, 0.0)
    t_mul, t_other = probe["n_simd"] / probe["mad_u64_u32_wave_inst_per_s"], probe["n_simd"] / probe["add_u32_wave_inst_per_s"]
    priced = (n_mul * t_mul + n_other * t_other) * k["SQ_WAVES"] / probe["n_simd"]
    return {"valu_wave_instructions_per_launch": k["SQ_INSTS_VALU"], "per_wave": per_wave, "multiplier_class_per_wave": n_mul,
            "other_per_wave": n_other, "multiplier_class_share": n_mul / per_wave,
            "ns_per_wave_inst_per_simd": {"multiplier_class": 1e9 * t_mul, "other": 1e9 * t_other},
            "kernel_ms_this_run": 1e3 * kernel_seconds, "utilisation": priced / kernel_seconds,
            "utilisation_flat_4_cycles": 4.0 * k["SQ_INSTS_VALU"] / (probe["n_simd"] * k["GRBM_GUI_ACTIVE"] / 8.0) if "GRBM_GUI_ACTIVE" in k else None,
            # rocprofv3's own derived metric VALUBusy = 100 * SQ_ACTIVE_INST_VALU / CU_NUM / max-over-XCDs(GRBM_GUI_ACTIVE) (counter_defs.yaml, gfx950);
            # the committed pass sums GRBM_GUI_ACTIVE over the 8 XCDs, hence the / 8
            "valu_busy_pct_rocprof_definition": (100.0 * k["SQ_ACTIVE_INST_VALU"] / 256.0 / (k["GRBM_GUI_ACTIVE"] / 8.0)
                                                 if "GRBM_GUI_ACTIVE" in k and "SQ_ACTIVE_INST_VALU" in k else None),
            "sq_wait_any_over_wave_cycles": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in k and k.get("SQ_WAVE_CYCLES") else None,
            "sq_wait_inst_any_over_wave_cycles": k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"] if "SQ_WAIT_INST_ANY" in k and k.get("SQ_WAVE_CYCLES") else None,
            "note": "a CONSISTENCY figure, not a measure of slack: the instruction mix priced with single-instruction probes (two waves per "
                    "SIMD) against the kernel's duration; values above 1 mean the probes over-price the mix.  Stall evidence is the wait "
                    "share next to it (SQ_WAIT_ANY / SQ_WAVE_CYCLES from the PMC pass).  Instruction counts come from "
                    "profiles/pmc_latest.json (see pmc_as_of), issue costs from bn254_probe_issue_rate in this process"}


def issue_probe(eng):
    """multiplier and plain-VALU issue rates of this device, measured now (two waves per SIMD like the pair kernels)"""
    mad, simds = eng.probe_issue_rate(0, 2)
    add, _ = eng.probe_issue_rate(1, 2)
    mul_lo, _ = eng.probe_issue_rate(2, 2)
    mad8, _ = eng.probe_issue_rate(0, 8)
    return {"n_simd": simds, "waves_per_simd": 2, "mad_u64_u32_wave_inst_per_s": mad, "add_u32_wave_inst_per_s": add,
            "mul_lo_u32_wave_inst_per_s": mul_lo, "mad_u64_u32_wave_inst_per_s_8_waves": mad8,
            "peak_mac32_measured": 64.0 * max(mad, mad8),
            "cycles_per_wave_inst_at_2p4GHz": {"mad_u64_u32": simds * 2.4e9 / mad, "add_u32": simds * 2.4e9 / add, "mul_lo_u32": simds * 2.4e9 / mul_lo}}


# Algorithmic Fq products of the other workloads' kernels (tests/test_workcount.py keeps them in sync with the device source):
FP_MUL_MILLER_KEYED = 8220         # keyed verify: two table lines per step, no twist-point arithmetic (2 508 dual + 348 single per lane)
FP_MUL_G1_MADD, FP_MUL_G2_MADD = 11, 22            # mixed additions of k_aggregate_pair (per tuple: both lanes of the pair together)
FP_MUL_AGG_TAIL = 6 + 16 + 16                      # G1 / G2 to affine, the final G1 addition of the two partial sums (its P = Q doubling runs only behind a wave vote: round 6)
HASH_MEAN_TRIES = 2.116                            # counters tested per message on average (p = 0.4726 per try)


def cpu_baseline_verify(msgs, sigs, pks, expected, sample_cap=8192):
    """`cpu_baseline` of a verify line: the oracle (oracle/bn254_oracle.c, the CPU restatement of the reference path — `kind: port`) on the
    first `sample_cap` tuples of the SAME batch, on every core the cgroup grants and on one; statuses compared with the expected pattern"""
    from oracle import c_oracle
    cores = effective_cores()
    sample = min(len(msgs), sample_cap)                  # ~11 CPU-seconds of work in total (1.4 ms per verify)
    t1 = time.perf_counter()
    st_cpu, _ = c_oracle.batch_verify(msgs[:sample], sigs[:64 * sample], pks[:128 * sample], flags=0, nthreads=cores)
    dt_all = time.perf_counter() - t1
    assert st_cpu == expected[:sample], "oracle disagrees with the expected pattern"
    one = min(sample, 256)
    t1 = time.perf_counter()
    c_oracle.batch_verify(msgs[:one], sigs[:64 * one], pks[:128 * one], flags=0, nthreads=1)
    dt_one = time.perf_counter() - t1
    return {"value": 2.0 * sample / dt_all, "unit": "pairings/s", "cores": cores, "kind": "port",
            "sample": "first %d tuples of the same batch, oracle/bn254_oracle.c (C restatement of the reference path, "
                      "4x64-bit Montgomery limbs, pthreads, gcc -O2); statuses equal the GPU's" % sample,
            "single_thread_value": 2.0 * one / dt_one}


def kernel_roofline(kernel, fp_mul_per_launch, kernel_ms, note=None):
    """roofline object of one kernel: algorithmic MAC32 per launch / its HIP-event duration in THIS run, against the
    VALU integer-multiply peak; `traffic` from the committed PMC pass of the same command when there is one"""
    achieved = fp_mul_per_launch * MAC32_PER_FP_MUL / (kernel_ms * 1e-3) / 1e12
    t = measured_traffic(kernel)
    r = {"bound": "valu", "kernel": kernel, "achieved": achieved, "peak": PEAK_MAC32_THEORETICAL / 1e12, "unit": "TMAC32/s",
         "frac": achieved / (PEAK_MAC32_THEORETICAL / 1e12), "traffic": (t or {}).get("bytes_per_launch"), "traffic_detail": t,
         "kernel_ms": kernel_ms, "fp_products_per_launch": fp_mul_per_launch}
    if note:
        r["note"] = note
    return r


def other_workloads(args, torch, eng, dev, stream):
    """configs[2] (aggregate), configs[4] (hash) and the other entry points of the verify path on one GPU; one JSON line
    with a `roofline` for the dominant kernel (HIP events in this run) and a `cpu_baseline` (the oracle on a bounded sample
    of the same inputs, results compared)"""
    from tests.datagen import make_verify_batch, sk_bytes
    sh = stream.cuda_stream

    def timed(fn, steps, warmup, per_step=None):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
            if per_step:
                per_step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def dev_bytes(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    kms = {"decode": 0.0, "hash_to_g1": 0.0, "miller_loop": 0.0, "final_exp": 0.0}

    def collect():
        ms = eng.last_kernel_ms()
        for key in kms:
            kms[key] += ms[key] / args.steps

    cpu = not args.no_cpu_baseline
    cores = effective_cores()
    out = {"workload": args.workload, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "data": "synthetic", "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32"}
    if args.workload == "verify-host":
        n = args.batch or BATCH
        msgs, sigs, pks, expected = make_verify_batch(eng, n)
        pin_threads = os.environ.get("BN254_PINNED_STAGING")             # A/B knob: threads of the pinned staging path (0 = off), default = the library's
        if pin_threads is not None:
            from bn254_amd.engine import OPT_PINNED_STAGING
            eng.set_option(OPT_PINNED_STAGING, int(pin_threads))
        assert eng.batch_verify(msgs, sigs, pks) == expected
        packed = bn254_pack(msgs)
        st = __import__("ctypes").create_string_buffer(n)
        lib, h = eng._lib, eng._h
        eng.set_profiling(True)
        dt = timed(lambda: lib.bn254_batch_verify(h, packed[0], packed[1], sigs, pks, n, 0, st), args.steps, args.warmup, collect)
        assert st.raw == expected
        out.update(metric="BN254 pairings/sec (batch verify, host buffers: H2D + kernels + D2H + sync)", value=2 * n / dt, unit="pairings/s",
                   ms_per_step=1e3 * dt, config={"workload": "configs[1] through the host-pointer entry point (PCIe-inclusive)", "batch": n,
                                                 "pinned_staging_threads": pin_threads},
                   roofline=kernel_roofline("k_miller_verify_pair", FP_MUL_MILLER * n, kms["miller_loop"]), kernel_ms=dict(kms))
        if cpu:
            from oracle import c_oracle
            sample = min(n, 8192)
            t1 = time.perf_counter()
            st_cpu, _ = c_oracle.batch_verify(msgs[:sample], sigs[:64 * sample], pks[:128 * sample], flags=0, nthreads=cores)
            dtc = time.perf_counter() - t1
            assert st_cpu == expected[:sample]
            out["cpu_baseline"] = {"value": 2.0 * sample / dtc, "unit": "pairings/s", "cores": cores, "kind": "port",
                                   "sample": "first %d tuples of the batch, oracle/bn254_oracle.c; statuses equal the GPU's" % sample}
    elif args.workload == "verify-keyed":
        # configs[1] tuples whose public keys are REGISTERED with the context (the batch draws from a pool of 256 keys): the keyed
        # verify reads the keys' line tables instead of recomputing the twist-point arithmetic (include/bn254_hip.h)
        from tests.datagen import KEY_POOL
        n = args.batch or BATCH
        msgs, sigs, pks, expected = make_verify_batch(eng, n)
        pool = min(KEY_POOL, n)
        t1 = time.perf_counter()
        assert eng.register_keys(pks[:128 * pool]) == bytes(pool)
        t_reg = time.perf_counter() - t1
        d_msgs, d_sigs = dev_bytes(b"".join(msgs)), dev_bytes(sigs)
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_idx = (torch.arange(n, dtype=torch.int64, device=dev) % pool).to(torch.int32)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        eng.reserve(n)
        eng.set_profiling(True)
        dt = timed(lambda: eng.batch_verify_keyed_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_idx.data_ptr(), n, d_st.data_ptr(),
                                                         stream=sh), args.steps, args.warmup, collect)
        assert bytes(d_st.cpu().numpy()) == expected
        out.update(metric="BN254 pairings/sec (batch verify, registered keys: line tables from HBM)", value=2 * n / dt, unit="pairings/s",
                   ms_per_step=1e3 * dt, kernel_ms=dict(kms),
                   config={"workload": "configs[1] with the public keys registered beforehand (bn254_ctx_register_keys): %d keys, 12.5 KB of line "
                                       "coefficients each, read per verify" % pool, "batch": n, "registered_keys": pool,
                           "registration_ms": 1e3 * t_reg, "line_table_bytes_read_per_step": 87 * 2 * 2 * 9 * 4 * n},
                   roofline=kernel_roofline("k_miller_verify_keyed_pair", FP_MUL_MILLER_KEYED * n, kms["miller_loop"]))
        if cpu:
            from oracle import c_oracle
            sample = min(n, 8192)
            t1 = time.perf_counter()
            st_cpu, _ = c_oracle.batch_verify(msgs[:sample], sigs[:64 * sample], pks[:128 * sample], flags=1, nthreads=cores)
            dtc = time.perf_counter() - t1
            assert st_cpu == expected[:sample]
            out["cpu_baseline"] = {"value": 2.0 * sample / dtc, "unit": "pairings/s", "cores": cores, "kind": "port",
                                   "sample": "first %d tuples with their keys expanded, oracle/bn254_oracle.c (subgroup check on, as registration "
                                             "does); statuses equal the GPU's" % sample}
    elif args.workload == "verify-compressed":
        # configs[1] tuples given as the compressed wire encodings (33-byte signatures, 65-byte public keys), device resident
        from bn254_amd import PublicKey, Signature
        n = args.batch or BATCH
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
        eng.set_profiling(True)
        dt = timed(lambda: lib.bn254_batch_verify_compressed_device(h, d_msgs.data_ptr(), d_off.data_ptr(), d_sc.data_ptr(), d_pc.data_ptr(), n,
                                                                    d_st.data_ptr(), sh), args.steps, args.warmup, collect)
        assert bytes(d_st.cpu().numpy()) == expected
        out.update(metric="BN254 pairings/sec (batch verify from compressed encodings: square roots + subgroup test on decode)", value=2 * n / dt,
                   unit="pairings/s", ms_per_step=1e3 * dt, kernel_ms=dict(kms),
                   config={"workload": "configs[1] from the 33- / 65-byte compressed encodings, device resident", "batch": n},
                   roofline=kernel_roofline("k_miller_verify_pair", FP_MUL_MILLER * n, kms["miller_loop"]))
        if cpu:
            from oracle import c_oracle
            sample = min(n, 8192)
            t1 = time.perf_counter()
            st_cpu, _ = c_oracle.batch_verify(msgs[:sample], sigs[:64 * sample], pks[:128 * sample], flags=1, nthreads=cores)
            dtc = time.perf_counter() - t1
            assert st_cpu == expected[:sample]
            out["cpu_baseline"] = {"value": 2.0 * sample / dtc, "unit": "pairings/s", "cores": cores, "kind": "port",
                                   "sample": "first %d tuples (uncompressed forms of the same points, subgroup check on), oracle/bn254_oracle.c; "
                                             "statuses equal the GPU's" % sample}
    elif args.workload == "verify-randomized":
        # opt-in randomised batch verification against the exact path on the same inputs;
        # generated in chunks so that the message list stays small on the host
        n = args.batch or (1 << 20)
        seed = hashlib.sha256(b"bench-seed").digest()
        from bn254_amd.engine import OPT_RAND_MIN_BATCH
        eng.set_option(OPT_RAND_MIN_BATCH, 0)                 # time the randomised kernels at every size
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
        eng.set_profiling(True)
        for name, flags in (("rand128", 0), ("rand128_glv", 0x200), ("rand64", 0x100)):
            for key in kms:
                kms[key] = 0.0
            dt = timed(lambda: eng.batch_verify_randomized_device(*ptrs, seed, d_st.data_ptr(), d_gr.data_ptr(), flags=flags, stream=sh),
                       args.steps, args.warmup, collect)
            assert int(d_st.max()) == 0 and int(d_gr.min()) == 1
            res[name] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt,
                         "kernel_ms": {"decode": kms["decode"], "hash_to_g1": kms["hash_to_g1"], "scalar_muls_and_miller_loops": kms["miller_loop"],
                                       "group_tails_final_exp_collect": kms["final_exp"]}}
        eng.set_profiling(False)
        dt = timed(lambda: eng.batch_verify_device(*ptrs, d_st.data_ptr(), flags=0, stream=sh), args.steps, args.warmup)
        assert int(d_st.max()) == 0
        res["exact"] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        # worst case: one corrupted item in every group of 64 -> every group fails and is re-verified exactly
        sig_view = d_sigs.view(n, 64)
        saved = sig_view[63::64].clone()
        sig_view[63::64] = sig_view[62::64]
        dt = timed(lambda: eng.batch_verify_randomized_device(*ptrs, seed, d_st.data_ptr(), d_gr.data_ptr(), stream=sh), args.steps, args.warmup)
        assert int(d_gr.max()) == 0 and int((d_st != 0).sum()) == n // 64 and int(d_st.view(-1)[63::64].min()) == 9
        sig_view[63::64] = saved
        res["rand128_every_group_fails"] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        out.update(metric="BN254 verifies/sec, randomised batch verification (groups of 64) vs exact, all-valid batch", unit="verifies/s",
                   value=res["rand128"]["verifies_per_s"], ms_per_step=res["rand128"]["ms_per_step"], modes=res,
                   config={"workload": "configs[1]-shaped tuples, opt-in randomised batch verification", "batch": n},
                   speedup_vs_exact=res["rand128"]["verifies_per_s"] / res["exact"]["verifies_per_s"],
                   kernel_ms=res["rand128"]["kernel_ms"])
        # dominant interval of the 128-bit mode: k_rand_scale (r_i H(m_i), r_i sig_i: two 128-bit ladders of 4-bit windows in G1) + the per-item
        # Miller loops (two items per lane pair sharing f^2 from 131 072 items on).  Products per item: DESIGN.md section 4c.
        two = n >= 131072
        fp_rand = 2 * (128 * 7 + 33 * 16) + (6800 if two else 8400)
        out["roofline"] = kernel_roofline("k_rand_scale + k_miller_rand2_pair" if two else "k_rand_scale + k_miller_rand_pair", fp_rand * n,
                                          res["rand128"]["kernel_ms"]["scalar_muls_and_miller_loops"],
                                          note="HIP-event interval of the two kernels together (the library's slot [2]); per item 2 x (128 doublings x 7 + 33 "
                                               "additions x 16) products for the two scalar multiplications + %d for its Miller loop; one final exponentiation per "
                                               "64 items is in the next slot" % (6800 if two else 8400))
        if cpu:
            from oracle import c_oracle
            sample = min(n, 2048)
            msg_list = [msgs[32 * i:32 * i + 32] for i in range(sample)]
            sg, pk = bytes(d_sigs[:64 * sample].cpu().numpy()), bytes(d_pks[:128 * sample].cpu().numpy())
            t1 = time.perf_counter()
            st_cpu, gr_cpu = c_oracle.batch_verify_randomized(msg_list, sg, pk, seed, flags=0)
            dtc = time.perf_counter() - t1
            st_gpu, gr_gpu = eng.batch_verify_randomized(msg_list, sg, pk, seed, flags=0)
            assert st_cpu == st_gpu and gr_cpu == gr_gpu, "oracle's randomised verdicts differ from the GPU's"
            out["cpu_baseline"] = {"value": sample / dtc, "unit": "verifies/s", "cores": 1, "kind": "port",
                                   "sample": "first %d tuples (%d groups of 64), oracle/bn254_oracle.c: the same randomised derivation restated (one thread); "
                                             "statuses and group verdicts equal the GPU's" % (sample, sample // 64)}
    elif args.workload == "verify-keyed-randomized":
        # opt-in: registered keys + the combined check of items that share a key (64 per pairing product), against the exact keyed path
        from tests.datagen import KEY_POOL
        n = args.batch or (1 << 20)
        seed = hashlib.sha256(b"bench-seed").digest()
        from bn254_amd.engine import OPT_RAND_MIN_BATCH
        eng.set_option(OPT_RAND_MIN_BATCH, 0)                 # the randomised kernels at every size
        chunk = 1 << 16
        parts = [make_verify_batch(eng, min(chunk, n - lo), corrupt_every=0, tag="bn254/msgK%d" % lo) for lo in range(0, n, chunk)]
        pool = min(KEY_POOL, n)
        assert eng.register_keys(parts[0][2][:128 * pool]) == bytes(pool)
        d_msgs, d_sigs = dev_bytes(b"".join(b"".join(p[0]) for p in parts)), dev_bytes(b"".join(p[1] for p in parts))
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_idx = ((torch.arange(n, dtype=torch.int64, device=dev) % chunk) % pool).to(torch.int32)
        d_st = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        eng.reserve(n + n // 64 + pool + 512)
        ptrs = (d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_idx.data_ptr(), n)
        res = {}
        eng.set_profiling(True)
        for name, flags in (("rand128", 0), ("rand128_glv", 0x200), ("rand64", 0x100)):
            for key in kms:
                kms[key] = 0.0
            dt = timed(lambda: eng.batch_verify_keyed_randomized_device(*ptrs, seed, d_st.data_ptr(), flags=flags, stream=sh), args.steps, args.warmup, collect)
            assert int(d_st.max()) == 0
            res[name] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt,
                         "kernel_ms": {"decode": kms["decode"], "hash_to_g1": kms["hash_to_g1"], "grouping_and_scalar_muls": kms["miller_loop"],
                                       "group_checks_and_rechecks": kms["final_exp"]}}
        dt = timed(lambda: eng.batch_verify_keyed_device(*ptrs, d_st.data_ptr(), stream=sh), args.steps, args.warmup)
        assert int(d_st.max()) == 0
        res["exact_keyed"] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        # one corrupted item in every 64: (nearly) every group fails and its items are re-verified exactly
        sig_view = d_sigs.view(n, 64)
        saved = sig_view[63::64].clone()
        sig_view[63::64] = sig_view[62::64]
        dt = timed(lambda: eng.batch_verify_keyed_randomized_device(*ptrs, seed, d_st.data_ptr(), stream=sh), args.steps, args.warmup)
        assert int((d_st != 0).sum()) == n // 64 and int(d_st.view(-1)[63::64].min()) == 9
        sig_view[63::64] = saved
        res["rand128_one_bad_item_in_64"] = {"verifies_per_s": n / dt, "ms_per_step": 1e3 * dt}
        out.update(metric="BN254 verifies/sec, keyed randomised batch verification (registered keys, groups of 64 per key) vs the exact keyed path",
                   unit="verifies/s", value=res["rand128"]["verifies_per_s"], ms_per_step=res["rand128"]["ms_per_step"], modes=res,
                   config={"workload": "configs[1]-shaped tuples over %d registered keys, opt-in randomised check of same-key groups" % pool, "batch": n},
                   speedup_vs_exact_keyed=res["rand128"]["verifies_per_s"] / res["exact_keyed"]["verifies_per_s"])
        print(json.dumps(out))
        return
    elif args.workload == "hash":
        n = args.batch or (1 << 24)                                 # config 4: 16 Mi messages
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        d_msgs = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, device=dev, generator=g)
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_st = torch.empty(n, dtype=torch.uint8, device=dev)
        eng.reserve(n)
        eng.set_profiling(True)
        dt = timed(lambda: eng.batch_hash_to_g1_device(d_msgs.data_ptr(), d_off.data_ptr(), n, d_pts.data_ptr(), d_st.data_ptr(), None, stream=sh),
                   args.steps, args.warmup, collect)
        assert int(d_st.max()) == 0
        fp_mul = (FP_MUL_HASH_FILTER * HASH_MEAN_TRIES + FP_MUL_HASH_FINISH) * n
        io = (32 + 8 + 64 + 1) * n
        # the library's event slots for this entry point: [0] filter rounds, [1] k_hash_finish, [2] encode (include/bn254_hip.h)
        k_hash = {"filter_rounds_init_round_resolve": kms["decode"], "finish_square_roots": kms["hash_to_g1"], "encode_points": kms["miller_loop"]}
        whole_ms = kms["decode"] + kms["hash_to_g1"]
        out.update(metric="hash_to_try_and_increment messages/sec", value=n / dt, unit="messages/s", ms_per_step=1e3 * dt, kernel_ms=k_hash,
                   config={"workload": "configs[4]: %d 32-byte messages -> G1 points (SHA-256 try-and-increment in rounds: Jacobi filter, then one "
                                       "square root per message)" % n, "batch": n},
                   roofline=kernel_roofline("k_hash_finish", FP_MUL_HASH_FINISH * n, kms["hash_to_g1"],
                                            note="per kernel: the square roots of k_hash_finish over ITS OWN duration (HIP events around the kernel)"))
        whole = kernel_roofline("k_hash_init / round / resolve / finish (whole sequence)", fp_mul, whole_ms,
                                note="whole sequence: every Fq product of the filter rounds and the square roots over the duration of all four "
                                     "kernels — SHA-256 and the Jacobi symbols of the filter rounds are 32-bit integer work outside the MAC32 unit, so "
                                     "this figure understates the utilisation")
        out["roofline"]["whole_sequence"] = {k: whole[k] for k in ("kernel", "achieved", "frac", "kernel_ms", "fp_products_per_launch", "note")}
        out["roofline"]["hbm"] = {"algorithmic_bytes_per_step": io, "achieved_GBps": io / dt / 1e9, "peak_GBps": HBM_PEAK_GBPS}
        out["roofline"]["fp_products_incl_filter_per_launch"] = fp_mul
        if cpu:
            from concurrent.futures import ThreadPoolExecutor
            from oracle import c_oracle
            sample = min(n, 65536)
            m_host = d_msgs[:32 * sample].cpu().numpy().tobytes()
            p_host = d_pts[:64 * sample].cpu().numpy().tobytes()

            def chunk(lo):
                return [c_oracle.hash_to_g1(m_host[32 * i:32 * i + 32])[1] for i in range(lo, min(lo + 1024, sample))]
            t1 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:                     # the oracle call releases the GIL
                pts = b"".join(b"".join(c) for c in ex.map(chunk, range(0, sample, 1024)))
            dtc = time.perf_counter() - t1
            assert pts == p_host, "oracle points differ from the GPU's"
            out["cpu_baseline"] = {"value": sample / dtc, "unit": "messages/s", "cores": cores, "kind": "port",
                                   "sample": "first %d messages, oracle/bn254_oracle.c hash_to_try_and_increment called per message from %d Python "
                                             "threads; points equal the GPU's byte for byte" % (sample, cores)}
    else:
        n = args.batch or (1 << 20)                                 # config 2: 1 Mi tuples, 1024 signers
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

        def call():
            eng.batch_aggregate_verify_device(d_msgs.data_ptr(), d_moff.data_ptr(), M, d_pk.data_ptr(), S, d_sig.data_ptr(), tuple_msg.data_ptr(),
                                              tuple_off.data_ptr(), signer_idx.data_ptr(), n, d_st.data_ptr(), stream=sh)
        from bn254_amd.engine import OPT_AGG_SORT_BY_MSG, OPT_AGG_SUBSET_MIN_TUPLES, OPT_AGG_WIDE_MIN_TUPLES
        eng.set_profiling(True)
        half = max(1, args.steps // 2)

        def side_run(option, value, restore):
            """the same batch with one option changed: (seconds per step, kernel ms per step)"""
            for key in kms:
                kms[key] = 0.0
            eng.set_option(option, value)
            t = timed(call, half, 1, collect)
            k = {kk: v * args.steps / half for kk, v in kms.items()}
            eng.set_option(option, restore)
            assert int(d_st.max()) == 0
            for key in kms:
                kms[key] = 0.0
            return t, k
        main_only = getattr(args, "agg_main_only", False)     # PMC passes: ONLY the default route runs, so that per-launch counters of k_aggregate_pair are its own
        if main_only:
            dt_unsorted = dt_narrow = float("nan")
            k_unsorted = k_narrow = {"decode": float("nan"), "hash_to_g1": float("nan")}
        else:
            # in the caller's (random) tuple order, without the device-side bucketing by message
            dt_unsorted, k_unsorted = side_run(OPT_AGG_SORT_BY_MSG, 0, 1)
            # without the widened tables (keys: 8 signers per table entry, signatures: 4 — rounds 3-4)
            dt_narrow, k_narrow = side_run(OPT_AGG_WIDE_MIN_TUPLES, 0, 262144)
        dt = timed(call, args.steps, args.warmup, collect)
        assert int(d_st.max()) == 0
        k_table = dict(kms)
        if main_only:
            out.update(metric="aggregate verifies/sec (1024 signers, ~512 per tuple)", value=n / dt, unit="verifies/s", ms_per_step=1e3 * dt,
                       kernel_ms={"pools_hash_table": k_table["decode"], "aggregate": k_table["hash_to_g1"], "miller_loop": k_table["miller_loop"], "final_exp": k_table["final_exp"]},
                       config={"workload": "configs[2], default route only (--agg-main-only: for counter passes)", "batch": n})
            print(json.dumps(out))
            return
        # REGISTERED pools (bn254_ctx_register_pools): decode, H(m) and every table once, outside the steps; a step is tuples in, statuses out
        t1 = time.perf_counter()
        eng.register_pools_device(d_msgs.data_ptr(), d_moff.data_ptr(), M, d_pk.data_ptr(), S, d_sig.data_ptr(), n, stream=sh)
        torch.cuda.synchronize()
        t_register = time.perf_counter() - t1

        def call_registered():
            eng.batch_aggregate_verify_registered_device(tuple_msg.data_ptr(), tuple_off.data_ptr(), signer_idx.data_ptr(), n, d_st.data_ptr(), stream=sh)
        d_st.fill_(0xEE)
        for key in kms:
            kms[key] = 0.0
        dt_registered = timed(call_registered, args.steps, args.warmup, collect)
        assert int(d_st.max()) == 0
        k_registered = dict(kms)
        eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, 0)               # every key added one by one (rounds 1-2)
        dt_direct = timed(call, half, 1)
        assert int(d_st.max()) == 0
        eng.set_option(OPT_AGG_SUBSET_MIN_TUPLES, 4096)
        total_signers = int(signer_idx.numel())
        groups = (S + 7) // 8
        # route of this batch (bn254_hip.hip: bn254_batch_aggregate_verify_device): key sums from the subset tables — 16 signers per entry
        # from 262 144 tuples on, else 8 — and, a message being shared by >= 64 tuples on average, signature sums from the per-message
        # tables — 8 signers per entry when a message is shared by >= 512 tuples, else 4
        sig_tables = n >= 64 * M
        wide_keys = n >= 262144
        wide_sigs = wide_keys and n >= 512 * M
        key_adds = (groups + 1) // 2 if wide_keys else groups
        sig_adds = groups if wide_sigs else 2 * groups
        agg_products = (FP_MUL_G1_MADD * sig_adds * n if sig_tables else FP_MUL_G1_MADD * total_signers) + (FP_MUL_G2_MADD * key_adds + FP_MUL_AGG_TAIL) * n
        out.update(metric="aggregate verifies/sec (1024 signers, ~512 per tuple)", value=n / dt, unit="verifies/s",
                   ms_per_step=1e3 * dt, kernel_ms={"pools_hash_table": k_table["decode"], "aggregate": k_table["hash_to_g1"],
                                                    "miller_loop": k_table["miller_loop"], "final_exp": k_table["final_exp"]},
                   config={"workload": "configs[2]: %d aggregate verifies over pools of %d signers x %d messages (random subsets, %.1f signers per "
                                       "tuple): G1 / G2 sums, then one verify each" % (n, S, M, total_signers / n), "batch": n,
                           "mean_signers_per_tuple": total_signers / n, "key_route": "subset sums of the key pool: %d table additions per tuple" % key_adds,
                           "signature_route": ("per-message subset tables: %d table additions per tuple" % sig_adds) if sig_tables else "one addition per signer"},
                   registered_pools={"verifies_per_s": n / dt_registered, "ms_per_step": 1e3 * dt_registered, "register_once_ms": 1e3 * t_register,
                                     "pools_hash_table_ms": k_registered["decode"], "aggregate_kernel_ms": k_registered["hash_to_g1"],
                                     "note": "bn254_ctx_register_pools_device once (pools decoded, messages hashed, all subset-sum tables built), then "
                                             "bn254_batch_aggregate_verify_registered_device per step: only the tuples cross the boundary — what a caller with a "
                                             "fixed validator set runs; `value` stays the raw-pool call that rebuilds everything per step"},
                   without_subset_sum_table={"verifies_per_s": n / dt_direct, "ms_per_step": 1e3 * dt_direct},
                   without_widened_tables={"verifies_per_s": n / dt_narrow, "ms_per_step": 1e3 * dt_narrow, "aggregate_kernel_ms": k_narrow["hash_to_g1"],
                                           "pools_hash_table_ms": k_narrow["decode"]},
                   without_bucketing_by_message={"verifies_per_s": n / dt_unsorted, "ms_per_step": 1e3 * dt_unsorted, "aggregate_kernel_ms": k_unsorted["hash_to_g1"],
                                                 "pools_hash_table_ms": k_unsorted["decode"]},
                   roofline=kernel_roofline("k_aggregate_pair", agg_products, k_table["hash_to_g1"],
                                            note="products per tuple: 11 per signature-table entry added + 22 per key-table entry added + 45 (round 6: the zero tests of the additions are weak reductions, not products); the walk over "
                                                 "the signer list (status checks, mask bits) is not MAC32 work; building the tables is in pools_hash_table"))
        out["roofline"]["whole_step"] = {"fp_products_per_tuple": agg_products / n + FP_MUL_MILLER + FP_MUL_FINAL_EXP,
                                         "frac": (agg_products / n + FP_MUL_MILLER + FP_MUL_FINAL_EXP) * MAC32_PER_FP_MUL * n / dt / PEAK_MAC32_THEORETICAL}
        if cpu:
            from oracle import c_oracle
            sample = min(n, 2048)
            hi = int(tuple_off[sample].item())
            t1 = time.perf_counter()
            st_cpu = c_oracle.batch_aggregate_verify(msgs, pk_pool, sig_pool, tuple_msg[:sample].cpu().numpy(), tuple_off[:sample + 1].cpu().numpy(),
                                                     signer_idx[:hi].cpu().numpy(), nthreads=cores)
            dtc = time.perf_counter() - t1
            assert st_cpu == bytes(d_st[:sample].cpu().numpy()), "oracle statuses differ from the GPU's"
            out["cpu_baseline"] = {"value": sample / dtc, "unit": "verifies/s", "cores": cores, "kind": "port",
                                   "sample": "first %d tuples (%d signer entries), oracle/bn254_oracle.c: Add of src/types.rs per signer, then verify; "
                                             "statuses equal the GPU's" % (sample, hi)}
    out["pmc_as_of"] = pmc_as_of()
    print(json.dumps(out))


def bn254_pack(msgs):
    from bn254_amd.engine import pack_messages
    return pack_messages(msgs)


# ------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without an external torchrun
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of an N-rank run.  Makes no GPU call and imports no torch: it starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py <argv>` as a CHILD process (never an
    exec), relays the single JSON line of rank 0 and returns non-zero if any rank failed or no line was printed."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print("bench.py: a rank failed (torchrun exit code %d)" % proc.returncode, file=sys.stderr)
        return proc.returncode
    if len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        return 1
    print(lines[0])
    return 0


class Rank:
    """what a rank knows: its place in the job, its device, the stream everything runs on"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.gpus > 1 and self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        # test knobs (not used by the driver): several ranks on ONE GPU over gloo exercise the rank logic on a 1-GPU box
        self.backend = os.environ.get("BN254_BENCH_BACKEND", "nccl")
        if os.environ.get("BN254_BENCH_SINGLE_DEVICE") == "1":
            self.local_rank = 0
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (bn254_amd has no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        # BN254_BENCH_FORCE_DIST=1 (test knob): run the collectives even in a 1-rank job, so that the RCCL path (stream
        # ordering, all_gather_into_tensor / all_reduce on device tensors) can be exercised on a one-GPU box
        self.dist_on = self.world > 1 or os.environ.get("BN254_BENCH_FORCE_DIST") == "1"
        if self.dist_on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.dev)
            else:
                dist.init_process_group(backend=self.backend)
            assert dist.get_world_size() == self.world
        # ONE explicit stream carries the kernels of the library (its handle is passed to every *_device call), the
        # torch ops that prepare the inputs, and the RCCL collectives (torch orders a collective after the current
        # stream) — no reliance on the null stream or on the context's private non-blocking stream.
        self.stream = torch.cuda.Stream(device=self.dev)
        self.own_elapsed = None          # this rank's own clock over the timed steps (the reported time is the MAX over ranks)
        self.coll_events = []            # (start, end) events around every collective of the timed steps (RCCL)
        self.coll_host_s = 0.0           # ... or its host time (gloo test mode: the collective is staged through the host)
        self.timing = False

    def _collective(self, fn):
        """run one collective; inside the timed steps also take its own time: HIP events on the rank's stream around it
        (torch orders an RCCL collective after the current stream and makes the stream wait for its result)"""
        if not self.timing:
            return fn()
        if self.backend == "nccl":
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            out = fn()
            e1.record(self.stream)
            self.coll_events.append((e0, e1))
            return out
        t0 = time.perf_counter()
        out = fn()
        self.coll_host_s += time.perf_counter() - t0
        return out

    def gather(self, local_status, out):
        """the one collective of a verify step: all-gather of the status bytes (RCCL over xGMI)"""
        from bn254_amd.sharding import gather_status
        if self.backend == "nccl":
            self._collective(lambda: gather_status(local_status, out=out))
        else:                                                       # gloo test mode: staged through the host
            self._collective(lambda: out.copy_(gather_status(local_status.cpu())))

    def allreduce_checksum(self, local_sum):
        from bn254_amd.sharding import allreduce_checksum
        return self._collective(lambda: allreduce_checksum(local_sum if self.backend == "nccl" else local_sum.cpu()))

    def time_steps(self, step, steps, warmup, after_warmup=None, per_step=None):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
        torch, dist = self.torch, self.dist
        with torch.cuda.stream(self.stream):
            for k in range(warmup):
                step(k)
            torch.cuda.synchronize()
            if after_warmup:
                after_warmup()
            if self.dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            self.timing = True
            t0 = time.perf_counter()
            for k in range(steps):
                step(warmup + k)
                if per_step:
                    per_step()
            torch.cuda.synchronize()
            self.own_elapsed = time.perf_counter() - t0            # this rank alone, before it waits for the others
            if self.dist_on:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            self.timing = False
        if self.dist_on:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def scaling_detail(self, steps, kernel_ms_total=None):
        """what makes an N-rank line diagnosable: every rank's own elapsed time over the timed steps (the headline uses the
        max), the sum of its kernel times (HIP events inside the library) and the time of its collectives, all per step —
        one all_gather of three doubles per rank AFTER the timed region.  None in a job without collectives."""
        if not self.dist_on:
            return None
        torch, dist = self.torch, self.dist
        coll_ms = 1e3 * self.coll_host_s
        if self.coll_events:
            torch.cuda.synchronize()
            coll_ms = sum(a.elapsed_time(b) for a, b in self.coll_events)
        mine = torch.tensor([self.own_elapsed or 0.0, (kernel_ms_total if kernel_ms_total is not None else float("nan")), coll_ms],
                            dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        rows = [[float(x) for x in t.cpu()] for t in every]

        def stats(vals):
            return {"min": min(vals), "max": max(vals), "mean": sum(vals) / len(vals), "per_rank": vals}
        el = [1e3 * r[0] / steps for r in rows]
        km = [r[1] / steps for r in rows]
        cm = [r[2] / steps for r in rows]
        return {"own_elapsed_ms_per_step": stats(el),
                "kernel_ms_per_step": stats(km) if kernel_ms_total is not None else None,
                "collective_ms_per_step": stats(cm),
                "collective_timed_by": "HIP events on the rank's stream around the collective" if self.coll_events else "host clock (gloo test mode)",
                "slowest_rank": max(range(len(el)), key=lambda i: el[i]),
                "spread_pct": 100.0 * (max(el) - min(el)) / max(el) if max(el) > 0 else 0.0,
                "note": "headline ms_per_step = max over ranks incl. the closing barrier; own_elapsed is each rank's clock before that barrier"}

    def finish(self):
        if self.dist_on:
            self.dist.barrier()
            self.dist.destroy_process_group()


def corrupt_phase(rank, step):
    """which residue (mod CORRUPT_EVERY) of a shard carries the wrong signature: depends on the rank AND on the
    step parity, so a gather that returned another rank's or a stale step's statuses is caught"""
    return (CORRUPT_EVERY - 1 - rank - (CORRUPT_EVERY // 2) * (step & 1)) % CORRUPT_EVERY


def run_verify(args, R):
    torch = R.torch
    import bn254_amd
    from tests.datagen import KEY_POOL, sk_bytes
    eng = bn254_amd.Engine(R.local_rank)
    n = args.batch or BATCH
    eng.reserve(2 * n)
    if args.split_miller:
        eng.set_option(bn254_amd.engine.OPT_SPLIT_MILLER, 1)
    if args.no_pair_lanes:
        eng.set_option(bn254_amd.engine.OPT_PAIR_LANES, 0)
    rank, world, dev = R.rank, R.world, R.dev

    # ---- synthetic inputs, generated on the GPU by the product's own sign / keygen kernels -------
    base = rank * n                                          # each rank owns a distinct shard
    msgs = [D("bn254/msg2", base + i) for i in range(n)]
    pool = min(KEY_POOL, n)
    sks = [sk_bytes(j) for j in range(pool)]
    pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), pool, reduce_scalar=True)
    assert st == bytes(pool)
    good, st = eng.batch_sign(msgs, b"".join(sks[(base + i) % pool] for i in range(n)))
    assert st == bytes(n)
    pks = b"".join(pk_pool[128 * ((base + i) % pool):128 * ((base + i) % pool) + 128] for i in range(n))

    def corrupted(phase):
        """signatures with every item at residue `phase` replaced by its neighbour's (valid point, wrong message)"""
        sigs = bytearray(good)
        for i in range(phase, n, CORRUPT_EVERY):
            j = i - 1 if i else i + 1
            sigs[64 * i:64 * i + 64] = good[64 * j:64 * j + 64]
        return bytes(sigs)

    def expected_for(r, step, length=n):
        e = bytearray(length)
        for i in range(corrupt_phase(r, step), length, CORRUPT_EVERY):
            e[i] = 9
        return bytes(e)

    with torch.cuda.stream(R.stream):
        def to_dev(b):
            return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        d_msgs = to_dev(b"".join(msgs))
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        sig_sets = [corrupted(corrupt_phase(rank, 0)), corrupted(corrupt_phase(rank, 1))]
        d_sigs = [to_dev(sig_sets[0]), to_dev(sig_sets[1])]
        d_pks = to_dev(pks)
        d_status = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_all = torch.zeros(n * world, dtype=torch.uint8, device=dev) if R.dist_on else None
    sh = R.stream.cuda_stream
    assert sh != 0

    def step(k):
        eng.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs[k & 1].data_ptr(), d_pks.data_ptr(), n, d_status.data_ptr(),
                                flags=0, stream=sh)
        if R.dist_on:
            R.gather(d_status, d_all)                        # the only collective

    checks = {"steps_checked": 0, "mismatches": 0}

    def check(step_index):
        """device statuses (own shard, and every shard of the gathered vector) == the pattern of that step"""
        torch.cuda.synchronize()
        got = bytes(d_status.cpu().numpy())
        bad = int(got != expected_for(rank, step_index))
        if R.dist_on:
            allst = bytes(d_all.cpu().numpy())
            for r in range(world):
                bad += int(allst[r * n:(r + 1) * n] != expected_for(r, step_index))
        checks["steps_checked"] += 1
        checks["mismatches"] += bad
        return bad == 0

    eng.set_profiling(True)
    kernel_ms = {"decode": 0.0, "hash_to_g1": 0.0, "miller_loop": 0.0, "final_exp": 0.0}
    pair = not (args.no_pair_lanes or args.split_miller)
    # the clock the chip sustains DURING the timed steps: the lane-pair kernels accumulate, per workgroup, shader-clock cycles and
    # constant-rate ticks between entry and exit (two scalar clock reads per workgroup); read and cleared after the warm-up, read again
    # after the last timed step (include/bn254_hip.h: BN254_OPT_CLOCK_PROBE, bn254_ctx_last_clocks)
    # Default (--clock-probe after): the timed steps run WITHOUT the probe — the headline binary path is stamp-free — and the clock is taken
    # over four more steps right after the timed region.  --clock-probe timed keeps the probe on inside the timed steps (the round-5 form);
    # the same-box A/B of the two is profiles/r06_*_ab_clock_probe.jsonl.
    clock_probe = False
    clock_capable = pair and n > 16384
    if clock_capable and args.clock_probe == "timed":
        try:
            eng.set_option(bn254_amd.engine.OPT_CLOCK_PROBE, 1)
            clock_probe = True
        except Exception:
            clock_probe = False

    def after_warmup():
        # parity gate before any timing is accepted
        assert args.warmup == 0 or check(args.warmup - 1), "GPU status bytes differ from the expected pattern"
        if clock_probe:
            eng.last_clocks()                                # clears the counters: what is read next belongs to the timed steps alone

    def per_step():
        ms = eng.last_kernel_ms()                            # HIP events on the launch stream (synchronises it)
        for key in kernel_ms:
            kernel_ms[key] += ms[key]

    elapsed = R.time_steps(step, args.steps, args.warmup, after_warmup, per_step)
    timed_clocks = eng.last_clocks() if clock_probe else None   # accumulated over the K timed steps, nothing else
    ok_last = check(args.warmup + args.steps - 1)            # the LAST step's pattern (differs from the one before)
    assert ok_last, "GPU status bytes of the last timed step differ from the expected pattern"
    clock_steps_after = 0
    if clock_capable and not clock_probe and args.clock_probe == "after":
        try:                                                 # four more steps, back to back, OUTSIDE the timed region, with the probe on
            eng.set_option(bn254_amd.engine.OPT_CLOCK_PROBE, 1)
            eng.last_clocks()
            clock_steps_after = 4
            for k in range(clock_steps_after):
                step(args.warmup + args.steps + k)
            torch.cuda.synchronize()
            timed_clocks = eng.last_clocks()                 # (the probe stays on: the measurements below, all outside the timed region, use it)
        except Exception:
            timed_clocks = None

    detail = R.scaling_detail(args.steps, sum(kernel_ms.values()))      # a collective: every rank takes part
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
                               "(32-byte messages, 1/64 corrupted, pattern alternates per step and differs per rank), 2 pairings per verify",
                   "batch_per_gpu": n, "verifies_per_s": verifies_per_s,
                   "bit_exact_vs_expected": checks["mismatches"] == 0, "status_vectors_checked": checks["steps_checked"] * (1 + (world if R.dist_on else 0)),
                   "collective": "all_gather_into_tensor(status bytes) per step over %s" % R.backend if R.dist_on else None},
    }

    if detail is not None:
        result["scaling_detail"] = detail
    if rank == 0:
        k_avg = {k: v / args.steps for k, v in kernel_ms.items()}
        dom = max(("miller_loop", "final_exp"), key=lambda k: k_avg[k])
        kname = {("miller_loop", True): "k_miller_verify_pair", ("miller_loop", False): "k_miller_verify",
                 ("final_exp", True): "k_final_exp_pair", ("final_exp", False): "k_final_exp"}[(dom, pair)]
        fp_mul = FP_MUL_MILLER if dom == "miller_loop" else FP_MUL_FINAL_EXP
        mac_per_launch = fp_mul * MAC32_PER_FP_MUL * n
        achieved = mac_per_launch / (k_avg[dom] * 1e-3) / 1e12
        io_bytes = BYTES_PER_VERIFY_IO * n
        probe = issue_probe(eng) if world == 1 else None
        # The clock the chip actually sustains under this load (it is power-limited below its nominal 2.4 GHz): a few more steps,
        # OUTSIDE the timed region, with the kernels' clock probe on — shader-clock cycles over constant-rate ticks between entry and
        # exit of every workgroup of the Miller kernel / the final exponentiation; the issue probe reports its own.
        sclk = None
        if timed_clocks is not None:
            try:
                sclk = {"miller_loop": round(timed_clocks["miller_loop"], 1), "final_exp": round(timed_clocks["final_exp"], 1)}
                if world == 1:
                    eng.probe_issue_rate(0, 2)
                    sclk["issue_probe"] = round(eng.last_clocks()["issue_probe"], 1)
                sclk["nominal"] = 2400.0
                sclk["method"] = ("sum over workgroups and launches of s_memtime cycles / s_memrealtime ticks x hipDeviceAttributeWallClockRate, accumulated "
                                  + ("over the %d TIMED steps themselves (BN254_OPT_CLOCK_PROBE on from before the warm-up; counters cleared after it)" % args.steps
                                     if clock_probe else
                                     "over %d steps run back to back right AFTER the timed region (the timed steps themselves run without the probe: "
                                     "--clock-probe after, the default)" % clock_steps_after))
                sclk["probe_inside_timed_region"] = bool(clock_probe)
            except Exception as exc:                               # never lose the bench line over the extra
                sclk = {"error": repr(exc)}
        leaf_floor = None
        lane_pair_kernels = pair and n > 16384                # batches of up to 16 384 take the small-batch kernels (BN254_OPT_TRIO_MAX_BATCH): other code
        if lane_pair_kernels and world == 1:
            try:                                                   # the product leaves of the two kernels alone, same launch shape (include/bn254_hip.h)
                from bn254_amd.engine import OPT_CLOCK_PROBE
                eng.set_option(OPT_CLOCK_PROBE, 1)
                leaf_floor = {"note": "kernels that run ONLY the product calls of a verify's Miller loop / final exponentiation (no tower additions, "
                                      "carries, twist point or LDS traffic): what the kernels would take if everything around their product leaves "
                                      "were free.  Round 5: the r04 probe's own loop (accumulator by reference in a real function) cost 20-25 % on top "
                                      "of the leaves it was meant to isolate — the floors below come from the inlined loops and are that much lower"}
                for mode, key, kernel_key, probe_counts, real_counts in ((0, "miller_loop", "miller_loop", (3219, 435, 348), (3194, 430, 348)),
                                                                         (1, "final_exp", "final_exp", (945, 1701, 0), (975, 1714, 0))):
                    variants = {}
                    # the same product calls in three loops: r04's probe (a real function taking the accumulator by reference, run-time trip
                    # counts, every product waiting for its predecessor), and one inlined loop with compile-time counts carrying one chain or
                    # four independent ones.  A floor must not lose to what it bounds: the figure quoted is the smallest.
                    for name, md in (("loop_in_a_real_function_ms", mode), ("one_chain_inlined_ms", 6 if mode == 0 else None), ("four_chains_inlined_ms", mode + 4)):
                        if md is None:
                            continue
                        eng.last_clocks()
                        variants[name] = {"ms": eng.probe_leaf_floor(n, md), "sclk_mhz": round(eng.last_clocks()["issue_probe"], 1)}
                    best = min(variants, key=lambda kk: variants[kk]["ms"])
                    floor_ms = variants[best]["ms"]
                    instr = lambda c: c[0] * 345 + c[1] * 258 + c[2] * 226          # noqa: E731
                    scaled = floor_ms * instr(real_counts) / instr(probe_counts)
                    leaf_floor[key] = {"probe_ms": floor_ms, "probe_variant": best, "variants": variants,
                                       "independent_chains_ms": variants["four_chains_inlined_ms"]["ms"],
                                       "probe_products_dual_sqr_scale": probe_counts, "kernel_products_dual_sqr_scale": real_counts,
                                       "ms_scaled_to_the_kernels_product_counts": scaled, "kernel_ms": k_avg[kernel_key],
                                       "floor_share_of_kernel": scaled / k_avg[kernel_key] if k_avg[kernel_key] else None,
                                       "probe_sclk_mhz": variants[best]["sclk_mhz"]}
                eng.set_option(OPT_CLOCK_PROBE, 0)
            except Exception as exc:
                leaf_floor = {"error": repr(exc)}
        fe_split = None
        if lane_pair_kernels and world == 1:
            try:
                # the final exponentiation BY ROUTINE, measured: the kernel's own interpreter on programs of one operation kind each
                # (include/bn254_hip.h: bn254_probe_fe_program), x the number of times the verify program (C_FE_CHECK) runs that operation
                LOAD, STORE, CSQR, MUL, CONJ, FROB, INV = 1, 2, 3, 4, 5, 6, 7
                counts = {"CSQR": 189, "MUL": 51, "CONJ": 50, "FROB": 4, "INV": 1, "STORE": 23, "LOAD": 7}      # gen_constants.py: C_FE_CHECK
                base = eng.probe_fe_program(n, [(STORE, 0)])                                                      # launch + load / store of f + one STORE
                reps = {"CSQR": 96, "MUL": 48, "CONJ": 96, "FROB": 12, "INV": 2, "STORE": 48, "LOAD": 48}
                prog = {"CSQR": [(STORE, 0)] + [(CSQR, 0)] * reps["CSQR"], "MUL": [(STORE, 0)] + [(MUL, 0)] * reps["MUL"],
                        "CONJ": [(STORE, 0)] + [(CONJ, 0)] * reps["CONJ"], "FROB": [(STORE, 0)] + [(FROB, 1 + (k % 3)) for k in range(reps["FROB"])],
                        "INV": [(STORE, 0)] + [(INV, 0)] * reps["INV"], "STORE": [(STORE, k % 10) for k in range(reps["STORE"] + 1)],
                        "LOAD": [(STORE, 0)] + [(LOAD, 0)] * reps["LOAD"]}
                per_op, total = {}, 0.0
                for name in counts:
                    t_op = max(0.0, eng.probe_fe_program(n, prog[name]) - base) / reps[name]
                    per_op[name] = {"ms_per_op": t_op, "ops_per_verify": counts[name], "ms": t_op * counts[name]}
                    total += t_op * counts[name]
                fe_split = {"per_routine": per_op, "sum_ms": total, "kernel_ms": k_avg["final_exp"], "launch_and_io_ms": base,
                            "note": "batch of %d lane pairs, two waves per SIMD; values are not meaningful (the routines are timed, not checked)" % n}
                # the same operations priced at their product leaves alone: dual product d and squaring s from the two leaf-floor kernels
                # (3219 d + 741 s = Miller floor, 945 d + 1701 s = final-exponentiation floor; a scaling counts as 0.88 squarings)
                try:
                    fm, ff = leaf_floor["miller_loop"]["probe_ms"], leaf_floor["final_exp"]["probe_ms"]
                    sq = (3219.0 * ff / 945.0 - fm) / (3219.0 * 1701.0 / 945.0 - 741.0)
                    du = (ff - 1701.0 * sq) / 945.0
                    fe_split["leaf_only_ms_per_op"] = {"dual_product": du, "squaring": sq, "CSQR_9_squarings": 9 * sq, "MUL_18_dual_products": 18 * du}
                    fe_split["non_leaf_share"] = {"CSQR": 1.0 - 9 * sq / per_op["CSQR"]["ms_per_op"], "MUL": 1.0 - 18 * du / per_op["MUL"]["ms_per_op"]}
                except Exception:
                    pass
            except Exception as exc:
                fe_split = {"error": repr(exc)}
        lane_products = lane_product_counts().get(kname)
        traffic = measured_traffic(kname)
        result["roofline"] = {
            "bound": "valu",                       # integer multiply issue (v_mad_u64_u32); not HBM, not MFMA
            "kernel": kname,
            "layout": "one verify per lane pair (Fq2 coefficients in adjacent lanes), two waves per SIMD" if pair else "one verify per lane",
            "achieved": achieved, "peak": PEAK_MAC32_THEORETICAL / 1e12, "unit": "TMAC32/s",
            "frac": achieved / (PEAK_MAC32_THEORETICAL / 1e12),
            "effective_sclk_mhz": sclk,
            "product_leaf_floor": leaf_floor,
            "final_exp_split": fe_split,
            "frac_at_effective_sclk": (achieved / (PEAK_MAC32_THEORETICAL / 1e12 * sclk[dom] / 2400.0)) if sclk and sclk.get(dom) else None,
            "peak_measured_in_this_run": probe["peak_mac32_measured"] / 1e12 if probe else None,
            "frac_of_measured_peak": achieved / (probe["peak_mac32_measured"] / 1e12) if probe else None,
            "issue_probe": probe,
            "traffic": (traffic or {}).get("bytes_per_launch"),   # HBM bytes per launch (PMC), private-segment traffic
            "traffic_over_algorithmic": ((traffic or {}).get("bytes_per_launch") / io_bytes) if (traffic or {}).get("bytes_per_launch") else None,
            "traffic_detail": traffic,
            "valu_issue": measured_valu_issue(kname, lane_products, probe, k_avg[dom] * 1e-3),
            "kernel_ms": k_avg,
            "mac32_per_verify": {"miller_loop": FP_MUL_MILLER * MAC32_PER_FP_MUL, "final_exp": FP_MUL_FINAL_EXP * MAC32_PER_FP_MUL,
                                 "hash_to_g1_mean": (FP_MUL_HASH_FILTER * 2.12 + FP_MUL_HASH_FINISH) * MAC32_PER_FP_MUL, "decode": FP_MUL_DECODE * MAC32_PER_FP_MUL},
            "hbm": {"algorithmic_bytes_per_step": io_bytes, "achieved_GBps": io_bytes / (1e-3 * 1e3 * elapsed / args.steps) / 1e9,
                    "peak_GBps": HBM_PEAK_GBPS, "note": "evidence that the path is not memory-bound"},
        }
        # north_star's "VALU-busy counters against gfx950 peak": rocprofv3's VALUBusy of the two hot kernels, at the top level of the roofline object
        vi = result["roofline"].get("valu_issue") or {}
        result["roofline"]["valu_busy_pct"] = {kname: vi.get("valu_busy_pct_rocprof_definition"), "source": "profiles/pmc_latest.json (see pmc_as_of)"}
        # the OTHER hot kernel of the step priced the same way (the dominant one is `roofline` itself): its own algorithmic MAC32 per launch
        # over its own HIP-event duration in this run, its own PMC traffic against the algorithmic bytes
        try:
            other = "final_exp" if dom == "miller_loop" else "miller_loop"
            okname = {("miller_loop", True): "k_miller_verify_pair", ("miller_loop", False): "k_miller_verify",
                      ("final_exp", True): "k_final_exp_pair", ("final_exp", False): "k_final_exp"}[(other, pair)]
            ofp = FP_MUL_MILLER if other == "miller_loop" else FP_MUL_FINAL_EXP
            oach = ofp * MAC32_PER_FP_MUL * n / (k_avg[other] * 1e-3) / 1e12
            otr = measured_traffic(okname)
            result["roofline"]["second_kernel"] = {
                "kernel": okname, "achieved": oach, "peak": PEAK_MAC32_THEORETICAL / 1e12, "unit": "TMAC32/s", "frac": oach / (PEAK_MAC32_THEORETICAL / 1e12),
                "frac_at_effective_sclk": (oach / (PEAK_MAC32_THEORETICAL / 1e12 * sclk[other] / 2400.0)) if sclk and sclk.get(other) else None,
                "kernel_ms": k_avg[other], "traffic": (otr or {}).get("bytes_per_launch"),
                "traffic_over_algorithmic": ((otr or {}).get("bytes_per_launch") / (432.0 * n)) if (otr or {}).get("bytes_per_launch") else None,
                "algorithmic_bytes_per_launch": 432.0 * n,
                "valu_busy_pct": (lambda k_: (100.0 * k_["SQ_ACTIVE_INST_VALU"] / 256.0 / (k_["GRBM_GUI_ACTIVE"] / 8.0)) if k_ and "GRBM_GUI_ACTIVE" in k_ and "SQ_ACTIVE_INST_VALU" in k_ else None)(_pmc(okname)),
                "note": "algorithmic bytes of this kernel alone: one Fq12 Miller value per verify (12 x 9 words) between the two kernels"}
            result["roofline"]["valu_busy_pct"][okname] = result["roofline"]["second_kernel"]["valu_busy_pct"]
        except Exception as exc:
            result["roofline"]["second_kernel"] = {"error": repr(exc)}
        if world == 1 and n >= 1024:
            # informational, outside the timed region: latency of a SMALL call through the host-pointer entry point (H2D, kernels,
            # D2H, sync) — batches of up to 16 384 verifies take the small-batch kernels
            try:
                lat = {}
                last = args.warmup + args.steps - 1
                for m in (1, 1024):
                    best = None
                    for _ in range(5):
                        t1 = time.perf_counter()
                        got = eng.batch_verify(msgs[:m], sig_sets[last & 1][:64 * m], pks[:128 * m], flags=0)
                        dt = time.perf_counter() - t1
                        best = dt if best is None or dt < best else best
                    assert got == expected_for(0, last, m), "small-batch statuses differ from the expected pattern"
                    lat["verifies_%d_ms" % m] = 1e3 * best
                # the same single verify on round 4's small-batch kernels (eight wave roles, nine lane pairs) — what the lane machine and the
                # eighteen-pair final exponentiation are measured against
                import re
                from bn254_amd.engine import OPT_LM_MAX_BATCH, OPT_NONET_WIDE
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_amd", "csrc", "bn254_ws.h")) as fh:
                    lm_default = int(re.search(r"#define\s+LM_MAX_BATCH_DEFAULT\s+(\d+)", fh.read()).group(1))
                eng.set_option(OPT_LM_MAX_BATCH, 0); eng.set_option(OPT_NONET_WIDE, 0)
                try:
                    best = None
                    for _ in range(5):
                        t1 = time.perf_counter()
                        got = eng.batch_verify(msgs[:1], sig_sets[last & 1][:64], pks[:128], flags=0)
                        dt = time.perf_counter() - t1
                        best = dt if best is None or dt < best else best
                    assert got == expected_for(0, last, 1), "small-batch statuses differ from the expected pattern"
                    lat["verifies_1_ms_wave_roles_and_nine_pairs"] = 1e3 * best
                finally:
                    eng.set_option(OPT_LM_MAX_BATCH, lm_default); eng.set_option(OPT_NONET_WIDE, 1)
                result["small_batch_latency"] = lat
            except AssertionError:
                raise
            except Exception as exc:                             # never lose the bench line over the extra
                result["small_batch_latency"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            last = args.warmup + args.steps - 1
            result["cpu_baseline"] = cpu_baseline_verify(msgs, sig_sets[last & 1], pks, expected_for(0, last))
        result["pmc_as_of"] = pmc_as_of()
        print(json.dumps(result))
    R.finish()


def run_verify_mgpu(args):
    """configs[1] on N GPUs from ONE process through the C ABI's multi-GPU layer (include/bn254_hip.h: bn254_mgpu_*): one context,
    stream and parked worker thread per device, shard g resident on device g, the only exchange the in-place gather of the status
    bytes (ncclAllGather on ncclCommInitAll communicators — RCCL's C API, no torch.distributed; peer copies when --mgpu-devices
    lists a device twice, which is how a one-GPU box rehearses it).  Same metric, same per-GPU batch, same checks as `--workload
    verify`; torch is used only to hold the device buffers."""
    import torch
    import bn254_amd
    from bn254_amd.engine import MGPU_OPT_GATHER, MGPU_OPT_TIMING
    from tests.datagen import KEY_POOL, sk_bytes
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload verify-mgpu is ONE process driving all devices: start it without torchrun")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (bn254_amd has no CPU fallback)")
    devices = [int(x) for x in args.mgpu_devices.split(",")] if args.mgpu_devices else list(range(args.gpus))
    G = len(devices)
    n = args.batch or BATCH
    N = n * G
    mg = bn254_amd.MultiEngine(devices)
    gather_mode = {"auto": 0, "rccl": 1, "copy": 2}[args.mgpu_gather]
    if gather_mode:
        mg.set_option(MGPU_OPT_GATHER, gather_mode)
    uses_rccl = gather_mode == 1 or (gather_mode == 0 and len(set(devices)) == G and G > 1)
    mg.reserve(2 * N, init_collectives=True)
    mg.set_option(MGPU_OPT_TIMING, 1)
    mg.engine(0).set_profiling(True)                     # entry 0's kernels by HIP events on its stream: the roofline of the line

    def corrupted(good, phase):
        sigs = bytearray(good)
        for i in range(phase, n, CORRUPT_EVERY):
            j = i - 1 if i else i + 1
            sigs[64 * i:64 * i + 64] = good[64 * j:64 * j + 64]
        return bytes(sigs)

    def expected_for(g, step, length=n):
        e = bytearray(length)
        for i in range(corrupt_phase(g, step), length, CORRUPT_EVERY):
            e[i] = 9
        return bytes(e)

    keep, host = [], []
    for g, d in enumerate(devices):
        eng = mg.engine(g)
        dev = torch.device("cuda", d)
        base = g * n                                         # every entry owns a distinct shard, as a rank does in run_verify
        msgs = [D("bn254/msg2", base + i) for i in range(n)]
        pool = min(KEY_POOL, n)
        sks = [sk_bytes(j) for j in range(pool)]
        pk_pool, st = eng.batch_g2_mul(None, b"".join(sks), pool, reduce_scalar=True)
        assert st == bytes(pool)
        good, st = eng.batch_sign(msgs, b"".join(sks[(base + i) % pool] for i in range(n)))
        assert st == bytes(n)
        pks = b"".join(pk_pool[128 * ((base + i) % pool):128 * ((base + i) % pool) + 128] for i in range(n))
        sig_sets = [corrupted(good, corrupt_phase(g, 0)), corrupted(good, corrupt_phase(g, 1))]

        def to_dev(b, dev=dev):
            return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        keep.append({"msgs": to_dev(b"".join(msgs)), "off": torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev),
                     "sigs": [to_dev(sig_sets[0]), to_dev(sig_sets[1])], "pks": to_dev(pks),
                     "all": torch.zeros(mg.gathered_len(N), dtype=torch.uint8, device=dev)})
        if g == 0:
            host = [msgs, sig_sets, pks]
    ptr = lambda key: [k[key].data_ptr() for k in keep]      # noqa: E731
    p_msgs, p_off, p_pks, p_all = ptr("msgs"), ptr("off"), ptr("pks"), ptr("all")
    p_sigs = [[k["sigs"][b].data_ptr() for k in keep] for b in (0, 1)]

    def sync_all():
        mg.synchronize()
        for d in set(devices):
            torch.cuda.synchronize(d)

    def step(k):
        mg.batch_verify_device(p_msgs, p_off, p_sigs[k & 1], p_pks, N, p_all, flags=0)

    checks = {"steps_checked": 0, "mismatches": 0}

    def check(step_index):
        sync_all()
        bad = 0
        for h in range(G):                                   # EVERY device's gathered buffer, every shard of it
            allst = bytes(keep[h]["all"].cpu().numpy())
            for g in range(G):
                bad += int(allst[g * n:(g + 1) * n] != expected_for(g, step_index))
        checks["steps_checked"] += 1
        checks["mismatches"] += bad
        return bad == 0

    for k in range(args.warmup):
        step(k)
    assert args.warmup == 0 or check(args.warmup - 1), "GPU status bytes differ from the expected pattern"
    comp_ms, coll_ms = [0.0] * G, [0.0] * G
    kernel_ms = {"decode": 0.0, "hash_to_g1": 0.0, "miller_loop": 0.0, "final_exp": 0.0}
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
        a, b = mg.last_timing()                              # HIP events on every device's stream (synchronises them), as run_verify's per_step
        comp_ms = [x + y for x, y in zip(comp_ms, a)]
        coll_ms = [x + y for x, y in zip(coll_ms, b)]
        ms0 = mg.engine(0).last_kernel_ms()                  # entry 0's own kernels (HIP events on the stream they were launched on)
        for key in kernel_ms:
            kernel_ms[key] += ms0[key]
    sync_all()
    elapsed = time.perf_counter() - t0
    assert check(args.warmup + args.steps - 1), "GPU status bytes of the last timed step differ from the expected pattern"

    verifies_per_s = N * args.steps / elapsed
    result = {
        "metric": "BN254 pairings/sec (batch verify)", "value": 2.0 * verifies_per_s, "unit": "pairings/s", "n_gpus": G, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "configs[1] through bn254_mgpu_batch_verify_device: ONE process, %d device entries %s, 65536-verify shard "
                               "resident on each (32-byte messages, 1/64 corrupted, pattern alternates per step and differs per shard), "
                               "2 pairings per verify" % (G, devices),
                   "batch_per_gpu": n, "verifies_per_s": verifies_per_s, "bit_exact_vs_expected": checks["mismatches"] == 0,
                   "status_vectors_checked": checks["steps_checked"] * G * G,
                   "collective": ("ncclAllGather (RCCL C API, ncclCommInitAll, in place) of %d status bytes per device and step" % n)
                   if uses_rccl else ("peer copies (--mgpu-gather %s; a one-device handle has nothing to gather, and RCCL refuses two ranks "
                                      "on one device)" % args.mgpu_gather)},
        "scaling_detail": {"compute_ms_per_step": {"per_device": [x / args.steps for x in comp_ms]},
                           "collective_ms_per_step": {"per_device": [x / args.steps for x in coll_ms]},
                           "timed_by": "HIP events on each device's stream: before its shard's first kernel, after its last, after the gather",
                           "note": "one process: ms_per_step is the host clock over K steps between synchronisations of every device"},
    }
    # roofline of the dominant kernel — per DEVICE (entry 0's launch: its shard of n verifies), HIP events in this run; cpu_baseline on shard 0
    k_avg = {k: v / args.steps for k, v in kernel_ms.items()}
    small = n <= 16384                                       # such shards take the small-batch kernels (other code, other counters)
    dom = max(("miller_loop", "final_exp"), key=lambda k: k_avg[k])
    kname = {"miller_loop": "k_miller_verify_pair", "final_exp": "k_final_exp_pair"}[dom] if not small else {"miller_loop": "small-batch Miller kernel", "final_exp": "small-batch final exponentiation"}[dom]
    fp_mul = FP_MUL_MILLER if dom == "miller_loop" else FP_MUL_FINAL_EXP
    if k_avg[dom] > 0:
        result["roofline"] = kernel_roofline(kname, fp_mul * n, k_avg[dom],
                                             note="per device: entry 0's launch over its shard of %d verifies, HIP events on its stream; %d device "
                                                  "entries run such a launch side by side (on distinct GPUs each at this rate; entries that share a GPU "
                                                  "share its VALUs and this figure with it)" % (n, G))
        result["roofline"]["kernel_ms"] = k_avg
        if result["roofline"].get("traffic"):
            result["roofline"]["traffic_over_algorithmic"] = result["roofline"]["traffic"] / (BYTES_PER_VERIFY_IO * n)
    if not args.no_cpu_baseline:
        last = args.warmup + args.steps - 1
        result["cpu_baseline"] = cpu_baseline_verify(host[0], host[1][last & 1], host[2], expected_for(0, last))
    # what the layer costs: the same shard through the single-GPU entry point on device entry 0 alone, same process, same steps
    eng0 = mg.engine(0)
    torch.cuda.synchronize(devices[0])
    t1 = time.perf_counter()
    for k in range(args.steps):
        eng0.batch_verify_device(p_msgs[0], p_off[0], p_sigs[k & 1][0], p_pks[0], n, p_all[0], flags=0)
        eng0.synchronize()
    direct = time.perf_counter() - t1
    result["single_gpu_direct"] = {"ms_per_step": 1e3 * direct / args.steps, "pairings_per_s": 2.0 * n * args.steps / direct,
                                   "mgpu_over_direct_ms": (1e3 * elapsed / args.steps) / (1e3 * direct / args.steps),
                                   "note": "bn254_batch_verify_device on entry 0's context alone, synchronised per step like the timed loop above; "
                                           "with one device entry the ratio is the overhead of the layer"}
    # the host-pointer form (whole batch in pageable memory, statuses straight into the caller's slices): PCIe-inclusive, informational
    try:
        msgs0, sig_sets0, pks0 = host
        reps = 3
        hm, hs, hp = msgs0 * G, sig_sets0[0] * G, pks0 * G            # G copies of shard 0's bytes: every shard sees the same tuples
        mg.batch_verify(hm, hs, hp)
        t1 = time.perf_counter()
        for _ in range(reps):
            got = mg.batch_verify(hm, hs, hp)
        dt = time.perf_counter() - t1
        assert got == expected_for(0, 0) * G, "host-pointer statuses differ from the expected pattern"
        result["host_pointers"] = {"pairings_per_s": 2.0 * N * reps / dt, "ms_per_call": 1e3 * dt / reps,
                                   "note": "bn254_mgpu_batch_verify incl. packing in Python; PCIe-inclusive, never `value`"}
    except AssertionError:
        raise
    except Exception as exc:
        result["host_pointers"] = {"error": repr(exc)}
    result["pmc_as_of"] = pmc_as_of()
    print(json.dumps(result))
    mg.close()


def lane_product_counts():
    """per-lane product counts of the pair kernels (dual-accumulated products / single products incl. squares) from
    the host instrumentation, committed by tests/test_workcount.py as profiles/lane_product_counts.json"""
    try:
        with open(os.path.join(ROOT, "profiles", "lane_product_counts.json")) as f:
            return json.load(f)
    except Exception:
        return {}


def pairing_indices(np, base, n, pool):
    """(P index, Q index) of global item g = base + i: period pool^2, so shards of different ranks differ"""
    g = np.arange(base, base + n, dtype=np.int64)
    return (g * 7 + 3) % pool, (g * 13 + 5 + (g // pool) * 29) % pool


def run_pairing(args, R):
    """BASELINE configs[3]: independent pairings e(P_i, Q_i) -> canonical Gt (384 B, stays in HBM) + status byte
    (Gt != 1), sharded over the ranks; per step: kernels, 64-bit additive Gt checksum, status all-gather, checksum
    all-reduce.  Unit = one Miller loop + one final exponentiation."""
    import numpy as np
    torch = R.torch
    import bn254_amd
    from bn254_amd.sharding import gt_checksum
    eng = bn254_amd.Engine(R.local_rank)
    n = args.batch or PAIRING_BATCH
    pool = min(4096, max(2, n))
    rank, world, dev = R.rank, R.world, R.dev
    sc = [hashlib.sha256(b"cfg4-%d" % j).digest() for j in range(2 * pool)]
    g1 = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    P, st1 = eng.batch_g1_mul(g1 * pool, b"".join(sc[:pool]), pool, reduce_scalar=True)
    Qs, st2 = eng.batch_g2_mul(None, b"".join(sc[pool:]), pool, reduce_scalar=True)
    assert st1 == bytes(pool) and st2 == bytes(pool)
    Pn = np.frombuffer(P, dtype=np.uint8).reshape(pool, 64)
    Qn = np.frombuffer(Qs, dtype=np.uint8).reshape(pool, 128)
    pi, qi = pairing_indices(np, rank * n, n, pool)
    with torch.cuda.stream(R.stream):
        d_g1 = torch.from_numpy(Pn[pi].reshape(-1).copy()).to(dev)
        d_g2 = torch.from_numpy(Qn[qi].reshape(-1).copy()).to(dev)
        d_gt = torch.empty(n * 384, dtype=torch.uint8, device=dev)
        d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_all = torch.zeros(n * world, dtype=torch.uint8, device=dev) if R.dist_on else None
    eng.reserve(n)
    sh = R.stream.cuda_stream
    state = {"sum": None, "sums": set()}

    def step(k):
        eng.batch_pairing_device(d_g1.data_ptr(), d_g2.data_ptr(), n, 1, d_gt.data_ptr(), d_st.data_ptr(), stream=sh)
        local = gt_checksum(d_gt)
        if R.dist_on:
            R.gather(d_st, d_all)
        state["sum"] = R.allreduce_checksum(local)           # 8-byte all-reduce (a host read: ends the step)
        state["sums"].add(state["sum"])

    def after_warmup():
        torch.cuda.synchronize()
        assert int(d_st.min()) == 9 and int(d_st.max()) == 9, "a pairing of two non-identity points came out as one"
        if R.dist_on:
            assert int(d_all.min()) == 9 and int(d_all.max()) == 9

    elapsed = R.time_steps(step, args.steps, args.warmup, after_warmup)
    torch.cuda.synchronize()
    # size-independent properties on the full shard: items with equal (P, Q) have equal Gt bytes; the checksum is
    # the same in every step (same inputs); statuses all "not one"
    gt = d_gt.view(n, 384)
    key = torch.from_numpy(pi * pool + qi).to(dev)
    order = torch.argsort(key)
    same = key[order][1:] == key[order][:-1]
    dup_ok = bool((gt[order][1:][same] == gt[order][:-1][same]).all()) if bool(same.any()) else True
    assert dup_ok, "items with identical inputs produced different Gt bytes"
    assert len(state["sums"]) == 1, "Gt checksum changed between steps"
    assert int(d_st.min()) == 9 and int(d_st.max()) == 9
    detail = R.scaling_detail(args.steps)
    total = n * world * args.steps
    result = {
        "metric": "BN254 pairings/sec (independent pairings, canonical Gt out)",
        "value": total / elapsed, "unit": "pairings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "configs[3]: independent pairings e(P_i,Q_i) sharded over the GPUs, %d per GPU (pool of %d P x %d Q combined by "
                               "index), 1 Miller loop + 1 final exponentiation each, Gt stays in HBM" % (n, pool, pool),
                   "batch_per_gpu": n, "gt_checksum_u64": "%016x" % state["sum"], "duplicate_inputs_equal_gt": dup_ok,
                   "collective": ("all_gather_into_tensor(status bytes) + all_reduce(sum, 8-byte Gt checksum) per step over %s" % R.backend) if R.dist_on else None},
    }
    if detail is not None:
        result["scaling_detail"] = detail
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        fp_mul = FP_MUL_MILLER_SINGLE + FP_MUL_FINAL_EXP_EXACT
        achieved = fp_mul * MAC32_PER_FP_MUL * n / (ms * 1e-3) / 1e12
        tr_m, tr_f = measured_traffic("k_miller_var_pair"), measured_traffic("k_final_exp_pair")
        result["roofline"] = {"bound": "valu", "kernel": "k_miller_var_pair + k_final_exp_pair (whole step)", "achieved": achieved,
                              "peak": PEAK_MAC32_THEORETICAL / 1e12, "unit": "TMAC32/s", "frac": achieved / (PEAK_MAC32_THEORETICAL / 1e12),
                              "traffic": (tr_m["bytes_per_launch"] + tr_f["bytes_per_launch"]) if tr_m and tr_f else None,
                              "traffic_detail": {"k_miller_var_pair": tr_m, "k_final_exp_pair": tr_f},
                              "hbm": {"algorithmic_bytes_per_step": BYTES_PER_PAIRING_IO * n, "achieved_GBps": BYTES_PER_PAIRING_IO * n / (ms * 1e-3) / 1e9,
                                      "peak_GBps": HBM_PEAK_GBPS}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c_oracle
            cores = effective_cores()
            sample = min(n, 65536)                           # >= 64 Ki Gt values per GPU against the oracle (~4 s on 16 cores)
            t1 = time.perf_counter()
            gt_cpu, st_cpu = c_oracle.batch_pairing(Pn[pi[:sample]].tobytes(), Qn[qi[:sample]].tobytes(), sample, 1, nthreads=cores)
            dt = time.perf_counter() - t1
            same_bytes = gt_cpu == gt[:sample].cpu().numpy().tobytes() and st_cpu == bytes([9]) * sample
            assert same_bytes, "oracle Gt bytes differ from the GPU's"
            one = min(sample, 128)                            # the same oracle on ONE thread (the multi-thread figure depends on the box's cgroup quota)
            t1 = time.perf_counter()
            c_oracle.batch_pairing(Pn[pi[:one]].tobytes(), Qn[qi[:one]].tobytes(), one, 1, nthreads=1)
            dt_one = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": sample / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
                                      "sample": "first %d pairings of the shard, oracle/bn254_oracle.c; canonical Gt bytes equal the GPU's" % sample,
                                      "single_thread_value": one / dt_one}
        result["pmc_as_of"] = pmc_as_of()
        print(json.dumps(result))
    R.finish()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="items per GPU per step (default: the size BASELINE.json names for the workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pair-lanes", action="store_true", help="one lane per verify instead of lane pairs (A/B)")
    ap.add_argument("--split-miller", action="store_true", help="one pairing per lane instead of the fused 2-pair Miller loop (A/B)")
    ap.add_argument("--workload", default="verify", choices=["verify", "verify-mgpu", "pairing", "verify-host", "verify-keyed", "verify-keyed-randomized", "verify-compressed", "verify-randomized", "hash", "aggregate"],
                    help="verify = the headline (configs[1]) and pairing = configs[3]: both run on N ranks; the others time configs 2, 4 or "
                         "other entry points on one GPU (informational — see DESIGN.md §4b)")
    ap.add_argument("--mgpu-devices", default="", help="--workload verify-mgpu: comma-separated HIP device list (default 0..gpus-1); a device "
                                                       "may be listed more than once (one-GPU rehearsal, gather by peer copies)")
    ap.add_argument("--agg-main-only", action="store_true", help="--workload aggregate: run only the default route (no side runs with options changed): "
                                                                 "what the counter passes profile, so that per-launch figures of k_aggregate_pair are that route's own")
    ap.add_argument("--clock-probe", default="after", choices=["after", "timed", "off"],
                    help="--workload verify: where the shader clock under load is measured — after = over four extra steps behind the timed region "
                         "(default: nothing rides in the timed kernels), timed = inside the timed steps (two scalar clock reads per workgroup), off")
    ap.add_argument("--mgpu-gather", default="auto", choices=["auto", "rccl", "copy"],
                    help="--workload verify-mgpu: auto = RCCL for two or more distinct devices, else peer copies; rccl with ONE device = the one-rank "
                         "rehearsal of the RCCL calls")
    args = ap.parse_args()
    global PMC_WORKLOAD
    PMC_WORKLOAD = "verify" if args.workload in ("verify-host", "verify-compressed") else args.workload   # same kernels, same batch as the headline command
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.workload == "verify-mgpu":                       # ONE process for all devices: no ranks, no torch.distributed
        PMC_WORKLOAD = "verify"
        return run_verify_mgpu(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.workload not in ("verify", "pairing"):
            raise SystemExit("--workload %s is a single-GPU measurement" % args.workload)
        sys.exit(launch_ranks(args, sys.argv[1:]))

    R = Rank(args)
    if args.workload == "verify":
        return run_verify(args, R)
    if args.workload == "pairing":
        return run_pairing(args, R)
    if R.world != 1:
        raise SystemExit("--workload %s is a single-GPU measurement" % args.workload)
    import bn254_amd
    eng = bn254_amd.Engine(R.local_rank)
    with R.torch.cuda.stream(R.stream):
        other_workloads(args, R.torch, eng, R.dev, R.stream)


if __name__ == "__main__":
    main()
