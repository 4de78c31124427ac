"""The N>1 path.
CPU (gloo, world size 2): the shard / gather / checksum functions bench.py's ranks use, with each rank's statuses
coming from the oracle on its own shard, and the launcher of `bench.py --gpus N` (the parent starts the ranks, relays
one JSON line, fails loudly when a rank fails — here every rank fails for want of a HIP device).
GPU (-m gpu): the exact command line `bench.py --gpus 2` for both sharded workloads, two ranks sharing the one GPU of
the box over gloo (BN254_BENCH_BACKEND / BN254_BENCH_SINGLE_DEVICE are test knobs), checked against the oracle."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bn254_amd.sharding import allreduce_checksum, failure_count, gather_status, gt_checksum, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 65536, 1000003):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                covered += list(range(lo, hi)) if n < 100 else [(lo, hi)]
            if n < 100:
                assert covered == list(range(n))
            else:
                assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json
    from oracle import c_oracle
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "derived_vectors.json")))["verify_cases"]
    n = len(cases)
    lo, hi = shard_range(n, rank, world)
    per = (n + world - 1) // world
    mine = cases[lo:hi]
    st, _ = c_oracle.batch_verify([bytes.fromhex(v["message_hex"]) for v in mine], b"".join(bytes.fromhex(v["sig"]) for v in mine),
                                  b"".join(bytes.fromhex(v["pk"]) for v in mine))
    local = torch.zeros(per, dtype=torch.uint8)
    local[: hi - lo] = torch.tensor(list(st), dtype=torch.uint8)
    padded = gather_status(local)                                   # rank-major, every shard padded to `per`
    allst = torch.cat([padded[r * per: r * per + (shard_range(n, r, world)[1] - shard_range(n, r, world)[0])] for r in range(world)])
    fails = failure_count(local[: hi - lo])
    # the 8-byte checksum all-reduce of the pairing workload: each rank sums the words of "its" bytes, wrapping mod 2^64
    blob = torch.frombuffer(bytearray(hashlib.sha256(b"gt-%d" % rank).digest() * 12), dtype=torch.uint8)   # 384 B
    total = allreduce_checksum(gt_checksum(blob) + torch.tensor([0x7FFFFFFFFFFFFF00 + rank], dtype=torch.int64))   # forces a wrap-around
    if rank == 0:
        q.put((bytes(allst.tolist()), fails, total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_gather_gloo(derived, world):
    """world size 2 and the full node's 8: contiguous shards, rank-major gather (the last shard is short: 8 does not divide
    the case count), failure count and wrapping checksum all-reduce"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, fails, total = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = bytes(v["status"] for v in derived["verify_cases"])
    assert got == want
    assert fails == sum(1 for s in want if s)
    ref = 0
    for r in range(world):
        b = hashlib.sha256(b"gt-%d" % r).digest() * 12
        ref += sum(int.from_bytes(b[i:i + 8], "little") for i in range(0, 384, 8)) + 0x7FFFFFFFFFFFFF00 + r
    assert total == ref % (1 << 64)


def _run_bench(extra, env_extra, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


@pytest.mark.timeout(600)
def test_bench_launcher_starts_ranks_and_fails_loudly_without_gpu():
    """`bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself; on this GPU-less host each rank stops with
    "needs a HIP device" (no CPU fallback), and the parent relays the failure: non-zero exit, no JSON line."""
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: covered by the gpu tests below")
    env = {"BN254_BENCH_BACKEND": "gloo", "BN254_BENCH_SINGLE_DEVICE": "1"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "128"],
                       env={**env_clean, **env}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540)
    assert p.returncode != 0
    assert '"metric"' not in p.stdout
    assert "needs a HIP device" in p.stderr
    # a mismatch between --gpus and an external launcher's world size is refused before any GPU call
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env={**env_clean, "WORLD_SIZE": "2", "RANK": "0"},
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_rank_selects_its_own_device_for_local_ranks_0_to_7(monkeypatch):
    """bench.py's Rank on an 8-GPU node, without the single-device test knob: LOCAL_RANK r -> torch.cuda.set_device(r), the
    RCCL process group is created with device_id cuda:r, and the rank's stream lives on that device (torch.cuda mocked:
    no GPU here; world size 8 has never run on hardware, so this pins the selection logic)."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = {}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.setdefault("set_device", []).append(d))
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: types.SimpleNamespace(device=device))
    monkeypatch.setattr(dist, "init_process_group", lambda backend=None, device_id=None, **kw: calls.setdefault("init", []).append((backend, device_id)))
    monkeypatch.setattr(dist, "get_world_size", lambda: 8)
    for k in ("BN254_BENCH_BACKEND", "BN254_BENCH_SINGLE_DEVICE", "BN254_BENCH_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    for r in range(8):
        monkeypatch.setenv("RANK", str(r)); monkeypatch.setenv("LOCAL_RANK", str(r)); monkeypatch.setenv("WORLD_SIZE", "8")
        R = bench.Rank(types.SimpleNamespace(gpus=8))
        assert R.rank == r and R.local_rank == r and R.world == 8 and R.dist_on and R.backend == "nccl"
        assert R.dev == torch.device("cuda", r) and R.stream.device == torch.device("cuda", r)
    assert calls["set_device"] == list(range(8))
    assert calls["init"] == [("nccl", torch.device("cuda", r)) for r in range(8)]
    # a launcher whose world size disagrees with --gpus is refused
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit):
        bench.Rank(types.SimpleNamespace(gpus=8))


def _two_rank_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BN254_BENCH_BACKEND="gloo", BN254_BENCH_SINGLE_DEVICE="1")
    return env


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus2_verify_command_line():
    """the driver-shaped command for configs[1], two ranks: n_gpus 2, every gathered status vector checked"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4096"],
                       env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["unit"] == "pairings/s"
    assert r["config"]["bit_exact_vs_expected"] is True and r["config"]["batch_per_gpu"] == 4096
    assert r["config"]["status_vectors_checked"] == 2 * 3                    # 2 checks x (own shard + 2 gathered shards)
    assert abs(r["value"] - 2 * 2 * 4096 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-6


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus2_pairing_command_line_checksum_vs_oracle():
    """configs[3] on two ranks: status gather, and the all-reduced 64-bit Gt checksum equals the oracle's over BOTH shards"""
    import numpy as np
    from oracle import c_oracle as c
    n = 1024
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "pairing", "--steps", "2", "--warmup", "1",
                        "--batch", str(n)], env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["batch_per_gpu"] == n and r["config"]["duplicate_inputs_equal_gt"] is True
    # rebuild both shards' inputs with the oracle alone and sum the Gt words
    spec = __import__("importlib.util").util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = __import__("importlib.util").util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pool = min(4096, n)
    sc = [hashlib.sha256(b"cfg4-%d" % j).digest() for j in range(2 * pool)]
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    red = lambda k: (int.from_bytes(k, "big") % R).to_bytes(32, "big")   # noqa: E731
    g1, g2 = c.g1_generator(), c.g2_generator()
    P = [c.g1_mul(g1, red(k)) for k in sc[:pool]]
    Qs = [c.g2_mul(g2, red(k)) for k in sc[pool:]]
    total = 0
    for rank in range(2):
        pi, qi = bench.pairing_indices(np, rank * n, n, pool)
        gt, st = c.batch_pairing(b"".join(P[i] for i in pi), b"".join(Qs[i] for i in qi), n, 1, nthreads=8)
        assert st == bytes([9]) * n
        total += int(np.frombuffer(gt, dtype="<u8").sum(dtype=np.uint64))
    assert "%016x" % (total % (1 << 64)) == r["config"]["gt_checksum_u64"]


def _check_scaling_detail(r, W, with_kernels):
    """the N > 1 line carries what a sub-linear curve would be diagnosed with: per-rank elapsed (min / max / mean, one value per
    rank), per-rank kernel time, the collectives' own time — consistent with the headline ms_per_step (= max over ranks)"""
    d = r["scaling_detail"]
    el = d["own_elapsed_ms_per_step"]
    assert len(el["per_rank"]) == W and el["min"] <= el["mean"] <= el["max"] and el["min"] > 0
    assert el["max"] <= r["ms_per_step"] * 1.0001                          # a rank's own clock stops before the closing barrier
    assert 0 <= d["slowest_rank"] < W and 0.0 <= d["spread_pct"] < 100.0
    co = d["collective_ms_per_step"]
    assert len(co["per_rank"]) == W and 0.0 <= co["min"] <= co["max"] <= r["ms_per_step"]
    if with_kernels:
        km = d["kernel_ms_per_step"]
        assert len(km["per_rank"]) == W and 0.0 < km["min"] and all(k <= e * 1.0001 for k, e in zip(km["per_rank"], el["per_rank"]))
    else:
        assert d["kernel_ms_per_step"] is None


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus4_both_workloads_all_shards_checked():
    """more ranks than two on the one GPU of the box: `bench.py --gpus 4` for both sharded workloads (gloo, single-device
    knobs).  FOUR, not the node's eight: the GPU pool allows at most six processes on a card at once and this test process
    is one of them (world size 8 itself: test_n_rank_gather_gloo[8] and test_rank_selects_its_own_device... on the CPU).
    Every rank checks every gathered shard; the all-reduced Gt checksum equals the oracle's over all four shards."""
    import numpy as np
    from oracle import c_oracle as c
    W = 4
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--steps", "2", "--warmup", "1", "--batch", "2048"],
                       env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == W and r["config"]["bit_exact_vs_expected"] is True and r["config"]["batch_per_gpu"] == 2048
    assert r["config"]["status_vectors_checked"] == 2 * (1 + W)              # 2 checks x (own shard + W gathered shards)
    assert abs(r["value"] - 2 * W * 2048 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-6
    _check_scaling_detail(r, W, with_kernels=True)
    n = 512
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--workload", "pairing", "--steps", "2", "--warmup", "1",
                        "--batch", str(n)], env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == W and r["config"]["batch_per_gpu"] == n
    _check_scaling_detail(r, W, with_kernels=False)
    spec = __import__("importlib.util").util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = __import__("importlib.util").util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pool = min(4096, n)
    sc = [hashlib.sha256(b"cfg4-%d" % j).digest() for j in range(2 * pool)]
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    red = lambda k: (int.from_bytes(k, "big") % R).to_bytes(32, "big")   # noqa: E731
    g1, g2 = c.g1_generator(), c.g2_generator()
    P = [c.g1_mul(g1, red(k)) for k in sc[:pool]]
    Qs = [c.g2_mul(g2, red(k)) for k in sc[pool:]]
    total = 0
    for rank in range(W):
        pi, qi = bench.pairing_indices(np, rank * n, n, pool)
        gt, st = c.batch_pairing(b"".join(P[i] for i in pi), b"".join(Qs[i] for i in qi), n, 1, nthreads=8)
        assert st == bytes([9]) * n
        total += int(np.frombuffer(gt, dtype="<u8").sum(dtype=np.uint64))
    assert "%016x" % (total % (1 << 64)) == r["config"]["gt_checksum_u64"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_rccl_collectives_on_one_rank():
    """the RCCL path itself (backend "nccl": all_gather_into_tensor of the status bytes, all_reduce of the checksum,
    barrier, all on the bench's explicit stream) in a 1-rank job — two ranks cannot share one GPU under RCCL, so this
    is what a one-GPU box can exercise of it; the N-rank logic is covered by the gloo tests above"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BN254_BENCH_BACKEND")}
    env.update(BN254_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, key in ((["--batch", "4096"], "bit_exact_vs_expected"), (["--workload", "pairing", "--batch", "2048"], "duplicate_inputs_equal_gt")):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540)
        assert p.returncode == 0, p.stderr[-2000:]
        r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert r["n_gpus"] == 1 and r["config"][key] is True and "nccl" in r["config"]["collective"]
        # the RCCL collectives are timed by HIP events on the rank's stream
        _check_scaling_detail(r, 1, with_kernels="--workload" not in extra)
        assert "HIP events" in r["scaling_detail"]["collective_timed_by"] and r["scaling_detail"]["collective_ms_per_step"]["max"] > 0
