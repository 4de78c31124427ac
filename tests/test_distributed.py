"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, produce their status shards and
all-gather them exactly as bench.py does over RCCL.  (The verify itself needs a GPU; here each rank's
statuses come from the oracle on its own shard, which is what the gather must reassemble.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bn254_amd.sharding import failure_count, gather_status, shard_range

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
    allst = gather_status(local, n_total=n)
    fails = failure_count(local[: hi - lo])
    if rank == 0:
        q.put((bytes(allst.tolist()), fails))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_gloo(derived):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, fails = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = bytes(v["status"] for v in derived["verify_cases"])
    assert got == want
    assert fails == sum(1 for s in want if s)
