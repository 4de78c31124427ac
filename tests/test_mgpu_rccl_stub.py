"""The RCCL branch of the library's own multi-GPU split (bn254_amd/csrc/bn254_mgpu.hip: gather()) with G = 4 ranks on ONE GPU.

RCCL refuses two ranks on one device and the pool has no multi-GPU box for the builder, so until now the G > 1 form of that branch —
ncclGroupStart / in-place ncclAllGather at offset g*S (+ ncclAllReduce of the Gt checksum) per communicator from one thread /
ncclGroupEnd — had never executed anywhere.  tests/rccl_stub/librccl.so.1 is a stand-in that implements the contract of those calls
with device-to-device copies and events (and checks that every rank posts the same collective in the group); the library under test
finds it through the very dlopen("librccl.so.1") it uses for the real thing.  The case runner is a process of its own WITHOUT torch
(torch would bring the real librccl into the process): tests/rccl_stub/run_cases.py.
Reference for the per-tuple semantics: /root/reference/src/ecdsa.rs:49-64."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "rccl_stub")


def test_stub_library_is_built_and_exports_the_calls_the_layer_binds():
    """CPU: the stand-in exists (built by __graft_entry__.build()) and has every symbol bn254_mgpu.hip resolves with dlsym"""
    import ctypes
    path = os.path.join(STUB_DIR, "librccl.so.1")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", STUB_DIR])
    lib = ctypes.CDLL(path)
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclAllGather", "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString",
                 "bn254_rccl_stub_shared_devices", "bn254_rccl_stub_stats"):
        assert hasattr(lib, name), name
    src = open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_mgpu.hip")).read()
    assert "#include <rccl" not in src                     # the product builds without the RCCL development headers
    assert "bn254_rccl_stub_shared_devices" in src         # ... and accepts a shared device only from a library that says it is the stand-in


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_branch_with_four_ranks_through_the_stand_in_vs_oracle():
    if not os.path.exists(os.path.join(STUB_DIR, "librccl.so.1")):       # normally built by __graft_entry__.build() and shipped with the tree
        subprocess.check_call(["make", "-C", STUB_DIR])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = STUB_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env.pop("BN254_RCCL_STUB_FAIL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(STUB_DIR, "run_cases.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                       timeout=840)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert len(r["cases"]) == 5 and r["stats"]["max_ranks_in_group"] == 4 and r["stats"]["failed"] == 1
    assert r["stats"]["allgather"] >= 8 and r["stats"]["allreduce"] == 2 and r["stats"]["inplace"] >= 32
    assert "ncclAllGather" in r["failing_rank"]
    seen = set()
    for case in r["cases"]:
        seen |= set(case["statuses_seen"])
    assert seen >= {0, 9}
