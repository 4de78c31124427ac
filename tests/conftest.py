import json
import os
import sys

import pytest

# torch first: its bundled HIP runtime and the system one libbn254hip.so links share a SONAME, and the process must
# end up with ONE runtime — the copy loaded first.  A test that imported torch only after the library had initialised
# the device found "No HIP GPUs are available" (two runtimes in one process).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - CPU-only environments without torch still run the oracle tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def derived():
    with open(os.path.join(GOLDEN, "derived_vectors.json")) as f:
        return json.load(f)


def ws_default(name):
    """the value of `#define <name> <integer>` in bn254_amd/csrc/bn254_ws.h — tests restore library defaults from the header, not from
    literals that would silently drift"""
    import re
    text = open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_ws.h")).read()
    m = re.search(r"#define\s+%s\s+(\d+)" % re.escape(name), text)
    assert m, name
    return int(m.group(1))
