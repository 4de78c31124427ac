"""ASan + UBSan over the host compilation of the device arithmetic and over the C oracle
(GPU sanitizers are unavailable on the pool; the CPU build is where they run — SURVEY.md §5).
UBSan's signed-overflow check is a run-time complement to the interval tracker of test_bounds.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_hostsim_and_oracle_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not (asan and ubsan):
        pytest.skip("sanitizer runtimes not installed")
    from tests import hostsim_binding
    hostsim_binding.build_all()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libbn254_oracle_asan.so"])
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_run.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitizer run ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
