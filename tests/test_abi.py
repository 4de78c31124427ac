"""The C-ABI library loads and exports every symbol include/bn254_hip.h declares; without a GPU it
refuses to create a context (no CPU fallback).  CPU only — no compute calls."""
import ctypes
import os
import re

import pytest

from bn254_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _native.build()
    return _native.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bn254_hip.h")).read()
    declared = set(re.findall(r"\b(bn254_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_version(lib):
    assert b"gfx950" in lib.bn254_version()


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.bn254_ctx_create(0, ctypes.byref(h)) == -10003     # BN254_E_NO_DEVICE
    with pytest.raises(Exception):
        import bn254_amd
        bn254_amd.Engine(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "bn254_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libbn254_oracle" not in src and "hostsim" not in src.replace("tests/hostsim", ""), f


def test_every_source_makes_the_library_stale(monkeypatch):
    """The staleness check covers every translation unit and header the compile command reads (a forgotten file would let
    tests and bench run against an old libbn254hip.so): with each dependency's mtime pushed past the library's, in turn,
    `_stale()` must say so; with none newer it must not."""
    deps = _native._dependencies()
    names = {os.path.basename(p) for p in deps}
    csrc = os.path.join(ROOT, "bn254_amd", "csrc")
    on_disk = {f for f in os.listdir(csrc) if f.endswith((".hip", ".h"))}
    assert on_disk <= names and {"bn254_quad.hip", "bn254_constants.h", "bn254_hip.h", "gen_constants.py"} <= names
    assert set(_native.translation_units()) == {os.path.join(csrc, f) for f in on_disk if f.endswith(".hip")}
    real = os.path.getmtime
    lib_t = 1000.0
    for newer in [None] + deps:
        monkeypatch.setattr(os.path, "getmtime", lambda p, newer=newer: lib_t + 1 if p == newer else (lib_t if p == _native.LIB_PATH else lib_t - 1))
        monkeypatch.setattr(os.path, "exists", lambda p: True)
        assert _native._stale() == (newer is not None), newer
    monkeypatch.setattr(os.path, "getmtime", real)


def test_option_numbers_of_header_and_python_mirror_agree():
    """every BN254_OPT_* of include/bn254_hip.h has the same number in bn254_amd.engine (OPT_*), and no number is used twice"""
    import re
    from bn254_amd import engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "bn254_hip.h")).read()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define BN254_OPT_(\w+) (\d+)", text)}
    assert len(set(header.values())) == len(header), header
    mirror = {k[4:]: v for k, v in vars(engine).items() if k.startswith("OPT_") and isinstance(v, int)}
    assert mirror, "bn254_amd.engine exports no OPT_* constants"
    for name, value in mirror.items():
        assert header.get(name) == value, (name, value, header.get(name))
    # the status codes the header names are the reference's Error variants, in order (src/error.rs:6-29)
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define BN254_ERR_(\w+) (\d+)", text)}
    assert sorted(codes.values()) == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11] and codes["INDEX_OUT_OF_BOUNDS"] == 2 and codes["VERIFICATION_FAILED"] == 9


def test_header_is_plain_c_and_the_dev_hooks_can_be_hidden(tmp_path):
    """include/bn254_hip.h compiles as C (the FFI of the reference's language binds a C header) and as C++; with BN254_NO_DEV_HOOKS the
    debug / probe entry points are not declared (they are not part of the drop-in ABI) while everything a binding needs still is"""
    import subprocess
    hdr = os.path.join(ROOT, "include", "bn254_hip.h")
    src = tmp_path / "t.c"
    src.write_text('#define BN254_NO_DEV_HOOKS 1\n#include "%s"\n'
                   'int use(bn254_ctx *c, bn254_mgpu *m) {\n'
                   '  (void)bn254_batch_verify; (void)bn254_mgpu_batch_verify; (void)bn254_mgpu_batch_verify_device; (void)bn254_ctx_reserve_host;\n'
                   '  return bn254_ctx_synchronize(c) + bn254_mgpu_device_count(m) + BN254_E_RCCL + BN254_ERR_VERIFICATION_FAILED;\n}\n'
                   '#ifdef TRY_HOOK\nint hook(bn254_ctx *c) { return bn254_debug_fp_op(c, 0, 0, 0, 0, 0, 0); }\n#endif\n' % hdr)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-unused-function", "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-fsyntax-only", str(src)])
    # with the hooks hidden, naming one is an error (implicit declarations are errors under -Werror=implicit-function-declaration)
    p = subprocess.run(["gcc", "-std=c99", "-Werror=implicit-function-declaration", "-DTRY_HOOK", "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert p.returncode != 0 and "bn254_debug_fp_op" in p.stderr


def test_routing_table_rows(tmp_path):
    """The batch-size -> layout routing is ONE table (bn254_amd/csrc/bn254_ws.h: bn_route / bn_route_table): its default rows are the documented
    ones, bn_route agrees with the rows on both sides of every boundary, and overriding a threshold moves exactly that boundary."""
    import subprocess
    src = tmp_path / "route.cpp"
    src.write_text(r'''
#define BN_WS_ROUTE_ONLY 1
#include "bn254_ws.h"
#include <cstdio>
static void dump(BnRouteLimits L) {
  size_t m[5]; BnRoute r[5];
  int rows = bn_route_table(L, m, r, 5);
  for (int i = 0; i < rows; ++i) printf("%zu:%d:%d ", m[i], r[i].miller, r[i].fe);
  // bn_route itself on both sides of every boundary
  for (int i = 0; i + 1 < rows; ++i) {
    BnRoute a = bn_route(L, m[i]), b = bn_route(L, m[i] + 1);
    if (a.miller != r[i].miller || a.fe != r[i].fe || b.miller != r[i + 1].miller || b.fe != r[i + 1].fe) printf("MISMATCH ");
  }
  printf("\n");
}
int main() {
  dump(BnRouteLimits{LM_MAX_BATCH_DEFAULT, NONET_WIDE_MAX_BATCH, NONET_MAX_BATCH_DEFAULT, TRIO_MAX_BATCH_DEFAULT});
  dump(BnRouteLimits{0, NONET_WIDE_MAX_BATCH, NONET_MAX_BATCH_DEFAULT, TRIO_MAX_BATCH_DEFAULT});      // lane machine off
  dump(BnRouteLimits{1u << 20, 0, 0, 1u << 20});                                                     // lane machine + octets at every small size
  dump(BnRouteLimits{LM_MAX_BATCH_DEFAULT, NONET_WIDE_MAX_BATCH, NONET_MAX_BATCH_DEFAULT, 0});        // small-batch family off
  dump(BnRouteLimits{1u << 20, NONET_WIDE_MAX_BATCH, 1u << 20, 8192});                                // everything clipped at the family's end
  return 0;
}''')
    exe = tmp_path / "route"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "bn254_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    big = str(2 ** 64 - 1)
    assert out[0].split() == ["1024:0:0", "1536:0:1", "3072:1:1", "16384:1:2", big + ":2:3"]
    assert out[1].split() == ["1024:1:0", "3072:1:1", "16384:1:2", big + ":2:3"]
    assert out[2].split() == ["1048576:0:2", big + ":2:3"]
    assert out[3].split() == [big + ":2:3"]
    assert out[4].split() == ["1024:0:0", "8192:0:1", big + ":2:3"]
    assert not any("MISMATCH" in ln for ln in out)


def test_generated_csqr_assembly_is_simulated_and_current():
    """The one hand-scheduled block of the product — the cyclotomic squaring of the final exponentiation as generated gfx950 assembly
    (bn254_amd/csrc/gen_step_asm.py -> bn254_csqr_asm.h) — is executed by the generator's own four-lane simulator against a big-integer
    model of the Granger-Scott formulas (random and extreme limbs, DPP hazards checked), and the committed header is what the generator
    prints today."""
    import subprocess
    import sys
    gen = os.path.join(ROOT, "bn254_amd", "csrc", "gen_step_asm.py")
    out = subprocess.check_output([sys.executable, gen, "selftest"], text=True)
    assert "selftest ok" in out
    assert subprocess.check_output([sys.executable, gen, "header"], text=True) == open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_csqr_asm.h")).read()
    stats = subprocess.check_output([sys.executable, gen, "stats"], text=True)
    assert "v_mov 16" in stats                               # nine leaf outputs' top limbs + setup: no operand shuffling between operations
    # ... and the MUL opcode's block (bn254_mul_asm.h): the Fq12 product of an LDS accumulator and a private-segment slot, its eighteen dual
    # products a subroutine inside the block, against the schoolbook product over the tower (random and extreme limbs, a carried c02)
    out = subprocess.check_output([sys.executable, gen, "selftest_mul"], text=True)
    assert "selftest mul ok" in out and "called 18 times" in out
    assert subprocess.check_output([sys.executable, gen, "header_mul"], text=True) == open(os.path.join(ROOT, "bn254_amd", "csrc", "bn254_mul_asm.h")).read()
