"""Build and load libbn254hip.so (the HIP kernels + C ABI of include/bn254_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is present, the
calls raise.  The library is built in-tree (bn254_amd/libbn254hip.so) with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.environ.get("BN254_LIB", os.path.join(_PKG, "libbn254hip.so"))   # BN254_LIB: A/B-test another build
import glob


def translation_units():
    """Every .hip file of csrc/ is a translation unit of the library (the compile command and the staleness check both
    come from this list, so they cannot drift apart)."""
    return sorted(glob.glob(os.path.join(_CSRC, "*.hip")))


def _dependencies():
    return translation_units() + sorted(glob.glob(os.path.join(_CSRC, "*.h"))) + sorted(glob.glob(os.path.join(_CSRC, "*.inc"))) + [
        os.path.join(_CSRC, "gen_constants.py"), os.path.join(_CSRC, "gen_step_asm.py"), os.path.join(os.path.dirname(_PKG), "include", "bn254_hip.h")]

# max-ilp: the AMDGPU machine scheduler's ILP-first strategy — these kernels are VALU-issue bound at a fixed occupancy
# (amdgpu_waves_per_eu), so the default strategy's occupancy-driven choices buy nothing; same-box A/B +1.3 % on the verify step
# (profiles/r02_c_ab_sched_strategy.log).  -Wl,--no-undefined: a missing translation unit fails at link time.
HIPCC_COMPILE_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
HIPCC_LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined"]
HIPCC_FLAGS = HIPCC_COMPILE_FLAGS + ["-shared", "-Wl,--no-undefined"]      # the one-command form (tools/build_variant.sh)
_OBJ = os.path.join(_PKG, "build")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _dependencies())


def build(force=False, verbose=False, jobs=None):
    """Compile the HIP extension for gfx950 (hipcc cross-compiles without a GPU): every translation unit to an object of its own, side by
    side (the eleven units are independent; one after the other they take over a minute), then one link."""
    const_h = os.path.join(_CSRC, "bn254_constants.h")
    gen = os.path.join(_CSRC, "gen_constants.py")
    if not os.path.exists(const_h) or os.path.getmtime(const_h) < os.path.getmtime(gen):
        subprocess.check_call(["python3", gen], stdout=None if verbose else subprocess.DEVNULL)
    asm_h, asm_gen = os.path.join(_CSRC, "bn254_csqr_asm.h"), os.path.join(_CSRC, "gen_step_asm.py")
    if not os.path.exists(asm_h) or os.path.getmtime(asm_h) < os.path.getmtime(asm_gen):
        subprocess.check_call(["python3", asm_gen, "selftest"], stdout=None if verbose else subprocess.DEVNULL)   # simulated before it is assembled
        with open(asm_h, "w") as f:
            subprocess.check_call(["python3", asm_gen, "header"], stdout=f)
    mul_h = os.path.join(_CSRC, "bn254_mul_asm.h")
    if not os.path.exists(mul_h) or os.path.getmtime(mul_h) < os.path.getmtime(asm_gen):
        subprocess.check_call(["python3", asm_gen, "selftest_mul"], stdout=None if verbose else subprocess.DEVNULL)
        with open(mul_h, "w") as f:
            subprocess.check_call(["python3", asm_gen, "header_mul"], stdout=f)
    if force or _stale():
        import time
        from concurrent.futures import ThreadPoolExecutor
        hipcc = os.environ.get("HIPCC", "hipcc")
        os.makedirs(_OBJ, exist_ok=True)
        units = translation_units()
        objs = [os.path.join(_OBJ, os.path.basename(u)[:-4] + ".o") for u in units]
        t0 = time.time()

        def compile_one(pair):
            cmd = [hipcc] + HIPCC_COMPILE_FLAGS + ["-c", "-o", pair[1], pair[0]]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, zip(units, objs)))
        cmd = [hipcc] + HIPCC_LINK_FLAGS + ["-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        if verbose:
            print("libbn254hip.so: %d translation units compiled and linked in %.1f s" % (len(units), time.time() - t0), flush=True)
    return LIB_PATH


_lib = None


def load():
    """Load the shared library and declare the C ABI.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libbn254hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "bn254_amd has no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
    L.bn254_version.restype = ctypes.c_char_p
    L.bn254_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.bn254_ctx_destroy.argtypes = [vp]
    L.bn254_ctx_destroy.restype = None
    L.bn254_ctx_reserve.argtypes = [vp, sz]
    L.bn254_ctx_reserve_host.argtypes = [vp, sz, sz]
    L.bn254_ctx_synchronize.argtypes = [vp]
    L.bn254_ctx_set_profiling.argtypes = [vp, i32]
    L.bn254_ctx_expect_msgs_len.argtypes = [vp, ctypes.c_uint64]
    L.bn254_ctx_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.bn254_ctx_set_option.argtypes = [vp, i32, i32]
    L.bn254_debug_route_table.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(i32), ctypes.POINTER(i32), i32]
    L.bn254_ctx_last_clocks.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.bn254_batch_verify.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp]
    L.bn254_batch_verify_device.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp, vp]
    L.bn254_ctx_register_keys.argtypes = [vp, vp, sz, u32, vp]
    L.bn254_batch_verify_keyed.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp]
    L.bn254_batch_verify_keyed_device.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp, vp]
    L.bn254_batch_verify_keyed_randomized.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp, vp]
    L.bn254_batch_verify_keyed_randomized_device.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp, vp, vp]
    L.bn254_batch_verify_compressed.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.bn254_batch_verify_compressed_device.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
    L.bn254_batch_verify_randomized.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp, vp, vp]
    L.bn254_batch_verify_randomized_device.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp, vp, vp, vp]
    L.bn254_batch_hash_to_g1.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.bn254_batch_hash_to_g1_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
    L.bn254_batch_pairing_check.argtypes = [vp, vp, vp, sz, sz, u32, vp]
    L.bn254_batch_pairing.argtypes = [vp, vp, vp, sz, sz, u32, vp, vp]
    L.bn254_batch_pairing_device.argtypes = [vp, vp, vp, sz, sz, u32, vp, vp, vp]
    L.bn254_batch_check_public_keys.argtypes = [vp, vp, vp, sz, u32, vp]
    L.bn254_batch_g1_add.argtypes = [vp, vp, vp, sz, vp, vp]
    L.bn254_batch_g2_add.argtypes = [vp, vp, vp, sz, vp, vp]
    L.bn254_batch_g1_mul.argtypes = [vp, vp, vp, sz, i32, vp, vp]
    L.bn254_batch_g2_mul.argtypes = [vp, vp, vp, sz, i32, vp, vp]
    L.bn254_batch_g1_mul_device.argtypes = [vp, vp, vp, sz, i32, vp, vp, vp]
    L.bn254_batch_g2_mul_device.argtypes = [vp, vp, vp, sz, i32, vp, vp, vp]
    L.bn254_batch_sign.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    L.bn254_batch_sign_device.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
    L.bn254_batch_g1_sum.argtypes = [vp, vp, vp, sz, vp, vp]
    L.bn254_batch_g2_sum.argtypes = [vp, vp, vp, sz, vp, vp]
    L.bn254_batch_aggregate_verify.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, sz, u32, vp]
    L.bn254_batch_aggregate_verify_device.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, sz, u32, vp, vp]
    L.bn254_ctx_register_pools.argtypes = [vp, vp, vp, sz, vp, sz, vp, u32, sz]
    L.bn254_ctx_register_pools_device.argtypes = [vp, vp, vp, sz, vp, sz, vp, u32, sz, vp]
    L.bn254_batch_aggregate_verify_registered.argtypes = [vp, vp, vp, vp, sz, vp]
    L.bn254_batch_aggregate_verify_registered_device.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    L.bn254_batch_g1_decompress.argtypes = [vp, vp, sz, vp, vp]
    L.bn254_batch_g2_decompress.argtypes = [vp, vp, sz, vp, vp]
    L.bn254_debug_fp_op.argtypes = [vp, i32, vp, vp, sz, vp, vp]
    L.bn254_debug_fp12_op.argtypes = [vp, i32, vp, vp, sz, vp]
    L.bn254_debug_final_exp_limbs.argtypes = [vp, i32, vp, sz, vp, vp]
    L.bn254_debug_miller_loop.argtypes = [vp, vp, vp, sz, vp]
    L.bn254_debug_hash_candidate.argtypes = [vp, vp, sz, vp, vp]
    L.bn254_probe_issue_rate.argtypes = [vp, i32, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i32)]
    L.bn254_probe_leaf_floor.argtypes = [vp, sz, i32, ctypes.POINTER(ctypes.c_float)]
    L.bn254_probe_fe_program.argtypes = [vp, sz, vp, sz, ctypes.POINTER(ctypes.c_float)]
    # multi-GPU layer (bn254_mgpu.hip)
    pp = ctypes.POINTER(vp)
    L.bn254_mgpu_create.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(vp)]
    L.bn254_mgpu_destroy.argtypes = [vp]
    L.bn254_mgpu_destroy.restype = None
    L.bn254_mgpu_device_count.argtypes = [vp]
    L.bn254_mgpu_ctx.argtypes = [vp, i32]
    L.bn254_mgpu_ctx.restype = vp
    L.bn254_mgpu_shard_len.argtypes = [vp, sz]
    L.bn254_mgpu_shard_len.restype = sz
    L.bn254_mgpu_gathered_len.argtypes = [vp, sz]
    L.bn254_mgpu_gathered_len.restype = sz
    L.bn254_mgpu_shard_range.argtypes = [vp, sz, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.bn254_mgpu_reserve.argtypes = [vp, sz, i32]
    L.bn254_mgpu_synchronize.argtypes = [vp]
    L.bn254_mgpu_set_option.argtypes = [vp, i32, i32]
    L.bn254_mgpu_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.bn254_mgpu_last_error.argtypes = [vp]
    L.bn254_mgpu_last_error.restype = ctypes.c_char_p
    L.bn254_mgpu_batch_verify.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp]
    L.bn254_mgpu_batch_verify_device.argtypes = [vp, pp, pp, pp, pp, sz, u32, pp, pp]
    L.bn254_mgpu_batch_pairing.argtypes = [vp, vp, vp, sz, sz, u32, vp, vp, ctypes.POINTER(ctypes.c_uint64)]
    L.bn254_mgpu_batch_pairing_device.argtypes = [vp, pp, pp, sz, sz, u32, pp, pp, pp, pp]
    L.bn254_mgpu_batch_hash_to_g1.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.bn254_mgpu_batch_verify_compressed.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.bn254_mgpu_register_keys.argtypes = [vp, vp, sz, u32, vp]
    L.bn254_mgpu_batch_verify_keyed.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp]
    L.bn254_mgpu_batch_aggregate_verify.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, sz, u32, vp]
    _lib = L
    return L


# every symbol include/bn254_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = [
    "bn254_version", "bn254_ctx_create", "bn254_ctx_destroy", "bn254_ctx_reserve", "bn254_ctx_reserve_host", "bn254_ctx_synchronize",
    "bn254_batch_verify", "bn254_batch_verify_device", "bn254_batch_verify_randomized", "bn254_batch_verify_randomized_device", "bn254_batch_verify_compressed", "bn254_batch_verify_compressed_device", "bn254_batch_hash_to_g1", "bn254_batch_hash_to_g1_device",
    "bn254_batch_pairing_check", "bn254_batch_pairing", "bn254_batch_pairing_device", "bn254_batch_check_public_keys",
    "bn254_batch_g1_add", "bn254_batch_g2_add", "bn254_batch_g1_mul", "bn254_batch_g2_mul", "bn254_batch_g1_mul_device",
    "bn254_batch_g2_mul_device", "bn254_batch_sign", "bn254_batch_sign_device", "bn254_batch_g1_sum", "bn254_batch_g2_sum",
    "bn254_batch_aggregate_verify", "bn254_batch_aggregate_verify_device", "bn254_ctx_register_pools", "bn254_ctx_register_pools_device",
    "bn254_batch_aggregate_verify_registered", "bn254_batch_aggregate_verify_registered_device", "bn254_batch_g1_decompress", "bn254_batch_g2_decompress", "bn254_debug_fp_op", "bn254_debug_fp12_op", "bn254_debug_final_exp_limbs", "bn254_debug_miller_loop", "bn254_debug_hash_candidate", "bn254_debug_route_table", "bn254_probe_issue_rate", "bn254_probe_leaf_floor", "bn254_probe_fe_program", "bn254_ctx_set_profiling", "bn254_ctx_last_kernel_ms", "bn254_ctx_set_option", "bn254_ctx_last_clocks", "bn254_ctx_expect_msgs_len", "bn254_ctx_register_keys", "bn254_batch_verify_keyed", "bn254_batch_verify_keyed_device", "bn254_batch_verify_keyed_randomized", "bn254_batch_verify_keyed_randomized_device",
    "bn254_mgpu_create", "bn254_mgpu_destroy", "bn254_mgpu_device_count", "bn254_mgpu_ctx", "bn254_mgpu_shard_len", "bn254_mgpu_shard_range",
    "bn254_mgpu_gathered_len", "bn254_mgpu_reserve", "bn254_mgpu_synchronize", "bn254_mgpu_set_option", "bn254_mgpu_last_timing",
    "bn254_mgpu_last_error", "bn254_mgpu_batch_verify", "bn254_mgpu_batch_verify_device", "bn254_mgpu_batch_pairing",
    "bn254_mgpu_batch_pairing_device", "bn254_mgpu_batch_hash_to_g1", "bn254_mgpu_batch_verify_compressed", "bn254_mgpu_register_keys",
    "bn254_mgpu_batch_verify_keyed", "bn254_mgpu_batch_aggregate_verify",
]
