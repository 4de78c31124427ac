"""Thin Python binding of the C ABI (include/bn254_hip.h): one `Engine` per GPU.

Host-pointer methods take/return `bytes`; `*_device` methods take raw device pointers (e.g.
`tensor.data_ptr()` of torch uint8 tensors resident in HBM) and only enqueue work.
"""
import ctypes

from . import _native

G1_BYTES, G2_BYTES, GT_BYTES, SCALAR_BYTES = 64, 128, 384, 32
FLAG_G2_SUBGROUP_CHECK = 1
FLAG_REJECT_IDENTITY = 2
FLAG_RAND64 = 0x100
FLAG_RAND_GLV = 0x200
OPT_SPLIT_MILLER = 1
OPT_HASH_MAX_TRIES = 2
OPT_RAND_ITEMS_PER_LANE = 3
OPT_PAIR_LANES = 4
OPT_RAND_MIN_BATCH = 5
OPT_TRIO_MAX_BATCH = 6
OPT_HASH_DIRECT_WIDTH = 7
OPT_TRIO_WAVE_ROLES = 8
OPT_AGG_SUBSET_MIN_TUPLES = 9
OPT_CLOCK_PROBE = 10
OPT_AGG_SORT_BY_MSG = 11
OPT_PINNED_STAGING = 12
OPT_NONET_MAX_BATCH = 13
OPT_LM_MAX_BATCH = 15
OPT_NONET_WIDE = 16
OPT_AGG_WIDE_MIN_TUPLES = 14
OPT_MAX_CHUNK = 17          # verify-shaped batches above this size are processed in slices (0 = only when the workspace would not fit)
OPT_ASSUME_FREE_MB = 18
OPT_G2_FIXED_BASE = 19      # developer option: key derivation through the comb table of the generator (default 1)     # test knob of the automatic slicing rule


class NativeError(RuntimeError):
    """A call into libbn254hip.so failed (HIP runtime error or bad argument) — not a per-item status."""

    def __init__(self, fn, rc):
        what = "HIP error %d" % (-rc) if -10000 < rc < 0 else {-10001: "bad argument", -10002: "misaligned device pointer",
                                                                -10003: "no HIP device (bn254_amd has no CPU fallback)"}.get(rc, "error")
        super().__init__("%s failed: %s (rc=%d)" % (fn, what, rc))
        self.rc = rc


def _check(fn, rc):
    if rc != 0:
        raise NativeError(fn, rc)


def pack_messages(messages):
    """list of bytes -> (concatenated bytes, ctypes uint64 offsets[n+1])"""
    n = len(messages)
    off = (ctypes.c_uint64 * (n + 1))()
    pos = 0
    for i, m in enumerate(messages):
        off[i] = pos
        pos += len(m)
    off[n] = pos
    return b"".join(bytes(m) for m in messages), off


class Engine:
    def __init__(self, device=0):
        self._lib = _native.load()
        h = ctypes.c_void_p()
        _check("bn254_ctx_create", self._lib.bn254_ctx_create(device, ctypes.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bn254_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def version(self):
        return self._lib.bn254_version().decode()

    def reserve(self, n):
        _check("bn254_ctx_reserve", self._lib.bn254_ctx_reserve(self._h, n))

    def reserve_host(self, n, msg_bytes):
        """presize workspace + staging for host-pointer batch_verify calls of up to n tuples / msg_bytes message bytes"""
        _check("bn254_ctx_reserve_host", self._lib.bn254_ctx_reserve_host(self._h, n, msg_bytes))

    def synchronize(self):
        _check("bn254_ctx_synchronize", self._lib.bn254_ctx_synchronize(self._h))

    def set_profiling(self, on):
        _check("bn254_ctx_set_profiling", self._lib.bn254_ctx_set_profiling(self._h, 1 if on else 0))

    def expect_msgs_len(self, msgs_len):
        """size in bytes of the d_msgs buffer of the NEXT *_device call that hashes messages: its spans get bounds-checked on the
        device (status 5 for a span outside the buffer) — include/bn254_hip.h: bn254_ctx_expect_msgs_len"""
        _check("bn254_ctx_expect_msgs_len", self._lib.bn254_ctx_expect_msgs_len(self._h, int(msgs_len)))

    def set_option(self, option, value):
        _check("bn254_ctx_set_option", self._lib.bn254_ctx_set_option(self._h, option, value))

    def probe_issue_rate(self, op, waves_per_simd=2):
        """(wave-instructions per second, SIMD count) of op 0 v_mad_u64_u32 / 1 v_add_u32 / 2 v_mul_lo_u32"""
        rate, simds = ctypes.c_double(0), ctypes.c_int(0)
        _check("bn254_probe_issue_rate", self._lib.bn254_probe_issue_rate(self._h, op, waves_per_simd, ctypes.byref(rate), ctypes.byref(simds)))
        return rate.value, simds.value

    def route_table(self):
        """developer hook: the context's batch-size -> layout routing table as it stands: [(max_n, miller, fe)], last row max_n = 2**64 - 1
        (miller: 0 lane machine, 1 wave roles, 2 lane pairs; fe: 0 eighteen lane pairs, 1 nine, 2 octets, 3 lane pairs)"""
        m, a, b = (ctypes.c_uint64 * 8)(), (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
        rows = self._lib.bn254_debug_route_table(self._h, m, a, b, 8)
        if rows < 0:
            _check("bn254_debug_route_table", rows)
        return [(int(m[i]), int(a[i]), int(b[i])) for i in range(rows)]

    def last_kernel_ms(self):
        ms = (ctypes.c_float * 4)()
        _check("bn254_ctx_last_kernel_ms", self._lib.bn254_ctx_last_kernel_ms(self._h, ms))
        return {"decode": ms[0], "hash_to_g1": ms[1], "miller_loop": ms[2], "final_exp": ms[3]}

    def probe_leaf_floor(self, n, mode=0):
        """ms of the kernel that runs only the product leaves of a verify's Miller loop (mode 0) / final exponentiation (mode 1), n lane
        pairs (include/bn254_hip.h)"""
        ms = ctypes.c_float()
        _check("bn254_probe_leaf_floor", self._lib.bn254_probe_leaf_floor(self._h, n, mode, ctypes.byref(ms)))
        return ms.value

    def probe_fe_program(self, n, steps):
        """ms of the final exponentiation's accumulator machine on the program `steps` = [(opcode, arg), ...] (include/bn254_hip.h)"""
        prog = bytes(b for st in steps for b in st)
        ms = ctypes.c_float()
        _check("bn254_probe_fe_program", self._lib.bn254_probe_fe_program(self._h, n, prog, len(steps), ctypes.byref(ms)))
        return ms.value

    def last_clocks(self):
        """OPT_CLOCK_PROBE on: achieved shader clock (MHz) of the last lane-pair Miller kernel, final exponentiation and issue probe"""
        mhz = (ctypes.c_double * 3)()
        _check("bn254_ctx_last_clocks", self._lib.bn254_ctx_last_clocks(self._h, mhz))
        return {"miller_loop": mhz[0], "final_exp": mhz[1], "issue_probe": mhz[2]}

    # ---- host-pointer entry points ------------------------------------------------------
    def batch_verify(self, messages, sigs, pks, flags=0):
        n = len(messages)
        assert len(sigs) == n * G1_BYTES and len(pks) == n * G2_BYTES
        msgs, off = pack_messages(messages)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_verify", self._lib.bn254_batch_verify(self._h, msgs, off, bytes(sigs), bytes(pks), n, flags, status))
        return status.raw[:n]

    def batch_verify_compressed(self, messages, sigs33, pks65):
        """verify from the compressed encodings (33-byte signatures, 65-byte public keys)"""
        n = len(messages)
        assert len(sigs33) == n * 33 and len(pks65) == n * 65
        msgs, off = pack_messages(messages)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_verify_compressed",
               self._lib.bn254_batch_verify_compressed(self._h, msgs, off, bytes(sigs33), bytes(pks65), n, status))
        return status.raw[:n]

    def batch_verify_randomized(self, messages, sigs, pks, seed32, flags=0):
        """opt-in randomised batch verification (include/bn254_hip.h) -> (status bytes, group_ok bytes)"""
        n = len(messages)
        assert len(sigs) == n * G1_BYTES and len(pks) == n * G2_BYTES and len(seed32) == 32
        msgs, off = pack_messages(messages)
        status = ctypes.create_string_buffer(max(n, 1))
        ng = (n + 63) // 64
        groups = ctypes.create_string_buffer(max(ng, 1))
        _check("bn254_batch_verify_randomized",
               self._lib.bn254_batch_verify_randomized(self._h, msgs, off, bytes(sigs), bytes(pks), n, flags, bytes(seed32), status, groups))
        return status.raw[:n], groups.raw[:ng]

    def batch_hash_to_g1(self, messages):
        n = len(messages)
        msgs, off = pack_messages(messages)
        pts = ctypes.create_string_buffer(max(n, 1) * G1_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        tries = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_hash_to_g1", self._lib.bn254_batch_hash_to_g1(self._h, msgs, off, n, pts, status, tries))
        return pts.raw[:n * G1_BYTES], status.raw[:n], tries.raw[:n]

    def batch_pairing_check(self, g1s, g2s, n, k, flags=0):
        assert len(g1s) == n * k * G1_BYTES and len(g2s) == n * k * G2_BYTES
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_pairing_check", self._lib.bn254_batch_pairing_check(self._h, bytes(g1s), bytes(g2s), n, k, flags, status))
        return status.raw[:n]

    def batch_pairing(self, g1s, g2s, n, k=1, flags=0):
        assert len(g1s) == n * k * G1_BYTES and len(g2s) == n * k * G2_BYTES
        gt = ctypes.create_string_buffer(max(n, 1) * GT_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_pairing", self._lib.bn254_batch_pairing(self._h, bytes(g1s), bytes(g2s), n, k, flags, gt, status))
        return gt.raw[:n * GT_BYTES], status.raw[:n]

    def batch_check_public_keys(self, pk_g2, pk_g1, n, flags=0):
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_check_public_keys", self._lib.bn254_batch_check_public_keys(self._h, bytes(pk_g2), bytes(pk_g1), n, flags, status))
        return status.raw[:n]

    def _binop(self, name, a, b, n, size):
        out = ctypes.create_string_buffer(max(n, 1) * size)
        status = ctypes.create_string_buffer(max(n, 1))
        _check(name, getattr(self._lib, name)(self._h, bytes(a), bytes(b), n, out, status))
        return out.raw[:n * size], status.raw[:n]

    def batch_g1_add(self, a, b, n):
        return self._binop("bn254_batch_g1_add", a, b, n, G1_BYTES)

    def batch_g2_add(self, a, b, n):
        return self._binop("bn254_batch_g2_add", a, b, n, G2_BYTES)

    def batch_g1_mul(self, points, scalars, n, reduce_scalar=False):
        """points=None multiplies the G1 generator (PublicKeyG1::from_private_key)."""
        out = ctypes.create_string_buffer(max(n, 1) * G1_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_g1_mul", self._lib.bn254_batch_g1_mul(self._h, None if points is None else bytes(points), bytes(scalars), n,
                                                               int(reduce_scalar), out, status))
        return out.raw[:n * G1_BYTES], status.raw[:n]

    def batch_g2_mul(self, points, scalars, n, reduce_scalar=False):
        """points=None multiplies the G2 generator (PublicKey::from_private_key)."""
        out = ctypes.create_string_buffer(max(n, 1) * G2_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_g2_mul", self._lib.bn254_batch_g2_mul(self._h, None if points is None else bytes(points), bytes(scalars), n,
                                                               int(reduce_scalar), out, status))
        return out.raw[:n * G2_BYTES], status.raw[:n]

    def batch_sign(self, messages, sks):
        n = len(messages)
        msgs, off = pack_messages(messages)
        sigs = ctypes.create_string_buffer(max(n, 1) * G1_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_sign", self._lib.bn254_batch_sign(self._h, msgs, off, bytes(sks), n, sigs, status))
        return sigs.raw[:n * G1_BYTES], status.raw[:n]

    def _sum(self, name, points, seg_off, size):
        n = len(seg_off) - 1
        off = (ctypes.c_uint64 * (n + 1))(*seg_off)
        out = ctypes.create_string_buffer(max(n, 1) * size)
        status = ctypes.create_string_buffer(max(n, 1))
        _check(name, getattr(self._lib, name)(self._h, bytes(points), off, n, out, status))
        return out.raw[:n * size], status.raw[:n]

    def batch_g1_sum(self, points, seg_off):
        return self._sum("bn254_batch_g1_sum", points, seg_off, G1_BYTES)

    def batch_g2_sum(self, points, seg_off):
        return self._sum("bn254_batch_g2_sum", points, seg_off, G2_BYTES)

    def batch_aggregate_verify(self, messages, pk_pool, sig_pool, tuple_msg, signer_lists, flags=0):
        """messages: M byte strings; pk_pool: S*128 B; sig_pool: M*S*64 B (signer s on message m at m*S+s);
        tuple i verifies messages[tuple_msg[i]] against the aggregate of signer_lists[i]."""
        n, n_msgs = len(tuple_msg), len(messages)
        n_signers = len(pk_pool) // G2_BYTES
        assert len(sig_pool) == n_msgs * n_signers * G1_BYTES and len(signer_lists) == n
        msgs, off = pack_messages(messages)
        t_off = (ctypes.c_uint64 * (n + 1))()
        flat = []
        for i, lst in enumerate(signer_lists):
            t_off[i] = len(flat)
            flat.extend(lst)
        t_off[n] = len(flat)
        idx = (ctypes.c_uint32 * max(len(flat), 1))(*flat)
        tm = (ctypes.c_uint32 * max(n, 1))(*tuple_msg)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_aggregate_verify",
               self._lib.bn254_batch_aggregate_verify(self._h, msgs, off, n_msgs, bytes(pk_pool), n_signers, bytes(sig_pool), tm, t_off, idx, n, flags, status))
        return status.raw[:n]

    def register_pools(self, messages, pk_pool, sig_pool, expect_tuples, flags=0):
        """the pools of an aggregate verify decoded, hashed and tabulated ONCE (bn254_ctx_register_pools): for a fixed validator set / message
        set whose tuples keep arriving; `expect_tuples` = the batch size the subset-sum tables are chosen for"""
        n_msgs = len(messages)
        n_signers = len(pk_pool) // G2_BYTES
        assert len(sig_pool) == n_msgs * n_signers * G1_BYTES
        msgs, off = pack_messages(messages)
        _check("bn254_ctx_register_pools",
               self._lib.bn254_ctx_register_pools(self._h, msgs, off, n_msgs, bytes(pk_pool), n_signers, bytes(sig_pool), flags, expect_tuples))

    def batch_aggregate_verify_registered(self, tuple_msg, signer_lists):
        """as batch_aggregate_verify on the registered pools: only the tuples cross the boundary"""
        n = len(tuple_msg)
        assert len(signer_lists) == n
        t_off = (ctypes.c_uint64 * (n + 1))()
        flat = []
        for i, lst in enumerate(signer_lists):
            t_off[i] = len(flat)
            flat.extend(lst)
        t_off[n] = len(flat)
        idx = (ctypes.c_uint32 * max(len(flat), 1))(*flat)
        tm = (ctypes.c_uint32 * max(n, 1))(*tuple_msg)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_aggregate_verify_registered", self._lib.bn254_batch_aggregate_verify_registered(self._h, tm, t_off, idx, n, status))
        return status.raw[:n]

    def register_pools_device(self, d_msgs, d_msg_off, n_msgs, d_pk_pool, n_signers, d_sig_pool, expect_tuples, flags=0, stream=None):
        _check("bn254_ctx_register_pools_device",
               self._lib.bn254_ctx_register_pools_device(self._h, d_msgs, d_msg_off, n_msgs, d_pk_pool, n_signers, d_sig_pool, flags, expect_tuples, stream))

    def batch_aggregate_verify_registered_device(self, d_tuple_msg, d_tuple_off, d_signer_idx, n, d_status, stream=None):
        _check("bn254_batch_aggregate_verify_registered_device",
               self._lib.bn254_batch_aggregate_verify_registered_device(self._h, d_tuple_msg, d_tuple_off, d_signer_idx, n, d_status, stream))

    def batch_g1_decompress(self, data, n):
        out = ctypes.create_string_buffer(max(n, 1) * G1_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_g1_decompress", self._lib.bn254_batch_g1_decompress(self._h, bytes(data), n, out, status))
        return out.raw[:n * G1_BYTES], status.raw[:n]

    def batch_g2_decompress(self, data, n):
        out = ctypes.create_string_buffer(max(n, 1) * G2_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_g2_decompress", self._lib.bn254_batch_g2_decompress(self._h, bytes(data), n, out, status))
        return out.raw[:n * G2_BYTES], status.raw[:n]

    # ---- test hooks -----------------------------------------------------------------------
    def debug_fp_op(self, op, a, b, n):
        out = ctypes.create_string_buffer(max(n, 1) * 32)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_debug_fp_op", self._lib.bn254_debug_fp_op(self._h, op, bytes(a), None if b is None else bytes(b), n, out, status))
        return out.raw[:n * 32], status.raw[:n]

    def debug_fp12_op(self, op, a, b, n):
        out = ctypes.create_string_buffer(max(n, 1) * GT_BYTES)
        _check("bn254_debug_fp12_op", self._lib.bn254_debug_fp12_op(self._h, op, bytes(a), None if b is None else bytes(b), n, out))
        return out.raw[:n * GT_BYTES]

    def debug_final_exp_limbs(self, layout, limbs, n, want_gt=False):
        """final exponentiation of layout 0..6 on n x 108 int32 limbs (include/bn254_hip.h) -> (Gt bytes or None, status bytes)"""
        assert len(limbs) == n * 108
        arr = (ctypes.c_int32 * max(len(limbs), 1))(*limbs)
        gt = ctypes.create_string_buffer(max(n, 1) * GT_BYTES) if want_gt else None
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_debug_final_exp_limbs", self._lib.bn254_debug_final_exp_limbs(self._h, layout, arr, n, gt, status))
        return (gt.raw[:n * GT_BYTES] if want_gt else None), status.raw[:n]

    def debug_miller_loop(self, g1s, g2s, n):
        out = ctypes.create_string_buffer(max(n, 1) * GT_BYTES)
        _check("bn254_debug_miller_loop", self._lib.bn254_debug_miller_loop(self._h, bytes(g1s), bytes(g2s), n, out))
        return out.raw[:n * GT_BYTES]

    def debug_hash_candidate(self, h, n):
        out = ctypes.create_string_buffer(max(n, 1) * G1_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_debug_hash_candidate", self._lib.bn254_debug_hash_candidate(self._h, bytes(h), n, out, status))
        return out.raw[:n * G1_BYTES], status.raw[:n]

    # ---- device-pointer entry points (inputs resident in HBM; enqueue only) ---------------
    def register_keys(self, pks, flags=0):
        """replace the context's registered key set (uncompressed G2, 128 B each); returns the per-key status bytes
        (what PublicKey::from_uncompressed reports, subgroup check included)"""
        n = len(pks) // G2_BYTES
        assert len(pks) == n * G2_BYTES
        st = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_ctx_register_keys", self._lib.bn254_ctx_register_keys(self._h, bytes(pks), n, flags, st))
        return st.raw[:n]

    def batch_verify_keyed(self, messages, sigs, key_idx, flags=0):
        n = len(messages)
        assert len(sigs) == n * G1_BYTES and len(key_idx) == n
        msgs, off = pack_messages(messages)
        idx = (ctypes.c_uint32 * max(n, 1))(*key_idx)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_verify_keyed", self._lib.bn254_batch_verify_keyed(self._h, msgs, off, bytes(sigs), idx, n, flags, status))
        return status.raw[:n]

    def batch_verify_keyed_randomized(self, messages, sigs, key_idx, seed32, flags=0):
        n = len(messages)
        assert len(sigs) == n * G1_BYTES and len(key_idx) == n and len(seed32) == 32
        msgs, off = pack_messages(messages)
        idx = (ctypes.c_uint32 * max(n, 1))(*key_idx)
        status = ctypes.create_string_buffer(max(n, 1))
        _check("bn254_batch_verify_keyed_randomized",
               self._lib.bn254_batch_verify_keyed_randomized(self._h, msgs, off, bytes(sigs), idx, n, flags, bytes(seed32), status))
        return status.raw[:n]

    def batch_verify_keyed_randomized_device(self, d_msgs, d_off, d_sigs, d_key_idx, n, seed32, d_status, flags=0, stream=None):
        assert len(seed32) == 32
        _check("bn254_batch_verify_keyed_randomized_device",
               self._lib.bn254_batch_verify_keyed_randomized_device(self._h, d_msgs, d_off, d_sigs, d_key_idx, n, flags, bytes(seed32), d_status, stream))

    def batch_verify_keyed_device(self, d_msgs, d_off, d_sigs, d_key_idx, n, d_status, flags=0, stream=None):
        _check("bn254_batch_verify_keyed_device",
               self._lib.bn254_batch_verify_keyed_device(self._h, d_msgs, d_off, d_sigs, d_key_idx, n, flags, d_status, stream))

    def batch_verify_device(self, d_msgs, d_off, d_sigs, d_pks, n, d_status, flags=0, stream=None):
        _check("bn254_batch_verify_device",
               self._lib.bn254_batch_verify_device(self._h, d_msgs, d_off, d_sigs, d_pks, n, flags, d_status, stream))

    def batch_verify_randomized_device(self, d_msgs, d_off, d_sigs, d_pks, n, seed32, d_status, d_group_ok=None, flags=0, stream=None):
        assert len(seed32) == 32
        _check("bn254_batch_verify_randomized_device",
               self._lib.bn254_batch_verify_randomized_device(self._h, d_msgs, d_off, d_sigs, d_pks, n, flags, bytes(seed32), d_status, d_group_ok, stream))

    def batch_hash_to_g1_device(self, d_msgs, d_off, n, d_points, d_status, d_tries=None, stream=None):
        _check("bn254_batch_hash_to_g1_device",
               self._lib.bn254_batch_hash_to_g1_device(self._h, d_msgs, d_off, n, d_points, d_status, d_tries, stream))

    def batch_pairing_device(self, d_g1, d_g2, n, k, d_gt, d_status, flags=0, stream=None):
        _check("bn254_batch_pairing_device", self._lib.bn254_batch_pairing_device(self._h, d_g1, d_g2, n, k, flags, d_gt, d_status, stream))

    def batch_aggregate_verify_device(self, d_msgs, d_msg_off, n_msgs, d_pk_pool, n_signers, d_sig_pool, d_tuple_msg, d_tuple_off, d_signer_idx, n,
                                      d_status, flags=0, stream=None):
        _check("bn254_batch_aggregate_verify_device",
               self._lib.bn254_batch_aggregate_verify_device(self._h, d_msgs, d_msg_off, n_msgs, d_pk_pool, n_signers, d_sig_pool, d_tuple_msg,
                                                             d_tuple_off, d_signer_idx, n, flags, d_status, stream))

    def batch_sign_device(self, d_msgs, d_off, d_sks, n, d_sigs, d_status, stream=None):
        _check("bn254_batch_sign_device", self._lib.bn254_batch_sign_device(self._h, d_msgs, d_off, d_sks, n, d_sigs, d_status, stream))

    def batch_g2_mul_device(self, d_points, d_scalars, n, d_out, d_status, reduce_scalar=False, stream=None):
        _check("bn254_batch_g2_mul_device",
               self._lib.bn254_batch_g2_mul_device(self._h, d_points, d_scalars, n, int(reduce_scalar), d_out, d_status, stream))

    def batch_g1_mul_device(self, d_points, d_scalars, n, d_out, d_status, reduce_scalar=False, stream=None):
        _check("bn254_batch_g1_mul_device",
               self._lib.bn254_batch_g1_mul_device(self._h, d_points, d_scalars, n, int(reduce_scalar), d_out, d_status, stream))


MGPU_OPT_GATHER = 1
MGPU_OPT_TIMING = 2
MGPU_GATHER_AUTO, MGPU_GATHER_RCCL, MGPU_GATHER_COPY = 0, 1, 2


def shard_range(n_total, g, n_dev):
    """contiguous slice [lo, hi) of ceil(n_total / n_dev) items owned by entry g — the arithmetic of bn254_mgpu_shard_range
    (include/bn254_hip.h), restated here so that it can be checked without a GPU"""
    per = (n_total + n_dev - 1) // n_dev
    lo = min(n_total, g * per)
    return lo, min(n_total, lo + per)


class _EngineView(Engine):
    """an Engine over a context OWNED by a MultiEngine (options, reserve, register_keys per device); never destroys it"""

    def __init__(self, lib, handle, device):
        self._lib, self._h, self.device = lib, handle, device

    def close(self):
        self._h = None


class MultiEngine:
    """The batch sharded over the GPUs of one node from ONE process (include/bn254_hip.h, section "Multi-GPU"): one context, one
    stream and one parked worker thread per entry of `devices`; the only exchange between devices is the gather of the status bytes
    (RCCL's C API; peer copies when a device is listed twice)."""

    def __init__(self, devices):
        self._lib = _native.load()
        self.devices = list(devices)
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        _check("bn254_mgpu_create", self._lib.bn254_mgpu_create(arr, len(self.devices), ctypes.byref(h)))
        self._h = h
        self.n_dev = len(self.devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bn254_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, fn, rc):
        if rc == -10004:
            raise NativeError(fn + " [" + self._lib.bn254_mgpu_last_error(self._h).decode(errors="replace") + "]", rc)
        _check(fn, rc)

    def engine(self, g):
        return _EngineView(self._lib, ctypes.c_void_p(self._lib.bn254_mgpu_ctx(self._h, g)), self.devices[g])

    def shard_len(self, n):
        return self._lib.bn254_mgpu_shard_len(self._h, n)

    def gathered_len(self, n):
        return self._lib.bn254_mgpu_gathered_len(self._h, n)

    def shard_range(self, n, g):
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        _check("bn254_mgpu_shard_range", self._lib.bn254_mgpu_shard_range(self._h, n, g, ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value

    def reserve(self, n_total, init_collectives=False):
        self._check("bn254_mgpu_reserve", self._lib.bn254_mgpu_reserve(self._h, n_total, int(init_collectives)))

    def synchronize(self):
        _check("bn254_mgpu_synchronize", self._lib.bn254_mgpu_synchronize(self._h))

    def set_option(self, option, value):
        _check("bn254_mgpu_set_option", self._lib.bn254_mgpu_set_option(self._h, option, value))

    def last_timing(self):
        """per device: (compute ms, collective ms) of the last call (MGPU_OPT_TIMING on)"""
        a, b = (ctypes.c_float * self.n_dev)(), (ctypes.c_float * self.n_dev)()
        _check("bn254_mgpu_last_timing", self._lib.bn254_mgpu_last_timing(self._h, a, b))
        return list(a), list(b)

    # ---- host-pointer entry points: the whole batch in, the whole result out ---------------------------------------------
    def batch_verify(self, messages, sigs, pks, flags=0):
        n = len(messages)
        assert len(sigs) == n * G1_BYTES and len(pks) == n * G2_BYTES
        msgs, off = pack_messages(messages)
        status = ctypes.create_string_buffer(max(n, 1))
        self._check("bn254_mgpu_batch_verify", self._lib.bn254_mgpu_batch_verify(self._h, msgs, off, bytes(sigs), bytes(pks), n, flags, status))
        return status.raw[:n]

    def batch_hash_to_g1(self, messages):
        n = len(messages)
        msgs, off = pack_messages(messages)
        pts = ctypes.create_string_buffer(max(n, 1) * G1_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        tries = ctypes.create_string_buffer(max(n, 1))
        self._check("bn254_mgpu_batch_hash_to_g1", self._lib.bn254_mgpu_batch_hash_to_g1(self._h, msgs, off, n, pts, status, tries))
        return pts.raw[:n * G1_BYTES], status.raw[:n], tries.raw[:n]

    def batch_verify_compressed(self, messages, sigs33, pks65):
        n = len(messages)
        assert len(sigs33) == n * 33 and len(pks65) == n * 65
        msgs, off = pack_messages(messages)
        status = ctypes.create_string_buffer(max(n, 1))
        self._check("bn254_mgpu_batch_verify_compressed",
                    self._lib.bn254_mgpu_batch_verify_compressed(self._h, msgs, off, bytes(sigs33), bytes(pks65), n, status))
        return status.raw[:n]

    def register_keys(self, pks, flags=0):
        """the whole key set on every device -> per-key status bytes"""
        n = len(pks) // G2_BYTES
        st = ctypes.create_string_buffer(max(n, 1))
        self._check("bn254_mgpu_register_keys", self._lib.bn254_mgpu_register_keys(self._h, bytes(pks), n, flags, st))
        return st.raw[:n]

    def batch_verify_keyed(self, messages, sigs, key_idx, flags=0):
        n = len(messages)
        assert len(sigs) == n * G1_BYTES and len(key_idx) == n
        msgs, off = pack_messages(messages)
        idx = (ctypes.c_uint32 * max(n, 1))(*key_idx)
        status = ctypes.create_string_buffer(max(n, 1))
        self._check("bn254_mgpu_batch_verify_keyed", self._lib.bn254_mgpu_batch_verify_keyed(self._h, msgs, off, bytes(sigs), idx, n, flags, status))
        return status.raw[:n]

    def batch_aggregate_verify(self, messages, pk_pool, sig_pool, tuple_msg, signer_lists, flags=0):
        """as Engine.batch_aggregate_verify, the tuples sharded over the devices"""
        n, n_msgs = len(tuple_msg), len(messages)
        n_signers = len(pk_pool) // G2_BYTES
        assert len(sig_pool) == n_msgs * n_signers * G1_BYTES and len(signer_lists) == n
        msgs, off = pack_messages(messages)
        t_off = (ctypes.c_uint64 * (n + 1))()
        flat = []
        for i, lst in enumerate(signer_lists):
            t_off[i] = len(flat)
            flat.extend(lst)
        t_off[n] = len(flat)
        idx = (ctypes.c_uint32 * max(len(flat), 1))(*flat)
        tm = (ctypes.c_uint32 * max(n, 1))(*tuple_msg)
        status = ctypes.create_string_buffer(max(n, 1))
        self._check("bn254_mgpu_batch_aggregate_verify",
                    self._lib.bn254_mgpu_batch_aggregate_verify(self._h, msgs, off, n_msgs, bytes(pk_pool), n_signers, bytes(sig_pool), tm, t_off, idx, n,
                                                                flags, status))
        return status.raw[:n]

    def batch_pairing(self, g1s, g2s, n, k=1, flags=0):
        """-> (Gt bytes, status bytes, checksum = sum mod 2^64 of the little-endian 64-bit words of all Gt bytes)"""
        assert len(g1s) == n * k * G1_BYTES and len(g2s) == n * k * G2_BYTES
        gt = ctypes.create_string_buffer(max(n, 1) * GT_BYTES)
        status = ctypes.create_string_buffer(max(n, 1))
        cs = ctypes.c_uint64(0)
        self._check("bn254_mgpu_batch_pairing",
                    self._lib.bn254_mgpu_batch_pairing(self._h, bytes(g1s), bytes(g2s), n, k, flags, gt, status, ctypes.byref(cs)))
        return gt.raw[:n * GT_BYTES], status.raw[:n], cs.value

    # ---- device-pointer entry points: per-device lists of raw device pointers (ints), enqueue only ---------------------
    def _ptrs(self, seq):
        if seq is None:
            return None
        assert len(seq) == self.n_dev
        return (ctypes.c_void_p * self.n_dev)(*[ctypes.c_void_p(p) if p else None for p in seq])

    def batch_verify_device(self, d_msgs, d_off, d_sigs, d_pks, n, d_status_all, flags=0, streams=None):
        self._check("bn254_mgpu_batch_verify_device",
                    self._lib.bn254_mgpu_batch_verify_device(self._h, self._ptrs(d_msgs), self._ptrs(d_off), self._ptrs(d_sigs), self._ptrs(d_pks),
                                                             n, flags, self._ptrs(d_status_all), self._ptrs(streams)))

    def batch_pairing_device(self, d_g1, d_g2, n, k, d_gt, d_status_all, d_checksum=None, flags=0, streams=None):
        self._check("bn254_mgpu_batch_pairing_device",
                    self._lib.bn254_mgpu_batch_pairing_device(self._h, self._ptrs(d_g1), self._ptrs(d_g2), n, k, flags, self._ptrs(d_gt),
                                                              self._ptrs(d_status_all), self._ptrs(d_checksum), self._ptrs(streams)))


_default = {}


def default_engine(device=0):
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]
