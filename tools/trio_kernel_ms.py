#!/usr/bin/env python3
"""per-kernel times (HIP events) of a small batch in the pair and the octet layout"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bn254_amd
from bn254_amd.engine import OPT_TRIO_MAX_BATCH, OPT_TRIO_WAVE_ROLES
from tests.datagen import make_verify_batch
eng = bn254_amd.Engine(0)
dev = torch.device("cuda", 0)
for n in [int(x) for x in os.environ.get("TRIO_SIZES", "64,1024,8192").split(",")]:
    msgs, sigs, pks, expected = make_verify_batch(eng, n)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_msgs, d_sigs, d_pks = t(b"".join(msgs)), t(sigs), t(pks)
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    eng.set_profiling(True)
    for name, lim, roles in (("pair", 0, 0), ("octet", 1 << 20, 0), ("roles", 1 << 20, 1), ("roles8", 1 << 20, 2)):
        eng.set_option(OPT_TRIO_MAX_BATCH, lim)
        eng.set_option(OPT_TRIO_WAVE_ROLES, roles)
        best = None
        for _ in range(4):
            with torch.cuda.stream(stream):
                eng.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_st.data_ptr(), flags=0, stream=stream.cuda_stream)
            ms = eng.last_kernel_ms()
            if best is None or sum(ms.values()) < sum(best.values()):
                best = ms
        assert bytes(d_st.cpu().numpy()) == expected
        print(json.dumps({"n": n, "layout": name, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
