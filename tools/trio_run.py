#!/usr/bin/env python3
"""a few device-resident verify calls of n items in the octet layout (argv: n [reps] [limit]) — the program to put under rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bn254_amd
from bn254_amd.engine import OPT_TRIO_MAX_BATCH
from tests.datagen import make_verify_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = bn254_amd.Engine(0)
if len(sys.argv) > 3:
    eng.set_option(OPT_TRIO_MAX_BATCH, int(sys.argv[3]))
dev = torch.device("cuda", 0)
msgs, sigs, pks, expected = make_verify_batch(eng, n)
t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_msgs, d_sigs, d_pks = t(b"".join(msgs)), t(sigs), t(pks)
d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
for _ in range(reps):
    eng.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_st.data_ptr(), flags=0)
    torch.cuda.synchronize()
assert bytes(d_st.cpu().numpy()) == expected
print("ok")
