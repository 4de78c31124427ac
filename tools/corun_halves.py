#!/usr/bin/env python3
"""do the kernels of two half batches on two streams (two contexts) co-run?  time: one 65 536 call vs two 32 768 calls enqueued back to back"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import torch
    import bn254_amd
    from tests.datagen import make_verify_batch
    dev = torch.device("cuda", 0)
    e1, e2 = bn254_amd.Engine(0), bn254_amd.Engine(0)
    n = 65536
    msgs, sigs, pks, expected = make_verify_batch(e1, n)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_msgs, d_sigs, d_pks = t(b"".join(msgs)), t(sigs), t(pks)
    d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    d_off2 = torch.arange(0, 32 * (n // 2 + 1), 32, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    h = n // 2
    def whole():
        e1.batch_verify_device(d_msgs.data_ptr(), d_off.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), n, d_st.data_ptr(), flags=0, stream=s1.cuda_stream)
    def halves():
        e1.batch_verify_device(d_msgs.data_ptr(), d_off2.data_ptr(), d_sigs.data_ptr(), d_pks.data_ptr(), h, d_st.data_ptr(), flags=0, stream=s1.cuda_stream)
        e2.batch_verify_device(d_msgs.data_ptr() + 32 * h, d_off2.data_ptr(), d_sigs.data_ptr() + 64 * h, d_pks.data_ptr() + 128 * h, h, d_st.data_ptr() + h, flags=0, stream=s2.cuda_stream)
    for name, fn in (("whole", whole), ("halves", halves), ("whole", whole), ("halves", halves)):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ok = bytes(d_st.cpu().numpy()) == expected
        print(json.dumps({"mode": name, "ms": round(1e3 * min(ts), 3), "ok": ok}), flush=True)


if __name__ == "__main__":
    main()
