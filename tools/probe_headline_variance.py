#!/usr/bin/env python3
"""Where does the run-to-run spread of the headline kernel (2.9 - 3.3 ms per launch) come from? (1) fresh allocations of the three
4 GiB arrays within one process (placement), (2) a long back-to-back run (clocks / temperature)."""
import importlib
import os
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
torch.cuda.set_device(0)
B, M = 1048576, 32
blob, desc = xs.descriptor(xs.F32, M, M, M)
L.libxsmm_amd_set_mfma(1)


def timed(a, b, c, n):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), 1024, 1024, 1024, B)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts


keep = []
for trial in range(5):
    a = torch.rand(B * 1024, device="cuda"); b = torch.rand(B * 1024, device="cuda"); c = torch.zeros(B * 1024, device="cuda")
    ts = timed(a, b, c, 12)[2:]
    print("allocation %d: a=%#x b=%#x c=%#x  min %.3f avg %.3f ms" % (trial, a.data_ptr(), b.data_ptr(), c.data_ptr(), min(ts), sum(ts) / len(ts)))
    pad = torch.empty((37 + 11 * trial) * 1024 * 1024 + 4096 * trial, dtype=torch.uint8, device="cuda")  # shifts where the next set lands
    keep.append(pad)
    del a, b, c
    torch.cuda.empty_cache()
a = torch.rand(B * 1024, device="cuda"); b = torch.rand(B * 1024, device="cuda"); c = torch.zeros(B * 1024, device="cuda")
ts = timed(a, b, c, 400)
for i in range(0, 400, 50):
    print("launches %3d-%3d: avg %.3f ms" % (i, i + 49, sum(ts[i:i + 50]) / 50))
