#!/usr/bin/env python3
"""Does the relative placement of A, B and C matter? One buffer, the three arrays at chosen byte offsets from 4 GiB-spaced bases;
several fresh allocations per pattern (physical placement changes with every allocation)."""
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
B = 1048576
N = B * 1024  # floats per array
blob, desc = xs.descriptor(xs.F32, 32, 32, 32)
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
PATTERNS = (("0 / 0", (0, 0)), ("256 / 512", (256, 512)), ("1K / 2K", (1024, 2048)), ("4K / 8K", (4096, 8192)), ("8K / 16K", (8192, 16384)), ("12K / 20K", (12288, 20480)),
            ("16K / 32K", (16384, 32768)), ("64K / 128K", (65536, 131072)), ("4K+256 / 8K+512", (4352, 8704)), ("2M+4K / 4M+8K", (2101248, 4202496)))
for name, (ob, oc) in PATTERNS:
    res = []
    for trial in range(5):
        if True:
            buf = torch.empty(3 * N + 8 * 1048576, device="cuda")
            buf.uniform_(0, 1)
            a = buf[0:N]; b = buf[N + ob // 4:2 * N + ob // 4]; c = buf[2 * N + oc // 4:3 * N + oc // 4]
        ts = timed(a, b, c, 8)[2:]
        res.append(sum(ts) / len(ts))
        keep.append(torch.empty((53 + 17 * len(keep)) * 1048576, dtype=torch.uint8, device="cuda"))
        del a, b, c, buf
        torch.cuda.empty_cache()
    print("%-18s %s" % (name, "  ".join("%.3f" % r for r in res)))
