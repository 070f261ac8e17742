#!/usr/bin/env python3
"""One libxsmm_?gemm call on device matrices (what an application relinked with --wrap=dgemm_ gets for its large products)."""
import ctypes as C
import importlib
import os
import sys
import time
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
torch.cuda.set_device(0)
for n in (int(v) for v in (sys.argv[1:] or ["256", "1024", "2048"])):
    for dt, fn, ct in ((torch.float64, L.libxsmm_dgemm, C.c_double), (torch.float32, L.libxsmm_sgemm, C.c_float)):
        a = torch.rand(n * n, device="cuda", dtype=dt); b = torch.rand(n * n, device="cuda", dtype=dt); c = torch.zeros(n * n, device="cuda", dtype=dt)
        i_n = C.c_int(n); al, be = ct(1.0), ct(0.0)
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn(b"N", b"N", C.byref(i_n), C.byref(i_n), C.byref(i_n), C.byref(al), xs.dptr(a), C.byref(i_n), xs.dptr(b), C.byref(i_n), C.byref(be), xs.dptr(c), C.byref(i_n))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ref = (a.view(n, n).T.double() @ b.view(n, n).T.double()).T.contiguous().view(-1)
        err = (c.double() - ref).abs().max().item() / ref.abs().max().item()
        print("%s %d^3: %s  %.3f ms  %.0f GFLOP/s  rel.err %.1e" % ("f64" if dt == torch.float64 else "f32", n, xs.last_kernel(), min(ts) * 1e3, 2.0 * n ** 3 / min(ts) / 1e9, err))
