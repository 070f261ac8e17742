#!/usr/bin/env python3
"""Where the shape-specialised kernels start to win over the pre-compiled generic one: strided batches of 32 ... 16384 items, fp64 23^3 and
fp32 32^3 (LIBXSMM_AMD_JIT_MINBATCH decides which of the two a batch gets).  python3 tools/bench_small_batches.py"""
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


def run(m, n, k, batch, dt, minbatch):
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = str(minbatch)
    ts = 8 if dt == torch.float64 else 4
    a = torch.rand(batch * m * k, device="cuda", dtype=dt); b = torch.rand(batch * k * n, device="cuda", dtype=dt); c = torch.zeros(batch * m * n, device="cuda", dtype=dt)
    blob, d = xs.descriptor(xs.F64 if ts == 8 else xs.F32, m, n, k, m, k, m, 1.0, 1.0, 0, 0)
    t = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); assert 0 == L.libxsmm_amd_gemm_batch_strided(d, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, batch); e1.record()
        torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    return min(t[2:]) * 1e3, xs.last_kernel()


for (m, n, k, dt) in ((23, 23, 23, torch.float64), (32, 32, 32, torch.float32), (13, 13, 13, torch.float64)):
    for batch in (32, 64, 128, 256, 512, 1024, 2048, 4096, 16384):
        tg, kg = run(m, n, k, batch, dt, 1 << 30)
        tj, kj = run(m, n, k, batch, dt, 1)
        print("%s %dx%dx%d batch %6d: generic %-24s %7.1f us   specialised %-24s %7.1f us" % (str(dt).split(".")[1], m, n, k, batch, kg, tg, kj, tj))
