#!/usr/bin/env python3
"""Shapes beyond 32 with gaps in the leading dimensions: the element-wise build of the one-wave-per-item matrix-core kernel against
the work-group form (XSMM_SMMJIT_MFMA_WAVE=0)."""
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
L.libxsmm_amd_set_mfma(1)
for dt, prec, ts in ((torch.float64, xs.F64, 8), (torch.float32, xs.F32, 4)):
    for (m, n, k, lda, ldb, ldc) in ((48, 48, 48, 56, 56, 56), (43, 9, 27, 48, 32, 48), (40, 64, 17, 40, 17, 44), (64, 64, 64, 72, 64, 64)):
        sa, sb, sc = lda * k, ldb * n, ldc * n
        batch = int(4e9 / (ts * (sa + sb + 2 * sc)))
        a = torch.rand(batch * sa + 4096, device="cuda", dtype=dt) - 0.5; b = torch.rand(batch * sb + 8192, device="cuda", dtype=dt) - 0.5; c = torch.zeros(batch * sc, device="cuda", dtype=dt)
        blob, desc = xs.descriptor(prec, m, n, k, lda, ldb, ldc)
        ts_ = []
        for it in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), sa, sb, sc, batch)
            e1.record(); torch.cuda.synchronize()
            if it >= 2:
                ts_.append(e0.elapsed_time(e1))
        t = min(ts_); by = batch * ts * (m * k + k * n + 2 * m * n)
        print("%s %2dx%2dx%2d ld %d/%d/%d  %-24s %.3f ms  %.0f GB/s algorithmic (%.1f %%)" % ("f64" if ts == 8 else "f32", m, n, k, lda, ldb, ldc, xs.last_kernel(), t, by / t / 1e6, by / t / 1e6 / 80.0))
        del a, b, c
