#!/usr/bin/env python3
"""How the pre-compiled generic kernel compares with the shape-specialised ones: the same fp64 23^3 batch (a) tight, specialised,
(b) tight, generic (LIBXSMM_AMD_JIT=0), (c) leading dimensions 24 (not tight), (d) K = 70, (e) a batch below the hiprtc threshold."""
import importlib
import os
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
torch.cuda.set_device(0)
L.libxsmm_amd_set_mfma(int(os.environ.get("XSMM_BENCH_MFMA", "0")))


def run(tag, m, n, k, lda, ldb, ldc, batch, dt=torch.float64, jit=True):
    if not jit:
        os.environ["LIBXSMM_AMD_JIT"] = "0"
    ts = 8 if dt == torch.float64 else 4
    a = torch.rand(batch * lda * k, device="cuda", dtype=dt); b = torch.rand(batch * ldb * n, device="cuda", dtype=dt); c = torch.zeros(batch * ldc * n, device="cuda", dtype=dt)
    blob, d = xs.descriptor(xs.F64 if ts == 8 else xs.F32, m, n, k, lda, ldb, ldc, 1.0, 1.0, 0, 0)
    t = []
    for it in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); assert 0 == L.libxsmm_amd_gemm_batch_strided(d, xs.dptr(a), xs.dptr(b), xs.dptr(c), lda * k, ldb * n, ldc * n, batch); e1.record()
        torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    tm = min(t[2:]); byt = batch * ts * (m * k + k * n + 2 * m * n)
    print("%-34s %-22s %8d items  %.3f ms  %6.0f GB/s (%4.1f%%)" % (tag, xs.last_kernel(), batch, tm, byt / tm / 1e6, byt / tm / 1e6 / 80))
    os.environ.pop("LIBXSMM_AMD_JIT", None)


B = 354442
run("f64 23^3 tight, specialised", 23, 23, 23, 23, 23, 23, B)
run("f64 23^3 tight, generic", 23, 23, 23, 23, 23, 23, B, jit=False)
run("f64 23^3 ld 24", 23, 23, 23, 24, 24, 24, B)
run("f64 23x23x70", 23, 23, 70, 23, 70, 23, 150000)
run("f64 23^3 tight, 10000 items", 23, 23, 23, 23, 23, 23, 10000)
run("f32 32^3 ld 40 (special kernel n/a)", 32, 32, 32, 40, 40, 40, 366210, torch.float32)
run("f64 13^3 ld 16", 13, 13, 13, 16, 16, 16, 1000000)
