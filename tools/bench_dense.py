#!/usr/bin/env python3
"""Developer benchmark of the dense batch kernels over shapes (BASELINE configs 1, 2, 5): strided batches in HBM.
usage: python3 tools/bench_dense.py [f64|f32|all] [reps]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
torch.cuda.set_device(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
TARGET_BYTES = float(os.environ.get("DENSE_BYTES", 6e9))  # traffic per launch


def run(dtype, m, n, k, mfma, beta=1.0):
    ts = 8 if dtype == torch.float64 else 4
    per_item = ts * (m * k + k * n + (2 if beta else 1) * m * n)
    batch = int(min(TARGET_BYTES / per_item, 4e6))
    a = torch.rand(batch * m * k, device="cuda", dtype=dtype, generator=g) - 0.5
    b = torch.rand(batch * k * n, device="cuda", dtype=dtype, generator=g) - 0.5
    c = torch.rand(batch * m * n, device="cuda", dtype=dtype, generator=g) - 0.5
    blob, desc = xs.descriptor(xs.F64 if ts == 8 else xs.F32, m, n, k, beta=beta)
    L.libxsmm_amd_set_mfma(mfma)
    times = []
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, batch)
        e1.record(); torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    t = min(times)
    gbs = batch * per_item / t / 1e6
    print("%-4s %2dx%2dx%2d beta=%g mfma=%d %-26s batch=%8d  %.3f ms  %6.0f GB/s (%4.1f%% of 8 TB/s)  %7.0f GFLOP/s"
          % ("f64" if ts == 8 else "f32", m, n, k, beta, mfma, xs.last_kernel(), batch, t, gbs, gbs / 80.0, 2.0 * m * n * k * batch / t / 1e6))
    del a, b, c


shapes = [(13, 13, 13), (23, 23, 23), (32, 32, 32), (13, 23, 32), (32, 13, 23), (64, 64, 64), (8, 8, 8), (16, 16, 16), (5, 5, 5)]
if os.environ.get("DENSE_SHAPES"):  # e.g. DENSE_SHAPES=5x5x5,13x13x13
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["DENSE_SHAPES"].split(",")]
for dt in ([torch.float64] if which == "f64" else [torch.float32] if which == "f32" else [torch.float64, torch.float32]):
    for (m, n, k) in shapes:
        for mfma in ((0, 1) if (dt == torch.float32 and (m, n, k) == (32, 32, 32)) else (0,)):
            run(dt, m, n, k, mfma)
if not os.environ.get("DENSE_SHAPES"):
    run(torch.float64, 23, 23, 23, 0, beta=0.0)
