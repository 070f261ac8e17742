#!/usr/bin/env python3
"""Developer benchmark of libxsmm_blocked_gemm_st (reference samples/blocked_gemm/blocked_gemm.c: one big GEMM in block layout,
every C block a chain over its k blocks): usage python3 tools/bench_blocked.py [m=2048] [bs=32] [f32|f64] [reps=5]"""
import ctypes as C
import importlib
import os
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
f64 = (sys.argv[3] == "f64") if len(sys.argv) > 3 else False
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
torch.cuda.set_device(0)
dt = torch.float64 if f64 else torch.float32
prec = xs.F64 if f64 else xs.F32
n = k = m
ib, one, order = C.c_int(bs), C.c_int(1), C.c_int(0)
al = (C.c_double if f64 else C.c_float)(1.0); be = (C.c_double if f64 else C.c_float)(1.0)
h = L.libxsmm_blocked_gemm_handle_create(1, prec, prec, m, n, k, C.byref(ib), C.byref(ib), C.byref(ib), C.byref(one), C.byref(one), C.byref(one), C.byref(one),
                                         C.byref(al), C.byref(be), None, None, C.byref(order))
assert h
a = torch.rand(m * k, device="cuda", dtype=dt) - 0.5; b = torch.rand(k * n, device="cuda", dtype=dt) - 0.5; c = torch.zeros(m * n, device="cuda", dtype=dt)
ba, bb, bc = torch.empty_like(a), torch.empty_like(b), torch.empty_like(c)
ld = C.c_int(m)
assert 0 == L.libxsmm_blocked_gemm_copyin_a(h, xs.dptr(a), C.byref(ld), xs.dptr(ba))
assert 0 == L.libxsmm_blocked_gemm_copyin_b(h, xs.dptr(b), C.byref(ld), xs.dptr(bb))
assert 0 == L.libxsmm_blocked_gemm_copyin_c(h, xs.dptr(c), C.byref(ld), xs.dptr(bc))
ts = []
for it in range(reps + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.libxsmm_blocked_gemm_st(h, xs.dptr(ba), xs.dptr(bb), xs.dptr(bc), 0, 0); e1.record(); torch.cuda.synchronize()
    if it >= 2:
        ts.append(e0.elapsed_time(e1))
t = min(ts)
# traffic as the reference's block loop sees it: every product reads an A and a B block, every C block is read and written once
nb = m // bs
byt = (8 if f64 else 4) * (2.0 * nb ** 3 * bs * bs + 2.0 * nb * nb * bs * bs)
print("blocked_gemm %s %dx%dx%d blocks of %d: %s  %.3f ms  %.0f GFLOP/s  (block traffic %.0f GB/s, mostly served by L2)"
      % ("f64" if f64 else "f32", m, n, k, bs, xs.last_kernel(), t, 2.0 * m * n * k / t / 1e6, byt / t / 1e6))
L.libxsmm_blocked_gemm_handle_destroy(h)
