#!/usr/bin/env python3
"""Odds and ends next to the hot path: blocked-GEMM layout copies (GB/s), batch-reduce kernel calls (us per call)."""
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


def ev(fn, reps=5):
    ts = []
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1))
    return min(ts)


for dt, prec, ts in ((torch.float32, xs.F32, 4), (torch.float64, xs.F64, 8)):
    m = 4096; bs = 32
    ib, one, order = C.c_int(bs), C.c_int(1), C.c_int(0)
    al = (C.c_double if ts == 8 else C.c_float)(1.0)
    h = L.libxsmm_blocked_gemm_handle_create(1, prec, prec, m, m, m, C.byref(ib), C.byref(ib), C.byref(ib), C.byref(one), C.byref(one), C.byref(one), C.byref(one),
                                             C.byref(al), C.byref(al), None, None, C.byref(order))
    a = torch.rand(m * m, device="cuda", dtype=dt); ba = torch.empty_like(a); ld = C.c_int(m)
    for name in ("copyin_a", "copyin_b", "copyin_c", "copyout_c", "transpose_b", "convert_b_to_a"):
        f = getattr(L, "libxsmm_blocked_gemm_" + name)
        t = ev(lambda: f(h, xs.dptr(a), C.byref(ld), xs.dptr(ba)))
        print("blocked_gemm_%-15s %s 4096^2: %.3f ms  %.0f GB/s" % (name, "f64" if ts == 8 else "f32", t, 2.0 * m * m * ts / t / 1e6))
    L.libxsmm_blocked_gemm_handle_destroy(h)

# batch-reduce kernel: C += sum_i A_i * B_i, called with `count` device pointers
m = n = k = 32; count = 64
L.libxsmm_dmmdispatch_reducebatch.restype = C.c_void_p
fn = L.libxsmm_dmmdispatch_reducebatch(m, n, k, None, None, None, None, None, None, None)
proto = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
a = torch.rand(count * m * k, device="cuda", dtype=torch.float64); b = torch.rand(count * k * n, device="cuda", dtype=torch.float64); c = torch.zeros(m * n, device="cuda", dtype=torch.float64)
pa = (C.c_void_p * count)(*[a.data_ptr() + i * m * k * 8 for i in range(count)]); pb = (C.c_void_p * count)(*[b.data_ptr() + i * k * n * 8 for i in range(count)])
cnt = C.c_ulonglong(count)
call = proto(fn)
call(pa, pb, c.data_ptr(), C.addressof(cnt)); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    call(pa, pb, c.data_ptr(), C.addressof(cnt))
torch.cuda.synchronize()
print("batch-reduce kernel 32^3 x %d (host pointer arrays, device matrices): %.1f us per call  [%s]" % (count, (time.perf_counter() - t0) * 1e6 / 200, xs.last_kernel()))
