#!/usr/bin/env python3
"""libxsmm_mmbatch_begin / N x libxsmm_dgemm on HOST matrices / libxsmm_mmbatch_end (the CP2K inner-loop style the reference's
auto-batching serves): wall time of recording and of the flush. usage: python3 tools/bench_autobatch.py [calls=20000]"""
import ctypes as C
import importlib
import os
import sys
import time
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
torch.cuda.set_device(0)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = n = k = 23
rng = np.random.default_rng(1)
a = rng.uniform(-1, 1, (calls, m * k)); b = rng.uniform(-1, 1, (calls, k * n)); c = np.zeros((calls, m * n))
im = C.c_int(m)
fn = L.libxsmm_dgemm
pa = [a[i].ctypes.data for i in range(calls)]; pb = [b[i].ctypes.data for i in range(calls)]; pc = [c[i].ctypes.data for i in range(calls)]
for rep in range(2):
    c[:] = 0
    t0 = time.perf_counter()
    L.libxsmm_mmbatch_begin(xs.F64, None, C.byref(im), C.byref(im), C.byref(im), None, None, None, None, None)
    for i in range(calls):
        fn(b"N", b"N", C.byref(im), C.byref(im), C.byref(im), None, pa[i], None, pb[i], None, None, pc[i], None)
    t1 = time.perf_counter()
    L.libxsmm_mmbatch_end()
    t2 = time.perf_counter()
    ref = np.einsum("bkm,bnk->bnm", a.reshape(calls, k, m), b.reshape(calls, n, k)).reshape(calls, -1)
    print("%d recorded dgemm calls on host matrices: recording %.1f ms (%.2f us per call, Python included), flush %.1f ms; max error %.1e"
          % (calls, (t1 - t0) * 1e3, (t1 - t0) * 1e6 / calls, (t2 - t1) * 1e3, np.max(np.abs(c - ref))))
