#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: the operands live in HOST memory, as an unchanged CPU caller has them.
(1) pageable memory (numpy): staged over PCIe by the library, synchronously; (2) libxsmm_malloc (pinned, GPU-mapped): processed
in place over PCIe, the call returns when the result is visible. Never reported as bench.py's `value`; DESIGN.md section 6."""
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
m = n = k = 32
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rng = np.random.default_rng(1)
idx = (np.arange(batch) * m * k).astype(np.int32)


def run(a, b, c):
    xs.gemm_batch(xs.F32, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, idx, idx, idx, batch)


def timed(a, b, c, reps=3):
    run(a, b, c); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(a, b, c); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


flops = 2.0 * m * n * k * batch; byt = 16384.0 * batch
a = rng.uniform(-1, 1, batch * m * k).astype(np.float32); b = rng.uniform(-1, 1, batch * k * n).astype(np.float32); c = np.zeros(batch * m * n, dtype=np.float32)
t = timed(a, b, c)
print("pageable host operands (staged):   %.1f ms  %.1f GFLOP/s  %.1f GB/s algorithmic  [%s]" % (t * 1e3, flops / t / 1e9, byt / t / 1e9, xs.last_kernel()))
L.libxsmm_malloc.restype = C.c_void_p; L.libxsmm_malloc.argtypes = [C.c_size_t]
L.libxsmm_free.argtypes = [C.c_void_p]
ptrs = [L.libxsmm_malloc(4 * batch * 1024) for _ in range(3)]
views = [np.ctypeslib.as_array((C.c_float * (batch * 1024)).from_address(p)) for p in ptrs]
views[0][:] = a; views[1][:] = b; views[2][:] = 0
t = timed(ptrs[0], ptrs[1], ptrs[2])
print("libxsmm_malloc operands (in place): %.1f ms  %.1f GFLOP/s  %.1f GB/s algorithmic  [%s]" % (t * 1e3, flops / t / 1e9, byt / t / 1e9, xs.last_kernel()))
for p in ptrs:
    L.libxsmm_free(p)
