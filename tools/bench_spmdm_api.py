#!/usr/bin/env python3
"""The reference's own spmdm flow on ONE large problem (samples/spmdm/spmdm.c defaults: M = N = K = 2048, 15 % non-zeros kept by
`r > 0.85`): libxsmm_spmdm_init, createSparseSlice_fp32_thread per block, compute_fp32_thread per block -- device operands.
usage: python3 tools/bench_spmdm_api.py [n=2048] [density=0.15] [reps=3]"""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
density = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.cuda.set_device(0)
M = N = K = n
g = torch.Generator(device="cuda"); g.manual_seed(1)
a = torch.rand(M * K, device="cuda", generator=g) - 0.5
a = torch.where(torch.rand(M * K, device="cuda", generator=g) < density, a, torch.zeros_like(a))
b = torch.rand(K * N, device="cuda", generator=g) - 0.5
c = torch.zeros(M * N, device="cuda")
h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
nc, nx = L.libxsmm_spmdm_get_num_createSparseSlice_blocks(C.byref(h)), L.libxsmm_spmdm_get_num_compute_blocks(C.byref(h))
alpha, beta = C.c_float(1.0), C.c_float(0.0)
for it in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for blk in range(nc):
        L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), b"N", xs.dptr(a), slices, blk, 0, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for blk in range(nx):
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(b), b"N", C.byref(beta), xs.dptr(c), blk, 0, 1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    nnz = float((a != 0).sum().item())
    print("spmdm %d^3 density %.2f: bm=%d bn=%d bk=%d  create %d blocks %.2f ms [%s]  compute %d blocks %.2f ms = %.0f GFLOP/s (sparse flops) [%s]"
          % (n, density, h.bm, h.bn, h.bk, nc, (t1 - t0) * 1e3, "", nx, (t2 - t1) * 1e3, 2.0 * nnz * N / (t2 - t1) / 1e9, xs.last_kernel()))
L.libxsmm_spmdm_destroy(C.byref(h))
