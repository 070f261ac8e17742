#!/usr/bin/env python3
"""The reference's own spmdm flow on ONE large problem (samples/spmdm/spmdm.c defaults: M = N = K = 2048, 15 % non-zeros kept by
`r > 0.85`): libxsmm_spmdm_init, createSparseSlice_fp32_thread per block, compute_fp32_thread per block -- device operands.
Timed: the per-block loops (one launch per block id) and the one-call extension libxsmm_amd_spmdm_*_all.
usage: python3 tools/bench_spmdm_api.py [n=2048] [density=0.15] [reps=10]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
density = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
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
def timed(fn, reps_):
    """average over reps_ back-to-back calls, HIP events on the engine's stream (the default stream)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps_):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps_


def create_blocks():
    for blk in range(nc):
        L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), b"N", xs.dptr(a), slices, blk, 0, 1)


def compute_blocks():
    for blk in range(nx):
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(b), b"N", C.byref(beta), xs.dptr(c), blk, 0, 1)


def create_all():
    L.libxsmm_amd_spmdm_createSparseSlice_all(C.byref(h), b"N", xs.dptr(a), slices)


def compute_all():
    L.libxsmm_amd_spmdm_compute_all(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(b), b"N", C.byref(beta), xs.dptr(c))


def bracketed(fn):
    """the same block calls inside libxsmm_amd_defer_begin/end: recorded, launched as one range when the bracket ends"""
    def run():
        L.libxsmm_amd_defer_begin(); fn(); L.libxsmm_amd_defer_end()
    return run


# the bfloat16 twins of the two loops (include/libxsmm_spmdm.h:98-133): operands are upper halves of floats
a16 = (a.view(torch.int32) >> 16).to(torch.int16); b16 = (b.view(torch.int32) >> 16).to(torch.int16)
alpha16, beta16 = C.c_ushort(0x3F80), C.c_ushort(0)


def create_blocks_bf16():
    for blk in range(nc):
        L.libxsmm_spmdm_createSparseSlice_bfloat16_thread(C.byref(h), b"N", xs.dptr(a16), slices, blk, 0, 1)


def compute_blocks_bf16():
    for blk in range(nx):
        L.libxsmm_spmdm_compute_bfloat16_thread(C.byref(h), b"N", b"N", C.byref(alpha16), slices, xs.dptr(b16), b"N", C.byref(beta16), xs.dptr(c), blk, 0, 1)


L.libxsmm_amd_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
nnz = float((a != 0).sum().item())
print("spmdm %d^3 density %.2f: bm=%d bn=%d bk=%d, %d create blocks, %d compute blocks" % (n, density, h.bm, h.bn, h.bk, nc, nx))
for name, fn, flops in (("createSparseSlice per block", create_blocks, 0.0), ("createSparseSlice_all", create_all, 0.0),
                        ("compute per block", compute_blocks, 2.0 * nnz * N), ("compute_all", compute_all, 2.0 * nnz * N),
                        ("create per block, bracket", bracketed(create_blocks), 0.0), ("compute per block, bracket", bracketed(compute_blocks), 2.0 * nnz * N),
                        ("bf16 create per block", create_blocks_bf16, 0.0), ("bf16 compute per block", compute_blocks_bf16, 2.0 * nnz * N),
                        ("bf16 create per block, bracket", bracketed(create_blocks_bf16), 0.0), ("bf16 compute per block, bracket", bracketed(compute_blocks_bf16), 2.0 * nnz * N)):
    ms = timed(fn, reps)
    extra = ("  %.0f GFLOP/s (sparse flops), %.0f GFLOP/s dense-equivalent" % (flops / ms / 1e6, 2.0 * M * N * K / ms / 1e6)) if flops else ("  %.0f GB/s of A" % (4.0 * M * K / ms / 1e6))
    print("  %-34s %8.3f ms  [%s]%s" % (name, ms, xs.last_kernel(), extra))
L.libxsmm_spmdm_destroy(C.byref(h))
