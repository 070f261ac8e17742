#!/usr/bin/env python3
"""Developer benchmark of the SOA kernels (EDGE/SeisSol-style element-local products, operands [row][col][v]):
the star-matrix product (A sparse 9 x 9, libxsmm_create_xcsr_soa) and a stiffness product (B sparse, CSC), fp64 and fp32,
batches of elements through libxsmm_amd_kernel_execute_batch. usage: python3 tools/bench_soa.py [elements=262144] [reps=5]"""
import ctypes as C
import importlib
import os
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
EDGE = os.path.join(ROOT, "tests", "golden", "mtx", "edge")
elems = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.cuda.set_device(0)


def read_mtx(path):
    rows = cols = 0; ent = []
    for line in open(path):
        if line.startswith("%") or not line.strip():
            continue
        t = line.split()
        if not rows:
            rows, cols = int(t[0]), int(t[1]); continue
        ent.append((int(t[0]) - 1, int(t[1]) - 1, float(t[2])))
    return rows, cols, ent


def compressed(rows, cols, ent, csr):
    major = rows if csr else cols
    ent = sorted(ent, key=(lambda e: (e[0], e[1])) if csr else (lambda e: (e[1], e[0])))
    ptr = np.zeros(major + 1, dtype=np.uint32); idx = np.zeros(len(ent), dtype=np.uint32); val = np.zeros(len(ent))
    for p, (r, c, v) in enumerate(ent):
        ptr[(r if csr else c) + 1] += 1; idx[p] = c if csr else r; val[p] = v
    return np.cumsum(ptr).astype(np.uint32), idx, val


def timed(fn):
    ts = []
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1))
    return min(ts)


for dtype, prec in ((np.float64, xs.F64), (np.float32, xs.F32)):
    v = L.libxsmm_amd_soa_width(prec); ts = np.dtype(dtype).itemsize
    tdt = torch.float64 if ts == 8 else torch.float32
    # (1) star matrix: C[9][20][v] += A(9x9 sparse) * B[9][20][v]
    r, c, ent = read_mtx(os.path.join(EDGE, "tet4_starMatrix_csr.mtx")); ptr, idx, val = compressed(r, c, ent, True)
    n = 20
    blob, d = xs.descriptor(prec, r, n, c, 0, n, n, 1.0, 1.0, 0, 0)
    fn = L.libxsmm_create_xcsr_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(val.astype(dtype)))
    assert fn
    dv = torch.from_numpy(val.astype(dtype)).cuda()
    B = torch.rand(elems * c * n * v, device="cuda", dtype=tdt); Cc = torch.zeros(elems * r * n * v, device="cuda", dtype=tdt)
    t = timed(lambda: L.libxsmm_amd_kernel_execute_batch(fn, xs.dptr(dv), xs.dptr(B), xs.dptr(Cc), c * n * v, r * n * v, elems))
    byt = elems * ts * v * n * (c + 2 * r)
    print("%s star matrix  %dx%d nnz %d, [.][%d][%d]: %s  %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)  %.0f GFLOP/s"
          % (np.dtype(dtype).name, r, c, len(val), n, v, xs.last_kernel(), t, byt / t / 1e6, byt / t / 1e6 / 80, 2.0 * len(val) * n * v * elems / t / 1e6))
    L.libxsmm_release_kernel(fn); del B, Cc
    # (2) stiffness: C[9][N][v] += A[9][K][v] * B(K x N sparse, CSC)
    r, c, ent = read_mtx(os.path.join(EDGE, "tet4_3_stiffT_0_csc.mtx")); ptr, idx, val = compressed(r, c, ent, False)
    m = 9
    blob, d = xs.descriptor(prec, m, c, r, r, 0, c, 1.0, 1.0, 0, 0)
    fn = L.libxsmm_create_xcsc_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(val.astype(dtype)))
    assert fn
    dv = torch.from_numpy(val.astype(dtype)).cuda()
    A = torch.rand(elems * m * r * v, device="cuda", dtype=tdt); Cc = torch.zeros(elems * m * c * v, device="cuda", dtype=tdt)
    t = timed(lambda: L.libxsmm_amd_kernel_execute_batch(fn, xs.dptr(A), xs.dptr(dv), xs.dptr(Cc), m * r * v, m * c * v, elems))
    used_rows = len(set(int(i) for i in idx))  # rows of B that hold an entry: only those columns of A are read
    used_cols = int(np.sum(np.diff(ptr.astype(np.int64)) > 0))  # columns of B with an entry: with beta = 1 a C column without one is
    # neither changed nor moved (the generated kernel stores what it loaded -- the compiler drops both), so it is not billed
    byt = elems * ts * v * m * (used_rows + 2 * used_cols)
    print("%s stiffness    %dx%d nnz %d, [9][.][%d]:  %s  %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)  %.0f GFLOP/s"
          % (np.dtype(dtype).name, r, c, len(val), v, xs.last_kernel(), t, byt / t / 1e6, byt / t / 1e6 / 80, 2.0 * len(val) * m * v * elems / t / 1e6))
    L.libxsmm_release_kernel(fn); del A, Cc
