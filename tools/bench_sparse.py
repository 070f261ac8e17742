#!/usr/bin/env python3
"""Developer benchmark of the sparse batch kernels only (spmdm create/compute at the BASELINE config-4 shape, fsspmdm at
config 3). Small and quick, meant to be run under rocprofv3 as well:  python3 tools/bench_sparse.py [spmdm|fsspmdm] [reps]"""
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
what = sys.argv[1] if len(sys.argv) > 1 else "spmdm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.cuda.set_device(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)


def timeit(fn, n):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sorted(ts)[len(ts) // 2]


if what == "spmdm":
    M, N, K = 64, 48, 64
    B = int(os.environ.get("SP_BATCH", "131072"))
    density = float(os.environ.get("SP_DENSITY", "0.5"))
    a = torch.rand(B * M * K, device="cuda", generator=g) - 0.5
    a = torch.where(torch.rand(B * M * K, device="cuda", generator=g) < density, a, torch.zeros_like(a))
    if os.environ.get("SP_PLACED", "1") != "0":  # B and C out of one allocation, C 8 KiB off (see profiles/r2_headline_placement.txt)
        nb_, nc_ = B * K * N, B * M * N
        ob_ = (nb_ + 63) // 64 * 64 + 2048
        pool = torch.rand(ob_ + nc_, device="cuda", generator=g) - 0.5
        b = pool[0:nb_]; c = pool[ob_:ob_ + nc_]; c.zero_()
    else:
        b = torch.rand(B * K * N, device="cuda", generator=g) - 0.5
        c = torch.zeros(B * M * N, device="cuda")
    sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, B)
    for beta_v in (0.0, 1.0):
        beta = C.c_float(beta_v)
        mn, md = timeit(lambda: L.libxsmm_amd_spmdm_batch_create_slices(sb, b"N", xs.dptr(a)), reps)
        nnz = float((a != 0).sum().item()) / B
        by = 4.0 * M * K + 6.0 * nnz + 2.0 * (M + 1)
        print("create  %-28s min %.4f ms median %.4f ms  %.0f GB/s" % (xs.last_kernel(), mn, md, B * by / mn / 1e6))
        mn, md = timeit(lambda: L.libxsmm_amd_spmdm_batch_compute(sb, b"N", xs.dptr(b), b"N", C.byref(beta), xs.dptr(c)), reps)
        by = 6.0 * nnz + 2.0 * (M + 1) + 4.0 * K * N + 4.0 * M * N * (2 if beta_v else 1)
        print("compute %-28s beta=%g nnz/item=%.0f min %.4f ms median %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)  %.0f GFLOP/s"
              % (xs.last_kernel(), beta_v, nnz, mn, md, B * by / mn / 1e6, B * by / mn / 1e6 / 80.0, B * 2 * nnz * N / mn / 1e6))
    L.libxsmm_amd_spmdm_batch_destroy(sb)
else:
    M, K, N = 35, 35, 96
    B = int(os.environ.get("SP_BATCH", "65536"))
    rng = np.random.default_rng(1)
    pal = np.array([0.25, -0.5, 0.75, 1.0, -1.25, 1.5, -2.0])
    A = np.where(rng.random((M, K)) < 0.15, pal[rng.integers(0, 7, (M, K))], 0.0)
    ntot = N * B
    for dtype, tdt, create, run, destroy in ((np.float64, torch.float64, L.libxsmm_dfsspmdm_create, L.libxsmm_amd_dfsspmdm_execute_batch, L.libxsmm_dfsspmdm_destroy),
                                             (np.float32, torch.float32, L.libxsmm_sfsspmdm_create, L.libxsmm_amd_sfsspmdm_execute_batch, L.libxsmm_sfsspmdm_destroy)):
        # SP_LDPAD: elements added to the leading dimension of the panels. With ld = ntot = 3 * 2^21 (65 536 items of 96 columns) the 35
        # rows of B and the 35 rows of C a thread touches all sit at the same offset modulo 2^24 bytes and more
        ld = ntot + int(os.environ.get("SP_LDPAD", "0"))
        Bm = torch.rand(K * ld, device="cuda", dtype=tdt, generator=g) - 0.5
        Cm = torch.zeros(M * ld, device="cuda", dtype=tdt)
        es = Bm.element_size()
        for beta in (1.0, 0.0):
            h = create(M, N, K, K, ld, ld, 1.0, beta, xs.dptr(np.ascontiguousarray(A.astype(dtype))))
            mn, md = timeit(lambda: run(h, xs.dptr(Bm), xs.dptr(Cm), B), reps)
            by = es * N * (K + (2 if beta else 1) * M)
            print("fsspmdm %-26s %s beta=%g min %.4f ms median %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)"
                  % (xs.last_kernel(), np.dtype(dtype).name, beta, mn, md, B * by / mn / 1e6, B * by / mn / 1e6 / 80.0))
            destroy(h)
        del Bm, Cm
