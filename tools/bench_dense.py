#!/usr/bin/env python3
"""Developer benchmark of the dense batch kernels over shapes (BASELINE configs 1, 2, 5): strided batches in HBM.
usage: python3 tools/bench_dense.py [f64|f32|all] [reps]"""
import importlib
import os
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

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
    # one allocation, the arrays 8 KiB / 16 KiB off their natural spacing (operand arrays at the same offset modulo the memory
    # interleave cost up to 12 %: profiles/r2_headline_placement.txt)
    na, nb, nc, sk, al = batch * m * k, batch * k * n, batch * m * n, 8192 // ts, 256 // ts
    ob = (na + al - 1) // al * al + sk; oc = (ob + nb + al - 1) // al * al + sk  # every array starts on a 256-byte boundary, as a separate allocation would
    pool = torch.rand(oc + nc, device="cuda", dtype=dtype, generator=g) - 0.5
    a = pool[0:na]; b = pool[ob:ob + nb]; c = pool[oc:oc + nc]
    blob, desc = xs.descriptor(xs.F64 if ts == 8 else xs.F32, m, n, k, beta=beta)
    L.libxsmm_amd_set_mfma(mfma)
    assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, batch)
    L.libxsmm_amd_jit_wait()  # the specialised kernel is compiled on a helper thread: not while timing
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
    del a, b, c, pool


def run_lowp(kind, m, n, k):
    """kind: 'bf16f32' | 'i16i32' | 'bf16'; A in pairs of k, tight leading dimensions, beta = 1"""
    import ctypes as C
    osz = 2 if kind == "bf16" else 4
    per_item = 2 * (m * k + k * n) + 2 * osz * m * n
    batch = int(min(TARGET_BYTES / per_item, 4e6))
    a = torch.randint(-100, 100, (batch * m * k,), device="cuda", dtype=torch.int16, generator=g)
    b = torch.randint(-100, 100, (batch * k * n,), device="cuda", dtype=torch.int16, generator=g)
    if kind != "i16i32":  # small bf16 values: 0x3C00..0x3FFF are 2^-7 .. 2
        a = (a.abs() % 1024 + 0x3C00).to(torch.int16); b = (b.abs() % 1024 + 0x3C00).to(torch.int16)
    c = torch.zeros(batch * m * n, device="cuda", dtype=torch.int16 if kind == "bf16" else torch.int32)
    ip, op = {"bf16f32": (xs.BF16, xs.F32), "i16i32": (xs.I16, xs.I32), "bf16": (xs.BF16, xs.BF16)}[kind]
    blob = xs.DescriptorBlob()
    L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
    L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
    desc = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), ip, op, m, n, k, m, k, m, 1.0, 1.0, 0, 0)
    times = []
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, batch)
        e1.record(); torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    t = min(times); gbs = batch * per_item / t / 1e6
    print("%-7s %2dx%2dx%2d beta=1 %-22s batch=%8d  %.3f ms  %6.0f GB/s (%4.1f%% of 8 TB/s)  %7.0f GFLOP/s"
          % (kind, m, n, k, xs.last_kernel(), batch, t, gbs, gbs / 80.0, 2.0 * m * n * k * batch / t / 1e6))
    # the same batch through the reference's libxsmm_mmbatch_kernel with index arrays (in elements of each operand's type; device arrays)
    L.libxsmm_xmmdispatch.restype = C.c_void_p; L.libxsmm_xmmdispatch.argtypes = [C.c_void_p]
    kern = L.libxsmm_xmmdispatch(C.c_void_p(desc))
    if kern:
        idx = torch.arange(batch, device="cuda", dtype=torch.int64)
        ia, ib, ic = (idx * (m * k)).to(torch.int32), (idx * (k * n)).to(torch.int32), (idx * (m * n)).to(torch.int32)
        L.libxsmm_mmbatch_kernel.restype = C.c_int
        L.libxsmm_mmbatch_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ubyte, C.c_ubyte, C.c_int]
        times = []
        for it in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert 0 == L.libxsmm_mmbatch_kernel(kern, 0, 4, xs.dptr(ia), xs.dptr(ib), xs.dptr(ic), xs.dptr(a), xs.dptr(b), xs.dptr(c), batch, 0, 1, 2, osz, 0)
            e1.record(); torch.cuda.synchronize()
            if it >= 2:
                times.append(e0.elapsed_time(e1))
        t = min(times); gbs = batch * per_item / t / 1e6
        print("%-7s %2dx%2dx%2d  index arrays (libxsmm_mmbatch_kernel) %-22s  %.3f ms  %6.0f GB/s (%4.1f%% of 8 TB/s)" % (kind, m, n, k, xs.last_kernel(), t, gbs, gbs / 80.0))


if which == "lowp":
    for kind in ("bf16f32", "i16i32", "bf16"):
        for (m, n, k) in ((32, 32, 32), (16, 16, 16), (48, 48, 48), (64, 64, 64)):
            run_lowp(kind, m, n, k)
    sys.exit(0)

shapes = [(13, 13, 13), (23, 23, 23), (32, 32, 32), (13, 23, 32), (32, 13, 23), (64, 64, 64), (8, 8, 8), (16, 16, 16), (5, 5, 5),
          (40, 40, 40), (48, 48, 48), (56, 56, 56), (64, 64, 32), (64, 32, 64)]
if os.environ.get("DENSE_SHAPES"):  # e.g. DENSE_SHAPES=5x5x5,13x13x13
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["DENSE_SHAPES"].split(",")]
for dt in ([torch.float64] if which == "f64" else [torch.float32] if which == "f32" else [torch.float64, torch.float32]):
    for (m, n, k) in shapes:
        for mfma in ((0, 1) if (dt == torch.float32 and (m, n, k) == (32, 32, 32)) or max(m, n) > 32 or os.environ.get("DENSE_BOTH") else (0,)):
            run(dt, m, n, k, mfma)
if not os.environ.get("DENSE_SHAPES"):
    run(torch.float64, 23, 23, 23, 0, beta=0.0)
