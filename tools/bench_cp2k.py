#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: CP2K-style stacks -- fp64 products of all 27 shapes (M,N,K) in {13,23,32}^3, grouped by
shape, every u consecutive products of a group accumulating into one C block (samples/cp2k/cp2k.cpp:155,328-360).
One libxsmm_gemm_batch call (index arrays) per shape group, all on the engine's stream.

usage: python3 tools/bench_cp2k.py [products=524288] [reps=7] [omp=0] [host_idx=0]   (omp=1: libxsmm_gemm_batch_omp, order of the sums relaxed)
Algorithmic bytes (the reference's bwsize, cp2k.cpp:156): sum over products 8*(M*K+K*N) + sum over C blocks 2*8*M*N."""
import importlib
import math
import os
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
products = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
OMP = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
HOST_IDX = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False  # index arrays in host memory, as an unchanged caller has them
ONLY_GROUPED = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False  # stop after the one-call form (profiling runs)
torch.cuda.set_device(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
L.libxsmm_amd_set_mfma(int(os.environ.get("CP2K_MFMA", "1")))  # 1 (the default policy): the run form on the matrix cores; 0: register-tiled run form

shapes = [(m, n, k) for m in (13, 23, 32) for n in (13, 23, 32) for k in (13, 23, 32)]
per = products // len(shapes)
_subset = os.environ.get("CP2K_SUBSET", "")  # developer knob: "light" / "heavy" = shapes whose operands are up to / beyond 900 elements per product
if _subset:
    shapes = [sh for sh in shapes if ((sh[0] * sh[2] + sh[2] * sh[1]) <= 900) == (_subset == "light")]
groups = []
tot_bytes = 0.0; tot_flops = 0.0; tot_runs = 0
for gi, (m, n, k) in enumerate(shapes):
    s = per + (products - per * len(shapes) if (m, n, k) == (32, 32, 32) else 0)
    u = max(1, math.isqrt(s * 160 // 240))
    nc = (s + u - 1) // u
    a = torch.rand(s * m * k, device="cuda", dtype=torch.float64, generator=g) - 0.5
    b = torch.rand(s * k * n, device="cuda", dtype=torch.float64, generator=g) - 0.5
    c = torch.zeros(nc * m * n, device="cuda", dtype=torch.float64)
    idx = torch.arange(s, device="cuda", dtype=torch.int64)
    ia = (idx * (m * k)).to(torch.int32); ib = (idx * (k * n)).to(torch.int32); ic = ((idx // u) * (m * n)).to(torch.int32)
    if HOST_IDX:
        ia, ib, ic = (x.cpu().numpy() for x in (ia, ib, ic))
    groups.append((m, n, k, s, a, b, c, ia, ib, ic))
    tot_bytes += s * 8.0 * (m * k + k * n) + nc * 16.0 * m * n
    tot_flops += 2.0 * m * n * k * s
    tot_runs += nc
print("products %d in %d groups, %d C blocks (runs of ~%d), %.2f GB algorithmic" % (products, len(groups), tot_runs, products // max(1, tot_runs), tot_bytes / 1e9))


import ctypes as C  # noqa: E402


def one_pass(streams):
    """Groups are independent (different C arrays): the caller may spread them over HIP streams (libxsmm_amd_set_stream).
    A batch call makes no host round trip, so the 27 calls are queued back to back and the groups overlap on the GPU."""
    if not streams:
        for (m, n, k, s, a, b, c, ia, ib, ic) in groups:
            xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s, omp=OMP)
        return
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(main)
    for st in streams:
        st.wait_event(fork)
    for gi, (m, n, k, s, a, b, c, ia, ib, ic) in enumerate(groups):
        st = streams[gi % len(streams)]
        L.libxsmm_amd_set_stream(C.c_void_p(st.cuda_stream))
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s, omp=OMP)
    L.libxsmm_amd_set_stream(C.c_void_p(main.cuda_stream))
    for st in streams:
        ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)


# ONE call for all groups: libxsmm_amd_gemm_batch_groups (one check launch + one multiplication launch, all chains resident)
if not HOST_IDX:
    shapes_ = [(g_[0], g_[1], g_[2]) for g_ in groups]

    def one_call():
        assert 0 == xs.gemm_batch_groups(xs.F64, shapes_, [g_[4] for g_ in groups], [g_[5] for g_ in groups], [g_[6] for g_ in groups],
                                         [g_[7] for g_ in groups], [g_[8] for g_ in groups], [g_[9] for g_ in groups], [g_[3] for g_ in groups], relaxed=OMP)
    import time as _time
    t0 = _time.perf_counter(); one_call(); torch.cuda.synchronize()
    print("first grouped call (hiprtc unless the code object is cached on disk): %.2f s" % (_time.perf_counter() - t0))
    times = []
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); one_call(); e1.record(); torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    t = sorted(times)[len(times) // 2]
    print("cp2k stacks%s, ONE grouped call: kernel %s  median %.3f ms (min %.3f)  %.0f GB/s (%.1f%% of 8 TB/s)  %.0f GFLOP/s"
          % (" (relaxed)" if OMP else "", xs.last_kernel(), t, min(times), tot_bytes / t / 1e6, tot_bytes / t / 1e6 / 80.0, tot_flops / t / 1e6))
    # the same call queued back to back (what bench.py times: the GPU never waits for the host as long as a call costs the host less
    # than the GPU) and the host's own time per call
    nq = max(4, reps)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = _time.perf_counter(); e0.record()
    for it in range(nq):
        one_call()
    e1.record(); h1 = _time.perf_counter(); torch.cuda.synchronize()
    tq = e0.elapsed_time(e1) / nq
    print("cp2k stacks%s, %d grouped calls queued back to back: %.3f ms per call on the GPU  %.0f GB/s (%.1f%% of 8 TB/s); host time per call %.3f ms"
          % (" (relaxed)" if OMP else "", nq, tq, tot_bytes / tq / 1e6, tot_bytes / tq / 1e6 / 80.0, (h1 - h0) / nq * 1e3))

if ONLY_GROUPED:
    sys.exit(0)
for nstreams in (0, 4, 8, 27):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    times = []
    names = set()
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one_pass(streams)
        e1.record(); torch.cuda.synchronize()
        names.add(xs.last_kernel())
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    t = sorted(times)[len(times) // 2]
    print("cp2k stacks%s, %2d streams: kernels %s  median %.3f ms (min %.3f)  %.0f GB/s (%.1f%% of 8 TB/s)  %.0f GFLOP/s"
          % (" (omp entry)" if OMP else "", nstreams if nstreams else 1, sorted(names), t, min(times), tot_bytes / t / 1e6, tot_bytes / t / 1e6 / 80.0, tot_flops / t / 1e6))
# the 27 calls captured once in a HIP graph and replayed (device index arrays only: no host work is left in the replay)
if not HOST_IDX:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        L.libxsmm_amd_set_stream(C.c_void_p(side.cuda_stream))
        one_pass([])  # warm-up on the capture stream
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            one_pass([])
        times = []
        for it in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side); graph.replay(); e1.record(side); side.synchronize()
            if it >= 2:
                times.append(e0.elapsed_time(e1))
    L.libxsmm_amd_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    t = sorted(times)[len(times) // 2]
    print("cp2k stacks%s, hipGraph replay of the 27 calls: median %.3f ms (min %.3f)  %.0f GB/s (%.1f%% of 8 TB/s)"
          % (" (omp entry)" if OMP else "", t, min(times), tot_bytes / t / 1e6, tot_bytes / t / 1e6 / 80.0))
# per-group breakdown (each group alone, synchronised)
for (m, n, k, s, a, b, c, ia, ib, ic) in groups[::13]:
    ts = []
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s, omp=OMP)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    u = max(1, math.isqrt(s * 160 // 240)); nc = (s + u - 1) // u
    byt = s * 8.0 * (m * k + k * n) + nc * 16.0 * m * n
    print("  %2dx%2dx%2d  %6d products, %4d runs  %s  %.3f ms  %.0f GB/s" % (m, n, k, s, nc, xs.last_kernel(), min(ts), byt / min(ts) / 1e6))
