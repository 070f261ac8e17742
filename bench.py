#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LIBXSMM engine.

Workload (BASELINE.json configs[1]): batched dense SMM, fp32, M=N=K=32, batch 1,048,576 per GPU, alpha=1, beta=1,
three contiguous operand arrays resident in HBM (the layout of samples/smm/specialized.cpp:143-146, streamed case).
One "step" = one pass of the hot path over the whole batch: a single call of the C-ABI entry point
libxsmm_amd_gemm_batch_strided (same kernel family as libxsmm_gemm_batch; --mode index goes through
libxsmm_gemm_batch with device index arrays).

Output: ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     -- dominant kernel's algorithmic bytes per launch / HIP-event launch time vs 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (a port of the reference arithmetic, NOT the product) timed on host cores, N=1 only
  secondary    -- spmdm CSR compute phase (BASELINE config 4 shape), fsspmdm (config 3), CP2K stacks (config 5) and 64^3 batches on this GPU.
Multi-GPU: one process per GPU (torch.distributed over RCCL); the batch axis shards with no data-path collective
("weak" scaling: fixed per-GPU batch). value = work of all ranks / max-over-ranks time.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1048576, help="items per GPU")
    ap.add_argument("--mfma", type=int, default=1, help="1: MFMA kernels where shapes allow, 0: scalar-FMA kernels only")
    ap.add_argument("--mode", default="strided", choices=["strided", "index"])
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-items", type=int, default=131072, help="bounded CPU-baseline sample (items)")
    return ap.parse_args()


def time_steps(torch, fn, steps, warmup, dist):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; returns (wall seconds, per-step ms list)"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for (e0, e1) in evs:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t1 - t0, [e0.elapsed_time(e1) for (e0, e1) in evs]


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        dist = dist_mod
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU compute path")
    torch.cuda.set_device(local)
    xs = importlib.import_module("libxsmm-1_amd")
    L = xs.lib()
    if L.libxsmm_amd_device_count() < 1:
        raise SystemExit("libxsmm.so sees no HIP device")
    L.libxsmm_amd_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))  # events and kernels on one stream
    L.libxsmm_amd_set_mfma(args.mfma)

    M = N = K = 32
    B = args.batch
    # synthetic data of the configured shape: uniform [-0.5, 0.5) (random data: zero/trivial operands flatter the clock)
    g = torch.Generator(device="cuda"); g.manual_seed(1 + rank)
    a = torch.rand(B * M * K, device="cuda", dtype=torch.float32, generator=g) - 0.5
    b = torch.rand(B * K * N, device="cuda", dtype=torch.float32, generator=g) - 0.5
    c = torch.rand(B * M * N, device="cuda", dtype=torch.float32, generator=g) - 0.5
    blob, desc = xs.descriptor(xs.F32, M, N, K, M, K, M, 1.0, 1.0)
    assert desc, "descriptor rejected"
    if args.mode == "index":
        ia = (torch.arange(B, device="cuda", dtype=torch.int32) * (M * K)).contiguous()
        ib = (torch.arange(B, device="cuda", dtype=torch.int32) * (K * N)).contiguous()
        ic = (torch.arange(B, device="cuda", dtype=torch.int32) * (M * N)).contiguous()

        def step():  # negative batchsize: no two items share a C (reference src/libxsmm_gemm.c:1338)
            xs.gemm_batch(xs.F32, "N", "N", M, N, K, 1.0, a, M, b, K, 1.0, c, M, 0, 4, ia, ib, ic, -B)
    else:
        pa, pb, pc = xs.dptr(a), xs.dptr(b), xs.dptr(c)

        def step():
            rc = L.libxsmm_amd_gemm_batch_strided(desc, pa, pb, pc, M * K, K * N, M * N, B)
            assert rc == 0

    wall, per_step = time_steps(torch, step, args.steps, args.warmup, dist)
    kernel_name = xs.last_kernel()
    wall_t = torch.tensor([wall], device="cuda", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())
    ms_per_step = 1e3 * wall_max / args.steps
    flops_item = 2.0 * M * N * K
    bytes_item = 4.0 * (M * K + K * N + 2 * M * N)  # A + B + C read + C write = 16384 B (specialized.cpp:91-92)
    total_items = float(B) * world
    gflops = total_items * flops_item / (wall_max / args.steps) / 1e9

    out = {
        "metric": "batched SMM fp32 32x32x32 GFLOP/s (whole job)", "value": round(gflops, 1), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: batched dense SMM fp32 M=N=K=32 alpha=1 beta=1, batch=%d per GPU, %s addressing, MFMA %s"
                   % (B, args.mode, "on" if args.mfma else "off"), "batch_per_gpu": B, "parallelism": "batch-shard x%d" % world},
        "hbm_gbs_per_gpu": round(float(B) * bytes_item / (wall_max / args.steps) / 1e9, 1),
        "gflops_per_gpu": round(gflops / world, 1),
    }
    if rank == 0:
        import statistics
        kms = statistics.mean(per_step)  # HIP events on the launch stream around each launch (one kernel per step)
        achieved = float(B) * bytes_item / (kms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                           "launch_ms_avg": round(kms, 4), "launch_ms_min": round(min(per_step), 4),
                           "algorithmic_bytes_per_launch": float(B) * bytes_item}
        # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950
        # correction applied) -- counters cannot be collected from inside this process, so the committed summary is used
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_summary.json")))
            entry = pmc.get(kernel_name)
            if entry and B == 1048576:
                out["roofline"]["traffic"] = entry["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r1_pmc_summary.json"
        except (OSError, ValueError, KeyError):
            pass
        # measured ceiling for this traffic mix: c += a + b over the same three arrays (3 reads : 1 write, no arithmetic)
        nbytes = B * M * K * 4

        def probe():
            assert 0 == L.libxsmm_amd_stream_probe(xs.dptr(a), xs.dptr(b), xs.dptr(c), nbytes)
        _, pt = time_steps(torch, probe, 5, 2, None)
        ceil = 4.0 * nbytes / (min(pt) * 1e-3) / 1e9
        out["roofline"]["stream_ceiling_gbs"] = round(ceil, 1)
        out["roofline"]["frac_of_stream_ceiling"] = round(achieved / ceil, 4)
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            out["secondary"] = secondary(torch, xs, L)
        except Exception as exc:  # secondary numbers must never take the headline down
            out["secondary"] = {"error": repr(exc)}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(M, N, K, args.cpu_items)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def secondary(torch, xs, L):
    """Other BASELINE configs on one GPU (reported, not the headline): spmdm compute phase and fsspmdm."""
    res = {}
    # config 4 shape: spmdm fp32 M=K=64 N=48, 50% zeros, beta=0
    M, N, K, B = 64, 48, 64, 131072
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    a = torch.rand(B * M * K, device="cuda", generator=g) - 0.5
    a = torch.where(torch.rand(B * M * K, device="cuda", generator=g) >= 0.5, a, torch.zeros_like(a))
    b = torch.rand(B * K * N, device="cuda", generator=g) - 0.5
    c = torch.zeros(B * M * N, device="cuda")
    sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, B)
    assert sb
    beta = C.c_float(0.0)

    def create():
        assert 0 == L.libxsmm_amd_spmdm_batch_create_slices(sb, b"N", xs.dptr(a))

    def compute():
        assert 0 == L.libxsmm_amd_spmdm_batch_compute(sb, b"N", xs.dptr(b), b"N", C.byref(beta), xs.dptr(c))
    _, t_create = time_steps(torch, create, 10, 2, None)
    k_create = xs.last_kernel()
    _, t_comp = time_steps(torch, compute, 10, 2, None)
    k_comp = xs.last_kernel()
    nnz = float((a != 0).sum().item()) / B
    by_create = 4.0 * M * K + 6.0 * nnz + 2.0 * (M + 1)
    by_comp = 6.0 * nnz + 2.0 * (M + 1) + 4.0 * K * N + 4.0 * M * N
    mc, mp = sum(t_create) / len(t_create) * 1e-3, sum(t_comp) / len(t_comp) * 1e-3  # averages, like the headline
    res["spmdm_f32_64x48x64_nnz50"] = {
        "batch": B, "nnz_per_item": round(nnz, 1),
        "create": {"kernel": k_create, "ms": round(mc * 1e3, 4), "hbm_gbs": round(B * by_create / mc / 1e9, 1), "frac": round(B * by_create / mc / 1e9 / HBM_PEAK_GBS, 4)},
        "compute": {"kernel": k_comp, "ms": round(mp * 1e3, 4), "hbm_gbs": round(B * by_comp / mp / 1e9, 1), "frac": round(B * by_comp / mp / 1e9 / HBM_PEAK_GBS, 4),
                    "gflops": round(B * 2.0 * nnz * N / mp / 1e9, 1)}}
    L.libxsmm_amd_spmdm_batch_destroy(sb)
    del a, b, c
    # config 3 shape: fsspmdm fp64 M=K=35, ~15% nnz, N=96 per item
    import numpy as np
    M, K, N, B = 35, 35, 96, 65536
    rng = np.random.default_rng(1)
    palette = np.array([0.25, -0.5, 0.75, 1.0, -1.25, 1.5, -2.0])
    A = np.where(rng.random((M, K)) < 0.15, palette[rng.integers(0, 7, (M, K))], 0.0)
    ntot = N * B
    Bm = torch.rand(K * ntot, device="cuda", dtype=torch.float64, generator=g) - 0.5
    Cm = torch.zeros(M * ntot, device="cuda", dtype=torch.float64)
    h = L.libxsmm_dfsspmdm_create(M, N, K, K, ntot, ntot, 1.0, 1.0, xs.dptr(np.ascontiguousarray(A)))
    assert h

    def run():
        assert 0 == L.libxsmm_amd_dfsspmdm_execute_batch(h, xs.dptr(Bm), xs.dptr(Cm), B)
    _, t = time_steps(torch, run, 10, 2, None)
    mt = sum(t) / len(t) * 1e-3
    by = 8.0 * N * (K + 2 * M)  # beta=1: B read + C read + C write = 80640 B per item
    res["fsspmdm_f64_35x96x35_nnz15"] = {"batch": B, "nnz": int((A != 0).sum()), "kernel": xs.last_kernel(), "ms": round(mt * 1e3, 4),
                                         "hbm_gbs": round(B * by / mt / 1e9, 1), "frac": round(B * by / mt / 1e9 / HBM_PEAK_GBS, 4)}
    L.libxsmm_dfsspmdm_destroy(h)
    del Bm, Cm
    # config 5 shape mix on one GPU: CP2K-style stacks, fp64, 27 shapes x 19418 products, runs of u products per C block
    # (samples/cp2k/cp2k.cpp:155,328-360); one libxsmm_gemm_batch (index arrays) per shape group, one stream
    import math
    old_mfma = L.libxsmm_amd_set_mfma(0)
    groups, byt, flops = [], 0.0, 0.0
    for (m, n, k) in [(m, n, k) for m in (13, 23, 32) for n in (13, 23, 32) for k in (13, 23, 32)]:
        s_ = 19418
        u = max(1, math.isqrt(s_ * 160 // 240)); nc = (s_ + u - 1) // u
        a = torch.rand(s_ * m * k, device="cuda", dtype=torch.float64, generator=g) - 0.5
        b = torch.rand(s_ * k * n, device="cuda", dtype=torch.float64, generator=g) - 0.5
        c = torch.zeros(nc * m * n, device="cuda", dtype=torch.float64)
        idx = torch.arange(s_, device="cuda", dtype=torch.int64)
        groups.append((m, n, k, s_, a, b, c, (idx * (m * k)).to(torch.int32), (idx * (k * n)).to(torch.int32), ((idx // u) * (m * n)).to(torch.int32)))
        byt += s_ * 8.0 * (m * k + k * n) + nc * 16.0 * m * n  # the reference's bwsize (cp2k.cpp:156)
        flops += 2.0 * m * n * k * s_

    def stacks():
        for (m, n, k, s_, a, b, c, ia, ib, ic) in groups:
            xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s_)
    _, t = time_steps(torch, stacks, 5, 2, None)
    mt = sum(t) / len(t) * 1e-3
    res["cp2k_stacks_f64_27shapes"] = {"products": 27 * 19418, "kernel": xs.last_kernel(), "streams": 1, "ms": round(mt * 1e3, 4),
                                       "hbm_gbs": round(byt / mt / 1e9, 1), "frac": round(byt / mt / 1e9 / HBM_PEAK_GBS, 4),
                                       "gflops": round(flops / mt / 1e9, 1)}
    # the same 27 calls with the shape groups (independent C arrays) spread over 8 caller streams: batch calls make no host
    # round trip, so the groups overlap on the GPU (each group alone is a set of sequential accumulation chains)
    main = torch.cuda.current_stream()
    pool = [torch.cuda.Stream() for _ in range(8)]

    def stacks_streams():
        fork = torch.cuda.Event(); fork.record(main)
        for st in pool:
            st.wait_event(fork)
        for gi, (m, n, k, s_, a, b, c, ia, ib, ic) in enumerate(groups):
            L.libxsmm_amd_set_stream(C.c_void_p(pool[gi % len(pool)].cuda_stream))
            xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s_)
        L.libxsmm_amd_set_stream(C.c_void_p(main.cuda_stream))
        for st in pool:
            ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)
    _, t8 = time_steps(torch, stacks_streams, 5, 2, None)
    m8 = sum(t8) / len(t8) * 1e-3
    res["cp2k_stacks_f64_27shapes"]["streams8"] = {"ms": round(m8 * 1e3, 4), "hbm_gbs": round(byt / m8 / 1e9, 1), "frac": round(byt / m8 / 1e9 / HBM_PEAK_GBS, 4)}
    # the multi-threaded entry point (libxsmm_gemm_batch_omp: the reference adds into a shared C under a lock, in no defined
    # order), one stream: few long runs are cut into segments whose sums join C with floating-point atomics
    def stacks_omp():
        for (m, n, k, s_, a, b, c, ia, ib, ic) in groups:
            xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s_, omp=True)
    _, to = time_steps(torch, stacks_omp, 5, 2, None)
    mo = sum(to) / len(to) * 1e-3
    res["cp2k_stacks_f64_27shapes"]["omp_entry"] = {"ms": round(mo * 1e3, 4), "hbm_gbs": round(byt / mo / 1e9, 1), "frac": round(byt / mo / 1e9 / HBM_PEAK_GBS, 4)}
    # the upper end of the (M,N,K) <= 64 family: strided batches of 64^3 on the matrix-core work-group kernels
    L.libxsmm_amd_set_mfma(1)
    for (name, dt, prec, ts) in (("smm_f32_64x64x64", torch.float32, xs.F32, 4), ("smm_f64_64x64x64", torch.float64, xs.F64, 8)):
        m = n = k = 64
        B64 = 65536 if ts == 4 else 32768
        a = torch.rand(B64 * m * k, device="cuda", dtype=dt, generator=g) - 0.5
        b = torch.rand(B64 * k * n, device="cuda", dtype=dt, generator=g) - 0.5
        c = torch.zeros(B64 * m * n, device="cuda", dtype=dt)
        blob, desc = xs.descriptor(prec, m, n, k, m, k, m, 1.0, 1.0)

        def dense64():
            assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, B64)
        _, td = time_steps(torch, dense64, 5, 2, None)
        md = sum(td) / len(td) * 1e-3
        byt64 = B64 * float(ts) * (m * k + k * n + 2 * m * n)
        res[name] = {"batch": B64, "kernel": xs.last_kernel(), "ms": round(md * 1e3, 4), "hbm_gbs": round(byt64 / md / 1e9, 1),
                     "frac": round(byt64 / md / 1e9 / HBM_PEAK_GBS, 4), "gflops": round(2.0 * m * n * k * B64 / md / 1e9, 1)}
        del a, b, c
    L.libxsmm_amd_set_mfma(old_mfma)
    return res


def cpu_baseline(M, N, K, items):
    """The oracle (a plain-C port of the reference's AVX2-path arithmetic, OpenMP over the batch) on the host cores."""
    import numpy as np
    import oracle_binding as orc
    try:
        cores = len(os.sched_getaffinity(0))  # the box grants a CPU share, not the whole host
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("XSMM_BENCH_CPU_THREADS", "16"))))
    rng = np.random.default_rng(1)
    a = rng.random(items * M * K, dtype=np.float32) - 0.5
    b = rng.random(items * K * N, dtype=np.float32) - 0.5
    c = rng.random(items * M * N, dtype=np.float32) - 0.5
    orc.gemm_batch_strided(orc.FMA, 0, M, N, K, M, K, M, a, b, c, M * K, K * N, M * N, min(items, 4096), cores)  # warm-up
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 200:
        t0 = time.perf_counter()
        orc.gemm_batch_strided(orc.FMA, 0, M, N, K, M, K, M, a, b, c, M * K, K * N, M * N, items, cores)
        t_total += time.perf_counter() - t0
        reps += 1
    gf = reps * items * 2.0 * M * N * K / t_total / 1e9
    return {"value": round(gf, 2), "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "%d items of the same fp32 32x32x32 beta=1 workload x %d passes (%.1f s), OpenMP static over the batch"
                      % (items, reps, t_total)}


if __name__ == "__main__":
    main()
