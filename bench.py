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

--config 4 / --config 5 run the two BASELINE configurations that have an exchange step, sharded the same way (readiness
harness for multi-GPU nodes; at N=1 they are single-GPU runs of the per-GPU shard):
  4: spmdm fp32 M=K=64 N=48, 50 % zeros, 131072 problems per GPU: createSparseSlice + compute chunk by chunk, chunk i's C
     all-gathered (RCCL all_gather_into_tensor) while chunk i+1 is computed; `collective_ms` = the gathers alone,
     `compute_ms` = the kernels alone, `ms_per_step` = both, overlapped.
  5: CP2K stacks fp64, 27 shapes, 524288 products per GPU in ONE grouped call (libxsmm_amd_gemm_batch_groups). Default
     partition: every rank owns its C blocks (no exchange); --c5-split measures the fallback for stacks cut across
     ranks: partial C + one fused all-reduce over all C blocks (`collective_ms`).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

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
    ap.add_argument("--no-variants", action="store_true", help="skip headline_variants (profiling runs: the variants launch the same kernel)")
    ap.add_argument("--cpu-items", type=int, default=131072, help="bounded CPU-baseline sample (items)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5], help="BASELINE configuration (2: the headline)")
    ap.add_argument("--chunks", type=int, default=4, help="config 4: chunks of the per-GPU shard (gather of chunk i overlaps compute of chunk i+1)")
    ap.add_argument("--c5-split", action="store_true", help="config 5: stacks cut across ranks: partial C + one fused all-reduce")
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


def launch_ranks(args):
    """`python bench.py --gpus N` outside a torchrun job: this process starts the N ranks itself (one process per GPU,
    `python -m torch.distributed.run`, rendezvous on 127.0.0.1), relays what they print -- rank 0's JSON line -- and exits with
    their status. It runs before anything here has touched the GPU (no torch import yet) and starts the ranks as CHILD processes
    (never an exec); a rank that fails is not restarted (--max-restarts 0): the job ends non-zero."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--max-restarts", "0",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        print("bench.py: --gpus %d but the job has %d ranks: reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (rank %d of %d): the engine has no CPU compute path" % (rank, world), file=sys.stderr, flush=True)
        if world > 1:
            time.sleep(2.0)  # (the launcher ends the other ranks as soon as one has failed: let each say why first)
        raise SystemExit(1)
    dist = None
    # BENCH_SHARED_GPU=1: a rehearsal of the N-rank path on a box with ONE card (launcher, barriers, max over ranks, the summed
    # value): every rank runs on cuda:0 and the ranks meet over gloo -- RCCL refuses two ranks on one device. Not a measurement.
    shared = (world > 1 and "1" == os.environ.get("BENCH_SHARED_GPU", ""))
    if shared:
        if args.config in (4, 5):
            raise SystemExit("BENCH_SHARED_GPU rehearses the headline configuration only (configs 4/5 exchange data over RCCL)")
        local = 0
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist_mod.init_process_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        dist = dist_mod
    torch.cuda.set_device(local)
    xs = importlib.import_module("libxsmm-1_amd")
    L = xs.lib()
    if L.libxsmm_amd_device_count() < 1:
        raise SystemExit("libxsmm.so sees no HIP device")
    L.libxsmm_amd_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))  # events and kernels on one stream
    L.libxsmm_amd_set_mfma(args.mfma)
    if args.config in (4, 5):
        out = (config4 if 4 == args.config else config5)(args, torch, xs, L, dist, rank, world)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    M = N = K = 32
    B = args.batch
    # synthetic data of the configured shape: uniform [-0.5, 0.5) (random data: zero/trivial operands flatter the clock)
    g = torch.Generator(device="cuda"); g.manual_seed(1 + rank)
    # Placement in HBM: the three arrays are pieces of one allocation, B and C 8 KiB and 16 KiB off the spacing of the arrays. A
    # wave reads a[i], b[i] and c[i] at the same time; when the three land at the same offset modulo the memory system's
    # interleave (three separate 4 GiB allocations do so in some processes and not in others) the kernel takes 3.3 ms instead of
    # 2.9-3.0 ms (tools/probe_headline_offsets.py, profiles/r2_headline_placement.txt). Items stay contiguous and 16-byte aligned.
    na, nb, nc = B * M * K, B * K * N, B * M * N
    skew = 2048  # floats = 8 KiB
    pool = torch.empty(na + nb + nc + 3 * skew, device="cuda", dtype=torch.float32)
    pool.uniform_(-0.5, 0.5, generator=g)
    a = pool[0:na]; b = pool[na + skew:na + skew + nb]; c = pool[na + nb + 2 * skew:na + nb + 2 * skew + nc]
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
            src = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))[-1]  # latest round
            pmc = json.load(open(os.path.join(ROOT, "profiles", src)))
            entry = pmc.get(kernel_name)
            if entry and B == 1048576:
                out["roofline"]["traffic"] = entry["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/" + src
        except (OSError, ValueError, KeyError):
            pass
        # a plain streaming kernel over the same three arrays (c += a + b: 3 reads : 1 write, no arithmetic) for comparison. NOT a
        # ceiling: the SMM kernel keeps more bytes in flight per CU and can be faster than this probe (round 2: 1.03 x).
        nbytes = B * M * K * 4

        def probe():
            assert 0 == L.libxsmm_amd_stream_probe(xs.dptr(a), xs.dptr(b), xs.dptr(c), nbytes)
        _, pt = time_steps(torch, probe, 5, 2, None)
        out["roofline"]["stream_probe_gbs"] = round(4.0 * nbytes / (min(pt) * 1e-3) / 1e9, 1)
        out["roofline"]["guide_achievable_gbs"] = 6300.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~6.3 TB/s achievable of the 8 TB/s peak
        out["roofline"]["frac_of_guide_achievable"] = round(achieved / 6300.0, 4)
        out["roofline"]["operands"] = "A, B, C cut from one allocation, B and C 8 / 16 KiB off the spacing of the arrays (see headline_variants for three separate allocations)"
    if rank == 0 and world == 1 and not args.no_secondary and not args.no_variants and args.mode == "strided":
        # The same workload (a) through the reference's own entry point, libxsmm_gemm_batch with index arrays (src/libxsmm_gemm.c:1878;
        # negative batchsize: no two items share a C), and (b) with A, B, C as three separate allocations -- what
        # samples/smm/specialized.cpp:143-146 does -- where the arrays may land at the same offset modulo the memory interleave.
        try:
            variants = {}
            ia = (torch.arange(B, device="cuda", dtype=torch.int32) * (M * K)).contiguous()
            ib = (torch.arange(B, device="cuda", dtype=torch.int32) * (K * N)).contiguous()
            ic = (torch.arange(B, device="cuda", dtype=torch.int32) * (M * N)).contiguous()

            def step_index():
                xs.gemm_batch(xs.F32, "N", "N", M, N, K, 1.0, a, M, b, K, 1.0, c, M, 0, 4, ia, ib, ic, -B)
            _, ti = time_steps(torch, step_index, 10, 2, None)
            mi = sum(ti) / len(ti)
            variants["libxsmm_gemm_batch_index_arrays"] = {"kernel": xs.last_kernel(), "ms": round(mi, 4), "hbm_gbs": round(float(B) * bytes_item / (mi * 1e-3) / 1e9, 1),
                                                            "frac": round(float(B) * bytes_item / (mi * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del ia, ib, ic
            # (c) BASELINE configs[1] names both policies ("MFMA off vs on"): the same strided call with the matrix cores switched off
            # (libxsmm_amd_set_mfma(0): the scalar-FMA kernel; bit-identical results)
            if 0 != args.mfma:
                old_policy = L.libxsmm_amd_set_mfma(0)
                try:
                    step(); torch.cuda.synchronize()
                    _, tf = time_steps(torch, step, 10, 2, None)
                    mf = sum(tf) / len(tf)
                    variants["matrix_cores_off"] = {"kernel": xs.last_kernel(), "ms": round(mf, 4), "hbm_gbs": round(float(B) * bytes_item / (mf * 1e-3) / 1e9, 1),
                                                    "frac": round(float(B) * bytes_item / (mf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                finally:
                    L.libxsmm_amd_set_mfma(old_policy)
            a2 = torch.empty(na, device="cuda", dtype=torch.float32).uniform_(-0.5, 0.5, generator=g)
            b2 = torch.empty(nb, device="cuda", dtype=torch.float32).uniform_(-0.5, 0.5, generator=g)
            c2 = torch.empty(nc, device="cuda", dtype=torch.float32).uniform_(-0.5, 0.5, generator=g)

            def step_sep():
                assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a2), xs.dptr(b2), xs.dptr(c2), M * K, K * N, M * N, B)
            _, tsp = time_steps(torch, step_sep, 10, 2, None)
            msp = sum(tsp) / len(tsp)
            variants["three_separate_allocations"] = {"kernel": xs.last_kernel(), "ms": round(msp, 4), "hbm_gbs": round(float(B) * bytes_item / (msp * 1e-3) / 1e9, 1),
                                                      "frac": round(float(B) * bytes_item / (msp * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                      "offsets_mod_64KiB": [int(x.data_ptr() % 65536) for x in (a2, b2, c2)]}
            del a2, b2, c2
            out["headline_variants"] = variants
        except Exception as exc:  # noqa: BLE001 (must never take the headline down)
            out["headline_variants"] = {"error": repr(exc)[:200]}
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
    if "|" in k_comp:  # both batch kernels are launched and the one that does not suit the density returns at once (the choice is made
        # on the device from sampled entry counts, threshold 28 %): name the one that did the work
        k_comp = "spmdm_compute_mfma" if 100.0 * nnz >= 28.0 * M * K else "spmdm_compute_wg_lds"
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
    M, K, N, B = 35, 35, 96, 262144  # the configuration's own batch (2 x 7.05 GB)
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
    old_mfma = L.libxsmm_amd_set_mfma(1)  # (the default policy: runs of products per C block on the matrix-core run form, bit-exact chain)
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

    shapes27 = [(q[0], q[1], q[2]) for q in groups]

    def stacks(relaxed=False):
        # ONE call for the 27 shape groups (sums per C block in batch order): one check launch + one fused run-kernel launch
        assert 0 == xs.gemm_batch_groups(xs.F64, shapes27, [q[4] for q in groups], [q[5] for q in groups], [q[6] for q in groups],
                                         [q[7] for q in groups], [q[8] for q in groups], [q[9] for q in groups], [q[3] for q in groups], relaxed=relaxed)
    stacks(); L.libxsmm_amd_jit_wait()  # (the fused kernel comes from lib/jit_cache, or is compiled now: not while timing)
    _, t = time_steps(torch, stacks, 5, 2, None)
    mt = sum(t) / len(t) * 1e-3
    res["cp2k_stacks_f64_27shapes"] = {"products": 27 * 19418, "entry": "libxsmm_amd_gemm_batch_groups (one call, batch order)", "kernel": xs.last_kernel(),
                                       "streams": 1, "ms": round(mt * 1e3, 4), "hbm_gbs": round(byt / mt / 1e9, 1), "frac": round(byt / mt / 1e9 / HBM_PEAK_GBS, 4),
                                       "gflops": round(flops / mt / 1e9, 1)}

    # the same stacks as 27 libxsmm_gemm_batch calls, one after the other on one stream (an unchanged CP2K-style caller)
    def stacks_calls():
        for (m, n, k, s_, a, b, c, ia, ib, ic) in groups:
            xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, s_)
    stacks_calls(); L.libxsmm_amd_jit_wait()
    _, tc = time_steps(torch, stacks_calls, 5, 2, None)
    mc_ = sum(tc) / len(tc) * 1e-3
    res["cp2k_stacks_f64_27shapes"]["call_per_shape"] = {"kernel": xs.last_kernel(), "ms": round(mc_ * 1e3, 4), "hbm_gbs": round(byt / mc_ / 1e9, 1), "frac": round(byt / mc_ / 1e9 / HBM_PEAK_GBS, 4)}
    # order of the sums relaxed (what libxsmm_gemm_batch_omp / ?gemm_batch_omp allow: the reference adds under a lock, in no
    # defined order): few long runs are cut into segments whose sums join C with floating-point atomics
    stacks(True); L.libxsmm_amd_jit_wait()
    _, to = time_steps(torch, lambda: stacks(True), 5, 2, None)
    mo = sum(to) / len(to) * 1e-3
    res["cp2k_stacks_f64_27shapes"]["relaxed_order"] = {"kernel": xs.last_kernel(), "ms": round(mo * 1e3, 4), "hbm_gbs": round(byt / mo / 1e9, 1), "frac": round(byt / mo / 1e9 / HBM_PEAK_GBS, 4)}
    # the upper end of the (M,N,K) <= 64 family: strided batches of 64^3 on the matrix-core work-group kernels
    L.libxsmm_amd_set_mfma(1)
    # ... and, between 32 and 64, the one-wave-per-item matrix-core kernel (48^3)
    for (name, dt, prec, ts, mnk) in (("smm_f32_64x64x64", torch.float32, xs.F32, 4, 64), ("smm_f64_64x64x64", torch.float64, xs.F64, 8, 64),
                                      ("smm_f32_48x48x48", torch.float32, xs.F32, 4, 48), ("smm_f64_48x48x48", torch.float64, xs.F64, 8, 48)):
        m = n = k = mnk
        B64 = (65536 if ts == 4 else 32768) * (64 // mnk) ** 2
        # (one allocation, B and C 8 KiB / 16 KiB off the spacing of the arrays: see the headline's operands)
        ne, sk = B64 * m * k, 8192 // ts
        pool = torch.empty(3 * ne + 3 * sk, device="cuda", dtype=dt)
        pool.uniform_(-0.5, 0.5, generator=g)
        a = pool[0:ne]; b = pool[ne + sk:2 * ne + sk]; c = pool[2 * ne + 2 * sk:3 * ne + 2 * sk]
        c.zero_()
        blob, desc = xs.descriptor(prec, m, n, k, m, k, m, 1.0, 1.0)

        def dense64():
            assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, B64)
        dense64(); L.libxsmm_amd_jit_wait()
        _, td = time_steps(torch, dense64, 5, 2, None)
        md = sum(td) / len(td) * 1e-3
        byt64 = B64 * float(ts) * (m * k + k * n + 2 * m * n)
        res[name] = {"batch": B64, "kernel": xs.last_kernel(), "ms": round(md * 1e3, 4), "hbm_gbs": round(byt64 / md / 1e9, 1),
                     "frac": round(byt64 / md / 1e9 / HBM_PEAK_GBS, 4), "gflops": round(2.0 * m * n * k * B64 / md / 1e9, 1)}
        del a, b, c, pool
    # bf16 inputs, fp32 result, 64^3 (reference: libxsmm_bsmmdispatch, src/libxsmm_main.c:2230-2244): the one-wave-per-item
    # matrix-core kernel on the widened operands (bit-identical to the gold loop's product-then-add)
    try:
        m = n = k = 64
        Bl = 131072
        a16 = (torch.randint(0, 1024, (Bl * m * k,), device="cuda", dtype=torch.int32, generator=g) + 0x3C00).to(torch.int16)
        b16 = (torch.randint(0, 1024, (Bl * k * n,), device="cuda", dtype=torch.int32, generator=g) + 0x3C00).to(torch.int16)
        c32 = torch.zeros(Bl * m * n, device="cuda", dtype=torch.float32)
        blob = xs.DescriptorBlob()
        L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
        L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
        dl = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), xs.BF16, xs.F32, m, n, k, m, k, m, 1.0, 1.0, 0, 0)

        def lowp64():
            assert 0 == L.libxsmm_amd_gemm_batch_strided(C.c_void_p(dl), xs.dptr(a16), xs.dptr(b16), xs.dptr(c32), m * k, k * n, m * n, Bl)
        lowp64(); L.libxsmm_amd_jit_wait()
        _, tl = time_steps(torch, lowp64, 5, 2, None)
        ml = sum(tl) / len(tl) * 1e-3
        bytl = Bl * (2.0 * (m * k + k * n) + 8.0 * m * n)
        res["smm_bf16f32_64x64x64"] = {"batch": Bl, "kernel": xs.last_kernel(), "ms": round(ml * 1e3, 4), "hbm_gbs": round(bytl / ml / 1e9, 1),
                                       "frac": round(bytl / ml / 1e9 / HBM_PEAK_GBS, 4), "gflops": round(2.0 * m * n * k * Bl / ml / 1e9, 1)}
        del a16, b16, c32
    except Exception as e:  # noqa: BLE001 (a secondary figure must not take the bench line down)
        res["smm_bf16f32_64x64x64"] = {"error": str(e)[:200]}
    L.libxsmm_amd_set_mfma(old_mfma)
    return res


def gather_ranks(value, dist, world):
    """[value of rank 0, value of rank 1, ...]"""
    import torch
    if dist is None:
        return [float(value)]
    t = torch.zeros(world, device="cuda", dtype=torch.float64)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t)
    return [float(x) for x in t.tolist()]


def config4(args, torch, xs, L, dist, rank, world):
    """BASELINE configs[3]: spmdm fp32 M=K=64 N=48, 50 % zeros, beta=0; the problems shard across the GPUs (no exchange while
    computing), every rank then holds the C of all problems: the per-GPU shard goes chunk by chunk -- createSparseSlice +
    compute on the chunk, and its C is all-gathered over xGMI while the next chunk is computed."""
    dist_mod = importlib.import_module("libxsmm-1_amd.dist")
    M, N, K = 64, 48, 64
    B = min(args.batch, 131072) if args.batch != 1048576 else 131072  # per GPU: 1 048 576 problems over 8 GPUs
    nch = max(1, args.chunks)
    per = B // nch
    B = per * nch
    g = torch.Generator(device="cuda"); g.manual_seed(1 + rank)
    a = torch.rand(B * M * K, device="cuda", generator=g) - 0.5
    a = torch.where(torch.rand(B * M * K, device="cuda", generator=g) >= 0.5, a, torch.zeros_like(a))
    b = torch.rand(B * K * N, device="cuda", generator=g) - 0.5
    c = torch.zeros(B * M * N, device="cuda")
    gathered = [torch.empty(world * per * M * N, device="cuda") for _ in range(nch)] if dist is not None else None
    sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, per)
    assert sb
    beta = C.c_float(0.0)

    def compute_chunk(i):
        assert 0 == L.libxsmm_amd_spmdm_batch_create_slices(sb, b"N", xs.dptr(a[i * per * M * K:]))
        assert 0 == L.libxsmm_amd_spmdm_batch_compute(sb, b"N", xs.dptr(b[i * per * K * N:]), b"N", C.byref(beta), xs.dptr(c[i * per * M * N:]))

    def local_chunk(i):
        return c[i * per * M * N:(i + 1) * per * M * N]

    def step():
        dist_mod.gather_chunks_overlapped(nch, compute_chunk, local_chunk, lambda i: gathered[i], dist)

    def compute_only():
        for i in range(nch):
            compute_chunk(i)

    def gather_only():
        dist_mod.gather_chunks_overlapped(nch, lambda i: None, local_chunk, lambda i: gathered[i], dist)
    wall, per_step = time_steps(torch, step, args.steps, args.warmup, dist)
    wall_max = dist_mod.max_over_ranks(wall, dist, "cuda")
    _, t_comp = time_steps(torch, compute_only, max(3, args.steps // 2), 1, dist)
    t_coll = [0.0]
    if dist is not None:
        _, t_coll = time_steps(torch, gather_only, max(3, args.steps // 2), 1, dist)
    nnz = float((a != 0).sum().item()) / B
    by_item = (4.0 * M * K + 6.0 * nnz + 2.0 * (M + 1)) + (6.0 * nnz + 2.0 * (M + 1) + 4.0 * K * N + 4.0 * M * N)  # create + compute
    ms = 1e3 * wall_max / args.steps
    L.libxsmm_amd_spmdm_batch_destroy(sb)
    return {
        "metric": "spmdm fp32 64x48x64 nnz50 GFLOP/s (sparse flops, whole job)", "value": round(world * B * 2.0 * nnz * N / (wall_max / args.steps) / 1e9, 1), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[3]: spmdm CSR A-sparse fp32 M=K=64 N=48, 50%% zeros, beta=0, %d problems per GPU in %d chunks, createSparseSlice + compute + all-gather of C"
                   % (B, nch), "batch_per_gpu": B, "parallelism": "batch-shard x%d, chunked all_gather_into_tensor overlapped with compute" % world},
        "compute_ms": round(sum(t_comp) / len(t_comp), 4), "collective_ms": round(sum(t_coll) / len(t_coll), 4),
        "ms_per_step_per_rank": [round(1e3 * w / args.steps, 4) for w in gather_ranks(wall, dist, world)],
        "hbm_gbs_per_gpu": round(B * by_item / (sum(t_comp) / len(t_comp) * 1e-3) / 1e9, 1),
        "gathered_bytes_per_gpu": (world - 1) * B * M * N * 4 if dist is not None else 0,
    }


def config5(args, torch, xs, L, dist, rank, world):
    """BASELINE configs[4]: CP2K-style stacks, fp64, 27 shapes (M,N,K) in {13,23,32}^3, 524288 products per GPU (4M over 8), every
    u consecutive products of a shape accumulate into one C block (samples/cp2k/cp2k.cpp:155,328-360). Every rank owns its
    C blocks and is handed the products that update them (no exchange); --c5-split: the stacks are cut across the ranks instead
    -- every rank sums into its own copy of all C blocks, one fused all-reduce joins them."""
    import math
    dist_mod = importlib.import_module("libxsmm-1_amd.dist")
    products = 524288 if args.batch == 1048576 else args.batch
    shapes = [(m, n, k) for m in (13, 23, 32) for n in (13, 23, 32) for k in (13, 23, 32)]
    per = products // len(shapes)
    g = torch.Generator(device="cuda"); g.manual_seed(1 + rank)
    groups, byt, flops, csizes = [], 0.0, 0.0, []
    for (m, n, k) in shapes:
        s_ = per + (products - per * len(shapes) if (m, n, k) == (32, 32, 32) else 0)
        u = max(1, math.isqrt(s_ * 160 // 240)); nc = (s_ + u - 1) // u
        a = torch.rand(s_ * m * k, device="cuda", dtype=torch.float64, generator=g) - 0.5
        b = torch.rand(s_ * k * n, device="cuda", dtype=torch.float64, generator=g) - 0.5
        idx = torch.arange(s_, device="cuda", dtype=torch.int64)
        groups.append([m, n, k, s_, a, b, None, (idx * (m * k)).to(torch.int32), (idx * (k * n)).to(torch.int32), ((idx // u) * (m * n)).to(torch.int32)])
        csizes.append(nc * m * n)
        byt += s_ * 8.0 * (m * k + k * n) + nc * 16.0 * m * n  # the reference's bwsize (cp2k.cpp:156)
        flops += 2.0 * m * n * k * s_
    # all C blocks of a rank in one array: the split fallback reduces it with ONE collective
    call = torch.zeros(sum(csizes), device="cuda", dtype=torch.float64)
    off = 0
    for q, cs in zip(groups, csizes):
        q[6] = call[off:off + cs]; off += cs

    def stacks():
        assert 0 == xs.gemm_batch_groups(xs.F64, shapes, [q[4] for q in groups], [q[5] for q in groups], [q[6] for q in groups],
                                         [q[7] for q in groups], [q[8] for q in groups], [q[9] for q in groups], [q[3] for q in groups])

    def step():
        stacks()
        if args.c5_split:
            dist_mod.reduce_partial_c(call, dist)
    stacks(); L.libxsmm_amd_jit_wait()
    wall, per_step = time_steps(torch, step, args.steps, args.warmup, dist)
    wall_max = dist_mod.max_over_ranks(wall, dist, "cuda")
    kernel = xs.last_kernel()
    _, t_comp = time_steps(torch, stacks, max(3, args.steps // 2), 1, dist)
    t_coll = [0.0]
    if dist is not None:
        _, t_coll = time_steps(torch, lambda: dist_mod.reduce_partial_c(call, dist), max(3, args.steps // 2), 1, dist)
    ms = 1e3 * wall_max / args.steps
    return {
        "metric": "CP2K stacks fp64 27 shapes GFLOP/s (whole job)", "value": round(world * flops / (wall_max / args.steps) / 1e9, 1), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[4]: CP2K-style stacks fp64, 27 shapes {13,23,32}^3, %d products per GPU, one grouped call, sums per C block in batch order; %s"
                   % (products, "stacks cut across ranks: partial C + one fused all-reduce" if args.c5_split else "every rank owns its C blocks (no exchange)"),
                   "products_per_gpu": products, "parallelism": "batch-shard x%d%s" % (world, ", all-reduce of %d C elements" % call.numel() if args.c5_split else "")},
        "kernel": kernel, "compute_ms": round(sum(t_comp) / len(t_comp), 4), "collective_ms": round(sum(t_coll) / len(t_coll), 4),
        "ms_per_step_per_rank": [round(1e3 * w / args.steps, 4) for w in gather_ranks(wall, dist, world)],
        "hbm_gbs_per_gpu": round(byt / (sum(t_comp) / len(t_comp) * 1e-3) / 1e9, 1),
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(byt / (sum(t_comp) / len(t_comp) * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(byt / (sum(t_comp) / len(t_comp) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": byt},
    }


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
