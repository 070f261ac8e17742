#!/usr/bin/env python3
"""Developer sweep of the fp32 32^3 kernels in ONE process (same allocation, interleaved rounds): grid occupancy,
MFMA variants, non-temporal hint. Prints median/min launch time per configuration. Run on the GPU box."""
import ctypes as C
import importlib
import os
import statistics
import sys
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")  # measurements: kernels are compiled in the calling thread (no helper-thread compile behind a timed loop)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
B = int(os.environ.get("SWEEP_BATCH", "1048576"))
M = N = K = 32
torch.cuda.set_device(0)
a = torch.rand(B * 1024, device="cuda") - 0.5
b = torch.rand(B * 1024, device="cuda") - 0.5
c = torch.rand(B * 1024, device="cuda") - 0.5
blob, desc = xs.descriptor(xs.F32, M, N, K, M, K, M, 1.0, 1.0)
blob0, desc0 = xs.descriptor(xs.F32, M, N, K, M, K, M, 1.0, 0.0)

configs = []  # variant 0: generic pointers (FLAT accesses), variant 1: global address space
for bpc in (2, 3, 4):
    for variant in (0, 1):
        for nt in ((0, 1) if bpc == 3 else (1,)):
            configs.append(("mfma %s nt%d bpc%d" % ("global" if variant else "flat  ", nt, bpc), dict(XSMM_SMM32_BPC=bpc, XSMM_SMM32_VARIANT=variant, XSMM_SMM32_NT=nt), 1, desc))
            configs.append(("fma  %s nt%d bpc%d" % ("global" if variant else "flat  ", nt, bpc), dict(XSMM_SMM32_BPC=bpc, XSMM_SMM32_VARIANT=variant, XSMM_SMM32_NT=nt), 0, desc))
configs.append(("mfma flat   nt1 bpc3 beta0", dict(XSMM_SMM32_BPC=3, XSMM_SMM32_VARIANT=0, XSMM_SMM32_NT=1), 1, desc0))
for bpc in (2, 3, 4, 6):
    configs.append(("stream probe bpc%d" % bpc, dict(XSMM_STREAM_BPC=bpc), -1, None))

times = {name: [] for name, _, _, _ in configs}
rounds = int(os.environ.get("SWEEP_ROUNDS", "5"))
for rnd in range(rounds + 1):
    for name, env, mfma, d in configs:
        for k, v in env.items():
            os.environ[k] = str(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if mfma >= 0:
            L.libxsmm_amd_set_mfma(mfma)
        e0.record()
        if mfma >= 0:
            assert 0 == L.libxsmm_amd_gemm_batch_strided(d, xs.dptr(a), xs.dptr(b), xs.dptr(c), 1024, 1024, 1024, B)
        else:
            assert 0 == L.libxsmm_amd_stream_probe(xs.dptr(a), xs.dptr(b), xs.dptr(c), B * 4096)
        e1.record()
        torch.cuda.synchronize()
        if rnd > 0:
            times[name].append(e0.elapsed_time(e1))
for name, env, mfma, d in configs:
    t = times[name]
    byts = B * (12288.0 if "beta0" in name else 16384.0)
    print("%-26s median %.4f ms  min %.4f ms  -> %.0f GB/s (median)" % (name, statistics.median(t), min(t), byts / statistics.median(t) / 1e6))
