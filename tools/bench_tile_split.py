#!/usr/bin/env python3
"""One libxsmm_gemm_batch call (index arrays on the device, fp64) over batches of 2 000 ... 60 000 products whose consecutive products share
a C block in runs of 1 ... 256: a wave per run (smm_f64_mfma_runs_jit) against a wave per run and 16 x 16 tile of C
(smm_f64_mfma_runs_tiles_jit, csrc/xsmm_jit_smm.cpp:smm_tile_split) -- the tiles as the groups of a grouped launch (XSMM_SMMJIT_TILESPLIT=1, the default)
or as the waves of one work-group (=2). The sweep splits every row (XSMM_SMMJIT_TILESPLIT_WAVES=4096).
usage: python3 tools/bench_tile_split.py"""
import importlib
import os
import sys
os.environ.setdefault("XSMM_SMMJIT_TILESPLIT_WAVES", "4096")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LIBXSMM_AMD_JIT_ASYNC", "0")
import torch
xs = importlib.import_module("libxsmm-1_amd")
torch.cuda.set_device(0)
def run(m, n, k, batch, runlen):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    a = torch.rand(batch * m * k, device="cuda", dtype=torch.float64, generator=g)
    b = torch.rand(batch * k * n, device="cuda", dtype=torch.float64, generator=g)
    nc = (batch + runlen - 1) // runlen
    c = torch.zeros(nc * m * n, device="cuda", dtype=torch.float64)
    idx = torch.arange(batch, device="cuda", dtype=torch.int64)
    ia = (idx * m * k).to(torch.int32); ib = (idx * k * n).to(torch.int32); ic = ((idx // runlen) * m * n).to(torch.int32)
    out = []
    for t in ("1", "2", "0"):
        os.environ["XSMM_SMMJIT_TILESPLIT"] = t
        ts = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, c, m, 0, 4, ia, ib, ic, batch); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append((t, xs.last_kernel(), sorted(ts)[len(ts) // 2]))
    byts = batch * 8.0 * (m * k + k * n) + nc * 16.0 * m * n
    print("%dx%dx%d batch %6d runs of %4d: " % (m, n, k, batch, runlen) + "  ".join("%s %.3f ms (%.0f GB/s)" % ({"1": "tiles/groups", "2": "tiles/wg", "0": "wave per run"}[t_] if "tiles" in nm or t_ == "0" else nm, ms, byts / ms / 1e6) for (t_, nm, ms) in out))
for (m, n, k) in ((32, 32, 32), (23, 23, 23)):
    for batch in (2000, 8000, 16000, 30000, 60000):
        for runlen in (1, 4, 32, 256):
            run(m, n, k, batch, runlen)
