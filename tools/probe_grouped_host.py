import importlib, sys, time, math, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
xs = importlib.import_module("libxsmm-1_amd"); L = xs.lib()
torch.cuda.set_device(0); L.libxsmm_amd_set_mfma(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
shapes = [(m, n, k) for m in (13, 23, 32) for n in (13, 23, 32) for k in (13, 23, 32)]
groups = []
for (m, n, k) in shapes:
    s = 19418; u = max(1, math.isqrt(s * 160 // 240)); nc = (s + u - 1) // u
    a = torch.rand(s * m * k, device="cuda", dtype=torch.float64, generator=g) - 0.5
    b = torch.rand(s * k * n, device="cuda", dtype=torch.float64, generator=g) - 0.5
    c = torch.zeros(nc * m * n, device="cuda", dtype=torch.float64)
    idx = torch.arange(s, device="cuda", dtype=torch.int64)
    groups.append((m, n, k, s, a, b, c, (idx * (m * k)).to(torch.int32), (idx * (k * n)).to(torch.int32), ((idx // u) * (m * n)).to(torch.int32)))
def one_call():
    assert 0 == xs.gemm_batch_groups(xs.F64, shapes, [q[4] for q in groups], [q[5] for q in groups], [q[6] for q in groups], [q[7] for q in groups], [q[8] for q in groups], [q[9] for q in groups], [q[3] for q in groups])
one_call(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): one_call()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host time per grouped call %.3f ms; 20 calls incl. drain %.3f ms per call" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
