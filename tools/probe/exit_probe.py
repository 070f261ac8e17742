import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
torch.cuda.set_device(0)
m, n, k, batch = 19, 27, 14, 4096
a = torch.rand(batch * m * k, device="cuda", dtype=torch.float64); b = torch.rand(batch * k * n, device="cuda", dtype=torch.float64)
c = torch.zeros(batch * m * n, device="cuda", dtype=torch.float64)
blob, desc = xs.descriptor(xs.F64, m, n, k)
assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, batch)
torch.cuda.synchronize()
print("KERNEL", xs.last_kernel())
if len(sys.argv) > 1:
    L.libxsmm_amd_jit_wait()
    assert 0 == L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), m * k, k * n, m * n, batch)
    torch.cuda.synchronize()
    print("KERNEL2", xs.last_kernel())
