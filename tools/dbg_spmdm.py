import importlib, os, sys, ctypes as C, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
xs = importlib.import_module("libxsmm-1_amd"); import oracle_binding as orc
L = xs.lib()
M, N, K, batch = 64, 48, 64, 97
rng = np.random.default_rng(17)
a = rng.uniform(-1, 1, batch * M * K).astype(np.float32)
a[rng.random(batch * M * K) < 0.5] = 0.0
a[0:K] = 0.0; a[5 * M * K:6 * M * K] = 0.0; a[7 * M * K + 3] = -0.0; a[8 * M * K:9 * M * K] = 1.5
b = rng.uniform(-1, 1, batch * K * N).astype(np.float32)
sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, batch)
da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
assert 0 == L.libxsmm_amd_spmdm_batch_create_slices(sb, b"N", xs.dptr(da))
for rep in range(3):
  for beta in (0.0, 1.0):
    c = rng.uniform(-1, 1, batch * M * N).astype(np.float32)
    if beta == 0.0: c[:] = np.nan
    ref = c.copy()
    orc.spmdm_exec_batch(orc.FMA, M, N, K, 48, "N", "N", "N", beta, a, b, ref, batch, 4)
    dc = torch.from_numpy(c).cuda(); be = C.c_float(beta)
    assert 0 == L.libxsmm_amd_spmdm_batch_compute(sb, b"N", xs.dptr(db), b"N", C.byref(be), xs.dptr(dc))
    torch.cuda.synchronize()
    out = dc.cpu().numpy()
    bad = np.nonzero(out.view(np.uint32) != ref.view(np.uint32))[0]
    print(xs.last_kernel(), "beta", beta, "mismatches", len(bad))
    if len(bad):
        items = np.unique(bad // (M * N)); print(" items", items[:20])
        for i in bad[:8]: print("  item %d m %d n %d got %r ref %r" % (i // (M*N), (i % (M*N)) // N, i % N, out[i], ref[i]))
    if len(bad):
        i0 = bad[0]; it = i0 // (M*N); m = (i0 % (M*N)) // N
        err = (out.astype(np.float64) - ref)[it*M*N + m*N: it*M*N + (m+1)*N]
        Bi = b[it*K*N:(it+1)*K*N].reshape(K, N).astype(np.float64)
        Ai = a[it*M*K:(it+1)*M*K].reshape(M, K)
        for k in range(K):
            ratio = err / Bi[k]
            if np.max(np.abs(ratio - ratio[0])) < 1e-3 * abs(ratio[0]):
                print("   row %d: err = %.6f * B[%d]; A[m][k] = %r" % (m, ratio[0], k, Ai[m, k]))
        cerr = err
        print("   err[:6]", err[:6])
