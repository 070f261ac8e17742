"""Randomised differential test of the dense batch paths against the oracle: shapes, leading dimensions (tight or with gaps),
beta, TRANS_B, the three addressing modes, distinct / run-wise / unordered C, strict and relaxed entry points, with and without
forcing the hiprtc-specialised kernels. Fixed seed; bit-exact wherever the order of the sums is defined, tolerance otherwise."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng, big_only=False):
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    big = big_only or rng.random() < 0.15
    m, n = int(rng.integers(1, 65 if big else 33)), int(rng.integers(1, 65 if big else 33))
    if big_only and max(m, n) <= 32:
        m = int(rng.integers(33, 65))
    k = int(rng.integers(1, 80 if rng.random() < 0.2 else 40))
    transb = rng.random() < 0.25
    gaps = rng.random() < 0.35
    lda = m + (int(rng.integers(0, 6)) if gaps else 0)
    ldb = (n if transb else k) + (int(rng.integers(0, 6)) if gaps else 0)
    ldc = m + (int(rng.integers(0, 6)) if gaps else 0)
    beta = 0.0 if rng.random() < 0.3 else 1.0
    mode = ("strided", "index", "pointer")[int(rng.integers(0, 3))]
    cpat = ("distinct", "runs", "unordered")[int(rng.integers(0, 3))] if (mode != "strided" and beta == 1.0) else "distinct"
    batch = int(rng.integers(1, 2500))
    relaxed = rng.random() < 0.4
    forced = big_only or rng.random() < 0.7
    return dtype, m, n, k, lda, ldb, ldc, transb, beta, mode, cpat, batch, relaxed, forced


@pytest.mark.parametrize("chunk", list(range(6)) + [100, 101, 102])
def test_dense_batches_fuzz(xs, orc, torch_gpu, chunk):
    """(chunks 100+: shapes beyond 32 only, matrix cores on, specialised kernels forced -- the one-wave-per-item forms with chunks,
    element by element, gaps, TRANS_B, index and pointer batches, next to the work-group and run forms)"""
    torch = torch_gpu
    rng = np.random.default_rng(20241004 + chunk)
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    big_only = chunk >= 100
    old_mfma = xs.lib().libxsmm_amd_set_mfma(1 if big_only else int(chunk % 2))
    try:
        for it in range(30):
            dtype, m, n, k, lda, ldb, ldc, transb, beta, mode, cpat, batch, relaxed, forced = _case(rng, big_only)
            if forced:
                os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
            else:
                os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
            prec = xs.F64 if dtype == np.float64 else xs.F32
            asz, bsz, csz = lda * k, ldb * (k if transb else n), ldc * n
            nc = batch if cpat == "distinct" else max(1, batch // int(rng.integers(2, 40)))
            a = rng.uniform(-1, 1, batch * asz).astype(dtype); b = rng.uniform(-1, 1, batch * bsz).astype(dtype)
            c = rng.uniform(-1, 1, nc * csz).astype(dtype)
            if cpat == "distinct":
                cidx = np.arange(batch)
            elif cpat == "runs":
                cidx = np.sort(rng.integers(0, nc, batch))
            else:
                cidx = rng.integers(0, nc, batch)
            if beta == 0.0:
                c.reshape(nc, n, ldc)[:, :, :m] = np.nan
            sa = (rng.permutation(batch) * asz).astype(np.int32) if mode != "strided" else (np.arange(batch) * asz).astype(np.int32)
            sb = (np.arange(batch) * bsz).astype(np.int32); sc = (cidx * csz).astype(np.int32)
            oflags = (orc.FLAG_BETA_0 if beta == 0.0 else 0) | (orc.FLAG_TRANS_B if transb else 0)
            ref = c.copy()
            assert 0 == orc.gemm_batch_idx(orc.FMA, oflags, m, n, k, lda, ldb, ldc, a, b, ref, 0, sa, sb, sc, batch)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            tag = (chunk, it, dtype.__name__, m, n, k, lda, ldb, ldc, transb, beta, mode, cpat, batch, relaxed, forced)
            tb = "T" if transb else "N"
            if mode == "strided":
                blob, d = xs.descriptor(prec, m, n, k, lda, ldb, ldc, 1.0, beta, xs.FLAG_TRANS_B if transb else 0, 0)
                assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(d, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch), tag
            elif mode == "index":
                xs.gemm_batch(prec, "N", tb, m, n, k, 1.0, da, lda, db, ldb, beta, dc, ldc, 0, 4, sa, sb, sc, batch, omp=relaxed)
            else:
                ts = a.itemsize
                pa = torch.from_numpy((da.data_ptr() + sa.astype(np.int64) * ts)).cuda()
                pb = torch.from_numpy((db.data_ptr() + sb.astype(np.int64) * ts)).cuda()
                pc = torch.from_numpy((dc.data_ptr() + sc.astype(np.int64) * ts)).cuda()
                ptrsize = np.array([8], dtype=np.int32)
                xs.gemm_batch(prec, "N", tb, m, n, k, 1.0, pa, lda, pb, ldb, beta, pc, ldc, 0, 0, ptrsize, ptrsize, ptrsize, batch, omp=relaxed)
            torch.cuda.synchronize()
            out = dc.cpu().numpy()
            ordered = (cpat == "distinct") or (cpat == "runs" and not relaxed)
            if ordered:
                assert np.array_equal(out.view(np.uint8), ref.view(np.uint8)), (tag, xs.last_kernel())
            else:
                terms = max(1, batch // max(1, nc)) * k * 8
                tol = np.finfo(dtype).eps * np.sqrt(terms) * 16
                good = ~np.isnan(ref)
                assert np.array_equal(np.isnan(out), np.isnan(ref)), (tag, xs.last_kernel())
                assert np.max(np.abs(out[good] - ref[good])) <= tol * max(1.0, np.max(np.abs(ref[good]))), (tag, xs.last_kernel())
    finally:
        xs.lib().libxsmm_amd_set_mfma(old_mfma)
        if old_env is None:
            os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env


@pytest.mark.parametrize("chunk", range(3))
def test_spmdm_batches_fuzz(xs, orc, torch_gpu, chunk):
    """spmdm batch extension over random geometries, densities, beta and transposes: gather kernel, matrix-core kernel (geometry
    permitting), generic kernel -- all bit-identical to the oracle's chain."""
    import ctypes as C
    torch = torch_gpu
    L = xs.lib()
    rng = np.random.default_rng(777 + chunk)
    old_mfma = L.libxsmm_amd_set_mfma(1)
    try:
        for it in range(16):
            M = int(rng.choice([64, 64, 48, 32, 16, int(rng.integers(1, 65))])); K = int(rng.choice([64, 64, 32, 60, int(rng.integers(1, 65))]))
            N = int(rng.choice([48, 16, 32, 64, 20, int(rng.integers(1, 65))]))
            batch = int(rng.integers(1, 200))
            density = float(rng.choice([0.02, 0.15, 0.5, 0.9, 1.0]))
            ta, tb, tc = ("N", "N", "N") if rng.random() < 0.7 else (("T", "N", "T") if rng.random() < 0.5 else ("N", "T", "N"))
            beta = float(rng.choice([0.0, 1.0, 0.5]))
            L.libxsmm_amd_set_mfma(int(rng.integers(0, 2)))
            a = rng.uniform(-1, 1, batch * M * K).astype(np.float32)
            a[rng.random(batch * M * K) >= density] = 0.0
            b = rng.uniform(-1, 1, batch * K * N).astype(np.float32); c = rng.uniform(-1, 1, batch * M * N).astype(np.float32)
            if beta == 0.0:
                c[:] = np.nan
            ref = c.copy()
            orc.spmdm_exec_batch(orc.FMA, M, N, K, 48, ta, tb, tc, beta, a, b, ref, batch, 4)
            sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, batch)
            assert sb
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            assert 0 == L.libxsmm_amd_spmdm_batch_create_slices(sb, ta.encode(), xs.dptr(da))
            be = C.c_float(beta)
            assert 0 == L.libxsmm_amd_spmdm_batch_compute(sb, tb.encode(), xs.dptr(db), tc.encode(), C.byref(be), xs.dptr(dc))
            torch.cuda.synchronize()
            out = dc.cpu().numpy()
            L.libxsmm_amd_spmdm_batch_destroy(sb)
            assert np.array_equal(out.view(np.uint8), ref.view(np.uint8)), (chunk, it, M, N, K, batch, density, ta, tb, tc, beta, xs.last_kernel())
    finally:
        L.libxsmm_amd_set_mfma(old_mfma)


@pytest.mark.parametrize("chunk", range(2))
def test_fsspmdm_fuzz(xs, orc, torch_gpu, chunk):
    """fsspmdm over random operators (shape, density, lda), precisions, beta, panel counts; the specialised operator kernel and
    (LIBXSMM_AMD_JIT=0) the generic CSR kernel -- both follow the oracle's per-element chain bit for bit (rows without
    non-zeros excluded for beta = 0: the reference's two code paths disagree there)."""
    import ctypes as C
    torch = torch_gpu
    L = xs.lib()
    rng = np.random.default_rng(4242 + chunk)
    old_jit = os.environ.get("LIBXSMM_AMD_JIT")
    try:
        for it in range(10):
            dtype = np.float64 if rng.random() < 0.5 else np.float32
            M, K = int(rng.integers(1, 60)), int(rng.integers(1, 60))
            N = 16 * int(rng.integers(1, 8)); panels = int(rng.integers(1, 9))
            lda = K + int(rng.integers(0, 4))
            density = float(rng.choice([0.05, 0.15, 0.5, 1.0]))
            beta = float(rng.choice([0.0, 1.0]))
            if rng.random() < 0.3:
                os.environ["LIBXSMM_AMD_JIT"] = "0"
            else:
                os.environ.pop("LIBXSMM_AMD_JIT", None)
            A = np.zeros((M, lda), dtype=dtype)
            A[:, :K] = np.where(rng.random((M, K)) < density, rng.uniform(-1, 1, (M, K)), 0.0)
            ntot = N * panels
            B = rng.uniform(-1, 1, (K, ntot)).astype(dtype); Cin = rng.uniform(-1, 1, (M, ntot)).astype(dtype)
            ref = Cin.copy()
            h = orc.Fsspmdm(A, M, N, K, lda, ntot, ntot, 1.0, beta, have_avx512=False)  # dense fallback: the plain chain for every row
            for p in range(panels):
                h.execute(B.reshape(-1)[p * N:], ref.reshape(-1)[p * N:])
            h.close()
            create = L.libxsmm_dfsspmdm_create if dtype == np.float64 else L.libxsmm_sfsspmdm_create
            execb = L.libxsmm_amd_dfsspmdm_execute_batch if dtype == np.float64 else L.libxsmm_amd_sfsspmdm_execute_batch
            destroy = L.libxsmm_dfsspmdm_destroy if dtype == np.float64 else L.libxsmm_sfsspmdm_destroy
            hd = create(M, N, K, lda, ntot, ntot, 1.0, beta, xs.dptr(A))
            assert hd
            dB, dC = torch.from_numpy(B).cuda(), torch.from_numpy(Cin).cuda()
            assert 0 == execb(hd, xs.dptr(dB), xs.dptr(dC), panels)
            torch.cuda.synchronize()
            out = dC.cpu().numpy()
            destroy(hd)
            assert np.array_equal(out.view(np.uint8), ref.view(np.uint8)), (chunk, it, dtype.__name__, M, N, K, lda, density, beta, panels, xs.last_kernel())
    finally:
        if old_jit is None:
            os.environ.pop("LIBXSMM_AMD_JIT", None)
        else:
            os.environ["LIBXSMM_AMD_JIT"] = old_jit


@pytest.mark.parametrize("chunk", range(2))
def test_matrix_core_work_group_kernels_fuzz(xs, orc, torch_gpu, chunk):
    """Random shapes of the class served by the matrix-core work-group kernels (32 < max(M, N) <= 64, K <= 64; tight or with
    gaps in the leading dimensions; odd K; fp32 and fp64; beta 0/1): bit for bit the oracle's k-ordered fma chain, signs of
    zeros included."""
    torch = torch_gpu
    from test_smm_gpu import make_inputs
    rng = np.random.default_rng(6464 + chunk)
    old = xs.lib().libxsmm_amd_set_mfma(1)
    try:
        for it in range(24):
            dtype = np.float64 if rng.random() < 0.5 else np.float32
            m, n = int(rng.integers(1, 65)), int(rng.integers(1, 65))
            if max(m, n) <= 32:
                m = int(rng.integers(33, 65))
            k = int(rng.integers(1, 65))
            gaps = rng.random() < 0.4
            lda, ldb, ldc = (m + int(rng.integers(0, 9)), k + int(rng.integers(0, 9)), m + int(rng.integers(0, 9))) if gaps else (m, k, m)
            beta = 0.0 if rng.random() < 0.3 else 1.0
            batch = int(rng.integers(1, 3000))
            a, b, c, asz, bsz, csz = make_inputs(rng, dtype, batch, m, n, k, lda, ldb, ldc, False, False, orc)
            a[:asz] = 0.0
            c[:csz] = -0.0
            if beta == 0.0:
                c[:] = np.nan
                for i in range(batch):
                    c[i * csz:(i + 1) * csz].reshape(n, ldc)[:, m:] = 0.5
            flags = xs.FLAG_BETA_0 if beta == 0.0 else 0
            ref = c.copy()
            orc.gemm_batch_strided(orc.FMA, flags, m, n, k, lda, ldb, ldc, a, b, ref, asz, bsz, csz, batch, 8)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            blob, desc = xs.descriptor(xs.F64 if dtype == np.float64 else xs.F32, m, n, k, lda, ldb, ldc, 1.0, beta)
            assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
            torch.cuda.synchronize()
            out = dc.cpu().numpy()
            case = (dtype.__name__, m, n, k, lda, ldb, ldc, beta, batch, xs.last_kernel())
            assert "mfma" in xs.last_kernel(), case
            bits = np.uint64 if dtype == np.float64 else np.uint32
            assert np.array_equal(out.view(bits), ref.view(bits)), case
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
