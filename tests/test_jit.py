"""Run-time specialised kernels (hiprtc): source generation + compilation without a GPU, numerics on the GPU.

The reference JITs one kernel per descriptor / per sparse operator (src/libxsmm_main.c:1246-1683); here the equivalent
step is HIP source generation + hiprtc. The CPU part proves that every generated source is valid gfx950 code; the GPU part
that the specialised kernels give the oracle's fma chain bit for bit.
"""
import ctypes as C
import os

import numpy as np
import pytest


SHAPES = [(23, 23, 23), (13, 13, 13), (32, 32, 32), (13, 23, 32), (32, 13, 23), (1, 1, 1), (5, 7, 3), (8, 8, 8), (16, 16, 16),
          (24, 9, 64), (31, 32, 2), (32, 32, 33)]


def test_generated_sources_compile_for_gfx950(xs):
    L = xs.lib()
    buf = C.create_string_buffer(1 << 17)
    for prec in (xs.F64, xs.F32):
        for (m, n, k) in [(23, 23, 23), (13, 23, 32), (32, 32, 64), (1, 1, 1)]:
            for beta, flags in ((1.0, 0), (0.0, 0), (1.0, xs.FLAG_TRANS_B)):
                blob, d = xs.descriptor(prec, m, n, k, beta=beta, flags=flags)
                rc = L.libxsmm_amd_smm_kernel_source(d, buf, len(buf), 1)
                if rc == -1:
                    pytest.skip("libhiprtc is not available here")
                assert rc == 0, (prec, m, n, k, beta, flags)
                src = buf.value.decode()
                assert "#define XM %d" % m in src and "xsmm_smm_op" in src
    # fixed-sparsity operator
    rng = np.random.default_rng(0)
    M, K = 35, 35
    A = np.where(rng.random((M, K)) < 0.15, rng.integers(1, 8, (M, K)) * 0.25, 0.0)
    rowptr = np.concatenate([[0], np.cumsum((A != 0).sum(axis=1))]).astype(np.uint32)
    colidx = np.nonzero(A)[1].astype(np.uint32); vals = A[A != 0].astype(np.float64)
    for ts, vec in ((8, 1), (8, 2), (4, 1), (4, 4)):
        for beta0 in (0, 1):
            assert 0 == L.libxsmm_amd_csr_kernel_source(ts, M, K, xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(vals), beta0, vec, buf, len(buf), 1)
    src = buf.value.decode()
    assert src.count("xf(") >= len(vals)  # one fma per non-zero (times the vector width)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_jit_dense_kernels_bitexact(xs, orc, torch_gpu, dtype, shape):
    torch = torch_gpu
    m, n, k = shape
    if (m, n, k) == (32, 32, 32) and dtype == np.float32:
        pytest.skip("served by the hand-tuned kernel")
    batch = 531
    rng = np.random.default_rng(m * 97 + n * 13 + k)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        for beta, flags in ((1.0, 0), (0.0, 0), (1.0, xs.FLAG_TRANS_B)):
            a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
            c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
            if beta == 0.0:
                c[:] = np.nan
            ref = c.copy()
            oflags = (orc.FLAG_BETA_0 if beta == 0.0 else 0) | (orc.FLAG_TRANS_B if flags else 0)
            ldb = n if flags else k
            orc.gemm_batch_strided(orc.FMA, oflags, m, n, k, m, ldb, m, a, b, ref, m * k, k * n, m * n, batch, 4)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            blob, desc = xs.descriptor(prec, m, n, k, beta=beta, flags=flags)
            assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), m * k, k * n, m * n, batch)
            torch.cuda.synchronize()
            assert xs.last_kernel().endswith("_jit_shape"), xs.last_kernel()
            assert np.array_equal(dc.cpu().numpy(), ref), (shape, beta, flags)
        # shared B (stride 0) stays on the specialised path too
        a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b1 = rng.uniform(-1, 1, k * n).astype(dtype); c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
        ref = c.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b1, ref, m * k, 0, m * n, batch, 4)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b1, c))
        blob, desc = xs.descriptor(prec, m, n, k)
        assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), m * k, 0, m * n, batch)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), ref)
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
        if old_env is None:
            del os.environ["LIBXSMM_AMD_JIT_MINBATCH"]
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env
