"""GPU parity of libxsmm_blocked_gemm_* (block-major GEMM with C-block accumulation) against the oracle.

Reference self-check: samples/blocked_gemm/blocked_gemm.c:121-190 (copy-in, blocked_gemm_omp, copy-out, compare with a
plain GEMM). Layout conversions are index work => bit-exact. The product itself differs from the single-threaded
reference only in association (device: chain continues from C; reference: thread-local partial sum added to C), so it is
held to the north_star tolerance.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5])
def test_blocked_gemm(xs, orc, torch_gpu, dtype, order):
    torch = torch_gpu
    m, n, k, bm, bn, bk = 128, 96, 160, 32, 32, 32
    ts = 8 if dtype == np.float64 else 4
    prec = xs.F64 if dtype == np.float64 else xs.F32
    rng = np.random.default_rng(21 + order)
    a = rng.uniform(-1, 1, m * k).astype(dtype); b = rng.uniform(-1, 1, k * n).astype(dtype); c = rng.uniform(-1, 1, m * n).astype(dtype)
    # oracle: copy-in, single-threaded blocked product, copy-out
    oh = orc.bgemm_init(ts, m, n, k, bm, bn, bk, order=order)
    assert oh is not None
    oa, ob, oc = np.zeros_like(a), np.zeros_like(b), np.zeros_like(c)
    orc.bgemm_copy(oh, "a", a, m, oa); orc.bgemm_copy(oh, "b", b, k, ob); orc.bgemm_copy(oh, "c", c, m, oc)
    oc_in = oc.copy()
    orc.bgemm_st(orc.FMA, oh, oa, ob, oc)
    oout = np.zeros_like(c); orc.bgemm_copy(oh, "out", oc, m, oout)
    L = xs.lib()
    ibm, ibn, ibk, one, iorder = (C.c_int(v) for v in (bm, bn, bk, 1, order))
    al = (C.c_double if ts == 8 else C.c_float)(1.0); be = (C.c_double if ts == 8 else C.c_float)(1.0)
    h = L.libxsmm_blocked_gemm_handle_create(1, prec, prec, m, n, k, C.byref(ibm), C.byref(ibn), C.byref(ibk),
                                             C.byref(one), C.byref(one), C.byref(one), C.byref(one),
                                             C.byref(al), C.byref(be), None, None, C.byref(iorder))
    assert h
    da, db, dc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda()
    ba, bb, bc = torch.empty_like(da), torch.empty_like(db), torch.empty_like(dc)
    ldm, ldk = C.c_int(m), C.c_int(k)
    assert 0 == L.libxsmm_blocked_gemm_copyin_a(h, xs.dptr(da), C.byref(ldm), xs.dptr(ba))
    assert 0 == L.libxsmm_blocked_gemm_copyin_b(h, xs.dptr(db), C.byref(ldk), xs.dptr(bb))
    assert 0 == L.libxsmm_blocked_gemm_copyin_c(h, xs.dptr(dc), C.byref(ldm), xs.dptr(bc))
    torch.cuda.synchronize()
    assert np.array_equal(ba.cpu().numpy(), oa) and np.array_equal(bb.cpu().numpy(), ob) and np.array_equal(bc.cpu().numpy(), oc_in)
    L.libxsmm_blocked_gemm_st(h, xs.dptr(ba), xs.dptr(bb), xs.dptr(bc), 0, 0)
    out = torch.empty_like(dc)
    assert 0 == L.libxsmm_blocked_gemm_copyout_c(h, xs.dptr(bc), C.byref(ldm), xs.dptr(out))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    L.libxsmm_blocked_gemm_handle_destroy(h)
    tol = 1e-12 if ts == 8 else 1e-6
    scale = np.max(np.abs(oout))
    assert np.max(np.abs(got.astype(np.float64) - oout.astype(np.float64))) <= tol * scale * 8  # k/bk = 5 partial sums re-associated
    # and the plain GEMM the sample compares with
    A = a.reshape(k, m).T.astype(np.float64); B = b.reshape(n, k).T.astype(np.float64); Cm = c.reshape(n, m).T.astype(np.float64)
    expect = A @ B + Cm
    assert np.max(np.abs(got.reshape(n, m).T - expect)) <= (1e-11 if ts == 8 else 2e-4)


def test_blocked_gemm_host_operands_and_invalid_blocks(xs, torch_gpu):
    L = xs.lib()
    m = n = k = 64
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, m * k); b = rng.uniform(-1, 1, k * n); c = np.zeros(m * n)
    bad = C.c_int(24)  # 64 % 24 != 0 => NULL (src/libxsmm_blocked_gemm.c:65-67)
    one = C.c_int(1)
    assert not L.libxsmm_blocked_gemm_handle_create(1, xs.F64, xs.F64, m, n, k, C.byref(bad), C.byref(bad), C.byref(bad),
                                                    C.byref(one), C.byref(one), C.byref(one), C.byref(one), None, None, None, None, None)
    blk = C.c_int(32)
    h = L.libxsmm_blocked_gemm_handle_create(4, xs.F64, xs.F64, m, n, k, C.byref(blk), C.byref(blk), C.byref(blk),
                                             C.byref(one), C.byref(one), C.byref(one), C.byref(one), None, None, None, None, None)
    assert h
    ba, bb, bc, out = np.zeros_like(a), np.zeros_like(b), np.zeros_like(c), np.zeros_like(c)
    assert 0 == L.libxsmm_blocked_gemm_copyin_a(h, xs.dptr(a), None, xs.dptr(ba))
    assert 0 == L.libxsmm_blocked_gemm_copyin_b(h, xs.dptr(b), None, xs.dptr(bb))
    assert 0 == L.libxsmm_blocked_gemm_copyin_c(h, xs.dptr(c), None, xs.dptr(bc))
    L.libxsmm_blocked_gemm_omp(h, xs.dptr(ba), xs.dptr(bb), xs.dptr(bc), 1)
    assert 0 == L.libxsmm_blocked_gemm_copyout_c(h, xs.dptr(bc), None, xs.dptr(out))
    L.libxsmm_blocked_gemm_handle_destroy(h)
    expect = a.reshape(k, m).T @ b.reshape(n, k).T
    assert np.max(np.abs(out.reshape(n, m).T - expect)) <= 1e-12 * np.max(np.abs(expect)) * 4


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("geom", [(128, 96, 160, 32, 32, 32), (96, 96, 96, 32, 32, 32), (64, 96, 48, 16, 32, 24)])
def test_blocked_permutations(xs, orc, torch_gpu, dtype, geom):
    """libxsmm_blocked_gemm_convert_b_to_a / _transpose_b (src/libxsmm_blocked_gemm.c:369-466): pure index work on
    block-major arrays => bit-exact; the second geometry takes transpose_b's square shortcut, the others its generic
    branch. Device and host operands."""
    torch = torch_gpu
    m, n, k, bm, bn, bk = geom
    ts = 8 if dtype == np.float64 else 4
    prec = xs.F64 if dtype == np.float64 else xs.F32
    rng = np.random.default_rng(m + n + k)
    oh = orc.bgemm_init(ts, m, n, k, bm, bn, bk)
    assert oh is not None
    L = xs.lib()
    ibm, ibn, ibk, one, iorder = (C.c_int(v) for v in (bm, bn, bk, 1, 0))
    al = (C.c_double if ts == 8 else C.c_float)(1.0); be = (C.c_double if ts == 8 else C.c_float)(1.0)
    h = L.libxsmm_blocked_gemm_handle_create(1, prec, prec, m, n, k, C.byref(ibm), C.byref(ibn), C.byref(ibk),
                                             C.byref(one), C.byref(one), C.byref(one), C.byref(one),
                                             C.byref(al), C.byref(be), None, None, C.byref(iorder))
    assert h
    try:
        for which, count in (("convert_b_to_a", m * n), ("transpose_b", k * n)):
            src = rng.uniform(-1, 1, count).astype(dtype)
            ref = np.full_like(src, np.nan); orc.bgemm_permute(oh, which, src, ref)
            assert not np.isnan(ref).any() and np.array_equal(np.sort(ref), np.sort(src))  # a permutation
            f = getattr(L, "libxsmm_blocked_gemm_" + which)
            dsrc = torch.from_numpy(src).cuda(); ddst = torch.empty_like(dsrc)
            assert 0 == f(h, xs.dptr(dsrc), None, xs.dptr(ddst))
            torch.cuda.synchronize()
            assert np.array_equal(ddst.cpu().numpy(), ref), which
            hdst = np.empty_like(src)
            assert 0 == f(h, xs.dptr(src), None, xs.dptr(hdst))
            assert np.array_equal(hdst, ref), which
        assert 0 != L.libxsmm_blocked_gemm_transpose_b(None, xs.dptr(src), None, xs.dptr(hdst))
    finally:
        L.libxsmm_blocked_gemm_handle_destroy(h)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("geom", [(256, 192, 320, 32, 32, 32), (256, 128, 384, 64, 64, 64), (192, 192, 192, 48, 24, 16), (128, 128, 4096, 32, 32, 32)])
@pytest.mark.parametrize("mfma", [0, 1])
def test_blocked_gemm_on_the_specialised_run_kernels(xs, torch_gpu, dtype, geom, mfma):
    """Large block GEMMs run on the hiprtc-specialised run kernels (wave / work-group per C block, segments for few long runs,
    the work-group-per-item form with uniform runs for blocks up to 64): forced here for small problems; compared with the
    plain GEMM the reference's sample checks against (samples/blocked_gemm/blocked_gemm.c:181). (128 x 128 x 4096: 16 C
    blocks with 128 k blocks each -- few long runs.)"""
    torch = torch_gpu
    m, n, k, bm, bn, bk = geom
    ts = 8 if dtype == np.float64 else 4
    prec = xs.F64 if ts == 8 else xs.F32
    rng = np.random.default_rng(m + n + k + bm)
    a = rng.uniform(-1, 1, m * k).astype(dtype); b = rng.uniform(-1, 1, k * n).astype(dtype); c = rng.uniform(-1, 1, m * n).astype(dtype)
    L = xs.lib()
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old = L.libxsmm_amd_set_mfma(mfma)  # on: blocks beyond 32 go to the matrix-core work-group kernel, a run of k blocks per unit
    try:
        ibm, ibn, ibk, one, iorder = (C.c_int(v) for v in (bm, bn, bk, 1, 0))
        al = (C.c_double if ts == 8 else C.c_float)(1.0); be = (C.c_double if ts == 8 else C.c_float)(1.0)
        h = L.libxsmm_blocked_gemm_handle_create(1, prec, prec, m, n, k, C.byref(ibm), C.byref(ibn), C.byref(ibk),
                                                 C.byref(one), C.byref(one), C.byref(one), C.byref(one), C.byref(al), C.byref(be), None, None, C.byref(iorder))
        assert h
        da, db, dc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda()
        ba, bb, bc = torch.empty_like(da), torch.empty_like(db), torch.empty_like(dc)
        ldm, ldk = C.c_int(m), C.c_int(k)
        assert 0 == L.libxsmm_blocked_gemm_copyin_a(h, xs.dptr(da), C.byref(ldm), xs.dptr(ba))
        assert 0 == L.libxsmm_blocked_gemm_copyin_b(h, xs.dptr(db), C.byref(ldk), xs.dptr(bb))
        assert 0 == L.libxsmm_blocked_gemm_copyin_c(h, xs.dptr(dc), C.byref(ldm), xs.dptr(bc))
        L.libxsmm_blocked_gemm_st(h, xs.dptr(ba), xs.dptr(bb), xs.dptr(bc), 0, 0)
        torch.cuda.synchronize()
        expect = "_jit_shape"
        if mfma and max(bm, bn) > 32:
            expect = "_mfma_wg_runs"
        elif mfma and ts == 4 and (bm, bn, bk) == (32, 32, 32):
            expect = "smm_f32_32x32x32_mfma_runs"
        elif mfma:  # blocks up to 32: the run form on the matrix cores (a wave per C block, C in the accumulators across its k blocks)
            expect = "_mfma_runs_"  # (_jit: a wave per run; _tiles_jit: small lists -- a wave per run and 16 x 16 tile of C)
        assert expect in xs.last_kernel(), xs.last_kernel()
        out = torch.empty_like(dc)
        assert 0 == L.libxsmm_blocked_gemm_copyout_c(h, xs.dptr(bc), C.byref(ldm), xs.dptr(out))
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        L.libxsmm_blocked_gemm_handle_destroy(h)
    finally:
        L.libxsmm_amd_set_mfma(old)
        if old_env is None:
            del os.environ["LIBXSMM_AMD_JIT_MINBATCH"]
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env
    A = a.reshape(k, m).T.astype(np.float64); B = b.reshape(n, k).T.astype(np.float64); Cm = c.reshape(n, m).T.astype(np.float64)
    expect = A @ B + Cm
    tol = np.finfo(dtype).eps * np.sqrt(k) * 8
    assert np.max(np.abs(got.reshape(n, m).T - expect)) <= tol * np.max(np.abs(expect))


def test_handle_rejects_low_precision(xs, torch_gpu):
    """libxsmm_blocked_gemm_handle_create with 16-bit inputs: the copy / compute kernels here move 4- and 8-byte elements only
    (the reference's I16 blocked GEMM, src/libxsmm_blocked_gemm.c:536-550, is not built) -- NULL, no handle that would read
    past the caller's buffers."""
    L = xs.lib()
    bm = C.c_int(32)
    for iprec, oprec in ((xs.I16, xs.I32), (xs.I16, xs.F32), (xs.BF16, xs.F32), (xs.BF16, xs.BF16), (xs.F32, xs.F64)):
        h = L.libxsmm_blocked_gemm_handle_create(1, iprec, oprec, 64, 64, 64, C.byref(bm), C.byref(bm), C.byref(bm),
                                                 None, None, None, None, None, None, None, None, None)
        assert not h, (iprec, oprec)
