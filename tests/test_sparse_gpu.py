"""GPU parity of the sparse paths (fsspmdm, csr_reg kernels, spmdm) through the C-ABI against the CPU oracle.

Reference self-checks mirrored here: samples/pyfr/pyfr_driver_asp_reg.c:275-347 (CSR gold loop, beta=0 and beta=1, operator
matrices from samples/pyfr/mats), samples/spmdm/spmdm.c:212-224,274-301 (inputs keep a value iff r > 0.85, naive gold,
NN / TN+transC / NT variants). Index structures (CSR slices) must be bit-exact; values follow the same fma chains as the
oracle, so they are compared for equality as well.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def palette_operator(rng, M, K, density, nvalues):
    pal = np.array([0.25, -0.5, 0.75, 1.0, -1.25, 1.5, -2.0, 3.0, -0.125, 0.0625][:nvalues])
    return np.where(rng.random((M, K)) < density, pal[rng.integers(0, nvalues, (M, K))], 0.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("beta", [1.0, 0.0])
@pytest.mark.parametrize("unique", ["palette", "random"])
def test_fsspmdm_synthetic(xs, orc, torch_gpu, dtype, beta, unique):
    """BASELINE config 3 operator shape (M=K=35, ~15% nnz, N=96 panels). palette -> the reference's csr_reg path,
    random values -> its dense fallback; both follow the same per-element chain as the device kernel."""
    torch = torch_gpu
    M, K, N, panels = 35, 35, 96, 7
    rng = np.random.default_rng(1)
    A = palette_operator(rng, M, K, 0.15, 7) if unique == "palette" else np.where(rng.random((M, K)) < 0.15, rng.uniform(-1, 1, (M, K)), 0.0)
    A[5, :] = 0.0  # a row without non-zeros (the beta == 0 quirk row)
    A = np.ascontiguousarray(A.astype(dtype))
    ntot = N * panels
    B = rng.uniform(-1, 1, (K, ntot)).astype(dtype)
    Cin = rng.uniform(-1, 1, (M, ntot)).astype(dtype)
    ref = Cin.copy()
    h = orc.Fsspmdm(A, M, N, K, K, ntot, ntot, 1.0, beta, have_avx512=True)
    assert h.sparse() == (1 if unique == "palette" else 0)
    for p in range(panels):  # pyfr_driver_asp_reg.c:295-309: execute(handle, B + i*N, C + i*N)
        h.execute(B.reshape(-1)[p * N:], ref.reshape(-1)[p * N:])
    h.close()
    L = xs.lib()
    create = L.libxsmm_dfsspmdm_create if dtype == np.float64 else L.libxsmm_sfsspmdm_create
    execute = L.libxsmm_dfsspmdm_execute if dtype == np.float64 else L.libxsmm_sfsspmdm_execute
    execb = L.libxsmm_amd_dfsspmdm_execute_batch if dtype == np.float64 else L.libxsmm_amd_sfsspmdm_execute_batch
    destroy = L.libxsmm_dfsspmdm_destroy if dtype == np.float64 else L.libxsmm_sfsspmdm_destroy
    hd = create(M, N, K, K, ntot, ntot, 1.0, beta, xs.dptr(A))
    assert hd
    dB = torch.from_numpy(B).cuda()
    # (1) panel by panel through the reference entry point, (2) all panels in one batch launch
    for batched in (False, True):
        dC = torch.from_numpy(Cin).cuda()
        if batched:
            assert 0 == execb(hd, xs.dptr(dB), xs.dptr(dC), panels)
        else:
            es = dC.element_size()
            for p in range(panels):
                execute(hd, C.c_void_p(dB.data_ptr() + p * N * es), C.c_void_p(dC.data_ptr() + p * N * es))
        torch.cuda.synchronize()
        out = dC.cpu().numpy()
        assert xs.last_kernel().startswith("fsspmdm_")
        rows = [r for r in range(M) if r != 5]
        assert np.array_equal(out[rows], ref[rows])
        if beta == 0.0:  # empty row: zeroed (the dense fallback's result); the reference's csr_reg path leaves it untouched
            assert np.all(out[5] == 0.0)
            assert np.array_equal(ref[5], Cin[5]) if unique == "palette" else np.all(ref[5] == 0.0)
        else:
            assert np.array_equal(out[5], Cin[5])
    destroy(hd)


def test_fsspmdm_pyfr_operators_sparse_equals_dense(xs, orc, torch_gpu):
    """Fixtures the reference holds: each PyFR operator exists as a sparse (-sp.mtx) and a dense (-de.mtx) file
    (samples/pyfr/mats). The operator read through the CSR reader and applied by fsspmdm must agree with the dense
    file applied as a plain GEMM -- on the oracle and on the device."""
    torch = torch_gpu
    files = sorted(glob.glob(os.path.join(GOLDEN, "mtx", "pyfr", "*-sp.mtx")))
    assert len(files) >= 8
    rng = np.random.default_rng(3)
    N = 96
    for sp in files:
        rowptr, colidx, vals, rows, cols, nnz = orc.read_csr(sp)
        dense = orc.read_dense_mtx(sp.replace("-sp.mtx", "-de.mtx"))
        assert dense.shape == (rows, cols)
        A = np.zeros((rows, cols))
        for r in range(rows):
            for q in range(rowptr[r], rowptr[r + 1]):
                A[r, colidx[q]] = vals[q]
        # the sparse file drops entries the dense file holds as (near) zeros; the operators agree to rounding of the text format
        assert np.max(np.abs(A - dense)) <= 1e-10 * max(1.0, np.max(np.abs(dense)))
        B = rng.uniform(-1, 1, (cols, N)); Cin = rng.uniform(-1, 1, (rows, N))
        for beta in (1.0, 0.0):
            expect = A @ B + beta * Cin
            hd = xs.lib().libxsmm_dfsspmdm_create(rows, N, cols, cols, N, N, 1.0, beta, xs.dptr(np.ascontiguousarray(A)))
            assert hd
            dB, dC = torch.from_numpy(B).cuda(), torch.from_numpy(Cin.copy()).cuda()
            xs.lib().libxsmm_dfsspmdm_execute(hd, xs.dptr(dB), xs.dptr(dC))
            torch.cuda.synchronize()
            got = dC.cpu().numpy()
            xs.lib().libxsmm_dfsspmdm_destroy(hd)
            # oracle on the same operator
            ref = Cin.copy()
            h = orc.Fsspmdm(np.ascontiguousarray(A), rows, N, cols, cols, N, N, 1.0, beta, have_avx512=False)
            h.execute(B, ref); h.close()
            scale = max(1.0, np.max(np.abs(expect)))
            assert np.max(np.abs(ref - expect)) <= 1e-12 * scale
            assert np.max(np.abs(got - expect)) <= 1e-12 * scale
            assert np.array_equal(got, ref)


def test_create_csr_reg_kernel_and_limits(xs, orc, torch_gpu):
    """libxsmm_create_dcsr_reg / scsr_reg (src/libxsmm_main.c:2523-2582): N must equal the reference's vector length,
    at most 31 unique values, rows without nnz are not touched; released with libxsmm_release_kernel."""
    torch = torch_gpu
    rng = np.random.default_rng(5)
    M, K = 20, 27
    A = palette_operator(rng, M, K, 0.2, 6); A[3, :] = 0.0
    rowptr = np.zeros(M + 1, dtype=np.uint32); cols = []; vals = []
    for r in range(M):
        rowptr[r] = len(cols)
        for c in range(K):
            if A[r, c] != 0: cols.append(c); vals.append(A[r, c])
    rowptr[M] = len(cols)
    colidx = np.array(cols, dtype=np.uint32); values = np.array(vals)
    for dtype, prec, vlen, creator in ((np.float64, xs.F64, 8, "libxsmm_create_dcsr_reg"), (np.float32, xs.F32, 16, "libxsmm_create_scsr_reg")):
        ldb, ldc = vlen + 3, vlen + 5
        for beta in (1.0, 0.0):
            blob, desc = xs.descriptor(prec, M, vlen, K, 0, ldb, ldc, 1.0, beta)
            fn = getattr(xs.lib(), creator)(desc, xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(values.astype(dtype)))
            assert fn
            B = rng.uniform(-1, 1, (K, ldb)).astype(dtype); Cin = rng.uniform(-1, 1, (M, ldc)).astype(dtype)
            ref = Cin.copy()
            assert 0 == orc.csr_reg(orc.FLAG_BETA_0 if beta == 0 else 0, M, vlen, K, ldb, ldc, rowptr, colidx, values.astype(dtype), B, ref)
            dB, dC = torch.from_numpy(B).cuda(), torch.from_numpy(Cin).cuda()
            xs.call_kernel(fn, None, dB, dC)
            torch.cuda.synchronize()
            assert np.array_equal(dC.cpu().numpy(), ref)       # includes row 3 untouched and the ld padding untouched
            hC = Cin.copy(); xs.call_kernel(fn, None, B, hC)   # host operands
            assert np.array_equal(hC, ref)
            xs.lib().libxsmm_release_kernel(fn)
        # wrong chunk width and too many unique values => NULL, as in the reference generator
        blob, desc = xs.descriptor(prec, M, vlen + 1, K, 0, ldb, ldc, 1.0, 1.0)
        assert not getattr(xs.lib(), creator)(desc, xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(values.astype(dtype)))
        many = rng.uniform(-1, 1, len(values)).astype(dtype)
        blob, desc = xs.descriptor(prec, M, vlen, K, 0, ldb, ldc, 1.0, 1.0)
        assert not getattr(xs.lib(), creator)(desc, xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(many))


def spmdm_inputs(M, N, K, keep, seed, orc):
    """samples/spmdm/spmdm.c:212-243: A first (value kept iff r > threshold), then B, then C, from rng seed 1"""
    orc.rng_seed(seed)
    a = np.zeros(M * K, dtype=np.float32)
    for i in range(M * K):
        r = orc.rng_f64()
        a[i] = np.float32(r) if r > keep else 0.0
    b = np.array([orc.rng_f64() for _ in range(K * N)], dtype=np.float32)
    c = np.array([orc.rng_f64() for _ in range(M * N)], dtype=np.float32)
    return a, b, c


@pytest.mark.parametrize("variant", [("N", "N", "N"), ("T", "N", "T"), ("N", "T", "N"), ("T", "T", "T")])
@pytest.mark.parametrize("beta", [0.0, 1.0, 0.5])
def test_spmdm_reference_api(xs, orc, torch_gpu, variant, beta):
    """libxsmm_spmdm_init / createSparseSlice_fp32_thread / compute_fp32_thread on one problem that spans several
    (mb, nb, kb) blocks with ragged edges; call sequence of samples/spmdm/spmdm.c:74-112."""
    torch = torch_gpu
    ta, tb, tc = variant
    M, N, K = 301, 131, 263
    a, b, c = spmdm_inputs(M, N, K, 0.85, 1, orc)
    if beta == 0.0:
        c[:] = np.nan  # beta == 0 never reads C (compute tpl :81-105)
    ref = c.copy()
    orc.spmdm_exec(orc.FMA, M, N, K, 48, ta, tb, tc, beta, a, b, ref)
    L = xs.lib()
    h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
    L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
    assert h.m == M and h.n == N and h.k == K and h.mb * h.bm >= M and h.kb * h.bk >= K and h.nb * h.bn >= N
    da, db, dc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda()
    alpha, be = C.c_float(1.0), C.c_float(beta)
    for blk in range(L.libxsmm_spmdm_get_num_createSparseSlice_blocks(C.byref(h))):
        L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), ta.encode(), xs.dptr(da), slices, blk, 0, 1)
    for blk in range(L.libxsmm_spmdm_get_num_compute_blocks(C.byref(h))):
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), ta.encode(), tb.encode(), C.byref(alpha), slices, xs.dptr(db), tc.encode(),
                                            C.byref(be), xs.dptr(dc), blk, 0, 1)
    torch.cuda.synchronize()
    out = dc.cpu().numpy()
    L.libxsmm_spmdm_destroy(C.byref(h))
    assert np.array_equal(out, ref)
    # and against the sample's own gold: a naive triple loop in float64 (samples/spmdm/spmdm.c:274-297), max abs error
    if beta == 0.0:
        A = a.reshape(K, M).T if ta == "T" else a.reshape(M, K)
        B = b.reshape(N, K).T if tb == "T" else b.reshape(K, N)
        gold = A.astype(np.float64) @ B.astype(np.float64)
        got = out.reshape(N, M).T if tc == "T" else out.reshape(M, N)
        assert np.max(np.abs(got - gold)) <= 1e-4


@pytest.mark.parametrize("variant", [("N", "N", "N"), ("T", "T", "T")])
@pytest.mark.parametrize("beta_bits", [0, 1, 0x3F80])
def test_spmdm_bfloat16_twins(xs, orc, torch_gpu, variant, beta_bits):
    """libxsmm_spmdm_createSparseSlice_bfloat16_thread / compute_bfloat16_thread (include/libxsmm_spmdm.h:98-133): inputs are
    upper halves of floats, slices/sums/C are float. `*beta` is taken as the number its 16 bits spell (pattern 1 is
    beta = 1; the bf16 encoding of 1.0 scales C by 16256) -- the reference template's behaviour, reproduced bit for bit.
    Device and host operands."""
    torch = torch_gpu
    ta, tb, tc = variant
    M, N, K = 150, 70, 200
    a32, b32, c = spmdm_inputs(M, N, K, 0.7, 1, orc)
    a = (a32.view(np.uint32) >> 16).astype(np.uint16); b = (b32.view(np.uint32) >> 16).astype(np.uint16)  # truncation keeps zeros zero
    if beta_bits == 0:
        c[:] = np.nan
    ref = c.copy()
    orc.spmdm_exec_bf16(orc.FMA, M, N, K, 48, ta, tb, tc, beta_bits, a, b, ref)
    L = xs.lib()
    for on_device in (True, False):
        h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
        L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
        if on_device:
            xa, xb, xc = torch.from_numpy(a.view(np.int16)).cuda(), torch.from_numpy(b.view(np.int16)).cuda(), torch.from_numpy(c).cuda()
        else:
            xa, xb, xc = a, b, c.copy()
        alpha, be = C.c_ushort(0x3F80), C.c_ushort(beta_bits)
        for blk in range(L.libxsmm_spmdm_get_num_createSparseSlice_blocks(C.byref(h))):
            L.libxsmm_spmdm_createSparseSlice_bfloat16_thread(C.byref(h), ta.encode(), xs.dptr(xa), slices, blk, 0, 1)
        for blk in range(L.libxsmm_spmdm_get_num_compute_blocks(C.byref(h))):
            L.libxsmm_spmdm_compute_bfloat16_thread(C.byref(h), ta.encode(), tb.encode(), C.byref(alpha), slices, xs.dptr(xb), tc.encode(),
                                                    C.byref(be), xs.dptr(xc), blk, 0, 1)
        torch.cuda.synchronize()
        out = xc.cpu().numpy() if on_device else xc
        L.libxsmm_spmdm_destroy(C.byref(h))
        assert np.array_equal(out, ref), on_device


@pytest.mark.parametrize("keep", [0.5, 0.85])
@pytest.mark.parametrize("variant", [("N", "N", "N"), ("T", "N", "T"), ("N", "T", "N")])
def test_spmdm_batch_config4(xs, orc, torch_gpu, keep, variant):
    """BASELINE config 4 problem (M=K=64, N=48) as a batch: CSR slices bit-exact per item, C equal to the oracle."""
    torch = torch_gpu
    ta, tb, tc = variant
    M, N, K, batch = 64, 48, 64, 97
    rng = np.random.default_rng(17)
    a = rng.uniform(-1, 1, batch * M * K).astype(np.float32)
    a[rng.random(batch * M * K) < keep] = 0.0
    a[0:K] = 0.0                      # an empty first row in item 0
    a[5 * M * K:6 * M * K] = 0.0      # item 5: all zeros
    a[7 * M * K + 3] = -0.0           # -0 counts as zero (LIBXSMM_FEQ)
    a[8 * M * K:9 * M * K] = 1.5      # item 8: fully dense (4096 nnz)
    b = rng.uniform(-1, 1, batch * K * N).astype(np.float32)
    L = xs.lib()
    sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, batch)
    assert sb
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert 0 == L.libxsmm_amd_spmdm_batch_create_slices(sb, ta.encode(), xs.dptr(da))
    assert xs.last_kernel().startswith("spmdm_create")
    ri = np.zeros(M + 1, dtype=np.uint16); ci = np.zeros(M * K, dtype=np.uint16); va = np.zeros(M * K, dtype=np.float32)
    for item in (0, 1, 5, 7, 8, 50, batch - 1):
        assert 0 == L.libxsmm_amd_spmdm_batch_get_slice(sb, item, xs.dptr(ri), xs.dptr(ci), xs.dptr(va), M * K)
        hnd, sl = orc.spmdm_slices(M, N, K, 48, ta, a[item * M * K:(item + 1) * M * K])
        assert hnd.mb == 1 and hnd.kb == 1
        oi, oc, ov = sl[0]
        nnz = int(oi[M])
        assert np.array_equal(ri, oi)
        assert np.array_equal(ci[:nnz], oc) and np.array_equal(va[:nnz].view(np.uint32), ov.view(np.uint32))
    for beta in (0.0, 1.0):
        c = rng.uniform(-1, 1, batch * M * N).astype(np.float32)
        if beta == 0.0:
            c[:] = np.nan
        ref = c.copy()
        orc.spmdm_exec_batch(orc.FMA, M, N, K, 48, ta, tb, tc, beta, a, b, ref, batch, 4)
        dc = torch.from_numpy(c).cuda()
        be = C.c_float(beta)
        assert 0 == L.libxsmm_amd_spmdm_batch_compute(sb, tb.encode(), xs.dptr(db), tc.encode(), C.byref(be), xs.dptr(dc))
        torch.cuda.synchronize()
        assert xs.last_kernel().startswith("spmdm_compute")
        assert np.array_equal(dc.cpu().numpy(), ref)
    L.libxsmm_amd_spmdm_batch_destroy(sb)


@pytest.mark.parametrize("beta", [0.0, 1.0, 0.5])
@pytest.mark.parametrize("mfma", [1, 0])
def test_spmdm_batch_nonfinite_b_under_zeros(xs, orc, torch_gpu, beta, mfma):
    """B holds inf / NaN in rows that meet only structural zeros of A (and, in other items, real entries). The reference
    skips absent entries (compute tpl :321-371: the loop runs over the CSR entries), so C stays finite where no entry meets
    the bad value and becomes inf / NaN exactly where one does. At 50 % density the batch goes to the matrix-core kernel, which
    multiplies a zero-filled slice (0 * inf = NaN): it has to notice such tiles and work them off entry by entry."""
    torch = torch_gpu
    L = xs.lib()
    M, N, K, batch = 64, 48, 64, 41
    rng = np.random.default_rng(29)
    a = rng.uniform(-1, 1, batch * M * K).astype(np.float32)
    a[rng.random(batch * M * K) < 0.5] = 0.0
    A = a.reshape(batch, M, K)
    b = rng.uniform(-1, 1, batch * K * N).astype(np.float32)
    Bt = b.reshape(batch, K, N)
    A[1, :, 7] = 0.0; Bt[1, 7, :] = np.inf                     # a whole row of B under a column of zeros: C of item 1 stays finite
    A[2, :, 9] = 0.0; Bt[2, 9, 5] = np.nan; Bt[2, 9, 40] = -np.inf
    Bt[3, 11, 3] = np.inf                                      # meets real entries too: inf / NaN in column 3 of those rows only
    A[4, 10:20, 30] = 0.0; Bt[4, 30, 17] = np.nan              # NaN reaches the rows with an entry in column 30, not rows 10..19
    Bt[batch - 1, K - 1, N - 1] = np.inf
    old = L.libxsmm_amd_set_mfma(mfma)
    try:
        sb = L.libxsmm_amd_spmdm_batch_create(M, N, K, batch)
        assert sb
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        assert 0 == L.libxsmm_amd_spmdm_batch_create_slices(sb, b"N", xs.dptr(da))
        c = rng.uniform(-1, 1, batch * M * N).astype(np.float32)
        if beta == 0.0:
            c[:] = np.nan
        ref = c.copy()
        orc.spmdm_exec_batch(orc.FMA, M, N, K, 48, "N", "N", "N", beta, a, b, ref, batch, 4)
        R = ref.reshape(batch, M, N)
        assert np.all(np.isfinite(R[1])) and np.all(np.isfinite(R[2])) and np.all(np.isfinite(R[4, 10:20]))  # the oracle skips absent entries
        assert not np.all(np.isfinite(R[3])) and not np.all(np.isfinite(R[4]))
        dc = torch.from_numpy(c).cuda()
        be = C.c_float(beta)
        assert 0 == L.libxsmm_amd_spmdm_batch_compute(sb, b"N", xs.dptr(db), b"N", C.byref(be), xs.dptr(dc))
        torch.cuda.synchronize()
        got = dc.cpu().numpy()
        finite = np.isfinite(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])  # (inf compares equal to inf of the same sign)
        assert np.array_equal(got[finite].view(np.uint32), ref[finite].view(np.uint32))
        L.libxsmm_amd_spmdm_batch_destroy(sb)
    finally:
        L.libxsmm_amd_set_mfma(old)


def _device_slices(xs, h, slices):
    """the handle's CSR slices copied from HBM: [(rowidx, colidx, values)] indexed kb*mb_count+mb"""
    L = xs.lib()
    out = []
    for blk in range(h.mb * h.kb):
        nrows = min(h.bm, h.m - (blk % h.mb) * h.bm)
        ri = np.zeros(nrows + 1, dtype=np.uint16)
        assert 0 == L.libxsmm_amd_memcpy_d2h(xs.dptr(ri), slices[blk].rowidx, ri.nbytes)
        nnz = int(ri[nrows])
        ci = np.zeros(max(nnz, 1), dtype=np.uint16); va = np.zeros(max(nnz, 1), dtype=np.float32)
        if nnz:
            assert 0 == L.libxsmm_amd_memcpy_d2h(xs.dptr(ci), slices[blk].colidx, 2 * nnz)
            assert 0 == L.libxsmm_amd_memcpy_d2h(xs.dptr(va), slices[blk].values, 4 * nnz)
        out.append((ri, ci[:nnz], va[:nnz]))
    return out


@pytest.mark.parametrize("transa", ["N", "T"])
def test_spmdm_reference_slices_bitexact(xs, orc, torch_gpu, transa):
    """createSparseSlice on one large matrix: every (kb, mb) slice -- row starts, uint16 column indexes, values -- equals the
    oracle's slice of the same block geometry (createSparseSlice tpl :47-141), block by block and through the one-call
    extension; -0 is dropped, NaN is kept; ragged last blocks in both directions."""
    torch = torch_gpu
    M, K, N = 1100, 203, 40
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, M * K).astype(np.float32)
    a[rng.random(M * K) < 0.8] = 0.0
    a[7] = -0.0; a[11] = np.nan
    a2d = a.reshape(K, M) if transa == "T" else a.reshape(M, K)
    if transa == "T":
        a2d[:, 600:664] = 1.25; a2d[:, 5] = 0.0   # 64 dense rows (a full tile of entries), an empty row
    else:
        a2d[600:664, :] = 1.25; a2d[5, :] = 0.0
    L = xs.lib()
    h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
    L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
    assert h.mb > 1 and h.kb > 1 and h.bm * h.bk <= 65535
    oh, osl = orc.spmdm_slices(M, N, K, 48, transa, a, handle=orc.spmdm_geometry(M, N, K, h.bm, h.bn, h.bk))
    da = torch.from_numpy(a).cuda()
    nblk = L.libxsmm_spmdm_get_num_createSparseSlice_blocks(C.byref(h))
    assert nblk == h.mb and h.mb * h.kb == len(osl)  # a create block is a row block of A: the slices of all its column blocks
    for how in ("blocks", "all", "host"):
        if how == "blocks":
            for blk in reversed(range(nblk)):
                L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), transa.encode(), xs.dptr(da), slices, blk, 0, 1)
            assert xs.last_kernel() == "spmdm_create_slice_wg"
        elif how == "all":
            assert 0 == L.libxsmm_amd_spmdm_createSparseSlice_all(C.byref(h), transa.encode(), xs.dptr(da), slices)
        else:  # pageable host matrix, block by block
            for blk in range(nblk):
                L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), transa.encode(), xs.dptr(a), slices, blk, 0, 1)
        torch.cuda.synchronize()
        got = _device_slices(xs, h, slices)
        for blk, ((ri, ci, va), (oi, oc, ov)) in enumerate(zip(got, osl)):
            assert np.array_equal(ri, oi), (how, blk)
            assert np.array_equal(ci, oc), (how, blk)
            assert np.array_equal(va.view(np.uint32), ov.view(np.uint32)), (how, blk)
        # clear the slices between the rounds: the next round must rewrite everything it claims
        nsl = h.mb * h.kb
        assert 0 == L.libxsmm_amd_memcpy_h2d(slices[0].values, xs.dptr(np.zeros(nsl * h.bm * h.bk, dtype=np.float32)), 4 * nsl * h.bm * h.bk)
    L.libxsmm_spmdm_destroy(C.byref(h))


def test_spmdm_block_contract(xs, orc, torch_gpu):
    """The per-block interface keeps the reference's contract (compute tpl :38-39, 509-558; createSparseSlice tpl :47-141): a
    call touches the C tile resp. the slice of its block id and nothing else, with the operands of THAT call.
    (i) every compute block with its own B and beta = 1; (ii) a subset of the blocks: the other tiles keep their NaN
    canaries bit for bit; (iii) B changed in place between two block calls; (iv) a subset of the create blocks after A
    changed: only the slices of those blocks change."""
    torch = torch_gpu
    M, N, K = 1100, 2300, 150   # mb = 3, nb = 2, kb = 3 with the engine's geometry
    a, b, c = spmdm_inputs(M, N, K, 0.85, 3, orc)
    L = xs.lib()
    h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
    L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
    ncreate, ncomp = L.libxsmm_spmdm_get_num_createSparseSlice_blocks(C.byref(h)), L.libxsmm_spmdm_get_num_compute_blocks(C.byref(h))
    assert ncreate >= 3 and ncomp >= 4 and h.nb >= 2 and h.mb >= 2
    geom = lambda: orc.spmdm_geometry(M, N, K, h.bm, h.bn, h.bk)
    da, db, dc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda()
    alpha = C.c_float(1.0)

    def create(blocks, src=None):
        for blk in blocks:
            L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), b"N", xs.dptr(da if src is None else src), slices, blk, 0, 1)

    def compute(blk, beta, bdev):
        be = C.c_float(beta)
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(bdev), b"N", C.byref(be), xs.dptr(dc), blk, 0, 1)

    create(range(ncreate))
    # (i) block i multiplies with its own B_i, beta = 1: one launch per call, each adds its own product to its own tile
    bs = [(b * (1.0 + 0.25 * i) + 0.125 * i).astype(np.float32) for i in range(ncomp)]
    dbs = [torch.from_numpy(x).cuda() for x in bs]
    launches = L.libxsmm_amd_launch_count()
    for i in range(ncomp):
        compute(i, 1.0, dbs[i])
    assert L.libxsmm_amd_launch_count() == launches + ncomp
    assert xs.last_kernel().startswith("spmdm_compute_tiled")
    torch.cuda.synchronize()
    ref = c.copy()
    for i in range(ncomp):
        orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 1.0, a, bs[i], ref, [i])
    assert np.array_equal(dc.cpu().numpy(), ref)
    # (ii) a subset of the blocks, beta = 0 then beta = 1: every other tile keeps its canary
    for beta in (0.0, 1.0):
        canary = np.full(M * N, np.nan, dtype=np.float32)
        canary.view(np.uint32)[:] = 0x7FC00000 + (np.arange(M * N, dtype=np.uint32) & 0xFFFF)  # NaNs with distinct payloads
        subset = [1, ncomp - 1]
        start = canary.copy()
        if beta != 0.0:  # the tiles that are computed hold numbers
            tmp = c.copy(); mask = np.zeros(M * N, dtype=bool)
            for blk in subset:
                mb, nb = blk // h.nb, blk % h.nb
                m2 = np.zeros((M, N), dtype=bool); m2[mb * h.bm:(mb + 1) * h.bm, nb * h.bn:(nb + 1) * h.bn] = True
                mask |= m2.reshape(-1)
            start[mask] = tmp[mask]
        dc.copy_(torch.from_numpy(start))
        for blk in subset:
            compute(blk, beta, db)
        torch.cuda.synchronize()
        ref = start.copy()
        orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", beta, a, b, ref, subset)
        out = dc.cpu().numpy()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), beta
        assert np.isnan(out).sum() == np.isnan(ref).sum() > 0
    # (iii) B modified in place between two block calls: the second call sees the new B
    dc.copy_(torch.from_numpy(c)); dbm = torch.from_numpy(b).cuda()
    compute(0, 0.0, dbm)
    b2 = (b * 0.5 + 0.25).astype(np.float32); dbm.copy_(torch.from_numpy(b2))
    compute(1, 0.0, dbm)
    torch.cuda.synchronize()
    ref = c.copy()
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.0, a, b, ref, [0])
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.0, a, b2, ref, [1])
    assert np.array_equal(dc.cpu().numpy(), ref)
    # (iv) A changes, only two create blocks are called: exactly those slices change
    before = _device_slices(xs, h, slices)
    a2 = a.copy(); a2[::7] = 0.0; a2[3::11] *= 2.0; a2[a2 == 0.0] = 0.0
    da2 = torch.from_numpy(a2).cuda()
    redo = [0, ncreate - 1]
    create(redo, da2)
    torch.cuda.synchronize()
    after = _device_slices(xs, h, slices)
    _, o2 = orc.spmdm_slices(M, N, K, 48, "N", a2, handle=geom())
    for blk in range(h.mb * h.kb):  # slice kb * mb_count + mb belongs to create block mb
        want = o2[blk] if (blk % h.mb) in redo else before[blk]
        for x, y in zip(after[blk], want):
            assert np.array_equal(x.view(np.uint16 if x.dtype == np.uint16 else np.uint32), y.view(np.uint16 if y.dtype == np.uint16 else np.uint32)), blk
    L.libxsmm_spmdm_destroy(C.byref(h))


def test_spmdm_block_calls_inside_a_bracket(xs, orc, torch_gpu):
    """Block calls on device operands inside libxsmm_amd_defer_begin/end are recorded and launched as rectangles of blocks
    (samples/spmdm/spmdm.c:99-109 is the loop; xsmm_sparse.cpp:record_block): same bits as a launch per call, the block contract
    (compute tpl :38-39, createSparseSlice tpl :47-141) unchanged. (i) a full sweep = one create launch + one compute launch,
    equal to the oracle; (ii) a subset of the blocks: every other tile keeps its canary; (iii) libxsmm_amd_flush, then B rewritten
    in place on the stream, then the next block: the second call sees the new B; (iv) other operands per block, and a block
    recorded twice with beta = 1: nothing merges that must not; (v) the bfloat16 twins; (vi) host operands inside a bracket are
    served at once."""
    torch = torch_gpu
    M, N, K = 1100, 2300, 150   # mb = 3, nb = 2, kb = 3 with the engine's geometry
    a, b, c = spmdm_inputs(M, N, K, 0.85, 3, orc)
    L = xs.lib()
    h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
    L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
    ncreate, ncomp = L.libxsmm_spmdm_get_num_createSparseSlice_blocks(C.byref(h)), L.libxsmm_spmdm_get_num_compute_blocks(C.byref(h))
    assert ncreate == 3 and ncomp == 6 and h.nb == 2
    geom = lambda: orc.spmdm_geometry(M, N, K, h.bm, h.bn, h.bk)
    da, db, dc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda()
    alpha = C.c_float(1.0)

    def create(blocks, src):
        for blk in blocks:
            L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), b"N", xs.dptr(src), slices, blk, 0, 1)

    def compute(blk, beta, bsrc, cdst):
        be = C.c_float(beta)
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(bsrc), b"N", C.byref(be), xs.dptr(cdst), blk, 0, 1)

    # (i) the sample's two loops inside one bracket
    launches = L.libxsmm_amd_launch_count()
    L.libxsmm_amd_defer_begin()
    create(range(ncreate), da)
    for blk in range(ncomp):
        compute(blk, 0.5, db, dc)
    assert L.libxsmm_amd_launch_count() <= launches + 1  # (the creates are launched when the first compute call arrives)
    L.libxsmm_amd_defer_end()
    assert L.libxsmm_amd_launch_count() == launches + 2
    torch.cuda.synchronize()
    ref = c.copy()
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.5, a, b, ref, list(range(ncomp)))
    assert np.array_equal(dc.cpu().numpy(), ref)
    # descending block ids: nothing merges, same result
    dc.copy_(torch.from_numpy(c))
    L.libxsmm_amd_defer_begin()
    for blk in reversed(range(ncomp)):
        compute(blk, 0.5, db, dc)
    L.libxsmm_amd_defer_end()
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)
    # (ii) a subset: the other tiles keep their canaries bit for bit
    canary = np.full(M * N, np.nan, dtype=np.float32)
    canary.view(np.uint32)[:] = 0x7FC00000 + (np.arange(M * N, dtype=np.uint32) & 0xFFFF)
    subset = [1, 2, 3, ncomp - 1]
    dc.copy_(torch.from_numpy(canary))
    launches = L.libxsmm_amd_launch_count()
    L.libxsmm_amd_defer_begin()
    for blk in subset:
        compute(blk, 0.0, db, dc)
    L.libxsmm_amd_defer_end()
    assert L.libxsmm_amd_launch_count() == launches + 3   # (0,1) | (1,0)+(1,1) | (2,1)
    torch.cuda.synchronize()
    ref = canary.copy()
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.0, a, b, ref, subset)
    out = dc.cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert np.isnan(out).sum() == np.isnan(ref).sum() > 0
    # (iii) the contract of the bracket: flush before own work on the stream that touches the operands
    dc.copy_(torch.from_numpy(c)); dbm = torch.from_numpy(b).cuda()
    b2 = (b * 0.5 + 0.25).astype(np.float32); db2 = torch.from_numpy(b2).cuda()
    L.libxsmm_amd_defer_begin()
    compute(0, 0.0, dbm, dc)
    L.libxsmm_amd_flush()
    dbm.copy_(db2)
    compute(1, 0.0, dbm, dc)
    L.libxsmm_amd_defer_end()
    torch.cuda.synchronize()
    ref = c.copy()
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.0, a, b, ref, [0])
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.0, a, b2, ref, [1])
    assert np.array_equal(dc.cpu().numpy(), ref)
    # (iv) other operands per call; one block twice (beta = 1: it must run twice)
    dc.copy_(torch.from_numpy(c))
    bs = [(b * (1.0 + 0.25 * i) + 0.125 * i).astype(np.float32) for i in range(ncomp)]
    dbs = [torch.from_numpy(x).cuda() for x in bs]
    L.libxsmm_amd_defer_begin()
    for i in range(ncomp):
        compute(i, 1.0, dbs[i], dc)
    compute(2, 1.0, dbs[ncomp - 1], dc)
    compute(2, 1.0, dbs[ncomp - 1], dc)
    L.libxsmm_amd_defer_end()
    torch.cuda.synchronize()
    ref = c.copy()
    for i in range(ncomp):
        orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 1.0, a, bs[i], ref, [i])
    for _ in range(2):
        orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 1.0, a, bs[ncomp - 1], ref, [2])
    assert np.array_equal(dc.cpu().numpy(), ref)
    # create blocks of a changed A, two of three inside a bracket: exactly those slices change
    before = _device_slices(xs, h, slices)
    a2 = a.copy(); a2[::7] = 0.0; a2[3::11] *= 2.0; a2[a2 == 0.0] = 0.0
    da2 = torch.from_numpy(a2).cuda()
    redo = [0, ncreate - 1]
    L.libxsmm_amd_defer_begin()
    create(redo, da2)
    L.libxsmm_amd_defer_end()
    torch.cuda.synchronize()
    after = _device_slices(xs, h, slices)
    _, o2 = orc.spmdm_slices(M, N, K, 48, "N", a2, handle=geom())
    for blk in range(h.mb * h.kb):
        want = o2[blk] if (blk % h.mb) in redo else before[blk]
        for x, y in zip(after[blk], want):
            assert np.array_equal(x.view(np.uint16 if x.dtype == np.uint16 else np.uint32), y.view(np.uint16 if y.dtype == np.uint16 else np.uint32)), blk
    # (vi) host operands inside the bracket: done on return
    hc = c.copy()
    L.libxsmm_amd_defer_begin()
    create(range(ncreate), a)
    for blk in range(ncomp):
        compute(blk, 0.5, b, hc)
        if blk == 0:
            snapshot = hc.copy()
    L.libxsmm_amd_defer_end()
    ref = c.copy()
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.5, a, b, ref, [0])
    assert np.array_equal(snapshot, ref)
    orc.spmdm_compute_blocks(orc.FMA, geom(), "N", "N", "N", 0.5, a, b, ref, list(range(1, ncomp)))
    assert np.array_equal(hc, ref)
    L.libxsmm_spmdm_destroy(C.byref(h))
    # (v) the bfloat16 twins, transposed and not: bracket == launch per call, bit for bit
    for ta, tb, tc in (("N", "N", "N"), ("T", "T", "T")):
        a16 = (a.view(np.uint32) >> 16).astype(np.uint16); b16 = (b.view(np.uint32) >> 16).astype(np.uint16)
        outs = []
        for bracket in (False, True):
            h2 = xs.SpmdmHandle(); sl2 = C.POINTER(xs.CSRSlice)()
            L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h2), C.byref(sl2))
            xa, xb, xc = torch.from_numpy(a16.view(np.int16)).cuda(), torch.from_numpy(b16.view(np.int16)).cuda(), torch.from_numpy(c).cuda()
            al, be = C.c_ushort(0x3F80), C.c_ushort(1)
            launches = L.libxsmm_amd_launch_count()
            if bracket:
                L.libxsmm_amd_defer_begin()
            for blk in range(ncreate):
                L.libxsmm_spmdm_createSparseSlice_bfloat16_thread(C.byref(h2), ta.encode(), xs.dptr(xa), sl2, blk, 0, 1)
            for blk in range(ncomp):
                L.libxsmm_spmdm_compute_bfloat16_thread(C.byref(h2), ta.encode(), tb.encode(), C.byref(al), sl2, xs.dptr(xb), tc.encode(),
                                                        C.byref(be), xs.dptr(xc), blk, 0, 1)
            if bracket:
                L.libxsmm_amd_defer_end()
                assert L.libxsmm_amd_launch_count() == launches + 4   # widen + create, widen + compute
            torch.cuda.synchronize()
            outs.append(xc.cpu().numpy())
            L.libxsmm_spmdm_destroy(C.byref(h2))
        ref = c.copy()
        orc.spmdm_exec_bf16(orc.FMA, M, N, K, 48, ta, tb, tc, 1, a16, b16, ref)
        assert np.array_equal(outs[0], ref) and np.array_equal(outs[1], ref), (ta, tb, tc)


def test_spmdm_blocks_and_per_call_kernels_inside_one_bracket(xs, orc, torch_gpu):
    """Recorded spmdm block calls and bursts of per-call dense kernels keep the order of the calls among themselves: a dense call
    that reads the C of a recorded spmdm block, and a spmdm block whose B a dense call of an open burst has written -- same bits as
    the same sequence with a launch per call (that path is checked against the oracle in the tests above)."""
    torch = torch_gpu
    L = xs.lib()
    M, N, K = 64, 32, 64
    a, b, c = spmdm_inputs(M, N, K, 0.6, 5, orc)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, 1024).astype(np.float32); y = rng.uniform(-1, 1, 1024).astype(np.float32); z = rng.uniform(-1, 1, 1024).astype(np.float32)
    fn = L.libxsmm_smmdispatch(32, 32, 32, None, None, None, None, None, None, None)
    assert fn
    alpha = C.c_float(1.0)
    outs = []
    for bracket in (False, True):
        h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
        L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
        assert L.libxsmm_spmdm_get_num_compute_blocks(C.byref(h)) == 1
        da, db, dc, dc2 = (torch.from_numpy(v.copy()).cuda() for v in (a, b, c, c))
        dx, dy, dz1, dz2 = (torch.from_numpy(v.copy()).cuda() for v in (x, y, z, z))
        be0 = C.c_float(0.0)
        if bracket:
            L.libxsmm_amd_defer_begin()
        xs.call_kernel(fn, dx, dy, dz1)                                   # opens a burst (inside the bracket)
        L.libxsmm_spmdm_createSparseSlice_fp32_thread(C.byref(h), b"N", xs.dptr(da), slices, 0, 0, 1)
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(db), b"N", C.byref(be0), xs.dptr(dc), 0, 0, 1)
        xs.call_kernel(fn, dc, dy, dz2)                                   # A = the first 1024 numbers of the spmdm result
        xs.call_kernel(fn, dx, dz2, db)                                   # ... and the first 1024 numbers of the spmdm B updated by a dense call
        L.libxsmm_spmdm_compute_fp32_thread(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(db), b"N", C.byref(be0), xs.dptr(dc2), 0, 0, 1)
        if bracket:
            L.libxsmm_amd_defer_end()
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (dz1, dc, dz2, db, dc2)])
        L.libxsmm_spmdm_destroy(C.byref(h))
    for u, v in zip(*outs):
        assert np.array_equal(u.view(np.uint32), v.view(np.uint32))
    assert not np.array_equal(outs[0][4], outs[0][1])  # (the second product really saw another B)


@pytest.mark.parametrize("case", [
    # M, N, K, keep-threshold, (ta, tb, tc), beta
    (700, 515, 330, 0.85, ("N", "N", "N"), 1.0),     # N % 4 != 0: element-wide global accesses
    (700, 516, 328, 0.85, ("N", "N", "N"), 0.5),     # 16-byte accesses, ragged tiles in every direction
    (130, 2304, 200, 0.0, ("N", "N", "N"), 0.0),     # fully dense A: 4096 entries per tile and column block (several windows)
    (520, 260, 132, 0.5, ("T", "T", "T"), 1.0),
    (520, 260, 132, 0.5, ("N", "T", "N"), 0.0),
    (520, 258, 131, 0.5, ("T", "N", "T"), 0.5),
    (64, 48, 64, 0.5, ("N", "N", "N"), 0.0),         # the config-4 problem as a single call
    (1, 1, 1, 0.0, ("N", "N", "N"), 1.0),
])
def test_spmdm_whole_problem_calls(xs, orc, torch_gpu, case):
    """libxsmm_amd_spmdm_createSparseSlice_all / _compute_all: the caller's two loops over block ids as one launch each;
    equal to the oracle bit for bit. Device operands and pageable host operands."""
    torch = torch_gpu
    M, N, K, keep, (ta, tb, tc), beta = case
    a, b, c = spmdm_inputs(M, N, K, keep, 4, orc) if M * K < 200000 else (None, None, None)
    if a is None:
        rng = np.random.default_rng(9)
        a = rng.uniform(0, 1, M * K).astype(np.float32); a[rng.random(M * K) < keep] = 0.0
        b = rng.uniform(0, 1, K * N).astype(np.float32); c = rng.uniform(0, 1, M * N).astype(np.float32)
    if beta == 0.0:
        c[:] = np.nan
    ref = c.copy()
    orc.spmdm_exec(orc.FMA, M, N, K, 48, ta, tb, tc, beta, a, b, ref)
    L = xs.lib()
    alpha, be = C.c_float(1.0), C.c_float(beta)
    for on_device in (True, False):
        h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
        L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
        xa, xb, xc = (torch.from_numpy(x).cuda() for x in (a, b, c)) if on_device else (a, b, c.copy())
        launches = L.libxsmm_amd_launch_count()
        assert 0 == L.libxsmm_amd_spmdm_createSparseSlice_all(C.byref(h), ta.encode(), xs.dptr(xa), slices)
        assert 0 == L.libxsmm_amd_spmdm_compute_all(C.byref(h), ta.encode(), tb.encode(), C.byref(alpha), slices, xs.dptr(xb), tc.encode(), C.byref(be), xs.dptr(xc))
        assert L.libxsmm_amd_launch_count() == launches + 2
        assert xs.last_kernel().startswith("spmdm_compute_tiled")
        torch.cuda.synchronize()
        out = xc.cpu().numpy() if on_device else xc
        L.libxsmm_spmdm_destroy(C.byref(h))
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), on_device


def test_spmdm_whole_problem_bfloat16(xs, orc, torch_gpu):
    """bfloat16 twins of the one-call extension (same widening and `*beta` reading as the *_thread functions)."""
    torch = torch_gpu
    M, N, K = 600, 136, 200
    a32, b32, c = spmdm_inputs(M, N, K, 0.7, 1, orc)
    a = (a32.view(np.uint32) >> 16).astype(np.uint16); b = (b32.view(np.uint32) >> 16).astype(np.uint16)
    ref = c.copy()
    orc.spmdm_exec_bf16(orc.FMA, M, N, K, 48, "N", "N", "N", 1, a, b, ref)
    L = xs.lib()
    h = xs.SpmdmHandle(); slices = C.POINTER(xs.CSRSlice)()
    L.libxsmm_spmdm_init(M, N, K, 1, C.byref(h), C.byref(slices))
    xa, xb, xc = torch.from_numpy(a.view(np.int16)).cuda(), torch.from_numpy(b.view(np.int16)).cuda(), torch.from_numpy(c).cuda()
    alpha, be = C.c_ushort(0x3F80), C.c_ushort(1)
    assert 0 == L.libxsmm_amd_spmdm_createSparseSlice_bfloat16_all(C.byref(h), b"N", xs.dptr(xa), slices)
    assert 0 == L.libxsmm_amd_spmdm_compute_bfloat16_all(C.byref(h), b"N", b"N", C.byref(alpha), slices, xs.dptr(xb), b"N", C.byref(be), xs.dptr(xc))
    torch.cuda.synchronize()
    L.libxsmm_spmdm_destroy(C.byref(h))
    assert np.array_equal(xc.cpu().numpy(), ref)
