"""Pins the CPU oracle (oracle/xsmm_oracle.c) -- runs without a GPU.

What the reference itself holds for this path and what is checked against it:
  * MatrixMarket fixtures: samples/generator/{left_sparse_test_csr,left_sparse_test_csc,right_sparse_test_csc}.mtx and
    operator pairs from samples/pyfr/mats (copied as data to tests/golden/mtx): reader known-answers (an independent
    Python parse of the same text) and the sparse-file == dense-file identity.
  * the gold loops of its self-checking samples (samples/generator/validation.c:203-209 naive GEMM,
    samples/spmdm/spmdm.c:274-297, samples/pyfr/pyfr_driver_asp_reg.c:275-293): restated here with numpy in float64
    as an implementation independent of the oracle's C code.
  * input generators: LIBXSMM_MATINIT closed form (include/libxsmm_frontend.h:414-431), libxsmm_rng_f64 == drand48
    (src/libxsmm_rng.c:131,256) checked against the C library's own drand48.
  * geometry observed from the reference during the survey: libxsmm_spmdm_init(64,48,64) gives
    bm=70 with one thread and bm=32/mb=2 with eight (bn=48, bk=128).
No outputs of a reference binary exist (it cannot be built under this round's rules), see DESIGN.md.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

GEN = os.path.join(GOLDEN, "mtx", "generator")


def parse_coo(path):
    rows = cols = nnz = None
    ent = []
    for line in open(path):
        if line.startswith("%"):
            continue
        t = line.split()
        if rows is None:
            rows, cols, nnz = int(t[0]), int(t[1]), int(t[2])
        else:
            ent.append((int(t[0]) - 1, int(t[1]) - 1, float(t[2])))
    assert len(ent) == nnz
    return rows, cols, ent


def dense_from(ent, rows, cols):
    A = np.zeros((rows, cols))
    for r, c, v in ent:
        A[r, c] = v
    return A


def test_csr_reader_known_answers(orc):
    path = os.path.join(GEN, "left_sparse_test_csr.mtx")
    rows, cols, ent = parse_coo(path)
    assert (rows, cols, len(ent)) == (84, 84, 686)  # header of the fixture
    rowptr, colidx, vals, r, c, z = orc.read_csr(path)
    assert (r, c, z) == (84, 84, 686)
    assert [e[0] for e in ent] == sorted(e[0] for e in ent)  # grouped by row, as the reader assumes
    counts = np.bincount([e[0] for e in ent], minlength=rows)
    assert np.array_equal(rowptr, np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32))
    assert np.array_equal(colidx, np.array([e[1] for e in ent], dtype=np.uint32))
    assert np.array_equal(vals, np.array([e[2] for e in ent]))
    assert (colidx[0], vals[0], colidx[1], vals[1]) == (1, 2.0, 5, 1.0)  # first lines of the file: "1 2 2", "1 6 1"


@pytest.mark.parametrize("name,shape", [("left_sparse_test_csc.mtx", (84, 84, 686)), ("right_sparse_test_csc.mtx", (9, 9, 24))])
def test_csc_reader_known_answers(orc, name, shape):
    path = os.path.join(GEN, name)
    rows, cols, ent = parse_coo(path)
    assert (rows, cols, len(ent)) == shape
    colptr, rowidx, vals, r, c, z = orc.read_csc(path)
    assert (r, c, z) == shape
    counts = np.bincount([e[1] for e in ent], minlength=cols)
    assert np.array_equal(colptr, np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32))  # includes empty columns
    assert np.array_equal(rowidx, np.array([e[0] for e in ent], dtype=np.uint32))
    assert np.array_equal(vals, np.array([e[2] for e in ent]))


def test_reader_rejects_malformed(orc, tmp_path):
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n")  # nnz mismatch (LIBXSMM_ERR_CSR_LEN)
    with pytest.raises(IOError):
        orc.read_csr(str(bad))
    with pytest.raises(IOError):
        orc.read_csr(str(tmp_path / "missing.mtx"))
    empty_rows = tmp_path / "gap.mtx"
    empty_rows.write_text("% comment\n4 3 2\n1 2 5.0\n4 1 -1.5\n")
    rowptr, colidx, vals, r, c, z = orc.read_csr(str(empty_rows))
    assert list(rowptr) == [0, 1, 1, 1, 2] and list(colidx) == [1, 0]  # empty rows back-filled (:158-163)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_generator_fixtures_sparse_kernels_vs_dense(orc, dtype, beta):
    """samples/generator/test_xGEMM.sh shapes: left sparse M=84,N=9,K=84 (row-major CSR; col-major CSC), right sparse
    M=20,N=9,K=9. Gold = dense product of the densified operator (validation.c:203-209)."""
    rng = np.random.default_rng(0)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    flags = orc.FLAG_BETA_0 if beta == 0.0 else 0
    # CSR A-sparse, row-major B (K x ldb) and C (M x ldc)
    rowptr, colidx, vals, M, K, _ = orc.read_csr(os.path.join(GEN, "left_sparse_test_csr.mtx"))
    A = dense_from(parse_coo(os.path.join(GEN, "left_sparse_test_csr.mtx"))[2], M, K)
    N, ldb, ldc = 9, 9, 9
    B = rng.uniform(-1, 1, (K, ldb)).astype(dtype); Cin = rng.uniform(-1, 1, (M, ldc)).astype(dtype)
    for arith in (orc.MULADD, orc.FMA):
        Cm = Cin.copy()
        orc.csr_asparse(arith, flags, M, N, K, ldb, ldc, rowptr, colidx, vals.astype(dtype), B, Cm)
        expect = A @ B.astype(np.float64) + beta * Cin
        assert np.max(np.abs(Cm - expect)) <= tol * max(1, np.max(np.abs(expect)))
    # CSC A-sparse, col-major B (ldb x N) and C (ldc x N)
    colptr, rowidx, vals, M, K, _ = orc.read_csc(os.path.join(GEN, "left_sparse_test_csc.mtx"))
    A = dense_from(parse_coo(os.path.join(GEN, "left_sparse_test_csc.mtx"))[2], M, K)
    B = rng.uniform(-1, 1, (N, K)).astype(dtype); Cin = rng.uniform(-1, 1, (N, M)).astype(dtype)  # stored column-major
    Cm = Cin.copy()
    orc.csc_asparse(orc.FMA, flags, M, N, K, K, M, colptr, rowidx, vals.astype(dtype), B, Cm)
    expect = (A @ B.T.astype(np.float64) + beta * Cin.T).T
    assert np.max(np.abs(Cm - expect)) <= tol * max(1, np.max(np.abs(expect)))
    # CSC B-sparse: C(20x9) = A(20x9) * B_sparse(9x9), col-major, lda = ldc = 20
    colptr, rowidx, vals, K, N, _ = orc.read_csc(os.path.join(GEN, "right_sparse_test_csc.mtx"))
    Bd = dense_from(parse_coo(os.path.join(GEN, "right_sparse_test_csc.mtx"))[2], K, N)
    M = 20
    Am = rng.uniform(-1, 1, (K, M)).astype(dtype); Cin = rng.uniform(-1, 1, (N, M)).astype(dtype)
    Cm = Cin.copy()
    orc.csc_bsparse(orc.FMA, flags, M, N, K, M, M, colptr, rowidx, Am, vals.astype(dtype), Cm)
    expect = (Am.T.astype(np.float64) @ Bd + beta * Cin.T).T
    assert np.max(np.abs(Cm - expect)) <= tol * max(1, np.max(np.abs(expect)))


def test_csr_asparse_quirks(orc):
    """beta == 0 zeroes ldc (not n) entries per row (generator_spgemm_csr_asparse.c:79); entries with col >= k are dropped (:136)."""
    rowptr = np.array([0, 2, 2, 3], dtype=np.uint32); colidx = np.array([0, 5, 1], dtype=np.uint32); vals = np.array([2.0, 100.0, 3.0])
    M, N, K, ldb, ldc = 3, 2, 4, 3, 4
    B = np.arange(1, 1 + 6 * ldb, dtype=np.float64).reshape(6, ldb)
    Cm = np.full((M, ldc), 7.0)
    orc.csr_asparse(orc.FMA, orc.FLAG_BETA_0, M, N, K, ldb, ldc, rowptr, colidx, vals, B, Cm)
    assert np.array_equal(Cm[0], [2 * B[0, 0], 2 * B[0, 1], 0, 0])  # col 5 >= K ignored; padding zeroed too
    assert np.array_equal(Cm[1], [0, 0, 0, 0]) and np.array_equal(Cm[2], [3 * B[1, 0], 3 * B[1, 1], 0, 0])


def test_csr_reg_semantics(orc):
    """generator_spgemm_csr_asparse_reg.c: N must equal the vector length (:187), <= 31 unique values (:146), rows without
    nnz untouched even for beta == 0 (:229,287)."""
    rng = np.random.default_rng(1)
    M, K, n = 6, 5, 8
    rowptr = np.array([0, 2, 2, 3, 5, 5, 6], dtype=np.uint32); colidx = np.array([0, 3, 1, 2, 4, 0], dtype=np.uint32)
    vals = np.array([1.5, -2.0, 1.5, 0.5, -2.0, 4.0])
    B = rng.uniform(-1, 1, (K, n)); Cin = rng.uniform(-1, 1, (M, n))
    Cm = Cin.copy()
    assert 0 == orc.csr_reg(orc.FLAG_BETA_0, M, n, K, n, n, rowptr, colidx, vals, B, Cm)
    A = np.zeros((M, K))
    for r in range(M):
        for q in range(rowptr[r], rowptr[r + 1]):
            A[r, colidx[q]] = vals[q]
    for r in range(M):
        if rowptr[r] == rowptr[r + 1]:
            assert np.array_equal(Cm[r], Cin[r])
        else:
            assert np.max(np.abs(Cm[r] - A[r] @ B)) <= 1e-14
    assert -1 == orc.csr_reg(0, M, 7, K, n, n, rowptr, colidx, vals, B, Cm.copy())
    assert 4 == orc.lib().xo_csr_reg_unique(vals.ctypes.data_as(C.c_void_p), 6)
    many = np.arange(40, dtype=np.float64)
    assert 40 == orc.lib().xo_csr_reg_unique(many.ctypes.data_as(C.c_void_p), 40)


def test_pyfr_fixture_pairs_and_fsspmdm_paths(orc):
    files = sorted(glob.glob(os.path.join(GOLDEN, "mtx", "pyfr", "*-sp.mtx")))
    assert len(files) >= 8
    rng = np.random.default_rng(2)
    for sp in files:
        rowptr, colidx, vals, rows, cols, nnz = orc.read_csr(sp)
        dense = orc.read_dense_mtx(sp.replace("-sp.mtx", "-de.mtx"))
        A = np.zeros((rows, cols))
        for r in range(rows):
            for q in range(rowptr[r], rowptr[r + 1]):
                A[r, colidx[q]] = vals[q]
        assert np.max(np.abs(A - dense)) <= 1e-10 * max(1.0, np.max(np.abs(dense)))
        N = 32
        B = rng.uniform(-1, 1, (cols, N)); Cin = rng.uniform(-1, 1, (rows, N))
        for beta in (1.0, 0.0):
            outs = []
            for avx512 in (True, False):  # csr_reg path where <= 31 unique values, else dense fallback; LIBXSMM_TARGET=hsw always dense
                h = orc.Fsspmdm(np.ascontiguousarray(A), rows, N, cols, cols, N, N, 1.0, beta, have_avx512=avx512)
                Cm = Cin.copy(); h.execute(B, Cm); h.close(); outs.append(Cm)
            expect = dense @ B + beta * Cin
            for Cm in outs:
                nonempty = [r for r in range(rows) if rowptr[r] != rowptr[r + 1]]
                assert np.max(np.abs(Cm[nonempty] - expect[nonempty])) <= 1e-10 * max(1.0, np.max(np.abs(expect)))
    with pytest.raises(ValueError):  # N % 16 != 0 violates the assert of libxsmm_fsspmdm.c:65
        orc.Fsspmdm(np.eye(4), 4, 24, 4, 4, 24, 24, 1.0, 1.0, have_avx512=True)


GEMM_TABLE = [  # (m, n, k, lda, ldb, ldc) rows of tests/gemm.c:75-82 (the alpha = 1 cases relevant to the SMM domain)
    (23, 23, 23, 23, 23, 23), (32, 32, 32, 32, 32, 32), (1, 1, 1, 1, 1, 1), (2, 2, 2, 2, 2, 2), (3, 3, 3, 3, 3, 3),
    (64, 8, 24, 64, 24, 64), (8, 64, 24, 8, 24, 8), (43, 9, 27, 48, 32, 48), (13, 70, 5, 16, 8, 16), (35, 16, 35, 35, 35, 40)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", GEMM_TABLE)
def test_dense_oracle_vs_numpy(orc, dtype, shape):
    m, n, k, lda, ldb, ldc = shape
    a = orc.matinit(42, m, k, lda, 1.0, dtype); b = orc.matinit(24, k, n, ldb, 1.0, dtype)  # tests/gemm.c:137-138 seeds
    A = a.reshape(k, lda)[:, :m].T.astype(np.float64); B = b.reshape(n, ldb)[:, :k].T.astype(np.float64)
    tol = 1e-13 if dtype == np.float64 else 1e-5
    for beta in (1.0, 0.0):
        results = []
        for arith in (orc.MULADD, orc.FMA):
            c = orc.matinit(22, m, n, ldc, 1.0, dtype)
            if beta == 0.0:
                c.reshape(n, ldc)[:, :m] = np.nan  # must not be read (tests/gemm.c:159-167)
            cin = c.copy()
            orc.smm(arith, orc.FLAG_BETA_0 if beta == 0.0 else 0, m, n, k, lda, ldb, ldc, a, b, c)
            got = c.reshape(n, ldc)[:, :m].T
            expect = A @ B + (0 if beta == 0.0 else cin.reshape(n, ldc)[:, :m].T)
            assert np.max(np.abs(got - expect)) <= tol * max(1.0, np.max(np.abs(expect)))
            assert np.array_equal(c.reshape(n, ldc)[:, m:], cin.reshape(n, ldc)[:, m:])  # padding rows untouched
            results.append(c)
        # TRANS_B: B given as n x k (ld n)
        bt = np.ascontiguousarray(B.T).reshape(-1).astype(dtype)  # column-major n x k == row-major (k, n)
        c1 = orc.matinit(22, m, n, ldc, 1.0, dtype); c2 = c1.copy()
        bt_cm = np.ascontiguousarray(B).reshape(-1).astype(dtype)  # B[k][n] at k*n_ld + n
        orc.smm(orc.FMA, orc.FLAG_TRANS_B, m, n, k, lda, n, ldc, a, bt_cm, c1)
        orc.smm(orc.FMA, 0, m, n, k, lda, ldb, ldc, a, b, c2)
        assert np.array_equal(c1, c2)
        del bt


def test_matinit_closed_form(orc):
    for dtype in (np.float64, np.float32):
        nrows, ncols, ld, seed, scale = 23, 7, 26, 42, 1.0 / 1024
        out = orc.matinit(seed, nrows, ncols, ld, scale, dtype).reshape(ncols, ld)
        idx = (np.arange(ncols)[:, None] * ld + np.arange(ld)[None, :]).astype(np.float64)
        expect = ((scale * seed + scale) / (1.0 + idx)).astype(dtype)
        assert np.array_equal(out[:, :nrows], expect[:, :nrows])
        assert np.all(out[:, nrows:] == dtype(seed))


def test_rng_is_posix_drand48(orc):
    libc = C.CDLL(None)
    libc.drand48.restype = C.c_double
    for seed in (1, 0, 12345):
        libc.srand48(C.c_long(seed)); orc.rng_seed(seed)
        assert [orc.rng_f64() for _ in range(64)] == [libc.drand48() for _ in range(64)]


def test_batch_addressing_modes_agree(orc):
    """libxsmm_mmbatch_kernel's modes (src/libxsmm_gemm.c:1333-1364 index arrays with base 0/1, :1426-1461 pointer arrays,
    NULL stride = shared operand) describe the same batch => identical C."""
    m, n, k, batch = 23, 23, 23, 64
    rng = np.random.default_rng(4)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c0 = rng.uniform(-1, 1, batch * m * n)
    ref = c0.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, m * k, k * n, m * n, batch)
    for base in (0, 1):
        sa = (np.arange(batch) * m * k + base).astype(np.int32); sb = (np.arange(batch) * k * n + base).astype(np.int32); sc = (np.arange(batch) * m * n + base).astype(np.int32)
        c = c0.copy(); assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, c, base, sa, sb, sc, batch)
        assert np.array_equal(c, ref)
        wa, wb, wc = (np.zeros(2 * batch, dtype=np.int32) for _ in range(3))  # index_stride = 8 bytes: every other slot
        wa[::2], wb[::2], wc[::2] = sa, sb, sc
        c = c0.copy(); assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, c, base, wa, wb, wc, batch, index_stride=8)
        assert np.array_equal(c, ref)
    pa = np.array([a.ctypes.data + 8 * i * m * k for i in range(batch)], dtype=np.uint64)
    pb = np.array([b.ctypes.data + 8 * i * k * n for i in range(batch)], dtype=np.uint64)
    c = c0.copy(); pc = np.array([c.ctypes.data + 8 * i * m * n for i in range(batch)], dtype=np.uint64)
    assert 0 == orc.gemm_batch_ptr(orc.FMA, 8, 0, m, n, k, m, k, m, pa, pb, pc, batch)
    assert np.array_equal(c, ref)
    # shared B (stride NULL) and a negative batchsize
    ref2 = c0.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, ref2, m * k, 0, m * n, batch)
    c = c0.copy(); pc = np.array([c.ctypes.data + 8 * i * m * n for i in range(batch)], dtype=np.uint64)
    assert 0 == orc.gemm_batch_ptr(orc.FMA, 8, 0, m, n, k, m, k, m, pa, pb, pc, -batch, db=None)
    assert np.array_equal(c, ref2)


def test_batch_reduce_equals_shared_c_walk(orc):
    m, n, k, cnt = 13, 23, 32, 11
    rng = np.random.default_rng(6)
    As = [rng.uniform(-1, 1, m * k) for _ in range(cnt)]; Bs = [rng.uniform(-1, 1, k * n) for _ in range(cnt)]
    c0 = rng.uniform(-1, 1, m * n)
    c1 = c0.copy(); orc.smm_reduce(orc.FMA, 0, m, n, k, m, k, m, As, Bs, c1)
    c2 = c0.copy()
    for x, y in zip(As, Bs):
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, x, y, c2)
    assert np.array_equal(c1, c2)
    expect = sum(x.reshape(k, m).T @ y.reshape(n, k).T for x, y in zip(As, Bs)) + c0.reshape(n, m).T
    assert np.max(np.abs(c1.reshape(n, m).T - expect)) <= 1e-12


def test_spmdm_geometry_and_slices(orc):
    h1 = orc.spmdm_init(64, 48, 64, 1, 48); h8 = orc.spmdm_init(64, 48, 64, 8, 48)
    assert (h1.bm, h1.bn, h1.bk, h1.mb, h1.nb, h1.kb) == (70, 48, 128, 1, 1, 1)   # observed from the reference
    assert (h8.bm, h8.mb) == (32, 2)
    h = orc.spmdm_init(2048, 2048, 2048, 8, 48)
    assert (h.bm, h.bn, h.bk) == (256, 48, 128)                                   # bm=256, bn=48, bk=128
    # slices: independent numpy construction (row scan, ascending column, != 0, -0 dropped, NaN kept)
    rng = np.random.default_rng(8)
    M, N, K = 150, 20, 300
    a = rng.uniform(-1, 1, (M, K)).astype(np.float32); a[rng.random((M, K)) < 0.8] = 0.0
    a[3, 7] = -0.0; a[4, 9] = np.nan
    for ta, arr in (("N", a), ("T", np.ascontiguousarray(a.T))):
        hnd, sl = orc.spmdm_slices(M, N, K, 48, ta, arr.reshape(-1), max_threads=4)
        for kb in range(hnd.kb):
            for mb in range(hnd.mb):
                blk = a[mb * hnd.bm:(mb + 1) * hnd.bm, kb * hnd.bk:(kb + 1) * hnd.bk]
                ri, ci, va = sl[kb * hnd.mb + mb]
                keep = ~(blk == 0)
                assert np.array_equal(ri, np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.uint16))
                assert np.array_equal(ci, np.nonzero(keep)[1].astype(np.uint16))
                assert np.array_equal(va.view(np.uint32), blk[keep].view(np.uint32))


@pytest.mark.parametrize("variant", [("N", "N", "N"), ("T", "N", "T"), ("N", "T", "N")])
def test_spmdm_compute_vs_gold(orc, variant):
    ta, tb, tc = variant
    M, N, K = 64, 48, 64
    orc.rng_seed(1)  # samples/spmdm/spmdm.c:212-243
    a = np.array([(lambda r: r if r > 0.85 else 0.0)(orc.rng_f64()) for _ in range(M * K)], dtype=np.float32)
    b = np.array([orc.rng_f64() for _ in range(K * N)], dtype=np.float32)
    c = np.zeros(M * N, dtype=np.float32)
    orc.spmdm_exec(orc.FMA, M, N, K, 48, ta, tb, tc, 0.0, a, b, c)
    A = a.reshape(K, M).T if ta == "T" else a.reshape(M, K)
    B = b.reshape(N, K).T if tb == "T" else b.reshape(K, N)
    got = c.reshape(N, M).T if tc == "T" else c.reshape(M, N)
    assert np.max(np.abs(got - A.astype(np.float64) @ B.astype(np.float64))) <= 1e-5  # the sample reports "max error" ~1e-6


def test_blocked_gemm_oracle(orc):
    for order in range(6):  # internal_bgemm_order is a bijection of the work items
        seen = set()
        for w in range(3 * 4 * 5):
            i2, j2, k2 = C.c_int(), C.c_int(), C.c_int()
            orc.lib().xo_bgemm_order(order, w, 3, 4, 5, C.byref(i2), C.byref(j2), C.byref(k2))
            assert 0 <= i2.value < 3 and 0 <= j2.value < 4 and 0 <= k2.value < 5
            seen.add((i2.value, j2.value, k2.value))
        assert len(seen) == 60
    m, n, k = 64, 96, 128
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, m * k); b = rng.uniform(-1, 1, k * n); c = rng.uniform(-1, 1, m * n)
    expect = a.reshape(k, m).T @ b.reshape(n, k).T + c.reshape(n, m).T
    for order in range(6):
        h = orc.bgemm_init(8, m, n, k, 32, 32, 32, order=order)
        ba, bb, bc, out = np.zeros_like(a), np.zeros_like(b), np.zeros_like(c), np.zeros_like(c)
        orc.bgemm_copy(h, "a", a, m, ba); orc.bgemm_copy(h, "b", b, k, bb); orc.bgemm_copy(h, "c", c, m, bc)
        assert sorted(ba) == sorted(a)  # pure permutation
        orc.bgemm_st(orc.FMA, h, ba, bb, bc)
        orc.bgemm_copy(h, "out", bc, m, out)
        assert np.max(np.abs(out.reshape(n, m).T - expect)) <= 1e-12
    assert orc.bgemm_init(8, 64, 64, 64, 24, 32, 32) is None  # 64 % 24 != 0 (libxsmm_blocked_gemm.c:65)


def test_blocked_permutations_oracle(orc):
    """convert_b_to_a / transpose_b restatements against their array-view meaning (reference templates
    libxsmm_blocked_gemm_convert_b_to_a.tpl.c:32-46, libxsmm_blocked_gemm_transpose_b.tpl.c:32-65)."""
    rng = np.random.default_rng(4)
    m, n, k, bm, bn, bk = 64, 96, 96, 16, 32, 32
    h = orc.bgemm_init(8, m, n, k, bm, bn, bk)
    mb, nb, kb = m // bm, n // bn, k // bk
    src = rng.uniform(-1, 1, m * n); dst = np.zeros_like(src)
    orc.bgemm_permute(h, "convert_b_to_a", src, dst)
    assert np.array_equal(dst.reshape(mb, nb, bn, bm), src.reshape(nb, mb, bn, bm).transpose(1, 0, 2, 3))
    # n == k and bn == bk: block transpose plus transpose inside each block
    src = rng.uniform(-1, 1, k * n); dst = np.zeros_like(src)
    orc.bgemm_permute(h, "transpose_b", src, dst)
    assert np.array_equal(dst.reshape(nb, kb, bn, bk), src.reshape(kb, nb, bk, bn).transpose(1, 0, 3, 2))
    # generic branch, small enough for a Python walk of the reference's index arithmetic
    m, n, k, bm, bn, bk = 8, 12, 6, 4, 4, 3
    h = orc.bgemm_init(8, m, n, k, bm, bn, bk)
    nb, kb = n // bn, k // bk
    src = rng.uniform(-1, 1, k * n); dst = np.full_like(src, np.nan)
    orc.bgemm_permute(h, "transpose_b", src, dst)
    expect = np.full_like(src, np.nan)
    s4 = src.reshape(kb, nb, bk, bn)
    for ikb in range(kb):
        for inb in range(nb):
            for ik in range(bk):
                for jn in range(bn):
                    job = (ikb * bk + ik) * n + (inb * bn + jn)
                    ii, jj = divmod(job, k)
                    q, r = divmod(jj * n + ii, k)
                    expect[(((q // bn) * kb + r // bk) * bn + q % bn) * bk + r % bk] = s4[ikb, inb, ik, jn]
    assert np.array_equal(dst, expect) and sorted(dst) == sorted(src)
