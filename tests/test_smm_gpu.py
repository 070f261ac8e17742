"""GPU parity of the batched dense SMM path, through the C-ABI, against the CPU oracle.

Mirrors how the reference tests this path: tests/gemm.c (shape table :75-82, NaN-filled C for beta=0 :159-167) and
samples/smm/specialized.cpp (MATINIT inputs :143-146, direct kernel calls vs libxsmm_gemm_batch, CHECK :224-238).
Bar: the scalar-FMA kernels must equal the oracle's k-ordered fma chain bit for bit; the MFMA kernels must be within
1e-6 (fp32) / 1e-12 (fp64) relative (north_star tolerance).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {np.float32: 1e-6, np.float64: 1e-12}


def rel_err(ref, tst):
    den = np.max(np.abs(ref))
    return float(np.max(np.abs(ref.astype(np.float64) - tst.astype(np.float64))) / (den if den > 0 else 1.0))


def make_inputs(rng, dtype, batch, m, n, k, lda, ldb, ldc, transb, matinit, orc):
    asz, bsz, csz = lda * k, ldb * (k if transb else n), ldc * n
    if matinit:  # samples/smm/specialized.cpp:143-146: seeds 42+i / 24+i / 22+i, scale 1/batch
        a = np.concatenate([orc.matinit(42 + i, m, k, lda, 1.0 / batch, dtype) for i in range(batch)])
        b = np.concatenate([orc.matinit(24 + i, (k if not transb else n), (n if not transb else k), ldb, 1.0 / batch, dtype) for i in range(batch)])
        c = np.concatenate([orc.matinit(22 + i, m, n, ldc, 1.0 / batch, dtype) for i in range(batch)])
    else:
        a = rng.uniform(-1, 1, batch * asz).astype(dtype)
        b = rng.uniform(-1, 1, batch * bsz).astype(dtype)
        c = rng.uniform(-1, 1, batch * csz).astype(dtype)
    return a, b, c, asz, bsz, csz


SHAPES = [  # (m, n, k, lda, ldb, ldc) -- includes rows of the reference's table tests/gemm.c:75-82
    (23, 23, 23, 23, 23, 23), (32, 32, 32, 32, 32, 32), (13, 13, 13, 13, 13, 13), (1, 1, 1, 1, 1, 1),
    (3, 5, 7, 3, 7, 3), (64, 64, 64, 64, 64, 64), (16, 35, 35, 96, 35, 96), (23, 29, 31, 32, 32, 32),
    (64, 8, 24, 64, 24, 64), (8, 64, 24, 8, 24, 8), (43, 9, 27, 48, 32, 48), (5, 13, 70, 5, 70, 5),
    (80, 40, 16, 80, 16, 80), (13, 23, 32, 13, 32, 13), (32, 13, 23, 32, 23, 32),
]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_strided_batch_bitexact(xs, orc, torch_gpu, dtype, shape, beta):
    torch = torch_gpu
    m, n, k, lda, ldb, ldc = shape
    batch = 37
    rng = np.random.default_rng(1234 + m * 7 + n * 3 + k)
    a, b, c, asz, bsz, csz = make_inputs(rng, dtype, batch, m, n, k, lda, ldb, ldc, False, False, orc)
    if beta == 0.0:
        c[:] = np.nan  # reads of C would poison the result (tests/gemm.c:159-167)
        for i in range(batch):  # padding rows (ldc > m) are never written: keep them finite for the comparison
            blk = c[i * csz:(i + 1) * csz].reshape(n, ldc)
            blk[:, m:] = 0.5
    prec = xs.F64 if dtype == np.float64 else xs.F32
    flags = xs.FLAG_BETA_0 if beta == 0.0 else 0
    ref = c.copy()
    orc.gemm_batch_strided(orc.FMA, flags, m, n, k, lda, ldb, ldc, a, b, ref, asz, bsz, csz, batch)
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        blob, desc = xs.descriptor(prec, m, n, k, lda, ldb, ldc, 1.0, beta)
        assert desc
        rc = xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
        assert rc == 0
        torch.cuda.synchronize()
        out = dc.cpu().numpy()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert xs.last_kernel().startswith("smm_")
    assert np.array_equal(out, ref), "max diff %g (kernel %s)" % (np.nanmax(np.abs(out - ref)), xs.last_kernel())


@pytest.mark.parametrize("beta", [1.0, 0.0])
@pytest.mark.parametrize("mfma", [0, 1])
def test_smm32_f32_special(xs, orc, torch_gpu, beta, mfma):
    """BASELINE config 2 shape (fp32 32^3, tight) through the tuned kernels, MFMA off and on: both bit-identical to the oracle's fma chain."""
    torch = torch_gpu
    m = n = k = 32
    batch = 4099  # not a multiple of anything
    rng = np.random.default_rng(7)
    a, b, c, asz, bsz, csz = make_inputs(rng, np.float32, batch, m, n, k, m, k, m, False, False, orc)
    if beta == 0.0:
        c[:] = np.nan
    ref = c.copy()
    flags = xs.FLAG_BETA_0 if beta == 0.0 else 0
    orc.gemm_batch_strided(orc.FMA, flags, m, n, k, m, k, m, a, b, ref, asz, bsz, csz, batch, 8)
    old = xs.lib().libxsmm_amd_set_mfma(mfma)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        blob, desc = xs.descriptor(xs.F32, m, n, k, m, k, m, 1.0, beta)
        assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
        torch.cuda.synchronize()
        out = dc.cpu().numpy()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert xs.last_kernel() == ("smm_f32_32x32x32_mfma" if mfma else "smm_f32_32x32x32_fma")
    # MFMA on: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain and the kernel feeds it k = 2s, 2s + 1 per step, so the matrix-core
    # kernel gives the reference's chain bit for bit as well (tolerance assert kept as the weaker, documented bar)
    assert rel_err(ref, out) <= TOL[np.float32]
    assert np.array_equal(out, ref), mfma


@pytest.mark.parametrize("beta", [1.0, 0.0])
@pytest.mark.parametrize("mode", ["strided", "index"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_smm64_mfma(xs, orc, torch_gpu, dtype, beta, mode):
    """64^3 (tight) on the matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64), one work-group per item: bit-identical to the oracle's k-ordered fma chain;
    strided and index-array addressing (items at odd element offsets: operands only 4-byte aligned), batch not a multiple of the grid."""
    torch = torch_gpu
    m = n = k = 64
    batch = 1543
    rng = np.random.default_rng(64)
    a, b, c, asz, bsz, csz = make_inputs(rng, dtype, batch, m, n, k, m, k, m, False, False, orc)
    if beta == 0.0:
        c[:] = np.nan
    flags = xs.FLAG_BETA_0 if beta == 0.0 else 0
    prec = xs.F64 if dtype == np.float64 else xs.F32
    old = xs.lib().libxsmm_amd_set_mfma(1)
    try:
        ref = c.copy()
        if mode == "strided":
            orc.gemm_batch_strided(orc.FMA, flags, m, n, k, m, k, m, a, b, ref, asz, bsz, csz, batch, 8)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            blob, desc = xs.descriptor(prec, m, n, k, m, k, m, 1.0, beta)
            assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
        else:
            # A and B shifted by one and three elements (operands only 4-byte aligned), walked out of order; distinct C blocks
            a1 = np.concatenate([np.zeros(1, dtype), a])
            b1 = np.concatenate([np.zeros(3, dtype), b])
            sa = (rng.permutation(batch) * asz + 1).astype(np.int32)
            sb = (rng.integers(0, batch, batch) * bsz + 3).astype(np.int32)
            sc = (np.arange(batch) * csz).astype(np.int32)
            assert 0 == orc.gemm_batch_idx(orc.FMA, flags, m, n, k, m, k, m, a1, b1, ref, 0, sa, sb, sc, batch)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a1, b1, c))
            # negative batch size: the caller's promise that C blocks are distinct (src/libxsmm_gemm.c:1338) -- no look at the order of C
            xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, beta, dc, m, 0, 4, sa, sb, sc, -batch)
        torch.cuda.synchronize()
        out = dc.cpu().numpy()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert xs.last_kernel() in (("smm_f64_mfma_wg", "smm_f64_mfma_wg_jit") if dtype == np.float64 else ("smm_f32_64x64x64_mfma",)), xs.last_kernel()
    assert np.array_equal(out, ref)


MFMA_WG_SHAPES = [  # 32 < max(M, N) <= 64, K <= 64: tight and with gaps, odd K (padded step), thin quadrants
    (64, 64, 64, 72, 64, 64), (48, 48, 48, 48, 48, 48), (40, 64, 17, 40, 17, 44), (64, 16, 32, 64, 32, 64),
    (16, 64, 64, 16, 64, 16), (33, 33, 33, 33, 33, 33), (64, 8, 24, 64, 24, 64), (57, 39, 1, 57, 1, 57), (43, 9, 27, 48, 32, 48),
]


def mfma_wave_serves(dtype, m, n, k, lda, ldb, ldc):
    """mirror of smm_mfma_wave_lds / smm_mfma_wave2_lds and the eligibility rules in csrc/xsmm_jit_smm.cpp: the one-wave-per-item
    matrix-core kernel takes tight operands whose LDS images leave room for four waves per CU ("wave": 16-byte chunks when M is a
    multiple of a chunk and K of four, else element by element); fp64 items too large for that are worked on in two halves of
    C's columns ("wave2", not for 64 x 64 x K). None: the work-group form."""
    ts = np.dtype(dtype).itemsize
    chunk = 16 // ts
    tight = (lda, ldb, ldc) == (m, k, m)
    span_loads = (lda * (k - 1) + m + 63) // 64 + (ldb * (n - 1) + k + 63) // 64 + (ldc * (n - 1) + m + 63) // 64
    if max(m, n) <= 32 or not (tight or (2 * lda <= 3 * m and 2 * ldb <= 3 * k and 2 * ldc <= 3 * m and span_loads <= 80)):
        return None  # (moderate gaps over short spans are served by the element-wise build)
    vec = chunk if (tight and m % chunk == 0 and k % 4 == 0) else 1
    kp4 = 4 * ((k + 3) // 4)
    ms = 16 if m <= 16 else (48 if m <= 48 else 64)
    ksd = (kp4 + vec - 1) // vec
    ksd = (ksd + 1 if ksd % 2 == 0 else ksd) * vec
    csd = m
    while not ((csd % 32 == 16) if ts == 8 else (csd % 16 in (4, 12))):
        csd += vec
    lds = (max(n * csd, kp4 * ms) + n * ksd + 64) * ts
    if 4 * lds <= 160 * 1024:
        return "wave"
    nh = n // 2
    if ts == 8 and vec == chunk and n % 2 == 0 and (k * nh) % vec == 0 and (m * nh) % vec == 0 and not (m == 64 and n == 64):
        ksd2 = (k + vec - 1) // vec
        ksd2 = (ksd2 + 1 if ksd2 % 2 == 0 else ksd2) * vec
        lds2 = (k * m + max(nh * ksd2, nh * m) + 64) * ts
        if 4 * lds2 <= 160 * 1024:
            return "wave2"
    return None


@pytest.mark.parametrize("shape", MFMA_WG_SHAPES)
@pytest.mark.parametrize("beta", [1.0, 0.0])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("specialised", [False, True])
def test_smm_mfma_wg(xs, orc, torch_gpu, dtype, shape, beta, specialised):
    """The general matrix-core kernel (work-group per item) against the oracle's fma chain, bit for bit -- including the sign of
    zeros: a batch of all-zero A with C = -0 (the padded k step of an odd K must not turn -0 into +0). Both builds of the one
    source (kernels/smm_mfma_wg.inc): pre-compiled with the shape as kernel arguments, and compiled by hiprtc with the
    descriptor baked in."""
    torch = torch_gpu
    old_jit = os.environ.get("LIBXSMM_AMD_JIT")
    os.environ["LIBXSMM_AMD_JIT"] = "1" if specialised else "0"
    m, n, k, lda, ldb, ldc = shape
    batch = 1100
    rng = np.random.default_rng(99 + m + 64 * n + k)
    a, b, c, asz, bsz, csz = make_inputs(rng, dtype, batch, m, n, k, lda, ldb, ldc, False, False, orc)
    a[:3 * asz] = 0.0
    c[:2 * csz] = -0.0
    b[:bsz] = -np.abs(b[:bsz])
    if beta == 0.0:
        c[:] = np.nan
        for i in range(batch):
            c[i * csz:(i + 1) * csz].reshape(n, ldc)[:, m:] = 0.5
    flags = xs.FLAG_BETA_0 if beta == 0.0 else 0
    ref = c.copy()
    orc.gemm_batch_strided(orc.FMA, flags, m, n, k, lda, ldb, ldc, a, b, ref, asz, bsz, csz, batch, 8)
    old = xs.lib().libxsmm_amd_set_mfma(1)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        blob, desc = xs.descriptor(xs.F64 if dtype == np.float64 else xs.F32, m, n, k, lda, ldb, ldc, 1.0, beta)
        assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
        torch.cuda.synchronize()
        out = dc.cpu().numpy()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
        if old_jit is None:
            del os.environ["LIBXSMM_AMD_JIT"]
        else:
            os.environ["LIBXSMM_AMD_JIT"] = old_jit
    form = (mfma_wave_serves(dtype, *shape) if specialised else None) or "wg"
    assert xs.last_kernel() == ("smm_f64_mfma_" if dtype == np.float64 else "smm_f32_mfma_") + form + ("_jit" if specialised else ""), xs.last_kernel()
    bits = np.uint64 if dtype == np.float64 else np.uint32
    assert np.array_equal(out.view(bits), ref.view(bits))


def test_smm_mfma_wave_index_batches(xs, orc, torch_gpu):
    """index batches whose C blocks the caller promises to be distinct (negative batch size) reach the one-wave-per-item kernel in
    its element-wise form: shuffled operands, index base 1"""
    torch = torch_gpu
    old_jit = os.environ.get("LIBXSMM_AMD_JIT")
    os.environ["LIBXSMM_AMD_JIT"] = "1"; os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old = xs.lib().libxsmm_amd_set_mfma(1)
    try:
        for dtype, prec in ((np.float64, xs.F64), (np.float32, xs.F32)):
            for (m, n, k) in ((40, 40, 40), (33, 35, 37)):
                batch = 1500
                rng = np.random.default_rng(m + n)
                a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype); c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
                pa, pb = rng.permutation(batch), rng.permutation(batch)
                sa = (pa * m * k + 1).astype(np.int32); sb = (pb * k * n + 1).astype(np.int32); sc = (np.arange(batch) * m * n + 1).astype(np.int32)
                ref = c.copy()
                orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 1, sa, sb, sc, batch)
                da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
                xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 1, 4, sa, sb, sc, -batch)
                torch.cuda.synchronize()
                assert xs.last_kernel() == ("smm_f64_mfma_wave_jit" if dtype == np.float64 else "smm_f32_mfma_wave_jit"), xs.last_kernel()
                assert np.array_equal(dc.cpu().numpy(), ref), (dtype, m, n, k)
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
        os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        if old_jit is None:
            del os.environ["LIBXSMM_AMD_JIT"]
        else:
            os.environ["LIBXSMM_AMD_JIT"] = old_jit


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(48, 48, 48), (40, 36, 20), (33, 35, 37), (56, 40, 32)])
def test_smm_mfma_wave_trans_b(xs, orc, torch_gpu, dtype, shape):
    """TRANS_B (B^T in memory, src/generator_gemm.c:219-223) beyond 32 on the one-wave-per-item kernel: the image of B is k-major like
    A's; the oracle's fma chain bit for bit, beta = 1 and 0."""
    torch = torch_gpu
    m, n, k = shape
    old_jit = os.environ.get("LIBXSMM_AMD_JIT")
    os.environ["LIBXSMM_AMD_JIT"] = "1"; os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old = xs.lib().libxsmm_amd_set_mfma(1)
    try:
        for beta in (1.0, 0.0):
            batch = 1300
            rng = np.random.default_rng(m + 3 * n + k)
            a = rng.uniform(-1, 1, batch * m * k).astype(dtype); bt = rng.uniform(-1, 1, batch * n * k).astype(dtype); c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
            flags = xs.FLAG_TRANS_B | (xs.FLAG_BETA_0 if beta == 0.0 else 0)
            ref = c.copy()
            orc.gemm_batch_strided(orc.FMA, flags, m, n, k, m, n, m, a, bt, ref, m * k, n * k, m * n, batch, 8)
            if beta == 0.0:
                c[:] = np.nan
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, bt, c))
            blob, desc = xs.descriptor(xs.F64 if dtype == np.float64 else xs.F32, m, n, k, m, n, m, 1.0, beta, flags=xs.FLAG_TRANS_B)
            assert desc
            assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), m * k, n * k, m * n, batch)
            torch.cuda.synchronize()
            assert xs.last_kernel() == ("smm_f64_mfma_wave_jit" if dtype == np.float64 else "smm_f32_mfma_wave_jit"), xs.last_kernel()
            bits = np.uint64 if dtype == np.float64 else np.uint32
            assert np.array_equal(dc.cpu().numpy().view(bits), ref.view(bits)), beta
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
        os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        if old_jit is None:
            del os.environ["LIBXSMM_AMD_JIT"]
        else:
            os.environ["LIBXSMM_AMD_JIT"] = old_jit


MFMA_WAVE_SHAPES = [(40, 40, 40), (48, 48, 48), (56, 56, 56), (36, 64, 8), (64, 20, 12), (44, 52, 36), (64, 64, 60), (16, 48, 64), (34, 40, 4),
                    (16, 64, 64), (56, 64, 48), (64, 56, 56), (33, 33, 33), (45, 37, 19), (57, 39, 1), (64, 5, 7), (35, 64, 62)]


@pytest.mark.parametrize("shape", MFMA_WAVE_SHAPES)
@pytest.mark.parametrize("beta", [1.0, 0.0])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_smm_mfma_wave(xs, orc, torch_gpu, dtype, shape, beta):
    """One wave per item on the matrix cores (csrc/xsmm_jit_smm.cpp, SMM_JIT_MFMA_WAVE_BODY: 16x16x4 tiles, operands and C as
    whole lines through LDS, stores deferred behind the next item's wait): the oracle's fma chain bit for bit, for batches
    smaller than, equal to and larger than the resident grid (every wave then walks several items) and with C = -0 / A = 0."""
    torch = torch_gpu
    m, n, k = shape
    form = mfma_wave_serves(dtype, m, n, k, m, k, m)
    if form is None:
        pytest.skip("shape is left to the work-group kernel")
    old_jit = os.environ.get("LIBXSMM_AMD_JIT")
    os.environ["LIBXSMM_AMD_JIT"] = "1"
    flags = xs.FLAG_BETA_0 if beta == 0.0 else 0
    old = xs.lib().libxsmm_amd_set_mfma(1)
    try:
        for batch in (1, 7, 1024, 5003):
            rng = np.random.default_rng(7 * batch + m + 64 * n + k)
            a, b, c, asz, bsz, csz = make_inputs(rng, dtype, batch, m, n, k, m, k, m, False, False, orc)
            a[:asz] = 0.0
            c[:csz] = -0.0
            if beta == 0.0:
                c[:] = np.nan
            ref = c.copy()
            orc.gemm_batch_strided(orc.FMA, flags, m, n, k, m, k, m, a, b, ref, asz, bsz, csz, batch, 8)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            blob, desc = xs.descriptor(xs.F64 if dtype == np.float64 else xs.F32, m, n, k, m, k, m, 1.0, beta)
            # (batches below the threshold of the specialised kernels: the descriptor asks for them explicitly)
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
            assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
            torch.cuda.synchronize()
            assert xs.last_kernel() == ("smm_f64_mfma_%s_jit" % form if dtype == np.float64 else "smm_f32_mfma_wave_jit"), (batch, xs.last_kernel())
            bits = np.uint64 if dtype == np.float64 else np.uint32
            assert np.array_equal(dc.cpu().numpy().view(bits), ref.view(bits)), batch
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
        os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        if old_jit is None:
            del os.environ["LIBXSMM_AMD_JIT"]
        else:
            os.environ["LIBXSMM_AMD_JIT"] = old_jit


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("index_base", [0, 1])
def test_index_arrays_match_reference_walk(xs, orc, torch_gpu, dtype, index_base):
    """libxsmm_gemm_batch with index arrays (src/libxsmm_gemm.c:1333-1364): shuffled A/B, distinct C, shared B via NULL."""
    torch = torch_gpu
    m, n, k = 23, 23, 23
    batch = 1024  # BASELINE config 1 shape
    rng = np.random.default_rng(5)
    a, b, c, asz, bsz, csz = make_inputs(rng, dtype, batch, m, n, k, m, k, m, False, True, orc)
    perm_a, perm_b, perm_c = rng.permutation(batch), rng.permutation(batch), np.arange(batch)
    sa = (perm_a * asz + index_base).astype(np.int32)
    sb = (perm_b * bsz + index_base).astype(np.int32)
    sc = (perm_c * csz + index_base).astype(np.int32)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    for stride_b in (sb, None):
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, index_base, sa, stride_b, sc, batch)
        old = xs.lib().libxsmm_amd_set_mfma(0)
        try:
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            # index arrays once on the host, once on the device
            for on_device in (False, True):
                dc.copy_(torch.from_numpy(c))
                ia = torch.from_numpy(sa).cuda() if on_device else sa
                ib = None if stride_b is None else (torch.from_numpy(stride_b).cuda() if on_device else stride_b)
                ic = torch.from_numpy(sc).cuda() if on_device else sc
                xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, index_base, 4, ia, ib, ic, batch)
                torch.cuda.synchronize()
                assert np.array_equal(dc.cpu().numpy(), ref)
        finally:
            xs.lib().libxsmm_amd_set_mfma(old)


def test_pointer_arrays_and_groups(xs, orc, torch_gpu):
    """index_stride == 0: arrays of pointers (src/libxsmm_gemm.c:1426-1461) and libxsmm_dgemm_batch groups (:1231-1246)."""
    torch = torch_gpu
    rng = np.random.default_rng(11)
    groups = [(13, 13, 13, 40), (23, 13, 32, 25), (32, 32, 32, 30)]
    mats, ptrs, refs = [], [], []
    tot = sum(g[3] for g in groups)
    pa = np.zeros(tot, dtype=np.uint64); pb = np.zeros(tot, dtype=np.uint64); pc = np.zeros(tot, dtype=np.uint64)
    ha, hb, hc = [], [], []
    j = 0
    for (m, n, k, cnt) in groups:
        # one pool per operand and group, the matrices of a group at increasing addresses with a gap in between (C blocks that
        # do not come in increasing order are taken for possible repeats and summed with atomics: tolerance, not bit, parity --
        # tests/test_smm_gpu.py::test_unsorted_duplicate_c_uses_atomics_within_tolerance)
        sa_, sb_, sc_ = m * k + 3, k * n + 5, m * n + 7
        pool_a = rng.uniform(-1, 1, cnt * sa_); pool_b = rng.uniform(-1, 1, cnt * sb_); pool_c = rng.uniform(-1, 1, cnt * sc_)
        dpa_, dpb_, dpc_ = torch.from_numpy(pool_a).cuda(), torch.from_numpy(pool_b).cuda(), torch.from_numpy(pool_c).cuda()
        for i in range(cnt):
            a = pool_a[i * sa_:i * sa_ + m * k]; b = pool_b[i * sb_:i * sb_ + k * n]; c = pool_c[i * sc_:i * sc_ + m * n]
            da, db, dc = dpa_[i * sa_:i * sa_ + m * k], dpb_[i * sb_:i * sb_ + k * n], dpc_[i * sc_:i * sc_ + m * n]
            mats.append((da, db, dc))
            pa[j], pb[j], pc[j] = da.data_ptr(), db.data_ptr(), dc.data_ptr()
            ref = c.copy(); orc.smm(orc.FMA, 0, m, n, k, m, k, m, a, b, ref); refs.append(ref)
            j += 1
    ng = len(groups)
    ta = (C.c_char * ng)(*[b"N"] * ng); tb = (C.c_char * ng)(*[b"N"] * ng)
    ms = (C.c_int * ng)(*[g[0] for g in groups]); ns = (C.c_int * ng)(*[g[1] for g in groups]); ks = (C.c_int * ng)(*[g[2] for g in groups])
    ldas = (C.c_int * ng)(*[g[0] for g in groups]); ldbs = (C.c_int * ng)(*[g[2] for g in groups]); ldcs = (C.c_int * ng)(*[g[0] for g in groups])
    al = (C.c_double * ng)(*[1.0] * ng); be = (C.c_double * ng)(*[1.0] * ng)
    gs = (C.c_int * ng)(*[g[3] for g in groups]); gc = C.c_int(ng)
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        xs.lib().libxsmm_dgemm_batch(ta, tb, ms, ns, ks, al, xs.dptr(pa), ldas, xs.dptr(pb), ldbs, be, xs.dptr(pc), ldcs, C.byref(gc), gs)
        torch.cuda.synchronize()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    for (da, db, dc), ref in zip(mats, refs):
        assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("omp", [False, True])
@pytest.mark.parametrize("second", ["reads_c_of_first_as_a", "reads_c_of_first_as_b", "updates_c_of_first", "beta0_first_then_update"])
def test_groups_that_depend_on_each_other_keep_the_group_order(xs, orc, torch_gpu, omp, second):
    """libxsmm_dgemm_batch[_omp] with two groups where the second depends on the first. The reference works the groups off
    strictly one after the other (src/libxsmm_gemm.c:1231-1262, each group a libxsmm_gemm_batch), so C1 += A B followed by
    C2 += C1 D -- or a second update of C1 -- is legal in one call; the fused launch must not be taken then."""
    torch = torch_gpu
    L = xs.lib()
    rng = np.random.default_rng(31)
    m = n = k = 16
    cnt = 300  # (large enough for the specialised kernels and the grouped launch to be candidates)
    sz = m * n
    A = rng.uniform(-1, 1, cnt * sz); B = rng.uniform(-1, 1, cnt * sz); D = rng.uniform(-1, 1, cnt * sz)
    C1 = rng.uniform(-1, 1, cnt * sz); C2 = rng.uniform(-1, 1, cnt * sz)
    beta0_first = (second == "beta0_first_then_update")
    r1 = C1.copy(); r2 = C2.copy()
    orc.gemm_batch_strided(orc.FMA, 16 if beta0_first else 0, m, n, k, m, k, m, A, B, r1, sz, sz, sz, cnt)
    if second == "reads_c_of_first_as_a":
        orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, r1, D, r2, sz, sz, sz, cnt)
    elif second == "reads_c_of_first_as_b":
        orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, D, r1, r2, sz, sz, sz, cnt)
    else:  # the second group updates the blocks of the first once more
        orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, D, B, r1, sz, sz, sz, cnt)
    dA, dB, dD, d1, d2 = (torch.from_numpy(x).cuda() for x in (A, B, D, C1, C2))
    blk = lambda t, i: t.data_ptr() + i * sz * 8
    pa = np.zeros(2 * cnt, dtype=np.uint64); pb = np.zeros(2 * cnt, dtype=np.uint64); pc = np.zeros(2 * cnt, dtype=np.uint64)
    for i in range(cnt):
        pa[i], pb[i], pc[i] = blk(dA, i), blk(dB, i), blk(d1, i)
        if second == "reads_c_of_first_as_a":
            pa[cnt + i], pb[cnt + i], pc[cnt + i] = blk(d1, i), blk(dD, i), blk(d2, i)
        elif second == "reads_c_of_first_as_b":
            pa[cnt + i], pb[cnt + i], pc[cnt + i] = blk(dD, i), blk(d1, i), blk(d2, i)
        else:
            pa[cnt + i], pb[cnt + i], pc[cnt + i] = blk(dD, i), blk(dB, i), blk(d1, i)
    ng = 2
    ta = (C.c_char * ng)(*[b"N"] * ng); tb = (C.c_char * ng)(*[b"N"] * ng)
    ms = (C.c_int * ng)(m, m); ns = (C.c_int * ng)(n, n); ks = (C.c_int * ng)(k, k)
    lds = (C.c_int * ng)(m, m)
    al = (C.c_double * ng)(1.0, 1.0); be = (C.c_double * ng)(0.0 if beta0_first else 1.0, 1.0)
    gs = (C.c_int * ng)(cnt, cnt); gc = C.c_int(ng)
    old = L.libxsmm_amd_set_mfma(0)
    try:
        f = L.libxsmm_dgemm_batch_omp if omp else L.libxsmm_dgemm_batch
        f(ta, tb, ms, ns, ks, al, xs.dptr(pa), lds, xs.dptr(pb), lds, be, xs.dptr(pc), lds, C.byref(gc), gs)
        torch.cuda.synchronize()
    finally:
        L.libxsmm_amd_set_mfma(old)
    assert np.array_equal(d1.cpu().numpy(), r1)
    assert np.array_equal(d2.cpu().numpy(), r2)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_shared_c_runs_accumulate_in_batch_order(xs, orc, torch_gpu, dtype):
    """CP2K-style stacks (samples/cp2k/cp2k.cpp:328-360): consecutive products update the same C; the sequential
    reference accumulates them in batch order -- the run kernel must give the identical chain."""
    torch = torch_gpu
    m, n, k = 23, 23, 23
    batch, nc = 600, 17
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    c = rng.uniform(-1, 1, nc * m * n).astype(dtype)
    cidx = np.sort(rng.integers(0, nc, batch))  # non-decreasing => runs
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    ref = c.copy()
    orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        out = dc.cpu().numpy()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert np.array_equal(out, ref)


def test_unsorted_duplicate_c_uses_atomics_within_tolerance(xs, orc, torch_gpu):
    torch = torch_gpu
    m, n, k = 16, 16, 16
    batch, nc = 500, 11
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, nc * m * n)
    cidx = rng.integers(0, nc, batch)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    ref = c.copy()
    orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
    torch.cuda.synchronize()
    assert rel_err(ref, dc.cpu().numpy()) <= 1e-12


def test_trans_b_and_general_fallback(xs, orc, torch_gpu):
    """TRANS_B is an SMM kernel (src/generator_gemm.c:219-223); alpha/beta outside {1}/{0,1} and TRANS_A are the BLAS
    fall-back domain of libxsmm_mmbatch (src/libxsmm_gemm.c:1842-1866) -- served by the general device kernel."""
    torch = torch_gpu
    m, n, k, batch = 20, 12, 28, 33
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, batch * m * k); bt = rng.uniform(-1, 1, batch * n * k); c = rng.uniform(-1, 1, batch * m * n)
    ref = c.copy()
    orc.gemm_batch_strided(orc.FMA, orc.FLAG_TRANS_B, m, n, k, m, n, m, a, bt, ref, m * k, n * k, m * n, batch)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * n * k).astype(np.int32); sc = (np.arange(batch) * m * n).astype(np.int32)
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, bt, c))
        xs.gemm_batch(xs.F64, "N", "T", m, n, k, 1.0, da, m, db, n, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), ref)
        # general: C = 0.5*A^T*B^T - 2*C, against numpy
        at = rng.uniform(-1, 1, batch * k * m)
        c2 = rng.uniform(-1, 1, batch * m * n)
        A = at.reshape(batch, m, k)           # column-major k x m  == row-major (m,k): A^T[m][k]
        B = bt.reshape(batch, k, n)           # column-major n x k (ld n) == row-major (k,n): B^T[k][n]
        Cm = c2.reshape(batch, n, m).transpose(0, 2, 1)
        expect = 0.5 * np.einsum("bmk,bkn->bmn", A, B) - 2.0 * Cm
        dat, dc2 = torch.from_numpy(at).cuda(), torch.from_numpy(c2).cuda()
        sat = (np.arange(batch) * k * m).astype(np.int32)
        xs.gemm_batch(xs.F64, "T", "T", m, n, k, 0.5, dat, k, db, n, -2.0, dc2, m, 0, 4, sat, sb, sc, batch)
        torch.cuda.synchronize()
        got = dc2.cpu().numpy().reshape(batch, n, m).transpose(0, 2, 1)
        assert np.max(np.abs(got - expect)) <= 1e-12 * max(1.0, np.max(np.abs(expect)))
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)


def test_dispatched_kernel_pointer_call_device_and_host(xs, orc, torch_gpu):
    """A dispatched kernel is a bare function pointer called as f(a,b,c[,pa,pb,pc]) (samples/smm/specialized.cpp:172-190):
    works on device operands (asynchronous) and on plain host memory (staged)."""
    torch = torch_gpu
    m, n, k = 23, 23, 23
    a = orc.matinit(42, m, k, m, 1.0, np.float64); b = orc.matinit(24, k, n, k, 1.0, np.float64); c = orc.matinit(22, m, n, m, 1.0, np.float64)
    ref = c.copy(); orc.smm(orc.FMA, 0, m, n, k, m, k, m, a, b, ref)
    fn = xs.lib().libxsmm_dmmdispatch(m, n, k, None, None, None, None, None, None, None)
    assert fn
    assert fn == xs.lib().libxsmm_dmmdispatch(m, n, k, None, None, None, None, None, None, None)  # registry hit: same pointer
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.call_kernel(fn, da, db, dc)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), ref)
        hc = c.copy()
        xs.call_kernel(fn, a, b, hc)  # host pointers: synchronous
        assert np.array_equal(hc, ref)
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)


def test_batch_reduce_kernel(xs, orc, torch_gpu):
    """libxsmm_dmmdispatch_reducebatch (src/libxsmm_main.c:2262-2276): C += sum_i A_i*B_i, one chain in batch order."""
    torch = torch_gpu
    m, n, k, cnt = 32, 13, 23, 9
    rng = np.random.default_rng(4)
    As = [rng.uniform(-1, 1, m * k) for _ in range(cnt)]; Bs = [rng.uniform(-1, 1, k * n) for _ in range(cnt)]
    c = rng.uniform(-1, 1, m * n)
    ref = c.copy(); orc.smm_reduce(orc.FMA, 0, m, n, k, m, k, m, As, Bs, ref)
    fn = xs.lib().libxsmm_dmmdispatch_reducebatch(m, n, k, None, None, None, None, None, None, None)
    assert fn
    dA = [torch.from_numpy(x).cuda() for x in As]; dB = [torch.from_numpy(x).cuda() for x in Bs]; dc = torch.from_numpy(c).cuda()
    pa = np.array([t.data_ptr() for t in dA], dtype=np.uint64); pb = np.array([t.data_ptr() for t in dB], dtype=np.uint64)
    count = np.array([cnt], dtype=np.uint64)
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        xs.call_kernel(fn, pa, pb, dc, count)
        torch.cuda.synchronize()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert np.array_equal(dc.cpu().numpy(), ref)


def test_host_operands_are_staged(xs, orc, torch_gpu):
    """An unchanged CPU caller passes malloc'ed memory: libxsmm_gemm_batch must still produce the reference result."""
    m, n, k, batch = 13, 13, 13, 200
    rng = np.random.default_rng(8)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, batch * m * n)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (np.arange(batch) * m * n).astype(np.int32)
    ref = c.copy(); orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        out = c.copy()
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, a, m, b, k, 1.0, out, m, 0, 4, sa, sb, sc, batch)
        assert np.array_equal(out, ref)
        # pointer arrays of host matrices
        out2 = c.copy()
        pa = np.array([a.ctypes.data + 8 * i * m * k for i in range(batch)], dtype=np.uint64)
        pb = np.array([b.ctypes.data + 8 * i * k * n for i in range(batch)], dtype=np.uint64)
        pc = np.array([out2.ctypes.data + 8 * i * m * n for i in range(batch)], dtype=np.uint64)
        eight = np.array([8], dtype=np.int32)
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, pa, m, pb, k, 1.0, pc, m, 0, 0, eight, eight, eight, batch)
        assert np.array_equal(out2, ref)
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)


def test_concurrent_threads_with_their_own_streams(xs, orc, torch_gpu):
    """Every entry point may be called from any thread (reference tests/threadsafety.c); the engine's stream is a per-thread
    setting, so threads can overlap independent batches on the GPU. Four threads, four shapes, stacks with runs of equal C."""
    import threading
    torch = torch_gpu
    shapes = [(23, 23, 23), (13, 32, 5), (32, 32, 32), (7, 9, 11)]
    results, errors = {}, []

    def work(tid, shape):
        try:
            m, n, k = shape
            batch, nc = 700 + 13 * tid, 31
            rng = np.random.default_rng(100 + tid)
            a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, nc * m * n)
            cidx = np.sort(rng.integers(0, nc, batch))
            sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
            ref = c.copy()
            for _ in range(3):
                orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
                dia, dib, dic = (torch.from_numpy(x).cuda() for x in (sa, sb, sc))
                stream.synchronize()
                xs.lib().libxsmm_amd_set_stream(C.c_void_p(stream.cuda_stream))
                for _ in range(3):
                    xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, dia, dib, dic, batch)
                stream.synchronize()
                results[tid] = (dc.cpu().numpy(), ref)
        except Exception as exc:  # surfaced in the main thread
            errors.append(repr(exc))

    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        threads = [threading.Thread(target=work, args=(i, s)) for i, s in enumerate(shapes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert not errors, errors
    for tid in range(len(shapes)):
        out, ref = results[tid]
        assert np.array_equal(out, ref), tid


@pytest.mark.gpu
@pytest.mark.parametrize("big", [False, True])
def test_mmbatch_tasks_on_threads_share_c_blocks(xs, orc, torch_gpu, big):
    """libxsmm_mmbatch(..., tid, ntasks) called concurrently from ntasks threads (reference src/libxsmm_gemm.c:1315-1324 slices,
    :1366-1423 a lock per C): every slice is strictly increasing in C on its own, yet all slices update the same C blocks.
    The engine cannot see the other slices, so with a positive batchsize every update is an atomic add (tolerance parity:
    the reference's order depends on thread timing as well)."""
    import threading
    torch = torch_gpu
    L = xs.lib()
    m, n, k = 23, 11, 17
    ntasks = 4
    per = 6000 if big else 300  # big: the hiprtc-specialised kernels (>= 16384 items would need no override; forced below)
    batch = ntasks * per
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, per * m * n)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32)
    sc = ((np.arange(batch) % per) * m * n).astype(np.int32)  # slice t: blocks 0 .. per-1, like every other slice
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    dsa, dsb, dsc = (torch.from_numpy(x).cuda() for x in (sa, sb, sc))
    torch.cuda.synchronize()
    errors = []
    one = C.c_double(1.0)
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    if big:
        os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old = L.libxsmm_amd_set_mfma(0)

    def work(tid):
        try:
            stream = torch.cuda.Stream()
            L.libxsmm_amd_set_stream(C.c_void_p(stream.cuda_stream))
            L.libxsmm_mmbatch(xs.F64, xs.F64, b"N", b"N", m, n, k, C.byref(one), da.data_ptr(), None, db.data_ptr(), None, C.byref(one), dc.data_ptr(), None,
                              0, 4, dsa.data_ptr(), dsb.data_ptr(), dsc.data_ptr(), batch, tid, ntasks)
            stream.synchronize()
        except Exception as exc:
            errors.append(repr(exc))
    try:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(ntasks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        L.libxsmm_amd_set_mfma(old)
        if big:
            if old_env is None:
                del os.environ["LIBXSMM_AMD_JIT_MINBATCH"]
            else:
                os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env
    assert not errors, errors
    torch.cuda.synchronize()
    out = dc.cpu().numpy()
    assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref))


@pytest.mark.parametrize("where", ["device", "host"])
def test_general_batch_repeated_c_is_sequential(xs, orc, torch_gpu, where):
    """General form (alpha = 2, beta = -1: the BLAS-fallback domain of libxsmm_mmbatch, src/libxsmm_gemm.c:1842-1866 ->
    libxsmm_mmbatch_blas :1778-1806, a sequential loop): items that share a C apply C = alpha*A_i*B_i + beta*C one after the
    other in batch order -- consecutive repeats, shuffled repeats, and stride_c == NULL (one C for the whole batch)."""
    torch = torch_gpu
    m, n, k, batch, nblocks = 9, 7, 11, 60, 13
    rng = np.random.default_rng(21)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32)
    cases = {
        "runs": np.sort(rng.integers(0, nblocks, batch)),
        "shuffled": rng.integers(0, nblocks, batch),
        "alternating": np.arange(batch) % 2,
        "single": None,
    }
    for name, blocks in cases.items():
        c = rng.uniform(-1, 1, nblocks * m * n)
        sc = None if blocks is None else (blocks * m * n).astype(np.int32)
        ref = c.copy()
        for i in range(batch):
            off = 0 if blocks is None else int(blocks[i]) * m * n
            A = a[i * m * k:(i + 1) * m * k].reshape(k, m).T; B = b[i * k * n:(i + 1) * k * n].reshape(n, k).T
            Cm = ref[off:off + m * n].reshape(n, m).T
            ref[off:off + m * n] = (2.0 * (A @ B) - 1.0 * Cm).T.reshape(-1)
        if where == "device":
            xa, xb, xc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        else:
            xa, xb, xc = a, b, c.copy()
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 2.0, xa, m, xb, k, -1.0, xc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        out = xc.cpu().numpy() if where == "device" else xc
        assert np.max(np.abs(out - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref))), name
    # arrays of pointers with repeats (device matrices, host pointer arrays)
    c = rng.uniform(-1, 1, nblocks * m * n); blocks = rng.integers(0, nblocks, batch)
    ref = c.copy()
    for i in range(batch):
        off = int(blocks[i]) * m * n
        A = a[i * m * k:(i + 1) * m * k].reshape(k, m).T; B = b[i * k * n:(i + 1) * k * n].reshape(n, k).T
        ref[off:off + m * n] = (2.0 * (A @ B) - 1.0 * ref[off:off + m * n].reshape(n, m).T).T.reshape(-1)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    pa = np.array([da.data_ptr() + 8 * i * m * k for i in range(batch)], dtype=np.uint64)
    pb = np.array([db.data_ptr() + 8 * i * k * n for i in range(batch)], dtype=np.uint64)
    pc = np.array([dc.data_ptr() + 8 * int(blocks[i]) * m * n for i in range(batch)], dtype=np.uint64)
    eight = np.array([8], dtype=np.int32)
    xs.gemm_batch(xs.F64, "N", "N", m, n, k, 2.0, pa, m, pb, k, -1.0, pc, m, 0, 0, eight, eight, eight, batch)
    torch.cuda.synchronize()
    assert np.max(np.abs(dc.cpu().numpy() - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("negative", [False, True])
def test_mmbatch_tasks_on_threads_with_host_operands(xs, orc, torch_gpu, negative):
    """libxsmm_mmbatch(..., tid, ntasks) from ntasks threads with operands in pageable host memory (an unchanged CPU caller):
    every task stages its slice through private device copies, so the staged slices take turns (the reference takes a lock
    per C, src/libxsmm_gemm.c:1366-1423). Shared C blocks (positive batchsize) and disjoint but interleaved C blocks
    (negative batchsize: the caller's promise that nothing is shared) -- no update and no finished block is lost."""
    import threading
    L = xs.lib()
    m, n, k = 13, 5, 7
    ntasks, per = 4, 50
    batch = ntasks * per
    rng = np.random.default_rng(8)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32)
    if negative:  # item i owns block perm[i]: the tasks' index ranges interleave, no block is shared
        blocks = rng.permutation(batch)
        c = rng.uniform(-1, 1, batch * m * n)
    else:         # every task updates the same `per` blocks
        blocks = np.arange(batch) % per
        c = rng.uniform(-1, 1, per * m * n)
    sc = (blocks * m * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    one = C.c_double(1.0)
    errors = []

    def work(tid):
        try:
            L.libxsmm_mmbatch(xs.F64, xs.F64, b"N", b"N", m, n, k, C.byref(one), a.ctypes.data, None, b.ctypes.data, None, C.byref(one), c.ctypes.data, None,
                              0, 4, sa.ctypes.data, sb.ctypes.data, sc.ctypes.data, -batch if negative else batch, tid, ntasks)
        except Exception as exc:
            errors.append(repr(exc))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(ntasks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert np.max(np.abs(c - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def test_shuffled_duplicate_c_in_pinned_host_memory(xs, orc, torch_gpu):
    """C in libxsmm_malloc memory (pinned host memory the GPU works on in place) with C blocks that repeat out of order:
    hardware floating-point atomics do not reach such memory, the sums join C by compare-and-swap (kernels/smm_generic.hip)."""
    torch = torch_gpu
    L = xs.lib()
    m, n, k, batch, nblocks = 8, 8, 8, 500, 17
    rng = np.random.default_rng(12)
    for dtype, prec, tol in ((np.float64, xs.F64, 1e-12), (np.float32, xs.F32, 1e-5)):
        a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
        c0 = rng.uniform(-1, 1, nblocks * m * n).astype(dtype)
        blocks = rng.integers(0, nblocks, batch)
        sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (blocks * m * n).astype(np.int32)
        ref = c0.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
        bufs = []
        for x in (a, b, c0):
            ptr = L.libxsmm_malloc(x.nbytes); assert ptr
            C.memmove(ptr, x.ctypes.data, x.nbytes); bufs.append(ptr)
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, bufs[0], m, bufs[1], k, 1.0, bufs[2], m, 0, 4, sa, sb, sc, batch)  # returns when C is written
        out = np.frombuffer((C.c_char * c0.nbytes).from_address(bufs[2]), dtype=dtype).copy()
        for ptr in bufs:
            L.libxsmm_free(ptr)
        assert np.max(np.abs(out.astype(np.float64) - ref.astype(np.float64))) <= tol * max(1.0, float(np.max(np.abs(ref)))) * 8
