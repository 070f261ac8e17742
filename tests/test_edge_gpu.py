"""Edge cases and full-size properties of the dense path on the GPU (through the C-ABI).

Edge cases the reference's own drivers exercise: empty and single-item batches, task slicing by (tid, nthreads)
(src/libxsmm_gemm.c:1321-1324), wide index strides (LIBXSMM_ACCESS byte stepping), negative batch sizes (:1338), shapes beyond
one tile / one K chunk, auto-batch recording (src/libxsmm_ext_gemm.c:1016-1135), libxsmm_?gemm (LIBXSMM_XGEMM).
Full size (BASELINE config 2: 1,048,576 items): properties that need no CPU pass over all data -- beta=0 idempotence,
exact power-of-two linearity, and a sampled comparison with the oracle.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def strided_ref(orc, dtype, flags, m, n, k, a, b, c, batch):
    ref = c.copy()
    orc.gemm_batch_strided(orc.FMA, flags, m, n, k, m, k, m, a, b, ref, m * k, k * n, m * n, batch, 4)
    return ref


def test_empty_and_single_item_batches(xs, orc, torch_gpu):
    torch = torch_gpu
    m, n, k = 23, 23, 23
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, m * k); b = rng.uniform(-1, 1, k * n); c = rng.uniform(-1, 1, m * n)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    blob, desc = xs.descriptor(xs.F64, m, n, k)
    launches = xs.lib().libxsmm_amd_launch_count()
    assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), m * k, k * n, m * n, 0)
    zero = np.zeros(1, dtype=np.int32)
    xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, zero, zero, zero, 0)
    torch.cuda.synchronize()
    assert xs.lib().libxsmm_amd_launch_count() == launches and np.array_equal(dc.cpu().numpy(), c)
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, zero, zero, zero, 1)
        torch.cuda.synchronize()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    ref = c.copy(); orc.smm(orc.FMA, 0, m, n, k, m, k, m, a, b, ref)
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(64, 64, 64), (100, 70, 40), (16, 16, 300), (70, 5, 129), (3, 130, 7)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_shapes_beyond_one_tile(xs, orc, torch_gpu, shape, dtype):
    """M,N > 64 loop over C tiles, large K over LDS chunks; the chain per element must stay k-ordered."""
    torch = torch_gpu
    m, n, k = shape
    batch = 9
    rng = np.random.default_rng(m + n + k)
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
    ref = strided_ref(orc, dtype, 0, m, n, k, a, b, c, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        blob, desc = xs.descriptor(prec, m, n, k)
        assert desc and 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), m * k, k * n, m * n, batch)
        torch.cuda.synchronize()
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)
    assert np.array_equal(dc.cpu().numpy(), ref)


def test_task_slices_wide_index_stride_and_negative_batch(xs, orc, torch_gpu):
    torch = torch_gpu
    m, n, k, batch = 13, 13, 13, 103
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, batch * m * n)
    ref = strided_ref(orc, np.float64, 0, m, n, k, a, b, c, batch)
    # index arrays with a 12-byte stride (an array of {a,b,c} int triples, the CP2K "stack" layout)
    trip = np.zeros((batch, 3), dtype=np.int32)
    trip[:, 0] = np.arange(batch) * m * k + 1; trip[:, 1] = np.arange(batch) * k * n + 1; trip[:, 2] = np.arange(batch) * m * n + 1
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    al, be = C.c_double(1.0), C.c_double(1.0)
    base = trip.ctypes.data
    old = xs.lib().libxsmm_amd_set_mfma(0)
    try:
        for nthreads, sign in ((1, 1), (4, 1), (3, -1)):
            dc = torch.from_numpy(c).cuda()
            for tid in range(nthreads):  # every task processes its own slice (reference: called from an OpenMP region)
                xs.lib().libxsmm_mmbatch(xs.F64, xs.F64, b"N", b"N", m, n, k, C.byref(al), xs.dptr(da), None, xs.dptr(db), None, C.byref(be),
                                         xs.dptr(dc), None, 1, 12, C.c_void_p(base), C.c_void_p(base + 4), C.c_void_p(base + 8),
                                         sign * batch, tid, nthreads)
            torch.cuda.synchronize()
            out = dc.cpu().numpy()
            if nthreads > 1 and sign > 0:
                # several tasks, positive batchsize: other tasks may update the same C blocks concurrently (the reference locks per
                # C, src/libxsmm_gemm.c:1366-1423), so every update is an atomic add of a product summed from zero: tolerance
                assert np.max(np.abs(out - ref)) <= 1e-12 * np.max(np.abs(ref)), (nthreads, sign)
            else:
                assert np.array_equal(out, ref), (nthreads, sign)
    finally:
        xs.lib().libxsmm_amd_set_mfma(old)


def test_autobatch_recording_and_single_gemm(xs, orc, torch_gpu):
    torch = torch_gpu
    m, n, k, cnt = 23, 23, 23, 50
    rng = np.random.default_rng(5)
    As = [rng.uniform(-1, 1, m * k) for _ in range(cnt)]; Bs = [rng.uniform(-1, 1, k * n) for _ in range(cnt)]
    c = rng.uniform(-1, 1, m * n)
    ref = c.copy(); orc.smm_reduce(orc.FMA, 0, m, n, k, m, k, m, As, Bs, ref)  # all products into one C, in call order
    dA = [torch.from_numpy(x).cuda() for x in As]; dB = [torch.from_numpy(x).cuda() for x in Bs]; dc = torch.from_numpy(c).cuda()
    L = xs.lib()
    im, i_n, ik = C.c_int(m), C.c_int(n), C.c_int(k)
    old = L.libxsmm_amd_set_mfma(0)
    try:
        launches = L.libxsmm_amd_launch_count()
        L.libxsmm_mmbatch_begin(xs.F64, None, C.byref(im), C.byref(i_n), C.byref(ik), None, None, None, None, None)
        for x, y in zip(dA, dB):  # CP2K inner loop style: one libxsmm_dgemm per product (samples/cp2k/cp2k.cpp:341-346)
            L.libxsmm_dgemm(b"N", b"N", C.byref(im), C.byref(i_n), C.byref(ik), None, xs.dptr(x), None, xs.dptr(y), None, None, xs.dptr(dc), None)
        assert L.libxsmm_amd_launch_count() == launches  # nothing ran yet
        L.libxsmm_mmbatch_end()
        torch.cuda.synchronize()
        assert L.libxsmm_amd_launch_count() <= launches + 2  # the order check and one batch launch
        assert np.array_equal(dc.cpu().numpy(), ref)
        # outside of a recording the same call executes immediately; alpha=2, beta=-1, 'T' go through the general kernel
        a, b = As[0], Bs[0]
        out = torch.from_numpy(c).cuda()
        al, be = C.c_double(2.0), C.c_double(-1.0)
        L.libxsmm_dgemm(b"T", b"N", C.byref(im), C.byref(i_n), C.byref(ik), C.byref(al), xs.dptr(dA[0]), None, xs.dptr(dB[0]), None, C.byref(be), xs.dptr(out), None)
        torch.cuda.synchronize()
        expect = 2.0 * (a.reshape(m, k) @ b.reshape(n, k).T) - c.reshape(n, m).T  # op(A) = A^T: stored k x m col-major == (m,k) row-major
        assert np.max(np.abs(out.cpu().numpy().reshape(n, m).T - expect)) <= 1e-12 * np.max(np.abs(expect))
    finally:
        L.libxsmm_amd_set_mfma(old)


@pytest.mark.parametrize("mfma", [1, 0])
def test_full_size_config2_properties(xs, orc, torch_gpu, mfma):
    """BASELINE config 2 at full size: 1,048,576 items of fp32 32^3 (12 GiB of operands)."""
    torch = torch_gpu
    m = n = k = 32
    batch = 1048576
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    a = torch.rand(batch * 1024, device="cuda", generator=g) - 0.5
    b = torch.rand(batch * 1024, device="cuda", generator=g) - 0.5
    c = torch.full((batch * 1024,), float("nan"), device="cuda")
    L = xs.lib()
    blob0, desc0 = xs.descriptor(xs.F32, m, n, k, beta=0.0)
    blob1, desc1 = xs.descriptor(xs.F32, m, n, k, beta=1.0)
    old = L.libxsmm_amd_set_mfma(mfma)
    try:
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc0, xs.dptr(a), xs.dptr(b), xs.dptr(c), 1024, 1024, 1024, batch)
        first = c.clone()
        assert not torch.isnan(first).any()                     # beta = 0 never reads C, every element was written
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc0, xs.dptr(a), xs.dptr(b), xs.dptr(c), 1024, 1024, 1024, batch)
        assert torch.equal(first, c)                             # idempotent and deterministic
        a2 = a * 2.0                                             # exact scaling => exactly doubled products
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc0, xs.dptr(a2), xs.dptr(b), xs.dptr(c), 1024, 1024, 1024, batch)
        assert torch.equal(c, first * 2.0)
        del a2
        # beta = 1 on top of the beta = 0 result, sampled against the oracle (items spread over the whole batch)
        assert 0 == L.libxsmm_amd_gemm_batch_strided(desc1, xs.dptr(a), xs.dptr(b), xs.dptr(c), 1024, 1024, 1024, batch)
        torch.cuda.synchronize()
        idx = np.unique(np.concatenate([[0, 1, batch - 1], np.random.default_rng(1).integers(0, batch, 253)]))
        sel = torch.from_numpy(idx).cuda()
        ha = a.view(batch, 1024)[sel].cpu().numpy().reshape(-1); hb = b.view(batch, 1024)[sel].cpu().numpy().reshape(-1)
        hc0 = (first * 2.0).view(batch, 1024)[sel].cpu().numpy().reshape(-1)
        got = c.view(batch, 1024)[sel].cpu().numpy().reshape(-1)
        ref = hc0.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, ha, hb, ref, 1024, 1024, 1024, len(idx), 4)
        if mfma:
            assert np.max(np.abs(got - ref)) <= 1e-6 * np.max(np.abs(ref))
        else:
            assert np.array_equal(got, ref)
    finally:
        L.libxsmm_amd_set_mfma(old)


# ---- the reference's own GEMM test table (tests/gemm.c:75-82: 36 cases incl. empty dimensions, n = 13824 / 65792, k = 1742,
# leading dimensions up to 9216; transposes NN, NT, TN, TT as in :84-88 with the dimension folding of :150-161) --------------------
GEMM_C_M = [0, 1, 0, 0, 1, 1, 2, 3, 3, 1, 8, 64, 64, 16, 80, 80, 80, 80, 16, 260, 260, 260, 260, 350, 350, 350, 350, 350, 5, 10, 12, 20, 32, 9, 13, 5]
GEMM_C_N = [0, 0, 1, 0, 1, 2, 2, 3, 1, 3, 1, 8, 239, 13824, 1, 3, 5, 7, 65792, 1, 3, 5, 7, 16, 1, 25, 4, 9, 13, 1, 10, 6, 33, 9, 13, 5]
GEMM_C_K = [0, 0, 0, 1, 1, 2, 2, 3, 2, 2, 0, 64, 64, 16, 1, 3, 6, 10, 16, 1, 3, 6, 10, 20, 1, 35, 4, 10, 70, 1, 12, 6, 192, 1742, 13, 5]
GEMM_C_LDA = [1, 1, 1, 1, 1, 1, 2, 3, 3, 1, 8, 64, 64, 16, 80, 80, 80, 80, 16, 260, 260, 260, 260, 350, 350, 350, 350, 350, 5, 22, 22, 22, 32, 9, 13, 5]
GEMM_C_LDB = [1, 1, 1, 1, 1, 2, 2, 3, 2, 2, 8, 9216, 240, 16, 1, 3, 5, 5, 16, 1, 3, 5, 7, 35, 35, 35, 35, 35, 70, 1, 20, 8, 2048, 1742, 13, 5]
GEMM_C_LDC = [1, 1, 1, 1, 1, 1, 2, 3, 3, 1, 8, 4096, 240, 16, 80, 80, 80, 80, 16, 260, 260, 260, 260, 350, 350, 350, 350, 350, 5, 22, 12, 20, 2048, 9, 13, 5]
GEMM_C_BETA = [0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 1, 0, 1, 1]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_reference_gemm_table(xs, orc, torch_gpu, dtype):
    """Every case of the reference's tests/gemm.c through libxsmm_?gemm on device memory: inputs by LIBXSMM_MATINIT (seeds 42 / 24
    over the flat maximum-size buffers, :137-138), C all-ones bytes (NaN) for beta = 0 and zero otherwise (:162-175), all four
    transpose pairs. The reference compares with a BLAS gold (absent here): the gold is numpy in float64."""
    torch = torch_gpu
    L = xs.lib()
    T = len(GEMM_C_M)
    lda = [max(GEMM_C_LDA[i], GEMM_C_M[i]) for i in range(T)]; ldb = [max(GEMM_C_LDB[i], GEMM_C_K[i]) for i in range(T)]
    ldc = [max(GEMM_C_LDC[i], GEMM_C_M[i]) for i in range(T)]
    size_a = max(lda[i] * GEMM_C_K[i] for i in range(T)); size_b = max(ldb[i] * GEMM_C_N[i] for i in range(T))
    size_c = max(ldc[i] * GEMM_C_N[i] for i in range(T))
    a = orc.matinit(42, size_a, 1, size_a, 1.0, dtype); b = orc.matinit(24, size_b, 1, size_b, 1.0, dtype)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    gemm = L.libxsmm_dgemm if dtype == np.float64 else L.libxsmm_sgemm
    ct = C.c_double if dtype == np.float64 else C.c_float
    eps = np.finfo(dtype).eps
    for i in range(T):
        for ta, tb in ("NN", "NT", "TN", "TT"):
            mi, ni, ki = GEMM_C_M[i], GEMM_C_N[i], GEMM_C_K[i]
            if ta != "N" and tb == "N":
                mi = ki = min(mi, ki)
            elif ta == "N" and tb != "N":
                ki = ni = min(ki, ni)
            elif ta != "N" and tb != "N":
                mi = ni = ki = min(mi, ni, ki)
            beta = float(GEMM_C_BETA[i])
            c = np.zeros(size_c, dtype=dtype)
            if beta == 0.0:
                c.view(np.uint8)[:] = 0xFF
            dc = torch.from_numpy(c).cuda()
            im, i_n, ik, ila, ilb, ilc = (C.c_int(v) for v in (mi, ni, ki, lda[i], ldb[i], ldc[i]))
            al, be = ct(1.0), ct(beta)
            gemm(ta.encode(), tb.encode(), C.byref(im), C.byref(i_n), C.byref(ik), C.byref(al), xs.dptr(da), C.byref(ila), xs.dptr(db), C.byref(ilb),
                 C.byref(be), xs.dptr(dc), C.byref(ilc))
            torch.cuda.synchronize()
            out = dc.cpu().numpy()
            if mi == 0 or ni == 0:
                assert np.array_equal(out.view(np.uint8), c.view(np.uint8)), (i, ta, tb)  # nothing to do, nothing touched
                continue
            A = (a[:lda[i] * ki].reshape(ki, lda[i])[:, :mi].T if ta == "N" else a[:lda[i] * mi].reshape(mi, lda[i])[:, :ki]).astype(np.float64)
            B = (b[:ldb[i] * ni].reshape(ni, ldb[i])[:, :ki].T if tb == "N" else b[:ldb[i] * ki].reshape(ki, ldb[i])[:, :ni]).astype(np.float64)
            gold = A @ B  # (beta = 1 starts from zeros)
            got = out[:ldc[i] * ni].reshape(ni, ldc[i])[:, :mi].T.astype(np.float64)
            scale = max(np.max(np.abs(A), initial=0.0) * np.max(np.abs(B), initial=0.0) * max(ki, 1), 1e-300)
            assert np.max(np.abs(got - gold), initial=0.0) <= 4 * eps * scale, (i, ta, tb, mi, ni, ki)
            pad = out[:ldc[i] * ni].reshape(ni, ldc[i])[:, mi:]
            assert np.array_equal(pad.view(np.uint8), c[:ldc[i] * ni].reshape(ni, ldc[i])[:, mi:].view(np.uint8)), (i, ta, tb)  # padding rows untouched
