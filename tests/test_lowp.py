"""Low-precision dense kernels: libxsmm_wimmdispatch (i16 -> i32), wsmmdispatch (i16 -> f32, scaled), bsmmdispatch
(bf16 -> f32), bmmdispatch (bf16 -> bf16) -- reference src/libxsmm_main.c:2198-2259.

The reference's own check for these kernels is the gold loop of its harness samples/xgemm/kernel.c (:915-927, :1007-1021,
:1104-1123, :1207-1229: A in pairs of k, terms in ascending k, int sums wrap, float forms round product and add
separately, a bf16 result is the upper half of the float sum). The oracle restates those loops in C; here they are
restated once more in numpy (independently) to pin the oracle, and the GPU kernels are compared with the oracle bit for bit.

PARITY UNPINNED beyond those gold loops: the reference tree holds no input/output vector of its low-precision JIT kernels
(its harness compares kernel and gold loop at run time, with a tolerance for the float kinds). What these tests guarantee is
the gold loops' arithmetic bit for bit; what the reference's AVX-512 JIT rounds differently from its own gold loop (e.g.
vdpbf16ps on Cooper Lake, vpmaddwd pairs for i16) is not covered by any fixture.
"""
import ctypes as C

import numpy as np
import pytest


def _bf16(x):
    """float32 array -> its bf16 truncation as uint16 (what the harness stores: the upper half of the float)"""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def _f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def _inputs(kind, m, n, k, lda, ldb, ldc, seed):
    rng = np.random.default_rng(seed)
    if kind in (0, 1):
        a = rng.integers(-300, 300, lda * k).astype(np.int16).view(np.uint16)
        b = rng.integers(-300, 300, ldb * n).astype(np.int16).view(np.uint16)
    else:
        a = _bf16(rng.uniform(-1, 1, lda * k)); b = _bf16(rng.uniform(-1, 1, ldb * n))
    if kind == 0:
        c = rng.integers(-1000, 1000, ldc * n).astype(np.int32)
    elif kind == 3:
        c = _bf16(rng.uniform(-1, 1, ldc * n))
    else:
        c = rng.uniform(-1, 1, ldc * n).astype(np.float32)
    return a, b, c


def _numpy_gold(kind, beta0, m, n, k, lda, ldb, ldc, a, b, c, scf):
    """kernel.c's gold loops, vectorised over the C tile: one term after the other in ascending k, every step rounded to
    float32 (numpy float32 arithmetic rounds each operation)."""
    out = c.copy()
    A = a.reshape(k // 2, lda, 2)   # a[(s*lda + i)*2 + k2]
    B = b.reshape(n, ldb)           # b[j*ldb + kk]
    C2 = out.reshape(n, ldc)
    if kind == 0:
        acc = np.zeros((n, m), dtype=np.int64) if beta0 else C2[:, :m].astype(np.int64)
        for kk in range(k):
            acc += np.outer(B[:, kk].view(np.int16).astype(np.int64), A[kk // 2, :m, kk % 2].view(np.int16).astype(np.int64))
        C2[:, :m] = (acc & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        return out
    if kind == 3:
        acc = np.zeros((n, m), dtype=np.float32) if beta0 else _f32(C2[:, :m])
    else:
        acc = np.zeros((n, m), dtype=np.float32) if beta0 else C2[:, :m].astype(np.float32)
    for kk in range(k):
        if kind == 1:
            iprod = np.outer(B[:, kk].view(np.int16).astype(np.int32), A[kk // 2, :m, kk % 2].view(np.int16).astype(np.int32))
            term = (iprod.astype(np.float32) * np.float32(scf)).astype(np.float32)
        else:
            term = np.outer(_f32(B[:, kk]), _f32(A[kk // 2, :m, kk % 2])).astype(np.float32)
        acc = (acc + term).astype(np.float32)
    C2[:, :m] = _bf16(acc) if kind == 3 else acc
    return out


CASES = [(16, 9, 8, 16, 8, 16), (32, 32, 32, 32, 32, 32), (16, 5, 6, 20, 10, 24), (48, 7, 64, 48, 64, 48)]


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("case", CASES)
def test_oracle_low_precision_gold_loops(orc, kind, case):
    m, n, k, lda, ldb, ldc = case
    for beta0 in (0, 1):
        a, b, c = _inputs(kind, m, n, k, lda, ldb, ldc, 17 * kind + m + n)
        scf = 0.0123 if kind == 1 else 1.0
        ref = c.copy()
        assert 0 == orc.gemm_lowp(kind, beta0, m, n, k, lda, ldb, ldc, a, b, ref, scf)
        gold = _numpy_gold(kind, beta0, m, n, k, lda, ldb, ldc, a, b, c, scf)
        assert np.array_equal(ref.view(np.uint8), gold.view(np.uint8))
    assert 0 != orc.gemm_lowp(kind, 0, m, n, k + 1, lda, ldb + 2, ldc, a, b, c, 1.0)  # odd k


def test_low_precision_dispatch_rules(xs):
    """k even, no TRANS_B, bf16 output needs m % 16 == 0 (src/generator_gemm.c:121-147,236-243); otherwise NULL."""
    L = xs.lib()
    for name in ("libxsmm_wimmdispatch", "libxsmm_wsmmdispatch", "libxsmm_bsmmdispatch", "libxsmm_bmmdispatch"):
        f = getattr(L, name)
        assert f(16, 8, 8, None, None, None, None, None, None, None)
        assert f(16, 8, 8, None, None, None, None, None, None, None) == f(16, 8, 8, None, None, None, None, None, None, None)
        assert not f(16, 8, 7, None, None, None, None, None, None, None)
        assert not f(16, 8, 8, None, None, None, None, None, C.byref(C.c_int(xs.FLAG_TRANS_B)), None)
    assert L.libxsmm_bsmmdispatch(13, 8, 8, None, None, None, None, None, None, None)
    assert not L.libxsmm_bmmdispatch(13, 8, 8, None, None, None, None, None, None, None)
    two = C.c_float(2.0)
    assert not L.libxsmm_bsmmdispatch(16, 8, 8, None, None, None, C.byref(two), None, None, None)  # alpha != 1


DISPATCH = {0: "libxsmm_wimmdispatch", 1: "libxsmm_wsmmdispatch", 2: "libxsmm_bsmmdispatch", 3: "libxsmm_bmmdispatch"}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("case", CASES + [(64, 64, 256, 64, 256, 64)])
def test_low_precision_kernels_match_the_gold_loops(xs, orc, torch_gpu, kind, case):
    """A dispatched kernel called like the harness does (samples/xgemm/kernel.c:262: kernel(a, b, c, NULL, NULL, NULL[, &scf]))
    on plain host memory (staged) and on device memory; beta = 1 and beta = 0 (C not read: NaN-safe).
    Parity unpinned: no reference-held vector for these kernels -- the gold loop (samples/xgemm/kernel.c) is the bar."""
    torch = torch_gpu
    L = xs.lib()
    m, n, k, lda, ldb, ldc = case
    scf = C.c_float(0.0123 if kind == 1 else 1.0)
    for beta0 in (0, 1):
        a, b, c = _inputs(kind, m, n, k, lda, ldb, ldc, 5 * kind + m + k)
        ref = c.copy()
        assert 0 == orc.gemm_lowp(kind, beta0, m, n, k, lda, ldb, ldc, a, b, ref, scf.value)
        if beta0 and kind in (1, 2):
            c[:] = np.nan
        ilda, ildb, ildc = (C.c_int(v) for v in (lda, ldb, ldc))
        one, zero, ione, izero = C.c_float(1.0), C.c_float(0.0), C.c_int(1), C.c_int(0)
        alpha = C.byref(ione if kind == 0 else one)
        beta = C.byref((izero if beta0 else ione) if kind == 0 else (zero if beta0 else one))
        fn = getattr(L, DISPATCH[kind])(m, n, k, C.byref(ilda), C.byref(ildb), C.byref(ildc), alpha, beta, None, None)
        assert fn
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
        # (1) plain host memory
        out = c.copy()
        proto(fn)(a.ctypes.data, b.ctypes.data, out.ctypes.data, None, None, None, C.addressof(scf))
        assert xs.last_kernel().endswith("_lowp"), xs.last_kernel()
        cols = np.arange(n * ldc).reshape(n, ldc)[:, :m].ravel()  # the m x n tile inside ldc
        assert np.array_equal(out[cols].view(np.uint8), ref[cols].view(np.uint8))
        # (2) device memory
        da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b))
        dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
        proto(fn)(da.data_ptr(), db.data_ptr(), dc.data_ptr(), None, None, None, C.addressof(scf))
        torch.cuda.synchronize()
        got = dc.cpu().numpy()
        got = got.view(np.uint16) if kind == 3 else got
        assert np.array_equal(got[cols].view(np.uint8), ref[cols].view(np.uint8))
        pad = np.setdiff1d(np.arange(n * ldc), cols)
        assert np.array_equal(got[pad].view(np.uint8), c[pad].view(np.uint8))  # padding rows of C are not touched


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 2, 3])
def test_low_precision_batches(xs, orc, torch_gpu, kind):
    """libxsmm_mmbatch_kernel with index arrays and with arrays of pointers (src/libxsmm_gemm.c:1333-1364, :1426-1461):
    shuffled operands, every item its own C. Parity unpinned beyond the gold loops (no reference-held vector)."""
    torch = torch_gpu
    L = xs.lib()
    m, n, k, batch = 16, 12, 24, 301
    rng = np.random.default_rng(kind)
    if kind == 0:
        a = rng.integers(-300, 300, batch * m * k).astype(np.int16).view(np.uint16); b = rng.integers(-300, 300, batch * k * n).astype(np.int16).view(np.uint16)
        c = rng.integers(-1000, 1000, batch * m * n).astype(np.int32)
    else:
        a = _bf16(rng.uniform(-1, 1, batch * m * k)); b = _bf16(rng.uniform(-1, 1, batch * k * n))
        c = rng.uniform(-1, 1, batch * m * n).astype(np.float32) if kind == 2 else _bf16(rng.uniform(-1, 1, batch * m * n))
    pa, pb = rng.permutation(batch), rng.permutation(batch)
    ref = c.copy()
    for i in range(batch):
        ci = ref[i * m * n:(i + 1) * m * n]
        assert 0 == orc.gemm_lowp(kind, 0, m, n, k, m, k, m, a[pa[i] * m * k:(pa[i] + 1) * m * k], b[pb[i] * k * n:(pb[i] + 1) * k * n], ci, 1.0)
    fn = getattr(L, DISPATCH[kind])(m, n, k, None, None, None, None, None, None, None)
    assert fn
    da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b))
    csize = 2 if kind == 3 else 4
    import os
    kern = C.c_void_p(fn)
    sa = (pa * m * k + 1).astype(np.int32); sb = (pb * k * n + 1).astype(np.int32); sc = (np.arange(batch) * m * n + 1).astype(np.int32)
    ptrsize = np.array([8], dtype=np.int32)
    for forced in (False, True):  # forced: the hiprtc-specialised streaming form, one k pair per access (the product takes it from 16 items on)
        old_min = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
        os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1" if forced else "100000"
        try:
            # index arrays (in elements of the operand's type, as the reference counts them), index_base 1
            dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
            rc = L.libxsmm_mmbatch_kernel(kern, 1, 4, xs.dptr(sa), xs.dptr(sb), xs.dptr(sc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, 0, 1, 2, csize, 0)
            assert rc == 0
            torch.cuda.synchronize()
            assert ("_jit_shape_lowp" in xs.last_kernel()) == forced, xs.last_kernel()
            got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), forced
            # arrays of pointers (device arrays)
            dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
            qa = torch.from_numpy((da.data_ptr() + pa.astype(np.int64) * m * k * 2).astype(np.int64)).cuda()
            qb = torch.from_numpy((db.data_ptr() + pb.astype(np.int64) * k * n * 2).astype(np.int64)).cuda()
            qc = torch.from_numpy((dc.data_ptr() + np.arange(batch, dtype=np.int64) * m * n * csize).astype(np.int64)).cuda()
            rc = L.libxsmm_mmbatch_kernel(kern, 0, 0, xs.dptr(ptrsize), xs.dptr(ptrsize), xs.dptr(ptrsize), qa.data_ptr(), qb.data_ptr(), qc.data_ptr(), batch, 0, 1, 2, csize, 0)
            assert rc == 0
            torch.cuda.synchronize()
            assert ("_jit_shape_lowp" in xs.last_kernel()) == forced, xs.last_kernel()
            got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), forced
        finally:
            if old_min is None:
                os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
            else:
                os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_min
    # contiguous items: libxsmm_amd_gemm_batch_strided on a low-precision descriptor (strides in elements of each operand's type)
    dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
    ident = (np.arange(batch) * m * k).astype(np.int64)  # (reference values for unpermuted operands)
    ref2 = c.copy()
    for i in range(batch):
        assert 0 == orc.gemm_lowp(kind, 0, m, n, k, m, k, m, a[i * m * k:(i + 1) * m * k], b[i * k * n:(i + 1) * k * n], ref2[i * m * n:(i + 1) * m * n], 1.0)
    blob = xs.DescriptorBlob()
    L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
    L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
    ip, op = {0: (xs.I16, xs.I32), 2: (xs.BF16, xs.F32), 3: (xs.BF16, xs.BF16)}[kind]
    desc = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), ip, op, m, n, k, m, k, m, 1.0, 1.0, 0, 0)
    assert desc and len(ident) == batch
    import os
    for forced in (False, True):  # forced: the hiprtc-specialised streaming form (normally for batches >= 16384)
        old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
        if forced:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
        try:
            dc.copy_(torch.from_numpy(c.view(np.int16) if kind == 3 else c))
            assert 0 == L.libxsmm_amd_gemm_batch_strided(C.c_void_p(desc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), m * k, k * n, m * n, batch)
            torch.cuda.synchronize()
        finally:
            if old_env is None:
                os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
            else:
                os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env
        assert xs.last_kernel().endswith("_lowp")
        assert ("_jit_shape_lowp" in xs.last_kernel()) == forced, xs.last_kernel()
        got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
        assert np.array_equal(got.view(np.uint8), ref2.view(np.uint8)), (kind, forced)
    assert xs.last_kernel().endswith("_lowp")
    got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
    assert np.array_equal(got.view(np.uint8), ref2.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 2, 3])
@pytest.mark.parametrize("shape", [(16, 16, 16), (8, 12, 6), (16, 8, 32)])
def test_small_low_precision_items_several_per_wave(xs, orc, torch_gpu, kind, shape):
    """Items laid out back to back (libxsmm_amd_gemm_batch_strided on a low-precision descriptor) of a few hundred bytes to 2 KB: the
    streaming form takes 2, 4 or 8 of them per wave and pass (XSMM_SMMJIT_LOWP_PACK; by default as many as make up 8-16 KB), the items
    that are left over one at a time -- bit-equal to the gold loops (samples/xgemm/kernel.c:915-927,1007-1021) whatever the packing.
    Parity unpinned beyond the gold loops (no reference-held vector)."""
    import os
    torch = torch_gpu
    L = xs.lib()
    m, n, k = shape
    batch = 37
    rng = np.random.default_rng(kind * 100 + m)
    if kind == 0:
        a = rng.integers(-300, 300, batch * m * k).astype(np.int16).view(np.uint16); b = rng.integers(-300, 300, batch * k * n).astype(np.int16).view(np.uint16)
        c = rng.integers(-1000, 1000, batch * m * n).astype(np.int32)
    else:
        a = _bf16(rng.uniform(-1, 1, batch * m * k)); b = _bf16(rng.uniform(-1, 1, batch * k * n))
        c = rng.uniform(-1, 1, batch * m * n).astype(np.float32) if kind == 2 else _bf16(rng.uniform(-1, 1, batch * m * n))
    if kind == 3 and 0 != m % 16:
        pytest.skip("a bf16 result needs m % 16 == 0")
    ref = c.copy()
    for i in range(batch):
        assert 0 == orc.gemm_lowp(kind, 0, m, n, k, m, k, m, a[i * m * k:(i + 1) * m * k], b[i * k * n:(i + 1) * k * n], ref[i * m * n:(i + 1) * m * n], 1.0)
    blob = xs.DescriptorBlob()
    L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
    L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
    ip, op = {0: (xs.I16, xs.I32), 2: (xs.BF16, xs.F32), 3: (xs.BF16, xs.BF16)}[kind]
    desc = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), ip, op, m, n, k, m, k, m, 1.0, 1.0, 0, 0)
    assert desc
    da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b))
    old = {key: os.environ.get(key) for key in ("LIBXSMM_AMD_JIT_MINBATCH", "XSMM_SMMJIT_LOWP_PACK")}
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    try:
        for pack in (0, 1, 2, 4, 8):
            os.environ["XSMM_SMMJIT_LOWP_PACK"] = str(pack)
            dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
            assert 0 == L.libxsmm_amd_gemm_batch_strided(C.c_void_p(desc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), m * k, k * n, m * n, batch)
            torch.cuda.synchronize()
            assert "_jit_shape_lowp" in xs.last_kernel(), xs.last_kernel()
            got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), pack
    finally:
        for key, val in old.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 3])
@pytest.mark.parametrize("shape", [(48, 48, 48), (64, 40, 56), (64, 64, 64), (16, 64, 8), (40, 33, 16), (45, 40, 26), (48, 50, 12)])
def test_low_precision_wave_kernel(xs, orc, torch_gpu, kind, shape):
    """bf16 inputs beyond 32 on the one-wave-per-item matrix-core kernel (csrc/xsmm_jit_smm.cpp, SMM_JIT_MFMA_WAVE_BODY with
    XLOWP): the fp32 instruction on the widened operands is the gold loop's product-then-add (samples/xgemm/kernel.c:1104-1123,
    1207-1229) bit for bit; a bf16 result is truncated once. Strided batches smaller and larger than the resident grid.
    Parity unpinned beyond the gold loops (no reference-held vector)."""
    import os
    torch = torch_gpu
    L = xs.lib()
    m, n, k = shape
    if kind == 3 and m % 16:
        pytest.skip("a bf16 result needs m % 16 == 0 (reference rule)")
    blob = xs.DescriptorBlob()
    L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
    L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
    ip, op = {2: (xs.BF16, xs.F32), 3: (xs.BF16, xs.BF16)}[kind]
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old_mfma = L.libxsmm_amd_set_mfma(1)
    try:
        for beta0 in (0, 1):
            desc = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), ip, op, m, n, k, m, k, m, 1.0, 0.0 if beta0 else 1.0, 0, 0)
            assert desc
            for batch in (1, 5, 2500):
                rng = np.random.default_rng(11 * batch + m + k + kind)
                a = _bf16(rng.uniform(-1, 1, batch * m * k)); b = _bf16(rng.uniform(-1, 1, batch * k * n))
                c = rng.uniform(-1, 1, batch * m * n).astype(np.float32) if kind == 2 else _bf16(rng.uniform(-1, 1, batch * m * n))
                ref = c.copy()
                for i in range(batch):
                    assert 0 == orc.gemm_lowp(kind, beta0, m, n, k, m, k, m, a[i * m * k:(i + 1) * m * k], b[i * k * n:(i + 1) * k * n], ref[i * m * n:(i + 1) * m * n], 1.0)
                if beta0 and kind == 2:
                    c[:] = np.nan
                da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b))
                dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
                assert 0 == L.libxsmm_amd_gemm_batch_strided(C.c_void_p(desc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), m * k, k * n, m * n, batch)
                torch.cuda.synchronize()
                wave = (max(m, n) > 32 and k % 8 == 0 and m % (8 if kind == 3 else 4) == 0)
                assert xs.last_kernel() == (("smm_bf16_mfma_wave_jit_lowp" if kind == 3 else "smm_bf16f32_mfma_wave_jit_lowp") if wave else xs.last_kernel()), xs.last_kernel()
                assert xs.last_kernel().endswith("_lowp")
                got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
                assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (beta0, batch)
    finally:
        L.libxsmm_amd_set_mfma(old_mfma)
        if old_env is None:
            os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 3])
@pytest.mark.parametrize("shape", [(32, 32, 32), (16, 12, 24), (64, 48, 64)])
def test_low_precision_batch_reduce(xs, orc, torch_gpu, kind, shape):
    """libxsmm_bsmmdispatch_reducebatch / libxsmm_bmmdispatch_reducebatch (src/libxsmm_main.c:2290-2315): kernel(a[], b[], c, &count),
    C (+)= sum_i A_i B_i with the sums kept in fp32 across the batch -- the chain of `count` calls of the bf16 -> f32 gold loop
    (samples/xgemm/kernel.c:1104-1123); a bf16 C is widened once and truncated once. No reference-held vector covers these
    kernels (they serve the RNN module): parity unpinned beyond that chain. Pointer arrays on the host and on the device."""
    torch = torch_gpu
    L = xs.lib()
    m, n, k = shape
    for name in ("libxsmm_bsmmdispatch_reducebatch", "libxsmm_bmmdispatch_reducebatch"):
        f = getattr(L, name); f.restype = C.c_void_p
        f.argtypes = [C.c_int] * 3 + [C.c_void_p] * 7
    disp = L.libxsmm_bsmmdispatch_reducebatch if kind == 2 else L.libxsmm_bmmdispatch_reducebatch
    rng = np.random.default_rng(3 * kind + m)
    for beta0, cnt in ((0, 1), (0, 7), (1, 5)):
        beta = C.c_float(0.0 if beta0 else 1.0)
        fn = disp(m, n, k, None, None, None, None, C.addressof(beta), None, None)
        assert fn
        As = [_bf16(rng.uniform(-1, 1, m * k)) for _ in range(cnt)]; Bs = [_bf16(rng.uniform(-1, 1, k * n)) for _ in range(cnt)]
        c = rng.uniform(-1, 1, m * n).astype(np.float32) if kind == 2 else _bf16(rng.uniform(-1, 1, m * n))
        chain = (np.zeros(m * n, dtype=np.float32) if beta0 else (c.copy() if kind == 2 else _f32(c)))  # fp32 sums across the batch
        for i in range(cnt):
            assert 0 == orc.gemm_lowp(2, 0, m, n, k, m, k, m, As[i], Bs[i], chain, 1.0)
        ref = chain if kind == 2 else _bf16(chain)
        if beta0 and kind == 2:
            c[:] = np.nan
        dA = [torch.from_numpy(x.view(np.int16)).cuda() for x in As]; dB = [torch.from_numpy(x.view(np.int16)).cuda() for x in Bs]
        pa = np.array([t.data_ptr() for t in dA], dtype=np.uint64); pb = np.array([t.data_ptr() for t in dB], dtype=np.uint64)
        count = np.array([cnt], dtype=np.uint64)
        for where in ("host", "device"):
            dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
            if where == "host":
                xs.call_kernel(fn, pa, pb, dc, count)
            else:
                qa = torch.from_numpy(pa.view(np.int64)).cuda(); qb = torch.from_numpy(pb.view(np.int64)).cuda()
                xs.call_kernel(fn, qa, qb, dc, count)
            torch.cuda.synchronize()
            assert xs.last_kernel() == ("smm_bf16f32_reduce_lowp" if kind == 2 else "smm_bf16_reduce_lowp")
            got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (beta0, cnt, where)
    # the reference's rules still hold: odd k and (for a bf16 result) m % 16 != 0 give NULL; int16 inputs have no batch-reduce form
    assert not disp(m, n, 7, None, None, None, None, None, None, None)
    if kind == 3:
        assert not disp(m + 4, n, k, None, None, None, None, None, None, None)


@pytest.mark.gpu
def test_low_precision_wave_kernel_fuzz(xs, orc, torch_gpu):
    """random shapes of the bf16 matrix-core form (M a multiple of 4 -- of 16 for a bf16 result --, K of 8, any N up to 64), batches
    around the size of the resident grid, beta 0 / 1: the gold loop bit for bit (parity unpinned beyond it: no reference-held vector)"""
    import os
    torch = torch_gpu
    L = xs.lib()
    rng = np.random.default_rng(424242)
    L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
    L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    old_mfma = L.libxsmm_amd_set_mfma(1)
    try:
        for it in range(14):
            kind = 2 if rng.random() < 0.5 else 3
            m = int(rng.integers(1, 5)) * 16 if kind == 3 else int(rng.integers(2, 17)) * 4
            n = int(rng.integers(1, 65)); k = int(rng.integers(1, 9)) * 8
            if max(m, n) < 32:
                n = int(rng.integers(32, 65))
            beta0 = int(rng.random() < 0.4)
            batch = int(rng.choice([1, 3, 257, 1100, 2100]))
            blob = xs.DescriptorBlob()
            ip, op = {2: (xs.BF16, xs.F32), 3: (xs.BF16, xs.BF16)}[kind]
            desc = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), ip, op, m, n, k, m, k, m, 1.0, 0.0 if beta0 else 1.0, 0, 0)
            assert desc, (kind, m, n, k)
            a = _bf16(rng.uniform(-1, 1, batch * m * k)); b = _bf16(rng.uniform(-1, 1, batch * k * n))
            c = rng.uniform(-1, 1, batch * m * n).astype(np.float32) if kind == 2 else _bf16(rng.uniform(-1, 1, batch * m * n))
            ref = c.copy()
            for i in range(batch):
                assert 0 == orc.gemm_lowp(kind, beta0, m, n, k, m, k, m, a[i * m * k:(i + 1) * m * k], b[i * k * n:(i + 1) * k * n], ref[i * m * n:(i + 1) * m * n], 1.0)
            da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b))
            dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
            assert 0 == L.libxsmm_amd_gemm_batch_strided(C.c_void_p(desc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), m * k, k * n, m * n, batch)
            torch.cuda.synchronize()
            got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
            assert xs.last_kernel().endswith("_lowp")
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (it, kind, m, n, k, beta0, batch, xs.last_kernel())
    finally:
        L.libxsmm_amd_set_mfma(old_mfma)
        if old_env is None:
            os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(48, 48, 48), (64, 64, 64), (40, 64, 16), (64, 33, 62)])
def test_i16_streaming_form_beyond_32(xs, orc, torch_gpu, shape):
    """i16 -> i32 beyond 32 x 32 on the specialised streaming kernel (a larger tile per lane, v_dot2_i32_i16 per k pair): the wrapping
    sums of the gold loop (samples/xgemm/kernel.c:915-927), exact in any order; strided batches, beta 1 and 0. (Integer sums: any
    correct implementation gives these bits; still no reference-held vector -- parity unpinned beyond the gold loop.)"""
    import os
    torch = torch_gpu
    L = xs.lib()
    m, n, k = shape
    blob = xs.DescriptorBlob()
    L.libxsmm_gemm_descriptor_dinit2.restype = C.c_void_p
    L.libxsmm_gemm_descriptor_dinit2.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_double, C.c_int, C.c_int]
    old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    try:
        for beta0 in (0, 1):
            desc = L.libxsmm_gemm_descriptor_dinit2(C.byref(blob), xs.I16, xs.I32, m, n, k, m, k, m, 1.0, 0.0 if beta0 else 1.0, 0, 0)
            assert desc
            for batch in (1, 700):
                rng = np.random.default_rng(batch + m + n)
                a = rng.integers(-30000, 30000, batch * m * k).astype(np.int16).view(np.uint16); b = rng.integers(-30000, 30000, batch * k * n).astype(np.int16).view(np.uint16)
                c = rng.integers(-2 ** 31, 2 ** 31 - 1, batch * m * n).astype(np.int32)
                ref = c.copy()
                for i in range(batch):
                    assert 0 == orc.gemm_lowp(0, beta0, m, n, k, m, k, m, a[i * m * k:(i + 1) * m * k], b[i * k * n:(i + 1) * k * n], ref[i * m * n:(i + 1) * m * n], 1.0)
                da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b)); dc = torch.from_numpy(c).cuda()
                assert 0 == L.libxsmm_amd_gemm_batch_strided(C.c_void_p(desc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), m * k, k * n, m * n, batch)
                torch.cuda.synchronize()
                assert xs.last_kernel() == "smm_i16i32_jit_shape_lowp", xs.last_kernel()
                assert np.array_equal(dc.cpu().numpy(), ref), (beta0, batch)
    finally:
        if old_env is None:
            os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_env


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 3])
@pytest.mark.parametrize("shape", [(48, 48, 48), (64, 40, 56), (32, 32, 32)])
def test_low_precision_index_and_pointer_batches_on_the_wave_kernel(xs, orc, torch_gpu, kind, shape):
    """libxsmm_mmbatch_kernel (index arrays in elements of 16 bits, arrays of pointers) with bf16 inputs from 32 x 32 up: the
    one-wave-per-item matrix-core kernel, as for strided batches -- the gold loop bit for bit (parity unpinned beyond it: no
    reference-held vector). Operands shuffled, every item its own C."""
    import os
    torch = torch_gpu
    L = xs.lib()
    m, n, k = shape
    batch = 600
    rng = np.random.default_rng(kind * 7 + m)
    a = _bf16(rng.uniform(-1, 1, batch * m * k)); b = _bf16(rng.uniform(-1, 1, batch * k * n))
    c = rng.uniform(-1, 1, batch * m * n).astype(np.float32) if kind == 2 else _bf16(rng.uniform(-1, 1, batch * m * n))
    pa, pb = rng.permutation(batch), rng.permutation(batch)
    ref = c.copy()
    for i in range(batch):
        assert 0 == orc.gemm_lowp(kind, 0, m, n, k, m, k, m, a[pa[i] * m * k:(pa[i] + 1) * m * k], b[pb[i] * k * n:(pb[i] + 1) * k * n], ref[i * m * n:(i + 1) * m * n], 1.0)
    fn = getattr(L, DISPATCH[kind])(m, n, k, None, None, None, None, None, None, None)
    assert fn
    kern = C.c_void_p(fn)
    da, db = (torch.from_numpy(x.view(np.int16)).cuda() for x in (a, b))
    csize = 2 if kind == 3 else 4
    sa = (pa * m * k).astype(np.int32); sb = (pb * k * n).astype(np.int32); sc = (np.arange(batch) * m * n).astype(np.int32)
    ptrsize = np.array([8], dtype=np.int32)
    old_min = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
    os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
    try:
        dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
        assert 0 == L.libxsmm_mmbatch_kernel(kern, 0, 4, xs.dptr(sa), xs.dptr(sb), xs.dptr(sc), da.data_ptr(), db.data_ptr(), dc.data_ptr(), batch, 0, 1, 2, csize, 0)
        torch.cuda.synchronize()
        assert "_mfma_wave_jit_lowp" in xs.last_kernel(), xs.last_kernel()
        got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        dc = torch.from_numpy(c.view(np.int16) if kind == 3 else c).cuda()
        qa = torch.from_numpy((da.data_ptr() + pa.astype(np.int64) * m * k * 2).astype(np.int64)).cuda()
        qb = torch.from_numpy((db.data_ptr() + pb.astype(np.int64) * k * n * 2).astype(np.int64)).cuda()
        qc = torch.from_numpy((dc.data_ptr() + np.arange(batch, dtype=np.int64) * m * n * csize).astype(np.int64)).cuda()
        assert 0 == L.libxsmm_mmbatch_kernel(kern, 0, 0, xs.dptr(ptrsize), xs.dptr(ptrsize), xs.dptr(ptrsize), qa.data_ptr(), qb.data_ptr(), qc.data_ptr(), batch, 0, 1, 2, csize, 0)
        torch.cuda.synchronize()
        assert "_mfma_wave_jit_lowp" in xs.last_kernel(), xs.last_kernel()
        got = dc.cpu().numpy(); got = got.view(np.uint16) if kind == 3 else got
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
    finally:
        if old_min is None:
            os.environ.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = old_min
