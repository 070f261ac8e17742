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
        for (m, n, k) in [(64, 64, 64), (48, 33, 200)]:  # work-group-per-item form (variant bit 16)
            for beta, flags in ((1.0, 0), (0.0, xs.FLAG_TRANS_B)):
                blob, d = xs.descriptor(prec, m, n, k, beta=beta, flags=flags)
                rc = L.libxsmm_amd_smm_kernel_source(d, 16, buf, len(buf), 1)
                if rc == -1:
                    pytest.skip("libhiprtc is not available here")
                assert rc == 0 and "#define XKC" in buf.value.decode()
        for (m, n, k) in [(23, 23, 23), (13, 23, 32), (32, 32, 64), (1, 1, 1)]:
            for beta, flags in ((1.0, 0), (0.0, 0), (1.0, xs.FLAG_TRANS_B)):
                blob, d = xs.descriptor(prec, m, n, k, beta=beta, flags=flags)
                # wide / element-wide / + wave-per-run / + work-group-per-run / wave-per-run that leaves long runs to the latter /
                # the relaxed-order twins (bit 32: few long runs are cut into segments that join C with atomics)
                for variant in ((0, 1, 3, 5, 11, 35, 37, 43) if beta == 1.0 else (0, 1)):
                    rc = L.libxsmm_amd_smm_kernel_source(d, variant, buf, len(buf), 1)
                    if rc == -1:
                        pytest.skip("libhiprtc is not available here")
                    assert rc == 0, (prec, m, n, k, beta, flags, variant)
                    src = buf.value.decode()
                    assert "#define XM %d" % m in src and "xsmm_smm_op" in src
                    assert "#define XRUNS %d" % (2 if variant & 4 else (variant >> 1) & 1) in src
    # matrix-core kernel with one wave per item (variant bit 16384): tight shapes beyond 32 with M and K multiples of four
    for prec, shapes in ((xs.F32, [(40, 40, 40), (48, 48, 48), (56, 56, 56), (36, 64, 8), (64, 20, 12)]), (xs.F64, [(40, 40, 40), (48, 48, 48), (34, 40, 4)])):
        for (m, n, k) in shapes:
            for beta in (1.0, 0.0):
                blob, d = xs.descriptor(prec, m, n, k, beta=beta)
                assert 0 == L.libxsmm_amd_smm_kernel_source(d, 16384, buf, len(buf), 1), (prec, m, n, k, beta)
                assert "xmfma" in buf.value.decode() and "#define XWPE" in buf.value.decode()
    # ... element by element (variant bits 16384 | 1): any shape, K padded to a multiple of four
    for prec in (xs.F32, xs.F64):
        for (m, n, k) in [(33, 33, 33), (45, 37, 19), (64, 5, 7), (40, 40, 40), (57, 39, 1)]:
            for beta in (1.0, 0.0):
                blob, d = xs.descriptor(prec, m, n, k, beta=beta)
                assert 0 == L.libxsmm_amd_smm_kernel_source(d, 16384 | 1, buf, len(buf), 1), (prec, m, n, k, beta)
                assert "#define XVEC 1\n" in buf.value.decode()
    # ... the fp64 form that works on the columns of C in two halves (variant bit 32768)
    for (m, n, k) in [(56, 56, 56), (16, 64, 64), (64, 56, 56)]:
        for beta in (1.0, 0.0):
            blob, d = xs.descriptor(xs.F64, m, n, k, beta=beta)
            assert 0 == L.libxsmm_amd_smm_kernel_source(d, 32768, buf, len(buf), 1), (m, n, k, beta)
            assert "#define XNSPLIT 2" in buf.value.decode()
    # ... and its bf16-input forms (variant bits 11..12: 2 = bf16 result, 3 = fp32 result)
    for (m, n, k) in [(48, 48, 48), (64, 40, 56), (16, 64, 8)]:
        for beta in (1.0, 0.0):
            blob, d = xs.descriptor(xs.F32, m, n, k, beta=beta)
            for lowp in (2, 3):
                assert 0 == L.libxsmm_amd_smm_kernel_source(d, 16384 | (lowp << 11), buf, len(buf), 1), (m, n, k, beta, lowp)
                assert "#define XLOWP %d\n" % lowp in buf.value.decode() and "widen8" in buf.value.decode()
    # several consecutive items per wave (variant bits 8..10 = log2 of the count): small and oddly sized shapes of tight strided batches
    for prec in (xs.F64, xs.F32):
        for (m, n, k), packs in (((5, 5, 5), (1, 2, 3, 4)), ((8, 8, 8), (1, 2)), ((13, 13, 13), (1, 2)), ((23, 23, 23), (1,)), ((5, 7, 3), (3,)),
                                ((13, 32, 13), (1, 2)), ((23, 32, 13), (1,))):  # (fp64, eight columns per lane: the stride search for B's image must end)
            for beta, flags in ((1.0, 0), (0.0, xs.FLAG_TRANS_B)):
                blob, d = xs.descriptor(prec, m, n, k, beta=beta, flags=flags)
                for lg in packs:
                    assert 0 == L.libxsmm_amd_smm_kernel_source(d, lg << 8, buf, len(buf), 1), (prec, m, n, k, beta, flags, lg)
                    assert "#define XPACK %d\n" % (1 << lg) in buf.value.decode()
    # 16-bit inputs widened on the way into LDS (variant bits 11..12: 1 = i16 -> i32, 2 = bf16 -> bf16, 3 = bf16 -> f32), wide and element-wide
    for (m, n, k) in [(32, 32, 32), (16, 12, 24), (23, 9, 64)]:
        for beta in (1.0, 0.0):
            blob, d = xs.descriptor(xs.F32, m, n, k, beta=beta)
            for variant in ((1 << 11), (2 << 11), (3 << 11), (1 << 11) | 1, (2 << 11) | 1, (3 << 11) | 1):
                assert 0 == L.libxsmm_amd_smm_kernel_source(d, variant, buf, len(buf), 1), (m, n, k, beta, variant)
                assert "#define XLOWP %d\n" % (variant >> 11) in buf.value.decode()
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


class _JitForced:
    """Lets small test batches take the run-time specialised kernels (normally reserved for batches >= 16384)."""
    def __init__(self, xs):
        self.xs = xs
    def __enter__(self):
        self.old_env = os.environ.get("LIBXSMM_AMD_JIT_MINBATCH")
        os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
        self.old = self.xs.lib().libxsmm_amd_set_mfma(0)
    def __exit__(self, *exc):
        self.xs.lib().libxsmm_amd_set_mfma(self.old)
        if self.old_env is None:
            del os.environ["LIBXSMM_AMD_JIT_MINBATCH"]
        else:
            os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = self.old_env


def test_grouped_source_compiles_for_gfx950(xs):
    """the text of a grouped launch (libxsmm_amd_gemm_batch_groups): the run forms of several shapes, each in its own
    namespace, behind one dispatching kernel -- valid gfx950 code (no device needed)"""
    L = xs.lib()
    buf = C.create_string_buffer(1 << 21)
    for prec in (xs.F64, xs.F32):
        shapes = [(13, 13, 13), (23, 13, 32), (32, 32, 32), (5, 7, 3), (13, 13, 13)]  # (a repeated shape shares its body)
        keep, arr = [], (C.c_void_p * len(shapes))()
        for i, (m, n, k) in enumerate(shapes):
            blob, d = xs.descriptor(prec, m, n, k)
            keep.append(blob); arr[i] = C.cast(d, C.c_void_p)
        rc = L.libxsmm_amd_smm_grouped_kernel_source(arr, len(shapes), buf, len(buf), 1)
        if rc == -1:
            pytest.skip("libhiprtc is not available here")
        assert rc == 0
        src = buf.value.decode()
        nbodies = src.count("namespace xg")
        assert "xsmm_smm_grouped" in src and 4 <= nbodies <= 8 and src.count("::xsmm_entry(") == nbodies


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("relaxed", [False, True])
def test_grouped_batches_one_launch(xs, orc, torch_gpu, dtype, relaxed):
    """libxsmm_amd_gemm_batch_groups: CP2K-style stacks of several shapes (samples/cp2k/cp2k.cpp:328-360) in one call. The
    C-ordering check of all groups is one launch and the multiplication another; every group equals libxsmm_gemm_batch on
    its own -- the sequential chain per C block, bit for bit (relaxed: the order of the sums is open, tolerance). Groups with
    runs of different lengths, a group of distinct C blocks, a group with one C (stride_c NULL), an empty group, host index
    arrays, and a group whose C blocks repeat out of order (its sums join C with atomics)."""
    torch = torch_gpu
    L = xs.lib()
    prec = xs.F64 if dtype == np.float64 else xs.F32
    rng = np.random.default_rng(77)
    shapes = [(13, 13, 13), (23, 23, 23), (32, 32, 32), (32, 13, 23), (13, 32, 32), (5, 7, 3), (13, 13, 13), (8, 8, 8)]
    sizes = [900, 700, 640, 811, 500, 333, 0, 600]
    groups, keep = [], []
    for gi, ((m, n, k), s) in enumerate(zip(shapes, sizes)):
        a = rng.uniform(-1, 1, max(s, 1) * m * k).astype(dtype); b = rng.uniform(-1, 1, max(s, 1) * k * n).astype(dtype)
        if gi == 4:       # every product its own C
            cidx = np.arange(s)
        elif gi == 5:     # one C for the whole group
            cidx = None
        elif gi == 7:     # C blocks repeat out of order
            cidx = rng.integers(0, 40, s)
        else:             # runs of ~ u consecutive products per C (cp2k.cpp:155)
            u = max(1, int(np.sqrt(max(s, 1) * 160 / 240)))
            cidx = np.arange(s) // u
        nc = 1 if cidx is None else (int(cidx.max()) + 1 if s else 1)
        c = rng.uniform(-1, 1, nc * m * n).astype(dtype)
        sa = (rng.permutation(max(s, 1))[:s] * m * k).astype(np.int32); sb = (np.arange(s) * k * n).astype(np.int32)
        sc = None if cidx is None else (cidx * m * n).astype(np.int32)
        ref = c.copy()
        if s:
            sc_ref = sc if sc is not None else np.zeros(s, dtype=np.int32)
            assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc_ref, s)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        on_host = (gi % 2 == 1)  # index arrays of every other group stay in host memory
        dsa, dsb = (x if on_host else torch.from_numpy(x).cuda() for x in (sa, sb))
        dsc = None if sc is None else (sc if on_host else torch.from_numpy(sc).cuda())
        groups.append((da, db, dc, dsa, dsb, dsc, ref, gi))
        keep.append((sa, sb, sc))
    with _JitForced(xs):
        launches = L.libxsmm_amd_launch_count()
        rc = xs.gemm_batch_groups(prec, shapes, [g[0] for g in groups], [g[1] for g in groups], [g[2] for g in groups],
                                  [g[3] for g in groups], [g[4] for g in groups], [g[5] for g in groups], sizes, relaxed=relaxed)
        assert rc == 0
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_jit_shape_runs_grouped"), xs.last_kernel()
        assert L.libxsmm_amd_launch_count() == launches + 1  # (note_launch counts the multiplication; the check is not a compute kernel)
    for (da, db, dc, dsa, dsb, dsc, ref, gi) in groups:
        out = dc.cpu().numpy()
        if relaxed or gi == 7:  # segments / out-of-order repeats: atomics, any order
            # two orders of the same sum: rounding errors random-walk, eps * sqrt(terms) with a margin of 4 -- the bound of
            # test_relaxed_order_cuts_long_runs_into_segments (for one product per C far inside north_star's 1e-6 / 1e-12)
            m, n, k = shapes[gi]; s = sizes[gi]
            sc = keep[gi][2]
            longest = s if sc is None else int(np.bincount(sc // (m * n)).max()) if s else 1  # products that meet in one C block
            tol = np.finfo(dtype).eps * np.sqrt(float(longest) * k) * 4
            assert np.max(np.abs(out.astype(np.float64) - ref.astype(np.float64))) <= tol * max(1.0, float(np.max(np.abs(ref)))), (gi, longest)
        else:
            assert np.array_equal(out, ref), gi


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(23, 23, 23), (13, 13, 13), (13, 23, 32), (32, 32, 32), (5, 7, 3)])
def test_jit_index_batches_and_runs_bitexact(xs, orc, torch_gpu, dtype, shape):
    """Index-array batches (src/libxsmm_gemm.c:1333-1364) on the specialised kernels: shuffled operands with distinct C
    (element-wide accesses), then CP2K-style stacks whose consecutive products share a C (samples/cp2k/cp2k.cpp:328-360):
    runs accumulate in registers, in batch order -- the reference's sequential chain, bit for bit."""
    torch = torch_gpu
    m, n, k = shape
    batch = 777
    rng = np.random.default_rng(m * 31 + n * 7 + k)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    with _JitForced(xs):
        # (1) permuted A/B, distinct C, index_base 1
        c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
        sa = (rng.permutation(batch) * m * k + 1).astype(np.int32); sb = (rng.permutation(batch) * k * n + 1).astype(np.int32)
        sc = (np.arange(batch) * m * n + 1).astype(np.int32)
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 1, sa, sb, sc, batch)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 1, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        # how the C blocks repeat is established on the device (no host round trip), so index batches always take the
        # run form of the kernel -- here every item is its own run
        assert xs.last_kernel() == "smm_f%d_jit_shape_runs" % (64 if dtype == np.float64 else 32), xs.last_kernel()
        assert np.array_equal(dc.cpu().numpy(), ref)
        # (2) runs of very different lengths (1 .. >64 so that a run crosses the 64-item scan chunks), some C untouched
        lens = [1, 1, 2, 3, 64, 65, 130, 1, 7, 200, 1, 1, 1, 1, 1]
        lens = np.array(lens + [batch - sum(lens)], dtype=np.int64)
        nc = len(lens)
        owners = np.sort(rng.choice(np.arange(nc + 5), size=nc, replace=False))  # increasing C blocks, with gaps
        cidx = np.repeat(owners, lens)
        assert len(cidx) == batch
        c2 = rng.uniform(-1, 1, (nc + 5) * m * n).astype(dtype)
        sa0 = (np.arange(batch) * m * k).astype(np.int32); sb0 = (np.arange(batch) * k * n).astype(np.int32); sc0 = (cidx * m * n).astype(np.int32)
        ref2 = c2.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref2, 0, sa0, sb0, sc0, batch)
        dc2 = torch.from_numpy(c2).cuda()
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc2, m, 0, 4, sa0, sb0, sc0, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_jit_shape_runs"), xs.last_kernel()
        assert np.array_equal(dc2.cpu().numpy(), ref2)


@pytest.mark.gpu
def test_jit_pointer_batches_with_runs(xs, orc, torch_gpu):
    """Arrays of pointers (src/libxsmm_gemm.c:1426-1461) with repeated consecutive C pointers."""
    torch = torch_gpu
    m, n, k = 23, 13, 32
    batch, nc = 300, 9
    rng = np.random.default_rng(77)
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, nc * m * n)
    cidx = np.sort(rng.integers(0, nc, batch))
    ref = c.copy()
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    pa = (da.data_ptr() + np.arange(batch, dtype=np.uint64) * np.uint64(m * k * 8)).astype(np.uint64)
    pb = (db.data_ptr() + np.arange(batch, dtype=np.uint64) * np.uint64(k * n * 8)).astype(np.uint64)
    pc = (dc.data_ptr() + cidx.astype(np.uint64) * np.uint64(m * n * 8)).astype(np.uint64)
    dpa, dpb, dpc = (torch.from_numpy(x.view(np.int64)).cuda() for x in (pa, pb, pc))
    ptrsize = np.array([8], dtype=np.int32)
    with _JitForced(xs):
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, dpa, m, dpb, k, 1.0, dpc, m, 0, 0, ptrsize, ptrsize, ptrsize, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_jit_shape_runs"), xs.last_kernel()
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(23, 23, 23), (13, 5, 7), (32, 32, 32)])
def test_relaxed_order_cuts_long_runs_into_segments(xs, orc, torch_gpu, dtype, shape):
    """libxsmm_gemm_batch_omp (src/libxsmm_ext_gemm.c:758-972: threads + a lock per C, i.e. no defined order of the sums)
    may cut a batch of few, long runs into segments that join C with atomics; libxsmm_gemm_batch on the same batch keeps
    the sequential chain bit for bit. Run lengths include 1 and lengths that straddle segment borders; the last segment
    is ragged; one C block in the middle is never referenced."""
    torch = torch_gpu
    m, n, k = shape
    rng = np.random.default_rng(m + 3 * n + 5 * k)
    lens = np.array([700, 1, 333, 8, 1201, 64, 15, 999, 1500, 2, 813], dtype=np.int64)
    batch, nc = int(lens.sum()), len(lens) + 1
    owners = np.array([0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11])
    cidx = np.repeat(owners, lens)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    c = rng.uniform(-1, 1, nc * m * n).astype(dtype)
    sa = (rng.permutation(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    da, db = (torch.from_numpy(x).cuda() for x in (a, b))
    with _JitForced(xs):
        dc = torch.from_numpy(c).cuda()
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), ref)  # strict entry point: batch order
        dc = torch.from_numpy(c).cuda()
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch, omp=True)
        torch.cuda.synchronize()
        assert "_jit_shape_runs" in xs.last_kernel(), xs.last_kernel()
    out = dc.cpu().numpy()
    # two orders of the same sum of up to 1500 * k terms: rounding errors random-walk, eps * sqrt(terms) with a margin of 4
    # (for one product per C this is far inside the north_star tolerance of 1e-6 / 1e-12 relative)
    tol = np.finfo(dtype).eps * np.sqrt(float(lens.max()) * k) * 4
    assert np.max(np.abs(out - ref)) <= tol * np.max(np.abs(ref))
    assert not np.array_equal(out, ref) or m * n * k < 1000  # the order really differs (else the segment form did not run)
    assert np.array_equal(out[5 * m * n:6 * m * n], c[5 * m * n:6 * m * n])  # the unreferenced block is untouched


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_jit_stands_down_for_out_of_order_repeats(xs, orc, torch_gpu, dtype):
    """C blocks that repeat out of order cannot be handled run by run: the run kernel reads the device-side verdict and
    returns, the generic kernel launched behind it accumulates with atomics (order of the sums is then free: north_star
    tolerance instead of bit-exactness)."""
    torch = torch_gpu
    m, n, k = 23, 23, 23
    batch, nc = 700, 13
    rng = np.random.default_rng(8)
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    c = rng.uniform(-1, 1, nc * m * n).astype(dtype)
    cidx = rng.integers(0, nc, batch)  # unsorted
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    with _JitForced(xs):
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
    out = dc.cpu().numpy()
    tol = 1e-12 if dtype == np.float64 else 1e-6
    assert np.max(np.abs(out - ref)) <= tol * max(1.0, np.max(np.abs(ref))) * 64


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(64, 64, 64), (48, 40, 56), (33, 64, 17), (64, 33, 100), (40, 8, 5), (23, 23, 70), (8, 8, 200)])
def test_jit_work_group_form_for_shapes_up_to_64(xs, orc, torch_gpu, dtype, shape):
    """32 < M or N <= 64: one work-group per item, K in chunks through LDS (any K): the k-ascending fma chain continues
    across the chunks, so the result is still the oracle's, bit for bit. Strided and index batches, beta 0/1, TRANS_B."""
    torch = torch_gpu
    m, n, k = shape
    batch = 301
    rng = np.random.default_rng(m + 3 * n + 7 * k)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    with _JitForced(xs):
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
            assert xs.last_kernel().endswith("_jit_shape_wg"), xs.last_kernel()
            assert np.array_equal(dc.cpu().numpy(), ref), (shape, beta, flags)
        # index batch with shuffled operands; the caller promises distinct C (negative batchsize)
        a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype); c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
        sa = (rng.permutation(batch) * m * k).astype(np.int32); sb = (rng.permutation(batch) * k * n).astype(np.int32); sc = (rng.permutation(batch) * m * n).astype(np.int32)
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, -batch)
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_jit_shape_wg"), xs.last_kernel()
        assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.gpu
def test_batch_calls_are_graph_capturable(xs, orc, torch_gpu):
    """A batch call makes no host round trip (the C-ordering verdict stays on the device), so a sequence of calls can be
    captured in a HIP graph and replayed: CP2K-style stacks (runs of equal C) of two shapes, replayed twice."""
    torch = torch_gpu
    rng = np.random.default_rng(21)
    groups = []
    for (m, n, k, batch, nc) in ((23, 23, 23, 500, 9), (13, 32, 23, 400, 400)):
        a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, nc * m * n)
        cidx = np.sort(rng.integers(0, nc, batch)) if nc < batch else np.arange(batch)
        sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
        dev = [torch.from_numpy(x).cuda() for x in (a, b, c, sa, sb, sc)]
        groups.append((m, n, k, batch, a, b, c, sa, sb, sc, dev))

    def calls():
        for (m, n, k, batch, a, b, c, sa, sb, sc, dev) in groups:
            xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, dev[0], m, dev[1], k, 1.0, dev[2], m, 0, 4, dev[3], dev[4], dev[5], batch)

    L = xs.lib()
    with _JitForced(xs):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        old_stream = L.libxsmm_amd_get_stream()
        try:
            with torch.cuda.stream(side):
                L.libxsmm_amd_set_stream(C.c_void_p(side.cuda_stream))
                calls()  # warm-up outside the capture: kernels get compiled and loaded, the flag ring is allocated
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    calls()
                graph.replay(); graph.replay()
                torch.cuda.synchronize()
        finally:
            L.libxsmm_amd_set_stream(C.c_void_p(old_stream))
    for (m, n, k, batch, a, b, c, sa, sb, sc, dev) in groups:
        ref = c.copy()
        for _ in range(3):  # warm-up + two replays (capturing itself does not execute)
            assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
        assert np.array_equal(dev[2].cpu().numpy(), ref)


@pytest.mark.gpu
def test_code_objects_are_cached_on_disk(xs, torch_gpu, tmp_path):
    """hiprtc output is kept under $LIBXSMM_AMD_CACHE: a second process loads the code objects instead of compiling (same
    results); LIBXSMM_AMD_CACHE=0 writes nothing."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import importlib, sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "xs = importlib.import_module('libxsmm-1_amd'); L = xs.lib(); L.libxsmm_amd_set_mfma(0)\n"
        "m, n, k, batch = 13, 9, 11, 20000\n"
        "g = torch.Generator(device='cuda'); g.manual_seed(3)\n"
        "a = torch.rand(batch*m*k, device='cuda', dtype=torch.float64, generator=g); b = torch.rand(batch*k*n, device='cuda', dtype=torch.float64, generator=g)\n"
        "c = torch.zeros(batch*m*n, device='cuda', dtype=torch.float64)\n"
        "blob, d = xs.descriptor(xs.F64, m, n, k)\n"
        "assert 0 == L.libxsmm_amd_gemm_batch_strided(d, xs.dptr(a), xs.dptr(b), xs.dptr(c), m*k, k*n, m*n, batch)\n"
        "torch.cuda.synchronize(); assert 'jit' in xs.last_kernel(), xs.last_kernel()\n"
        "print('%%.17g' %% float(c.sum().item()))\n" % root)
    cache = tmp_path / "cache"
    outs = []
    for _ in range(2):
        env = dict(os.environ, LIBXSMM_AMD_CACHE=str(cache))
        res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.strip().splitlines()[-1])
        files = sorted(os.listdir(cache))
        assert files and all(f.endswith(".hsaco") for f in files), files
    assert outs[0] == outs[1]
    nfiles = len(os.listdir(cache))
    off = tmp_path / "off"
    env = dict(os.environ, LIBXSMM_AMD_CACHE="0", HOME=str(off))
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().splitlines()[-1] == outs[0]
    assert len(os.listdir(cache)) == nfiles and not (off / ".cache" / "libxsmm-amd").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(23, 23, 23, 24, 24, 24), (13, 9, 17, 16, 20, 13), (32, 32, 32, 40, 32, 48), (5, 7, 3, 8, 8, 8), (16, 31, 35, 16, 35, 24)])
def test_jit_leading_dimensions_with_gaps(xs, orc, torch_gpu, dtype, shape):
    """Leading dimensions larger than the matrix (tests/gemm.c rows such as m=10,lda=22 / ldc=12..20): the specialised wave forms
    fetch an operand's span and drop the gaps; rows between m and ldc are never written (checked with NaN-poisoned and
    sentinel-filled gaps), beta = 0 and 1, TRANS_B, strided batches and index batches with runs."""
    torch = torch_gpu
    m, n, k, lda, ldb, ldc = shape
    batch = 333
    rng = np.random.default_rng(m * 5 + n * 3 + k + lda)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    asz, csz = lda * k, ldc * n
    with _JitForced(xs):
        for beta, transb in ((1.0, False), (0.0, False), (1.0, True)):
            ldb_ = max(ldb, n) if transb else ldb
            bsz = ldb_ * (k if transb else n)
            a = rng.uniform(-1, 1, batch * asz).astype(dtype); b = rng.uniform(-1, 1, batch * bsz).astype(dtype)
            c = rng.uniform(-1, 1, batch * csz).astype(dtype)
            if beta == 0.0:
                cc = c.reshape(batch, n, ldc); cc[:, :, :m] = np.nan  # C itself is not read; the gaps keep their values
            flags = (xs.FLAG_TRANS_B if transb else 0)
            oflags = (orc.FLAG_BETA_0 if beta == 0.0 else 0) | (orc.FLAG_TRANS_B if transb else 0)
            ref = c.copy()
            orc.gemm_batch_strided(orc.FMA, oflags, m, n, k, lda, ldb_, ldc, a, b, ref, asz, bsz, csz, batch, 4)
            da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
            blob, d = xs.descriptor(prec, m, n, k, lda, ldb_, ldc, 1.0, beta, flags, 0)
            assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(d, xs.dptr(da), xs.dptr(db), xs.dptr(dc), asz, bsz, csz, batch)
            torch.cuda.synchronize()
            assert "_jit_shape" in xs.last_kernel(), xs.last_kernel()
            assert np.array_equal(dc.cpu().numpy().view(np.uint8), ref.view(np.uint8)), (beta, transb)
        # index batch with runs of equal C (sorted block ids), gaps in C
        nc = 17
        a = rng.uniform(-1, 1, batch * asz).astype(dtype); b = rng.uniform(-1, 1, batch * ldb * n).astype(dtype); c = rng.uniform(-1, 1, nc * csz).astype(dtype)
        cidx = np.sort(rng.integers(0, nc, batch))
        sa = (rng.permutation(batch) * asz).astype(np.int32); sb = (np.arange(batch) * ldb * n).astype(np.int32); sc = (cidx * csz).astype(np.int32)
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, lda, ldb, ldc, a, b, ref, 0, sa, sb, sc, batch)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, lda, db, ldb, 1.0, dc, ldc, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert "_jit_shape_runs" in xs.last_kernel(), xs.last_kernel()
        assert np.array_equal(dc.cpu().numpy().view(np.uint8), ref.view(np.uint8))


@pytest.mark.gpu
def test_jit_never_blocks_a_batch_call(xs, orc, torch_gpu, tmp_path):
    """The compiler works on a helper thread (the product's default, LIBXSMM_AMD_JIT_ASYNC unset): in a cold process with an
    empty code-object cache no batch call waits for hiprtc -- the first calls of a new shape are served by the pre-compiled
    kernel (same bits), after libxsmm_amd_jit_wait the specialised kernel takes over (same bits again). Also a grouped call
    of several new shapes. Run in a child process: environment and caches of this one stay as they are."""
    import subprocess
    import sys
    script = r'''
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch
import oracle_binding as orc
xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
torch.cuda.set_device(0)
L.libxsmm_amd_set_mfma(0)
rng = np.random.default_rng(1)
def stack(m, n, k, batch):
    a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, (batch // 7 + 1) * m * n)
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = ((np.arange(batch) // 7) * m * n).astype(np.int32)
    ref = c.copy(); assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    return [torch.from_numpy(x).cuda() for x in (a, b, c, sa, sb, sc)] + [ref, c]
# warm-up of the runtime itself (first kernel launch, pinned rings, verdict slots): a batch too small to be specialised
w = stack(3, 3, 3, 100)
xs.gemm_batch(xs.F64, "N", "N", 3, 3, 3, 1.0, w[0], 3, w[1], 3, 1.0, w[2], 3, 0, 4, w[3], w[4], w[5], 100); torch.cuda.synchronize()
m, n, k, batch = 29, 17, 21, 20000
d = stack(m, n, k, batch)
t0 = time.perf_counter()
xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, d[0], m, d[1], k, 1.0, d[2], m, 0, 4, d[3], d[4], d[5], batch)
blocked = time.perf_counter() - t0
first = xs.last_kernel()
torch.cuda.synchronize()
assert np.array_equal(d[2].cpu().numpy(), d[6]), "first call"
L.libxsmm_amd_jit_wait()
d[2].copy_(torch.from_numpy(d[7]))
xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, d[0], m, d[1], k, 1.0, d[2], m, 0, 4, d[3], d[4], d[5], batch); torch.cuda.synchronize()
second = xs.last_kernel()
assert np.array_equal(d[2].cpu().numpy(), d[6]), "second call"
# grouped call of three new shapes
shapes = [(11, 19, 7), (27, 5, 30), (9, 9, 9)]
g = [stack(mm, nn, kk, 5000) for (mm, nn, kk) in shapes]
t0 = time.perf_counter()
assert 0 == xs.gemm_batch_groups(xs.F64, shapes, [q[0] for q in g], [q[1] for q in g], [q[2] for q in g], [q[3] for q in g], [q[4] for q in g], [q[5] for q in g], [5000] * 3)
gblocked = time.perf_counter() - t0
gfirst = xs.last_kernel()
torch.cuda.synchronize()
for q in g: assert np.array_equal(q[2].cpu().numpy(), q[6]), "grouped, first call"
L.libxsmm_amd_jit_wait()
for q in g: q[2].copy_(torch.from_numpy(q[7]))
assert 0 == xs.gemm_batch_groups(xs.F64, shapes, [q[0] for q in g], [q[1] for q in g], [q[2] for q in g], [q[3] for q in g], [q[4] for q in g], [q[5] for q in g], [5000] * 3)
torch.cuda.synchronize()
gsecond = xs.last_kernel()
for q in g: assert np.array_equal(q[2].cpu().numpy(), q[6]), "grouped, second call"
print("RESULT", blocked, first, second, gblocked, gfirst, gsecond)
'''
    env = dict(os.environ)
    env.pop("LIBXSMM_AMD_JIT_ASYNC", None)
    env.pop("LIBXSMM_AMD_JIT_MINBATCH", None)
    env["LIBXSMM_AMD_CACHE"] = str(tmp_path / "cold_cache")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", script, root], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    blocked, first, second, gblocked, gfirst, gsecond = float(line[1]), line[2], line[3], float(line[4]), line[5], line[6]
    assert blocked < 0.05 and gblocked < 0.05, (blocked, gblocked)   # no call waits for the compiler (0.3 s and more per kernel)
    assert "generic" in first and "generic" in gfirst, (first, gfirst)
    assert second.endswith("_jit_shape_runs") and gsecond.endswith("_jit_shape_runs_grouped"), (second, gsecond)
    assert len(os.listdir(str(tmp_path / "cold_cache"))) >= 2     # the compiled code objects were kept for the next process


@pytest.mark.gpu
def test_process_may_exit_while_the_compiler_thread_is_busy(xs, torch_gpu, tmp_path):
    """A short-lived process that has just handed a kernel to the compiler thread exits cleanly (the helper thread is not left
    inside hiprtc / the HIP runtime while their static objects are torn down), and the code object it was building is on disk."""
    import subprocess
    import sys
    script = r'''
import importlib, os, sys
sys.path.insert(0, sys.argv[1])
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
'''
    env = dict(os.environ)
    env.pop("LIBXSMM_AMD_JIT_ASYNC", None)
    env["LIBXSMM_AMD_CACHE"] = str(tmp_path / "exit_cache")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", script, root], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-2000:])
    assert "KERNEL smm_f64_generic" in res.stdout            # served by the pre-compiled kernel, the compiler was still busy
    assert len(os.listdir(str(tmp_path / "exit_cache"))) >= 1  # ... and finished its job before the process was gone
