"""The run form on the matrix cores (csrc/xsmm_jit_smm.cpp, SMM_JIT_MFMA_RUNS_*): batches whose consecutive products share a C
block -- CP2K stacks (samples/cp2k/cp2k.cpp:328-360), batch-reduce, blocked GEMM work lists -- with M, N <= 32. A wave owns a
run and keeps C in the accumulators of v_mfma_{f32,f64}_16x16x4 across its products; both instructions are k-ordered fma
chains, so C must equal the oracle's sequential chain (products in batch order, k ascending) BIT FOR BIT, exactly as the
register-tiled run form does. Covered: index and pointer batches, runs that cross the 64-item scan chunks, untouched C blocks,
odd shapes (K not a multiple of four: the -0 / +0 padding), leading dimensions with gaps, the relaxed order (segments joined
with atomics: tolerance), and the 27-shape grouped launch of BASELINE config 5 under both policies.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Jit:
    """small test batches on the run-time specialised kernels, matrix cores on (the default policy) or off; tiles: a batch of few runs
    whose C has several 16 x 16 tiles runs a wave per run and TILE (xsmm_jit_smm.cpp:smm_tile_split; the default) or per run"""
    def __init__(self, xs, mfma=1, tiles=1):
        self.xs = xs; self.mfma = mfma; self.tiles = tiles

    def __enter__(self):
        self.old_env = {k: os.environ.get(k) for k in ("LIBXSMM_AMD_JIT_MINBATCH", "XSMM_SMMJIT_TILESPLIT")}
        os.environ["LIBXSMM_AMD_JIT_MINBATCH"] = "1"
        os.environ["XSMM_SMMJIT_TILESPLIT"] = str(self.tiles)
        self.old = self.xs.lib().libxsmm_amd_set_mfma(self.mfma)

    def __exit__(self, *exc):
        self.xs.lib().libxsmm_amd_set_mfma(self.old)
        for k, v in self.old_env.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def runs_kernel(dtype, m, n, tiles=1):
    """the kernel a small index batch with shared C blocks lands on"""
    split = tiles and (m > 16 or n > 16)
    return "smm_f%d_mfma_runs_%sjit" % (64 if np.dtype(dtype) == np.float64 else 32, "tiles_" if split else "")


SHAPES = [(13, 13, 13), (23, 23, 23), (32, 32, 32), (32, 13, 23), (13, 32, 32), (5, 7, 3), (16, 16, 16), (1, 1, 1), (17, 31, 29), (32, 32, 64), (9, 20, 2),
          (31, 2, 63)]


@pytest.mark.parametrize("tiles", [1, 2, 0])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", SHAPES)
def test_runs_on_the_matrix_cores_bitexact(xs, orc, torch_gpu, dtype, shape, tiles):
    """index batches: (1) permuted operands, every item its own C (index_base 1); (2) runs of 1 ... 200 products with gaps between
    the C blocks that are used. A wave per run (tiles = 0), or -- these batches are small -- per run and 16 x 16 tile of C, the tiles
    as the groups of a grouped launch (1, the default) or as the waves of one work-group (2): the same bits"""
    torch = torch_gpu
    m, n, k = shape
    batch = 777
    rng = np.random.default_rng(m * 31 + n * 7 + k)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    want = runs_kernel(dtype, m, n, tiles)
    with _Jit(xs, tiles=tiles):
        c = rng.uniform(-1, 1, batch * m * n).astype(dtype)
        sa = (rng.permutation(batch) * m * k + 1).astype(np.int32); sb = (rng.permutation(batch) * k * n + 1).astype(np.int32)
        sc = (np.arange(batch) * m * n + 1).astype(np.int32)
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 1, sa, sb, sc, batch)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 1, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel() == want, xs.last_kernel()
        assert np.array_equal(dc.cpu().numpy().view(np.uint8), ref.view(np.uint8))
        lens = [1, 1, 2, 3, 64, 65, 130, 1, 7, 200, 1, 1, 1, 1, 1]
        lens = np.array(lens + [batch - sum(lens)], dtype=np.int64)
        nc = len(lens)
        owners = np.sort(rng.choice(np.arange(nc + 5), size=nc, replace=False))
        cidx = np.repeat(owners, lens)
        c2 = rng.uniform(-1, 1, (nc + 5) * m * n).astype(dtype)
        sa0 = (np.arange(batch) * m * k).astype(np.int32); sb0 = (np.arange(batch) * k * n).astype(np.int32); sc0 = (cidx * m * n).astype(np.int32)
        ref2 = c2.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref2, 0, sa0, sb0, sc0, batch)
        dc2 = torch.from_numpy(c2).cuda()
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc2, m, 0, 4, sa0, sb0, sc0, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel() == want, xs.last_kernel()
        assert np.array_equal(dc2.cpu().numpy().view(np.uint8), ref2.view(np.uint8))


def _random_run_shapes(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        m, n, k = int(rng.integers(1, 33)), int(rng.integers(1, 33)), int(rng.integers(1, 65))
        lda = m + int(rng.choice([0, 0, 1, 3, 8])); ldb = k + int(rng.choice([0, 0, 1, 5])); ldc = m + int(rng.choice([0, 0, 2, 7]))
        out.append((m, n, k, lda, ldb, ldc))
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", _random_run_shapes(10, 20251005))
def test_random_shapes_and_leading_dimensions_bitexact(xs, orc, torch_gpu, dtype, shape):
    """shapes, leading dimensions and run structures drawn at random (fixed seed): every combination of tile counts (1, 2, 4 tiles of C),
    K padding (K % 4), fragment windows (offsets beyond 4 KiB) and spans of B the generator can meet -- a wave per run and tile, then
    a wave per run, both bit-equal to the oracle's sequential chain; elements of C in the gaps stay untouched"""
    torch = torch_gpu
    m, n, k, lda, ldb, ldc = shape
    rng = np.random.default_rng(m * 1009 + n * 31 + k)
    batch = 333
    lens = []
    while sum(lens) < batch:
        lens.append(int(rng.choice([1, 1, 2, 5, 17, 70, 130])))
    lens[-1] -= sum(lens) - batch
    lens = np.array(lens, dtype=np.int64); nc = len(lens)
    owners = np.sort(rng.choice(np.arange(nc + 3), size=nc, replace=False))
    cidx = np.repeat(owners, lens)
    a = rng.uniform(-1, 1, batch * lda * k).astype(dtype); b = rng.uniform(-1, 1, batch * ldb * n).astype(dtype)
    c = rng.uniform(-1, 1, (nc + 3) * ldc * n).astype(dtype)
    sa = (rng.permutation(batch) * lda * k).astype(np.int32); sb = (rng.permutation(batch) * ldb * n).astype(np.int32); sc = (cidx * ldc * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, lda, ldb, ldc, a, b, ref, 0, sa, sb, sc, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for tiles in (1, 0):
        with _Jit(xs, tiles=tiles):
            dc = torch.from_numpy(c).cuda()
            xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, lda, db, ldb, 1.0, dc, ldc, 0, 4, sa, sb, sc, batch)
            torch.cuda.synchronize()
            assert xs.last_kernel() == runs_kernel(dtype, m, n, tiles), (xs.last_kernel(), tiles)
        assert np.array_equal(dc.cpu().numpy().view(np.uint8), ref.view(np.uint8)), tiles


def test_batch_calls_captured_in_a_graph(xs, orc, torch_gpu):
    """libxsmm_gemm_batch calls on device index arrays recorded into a HIP graph and replayed (an index batch makes no host round trip:
    ordering check, verdict and multiplication are all launches): a batch that runs a wave per run and tile -- its table travels as a
    kernel argument, nothing is staged -- and one that runs a wave per run; every replay adds the products once more, bit-equal to the
    oracle applied as often"""
    torch = torch_gpu
    L = xs.lib()
    rng = np.random.default_rng(3)
    cases = []
    for (m, n, k) in ((23, 23, 23), (13, 13, 13)):
        batch, runlen = 600, 50
        a = rng.uniform(-1, 1, batch * m * k); b = rng.uniform(-1, 1, batch * k * n); c = rng.uniform(-1, 1, (batch // runlen) * m * n)
        sa = (rng.permutation(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = ((np.arange(batch) // runlen) * m * n).astype(np.int32)
        cases.append((m, n, k, batch, a, b, c, sa, sb, sc))
    with _Jit(xs):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        dev = []
        for (m, n, k, batch, a, b, c, sa, sb, sc) in cases:
            dev.append([torch.from_numpy(x).cuda() for x in (a, b, c, sa, sb, sc)])
        torch.cuda.synchronize()

        def calls():
            for (m, n, k, batch, *_), (da, db, dc, dsa, dsb, dsc) in zip(cases, dev):
                xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, dsa, dsb, dsc, batch)
        names = []
        try:
            with torch.cuda.stream(side):
                L.libxsmm_amd_set_stream(C.c_void_p(side.cuda_stream))
                calls()  # warm-up on the capture stream (kernels resolved, scratch in place): application 1
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    calls()  # (captured, not run)
                names.append(xs.last_kernel())
                graph.replay(); graph.replay()  # applications 2 and 3
                side.synchronize()
        finally:
            L.libxsmm_amd_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert names[0] == "smm_f64_mfma_runs_jit"  # (the last call of the capture: 13^3, one tile)
        for (m, n, k, batch, a, b, c, sa, sb, sc), (da, db, dc, *_) in zip(cases, dev):
            ref = c.copy()
            for _ in range(3):
                assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
            assert np.array_equal(dc.cpu().numpy().view(np.uint64), ref.view(np.uint64)), (m, n, k)


def test_signs_of_zeros_survive_the_k_padding(xs, orc, torch_gpu):
    """K = 13 is padded to 16 with A = -0, B = +0: the padded products are -0, the identity of the addition for every sum --
    also for a C that is -0 and stays untouched by real products of zero"""
    torch = torch_gpu
    m, n, k = 13, 13, 13
    batch = 64
    a = np.zeros(batch * m * k); b = np.zeros(batch * k * n)
    a[::3] = -0.0
    c = np.full(batch * m * n, -0.0)  # 0 * 0 = +0 added to -0 gives +0 in the reference's chain as well: whatever comes out must match the oracle's bits
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = ((np.arange(batch) // 4) * m * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    with _Jit(xs):
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(xs.F64, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel() == "smm_f64_mfma_runs_jit"
    assert np.array_equal(dc.cpu().numpy().view(np.uint64), ref.view(np.uint64))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pointer_batches_with_runs_on_the_matrix_cores(xs, orc, torch_gpu, dtype):
    """arrays of pointers (src/libxsmm_gemm.c:1426-1461) with repeated consecutive C pointers"""
    torch = torch_gpu
    m, n, k = 23, 13, 32
    batch, nc = 300, 9
    ts = np.dtype(dtype).itemsize
    rng = np.random.default_rng(77)
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype); c = rng.uniform(-1, 1, nc * m * n).astype(dtype)
    cidx = np.sort(rng.integers(0, nc, batch))
    ref = c.copy()
    sa = (np.arange(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    pa = (da.data_ptr() + np.arange(batch, dtype=np.uint64) * np.uint64(m * k * ts)).astype(np.uint64)
    pb = (db.data_ptr() + np.arange(batch, dtype=np.uint64) * np.uint64(k * n * ts)).astype(np.uint64)
    pc = (dc.data_ptr() + cidx.astype(np.uint64) * np.uint64(m * n * ts)).astype(np.uint64)
    dpa, dpb, dpc = (torch.from_numpy(x.view(np.int64)).cuda() for x in (pa, pb, pc))
    ptrsize = np.array([8], dtype=np.int32)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    with _Jit(xs):
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, dpa, m, dpb, k, 1.0, dpc, m, 0, 0, ptrsize, ptrsize, ptrsize, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_mfma_runs_jit"), xs.last_kernel()
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(23, 23, 23, 24, 24, 24), (13, 9, 17, 16, 20, 13), (32, 32, 32, 40, 32, 48), (5, 7, 3, 8, 8, 8), (16, 31, 35, 16, 35, 24)])
def test_runs_with_gaps_in_the_leading_dimensions(xs, orc, torch_gpu, dtype, shape):
    """lda > m, ldb > k, ldc > m: A's fragments and C are addressed through their leading dimensions, B's span is fetched whole and
    the gaps are dropped on the way into LDS; elements of C in the gaps stay untouched"""
    torch = torch_gpu
    m, n, k, lda, ldb, ldc = shape
    batch, nc = 500, 23
    rng = np.random.default_rng(lda + ldb + ldc)
    a = rng.uniform(-1, 1, batch * lda * k).astype(dtype); b = rng.uniform(-1, 1, batch * ldb * n).astype(dtype)
    c = rng.uniform(-1, 1, nc * ldc * n).astype(dtype)
    cidx = np.sort(rng.integers(0, nc, batch))
    sa = (np.arange(batch) * lda * k).astype(np.int32); sb = (np.arange(batch) * ldb * n).astype(np.int32); sc = (cidx * ldc * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, lda, ldb, ldc, a, b, ref, 0, sa, sb, sc, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    with _Jit(xs):
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, lda, db, ldb, 1.0, dc, ldc, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert xs.last_kernel() == runs_kernel(dtype, m, n), xs.last_kernel()
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(23, 23, 23), (13, 5, 7), (32, 32, 32)])
def test_relaxed_order_on_the_matrix_cores(xs, orc, torch_gpu, dtype, shape):
    """libxsmm_gemm_batch_omp leaves the order of the sums open: few long runs are cut into segments whose sums join C with
    floating-point atomics -- tolerance eps * sqrt(terms) * 4 as in tests/test_jit.py; the strict entry point stays bit-exact"""
    torch = torch_gpu
    m, n, k = shape
    rng = np.random.default_rng(5 + m)
    lens = np.array([1500, 900, 1, 1200, 700], dtype=np.int64)
    batch = int(lens.sum()); nc = len(lens) + 1
    cidx = np.repeat(np.array([0, 1, 2, 3, 4]), lens)
    a = rng.uniform(-1, 1, batch * m * k).astype(dtype); b = rng.uniform(-1, 1, batch * k * n).astype(dtype)
    c = rng.uniform(-1, 1, nc * m * n).astype(dtype)
    sa = (rng.permutation(batch) * m * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (cidx * m * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    da, db = (torch.from_numpy(x).cuda() for x in (a, b))
    with _Jit(xs):
        dc = torch.from_numpy(c).cuda()
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), ref)
        dc = torch.from_numpy(c).cuda()
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, m, db, k, 1.0, dc, m, 0, 4, sa, sb, sc, batch, omp=True)
        torch.cuda.synchronize()
        assert xs.last_kernel() == runs_kernel(dtype, m, n), xs.last_kernel()
    out = dc.cpu().numpy()
    tol = np.finfo(dtype).eps * np.sqrt(float(lens.max()) * k) * 4
    assert np.max(np.abs(out - ref)) <= tol * np.max(np.abs(ref))
    assert np.array_equal(out[5 * m * n:], c[5 * m * n:])  # the unreferenced block is untouched


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("host_indexes", [False, True])
def test_cp2k_27_shape_grouped_launch_bitexact(xs, orc, torch_gpu, mfma, host_indexes):
    """The launch BASELINE config 5 is measured on (bench.py, tools/bench_cp2k.py), at a size the oracle finishes in seconds: all
    27 shapes (M, N, K) in {13, 23, 32}^3 in ONE libxsmm_amd_gemm_batch_groups call, every group a stack whose u = isqrt(s 160 / 240)
    consecutive products accumulate into one C block (samples/cp2k/cp2k.cpp:155,328-360; reference entry: groups of
    libxsmm_dgemm_batch, src/libxsmm_gemm.c:1231-1262, src/libxsmm_ext_gemm.c:758-972). One fused multiplication launch -- the
    27-body code object -- and every C block equal to the oracle's sequential chain bit for bit, matrix cores on and off."""
    torch = torch_gpu
    L = xs.lib()
    shapes = [(m, n, k) for m in (13, 23, 32) for n in (13, 23, 32) for k in (13, 23, 32)]
    rng = np.random.default_rng(2718)
    groups, sizes = [], []
    for gi, (m, n, k) in enumerate(shapes):
        s = 1500 + 37 * gi  # (group sizes differ: the table of work-group ranges is not uniform)
        u = max(1, math.isqrt(s * 160 // 240)); nc = (s + u - 1) // u
        a = rng.uniform(-1, 1, s * m * k); b = rng.uniform(-1, 1, s * k * n); c = rng.uniform(-1, 1, nc * m * n)
        idx = np.arange(s)
        sa = (rng.permutation(s) * m * k).astype(np.int32); sb = (idx * k * n).astype(np.int32); sc = ((idx // u) * m * n).astype(np.int32)
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, s)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        ix = (sa, sb, sc) if host_indexes else tuple(torch.from_numpy(x).cuda() for x in (sa, sb, sc))
        groups.append((da, db, dc, ix, ref)); sizes.append(s)
    with _Jit(xs, mfma):
        before = L.libxsmm_amd_launch_count()
        rc = xs.gemm_batch_groups(xs.F64, shapes, [g[0] for g in groups], [g[1] for g in groups], [g[2] for g in groups],
                                  [g[3][0] for g in groups], [g[3][1] for g in groups], [g[3][2] for g in groups], sizes)
        assert rc == 0
        torch.cuda.synchronize()
        assert xs.last_kernel() == "smm_f64_jit_shape_runs_grouped", xs.last_kernel()
        assert L.libxsmm_amd_launch_count() == before + 1  # ONE multiplication launch for the 27 groups
    for gi, (da, db, dc, ix, ref) in enumerate(groups):
        assert np.array_equal(dc.cpu().numpy().view(np.uint64), ref.view(np.uint64)), shapes[gi]


def test_more_groups_than_one_launch_takes(xs, orc, torch_gpu):
    """A fused launch takes up to 32 groups (the ordering check and the grouped kernel carry their tables as kernel arguments); a call
    of 40 groups (five shapes, eight stacks each) becomes two fused launches, in the order of the groups, bit-equal to the oracle"""
    torch = torch_gpu
    L = xs.lib()
    base = [(13, 13, 13), (23, 23, 23), (32, 32, 32), (13, 32, 23), (5, 7, 9)]
    shapes = [base[i % len(base)] for i in range(40)]
    rng = np.random.default_rng(40)
    groups, sizes = [], []
    for gi, (m, n, k) in enumerate(shapes):
        s = 200 + 11 * gi
        u = 1 + gi % 7; nc = (s + u - 1) // u
        a = rng.uniform(-1, 1, s * m * k); b = rng.uniform(-1, 1, s * k * n); c = rng.uniform(-1, 1, nc * m * n)
        idx = np.arange(s)
        sa = (rng.permutation(s) * m * k).astype(np.int32); sb = (idx * k * n).astype(np.int32); sc = ((idx // u) * m * n).astype(np.int32)
        ref = c.copy()
        assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, s)
        groups.append(tuple(torch.from_numpy(x).cuda() for x in (a, b, c, sa, sb, sc)) + (ref,)); sizes.append(s)
    with _Jit(xs):
        before = L.libxsmm_amd_launch_count()
        assert 0 == xs.gemm_batch_groups(xs.F64, shapes, [g[0] for g in groups], [g[1] for g in groups], [g[2] for g in groups],
                                         [g[3] for g in groups], [g[4] for g in groups], [g[5] for g in groups], sizes)
        torch.cuda.synchronize()
        assert xs.last_kernel() == "smm_f64_jit_shape_runs_grouped", xs.last_kernel()
        assert L.libxsmm_amd_launch_count() == before + 2
    for gi, g in enumerate(groups):
        assert np.array_equal(g[2].cpu().numpy().view(np.uint64), g[6].view(np.uint64)), (gi, shapes[gi])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(23, 23, 23, 24, 24, 24), (13, 13, 13, 16, 16, 16), (32, 32, 32, 40, 40, 40), (5, 7, 3, 8, 8, 8), (16, 31, 35, 16, 35, 24),
                                   (23, 23, 23, 23, 23, 23), (32, 32, 32, 32, 32, 32), (1, 1, 1, 1, 1, 1), (31, 2, 63, 33, 64, 31),
                                   # K beyond 64: through the registers and B's image in chunks of at most 32 (the last chunk padded with -0 / +0)
                                   (23, 23, 70, 23, 70, 23), (32, 32, 128, 32, 128, 32), (13, 17, 200, 16, 203, 20), (5, 3, 65, 5, 65, 5), (16, 16, 256, 16, 256, 16)])
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_independent_items_on_the_matrix_core_streaming_form(xs, orc, torch_gpu, dtype, shape, beta):
    """strided batches whose items own their C (SYNC_NONE), with and without gaps in the leading dimensions, on the streaming form of
    the matrix-core kernel: a wave per item, A's fragments and C addressed through their leading dimensions. Bit-equal to the
    oracle's chain; gap elements of C untouched; beta = 0 never reads C (NaN canaries); batches smaller and larger than the grid."""
    torch = torch_gpu
    m, n, k, lda, ldb, ldc = shape
    rng = np.random.default_rng(lda * 3 + ldb + ldc + int(beta))
    prec = xs.F64 if dtype == np.float64 else xs.F32
    old_env = os.environ.get("XSMM_SMMJIT_GAPS_MFMA")
    os.environ["XSMM_SMMJIT_GAPS_MFMA"] = "2"
    try:
        with _Jit(xs):
            for batch in (7, 5000):
                a = rng.uniform(-1, 1, batch * lda * k).astype(dtype); b = rng.uniform(-1, 1, batch * ldb * n).astype(dtype)
                c = rng.uniform(-1, 1, batch * ldc * n).astype(dtype)
                if beta == 0.0:  # the elements a product writes are NaN before: C must not be read
                    cv = c.reshape(batch, n, ldc); cv[:, :, :m] = np.nan
                ref = c.copy()
                orc.gemm_batch_strided(orc.FMA, 0 if beta else 16, m, n, k, lda, ldb, ldc, a, b, ref, lda * k, ldb * n, ldc * n, batch)
                blob, d = xs.descriptor(prec, m, n, k, lda, ldb, ldc, 1.0, beta)
                da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
                assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(d, xs.dptr(da), xs.dptr(db), xs.dptr(dc), lda * k, ldb * n, ldc * n, batch)
                torch.cuda.synchronize()
                if not (dtype == np.float32 and shape == (32, 32, 32, 32, 32, 32)):  # (tight fp32 32^3 has its hand-tuned kernel, tried first)
                    assert xs.last_kernel().endswith("_mfma_stream_jit"), xs.last_kernel()
                assert np.array_equal(dc.cpu().numpy().view(np.uint8), ref.view(np.uint8)), batch
    finally:
        if old_env is None:
            del os.environ["XSMM_SMMJIT_GAPS_MFMA"]
        else:
            os.environ["XSMM_SMMJIT_GAPS_MFMA"] = old_env


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_streaming_form_with_index_and_pointer_batches(xs, orc, torch_gpu, dtype):
    """the caller's promise of a negative batchsize (no two items share a C, reference src/libxsmm_gemm.c:1338,1430) puts index and
    pointer batches with gaps in the leading dimensions on the streaming form as well: every wave looks its item's addresses up"""
    torch = torch_gpu
    m, n, k, lda, ldb, ldc = 23, 17, 29, 24, 32, 28
    batch = 3000
    ts = np.dtype(dtype).itemsize
    rng = np.random.default_rng(41)
    a = rng.uniform(-1, 1, batch * lda * k).astype(dtype); b = rng.uniform(-1, 1, batch * ldb * n).astype(dtype)
    c = rng.uniform(-1, 1, batch * ldc * n).astype(dtype)
    pa, pb, pc = rng.permutation(batch), rng.permutation(batch), rng.permutation(batch)
    sa = (pa * lda * k).astype(np.int32); sb = (pb * ldb * n).astype(np.int32); sc = (pc * ldc * n).astype(np.int32)
    ref = c.copy()
    assert 0 == orc.gemm_batch_idx(orc.FMA, 0, m, n, k, lda, ldb, ldc, a, b, ref, 0, sa, sb, sc, batch)
    prec = xs.F64 if dtype == np.float64 else xs.F32
    with _Jit(xs):
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, da, lda, db, ldb, 1.0, dc, ldc, 0, 4, sa, sb, sc, -batch)
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_mfma_stream_jit"), xs.last_kernel()
        assert np.array_equal(dc.cpu().numpy(), ref)
        dc = torch.from_numpy(c).cuda()
        qa = torch.from_numpy((da.data_ptr() + sa.astype(np.int64) * ts).astype(np.int64)).cuda()
        qb = torch.from_numpy((db.data_ptr() + sb.astype(np.int64) * ts).astype(np.int64)).cuda()
        qc = torch.from_numpy((dc.data_ptr() + sc.astype(np.int64) * ts).astype(np.int64)).cuda()
        ptrsize = np.array([8], dtype=np.int32)
        xs.gemm_batch(prec, "N", "N", m, n, k, 1.0, qa, lda, qb, ldb, 1.0, qc, ldc, 0, 0, ptrsize, ptrsize, ptrsize, -batch)
        torch.cuda.synchronize()
        assert xs.last_kernel().endswith("_mfma_stream_jit"), xs.last_kernel()
        assert np.array_equal(dc.cpu().numpy(), ref)
