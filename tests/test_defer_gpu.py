"""Per-call invocations of a dispatched kernel on device memory (the reference's canonical caller,
samples/smm/specialized.cpp:172-190: `kernel(a + i * asize, b + i * bsize, c + i * csize)` once per product) inside the
opt-in bracket libxsmm_amd_defer_begin/end are recorded into stream-ordered bursts instead of costing a launch each
(libxsmm-1_amd/csrc/xsmm_defer.cpp). What must hold: the results equal the oracle's sequential loop bit for bit whatever
the calls alias, and anything the caller queues or waits for after the last call -- without a further call into the
library -- sees them. (The default, a launch per call, and work of the caller's own between two calls:
tests/test_call_order_gpu.py.)

The calling loop is a few lines of C compiled here (a Python loop is too slow to keep a burst open).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOOP_C = r"""
typedef void (*fn3)(const void*, const void*, void*);
void call_loop(fn3 f, const char* a, const char* b, char* c, const long long* ia, const long long* ib, const long long* ic, long long n, int ts)
{ long long i; for (i = 0; i < n; ++i) f(a + ia[i] * ts, b + ib[i] * ts, c + ic[i] * ts); }
typedef void (*fnx)(const void*, const void*, void*);
/* the PyFR driver's loop (samples/pyfr/pyfr_driver_asp_reg.c:300-308): execute(handle, B + z, C + z) for z = 0, nblock, 2 nblock, ... */
void panel_loop(fnx execute, const void* handle, const char* b, char* c, long long n, long long nblock, int ts, int backwards)
{ long long z; if (!backwards) for (z = 0; z < n; z += nblock) execute(handle, b + z * ts, c + z * ts); else for (z = n - nblock; z >= 0; z -= nblock) execute(handle, b + z * ts, c + z * ts); }
void call_loop2(fn3 f0, fn3 f1, int period, const char* a, const char* b, char* c, const long long* ia, const long long* ib, const long long* ic, long long n, int ts)
{ long long i; for (i = 0; i < n; ++i) (((i / period) & 1) ? f1 : f0)(a + ia[i] * ts, b + ib[i] * ts, c + ic[i] * ts); }
"""


@pytest.fixture(scope="module")
def loop(tmp_path_factory):
    d = tmp_path_factory.mktemp("loop")
    src = d / "loop.c"
    src.write_text(LOOP_C)
    so = d / "loop.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    vp, ll = C.c_void_p, C.c_longlong
    lib.call_loop.argtypes = [vp, vp, vp, vp, vp, vp, vp, ll, C.c_int]; lib.call_loop.restype = None
    lib.call_loop2.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, ll, C.c_int]; lib.call_loop2.restype = None
    lib.panel_loop.argtypes = [vp, vp, vp, vp, ll, ll, C.c_int, C.c_int]; lib.panel_loop.restype = None
    return lib


@pytest.fixture(autouse=True)
def bracket(xs):
    """every test of this module runs inside the calling thread's opt-in bracket"""
    L = xs.lib()
    assert 0 == L.libxsmm_amd_defer_active()  # off by default
    L.libxsmm_amd_defer_begin()
    assert 1 == L.libxsmm_amd_defer_active()
    yield
    L.libxsmm_amd_defer_end()
    assert 0 == L.libxsmm_amd_defer_active()


def _dispatch(xs, dtype, m, n, k, flags=None):
    f = xs.lib().libxsmm_dmmdispatch if dtype == np.float64 else xs.lib().libxsmm_smmdispatch
    fl = None if flags is None else C.byref(C.c_int(flags))
    fn = f(m, n, k, None, None, None, None, None, fl, None)
    assert fn
    return fn


def _idx(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.int64))


def _run(loop, fn, da, db, dc, ia, ib, ic, ts):
    ia, ib, ic = _idx(ia), _idx(ib), _idx(ic)
    loop.call_loop(fn, da.data_ptr(), db.data_ptr(), dc.data_ptr(), ia.ctypes.data, ib.ctypes.data, ic.ctypes.data, len(ia), ts)


@pytest.fixture()
def scalar_kernels(xs):
    old = xs.lib().libxsmm_amd_set_mfma(0)  # bit-exact against the oracle's fma chain
    yield
    xs.lib().libxsmm_amd_set_mfma(old)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape,calls", [((23, 23, 23), 20000), ((13, 5, 7), 9000), ((32, 32, 32), 3000), ((40, 40, 40), 700)])
def test_call_per_product_equals_the_batch(xs, orc, torch_gpu, loop, scalar_kernels, dtype, shape, calls):
    """independent C blocks, more calls than one burst holds; afterwards only the caller's own synchronisation"""
    torch = torch_gpu
    m, n, k = shape
    rng = np.random.default_rng(calls)
    a = rng.uniform(-1, 1, calls * m * k).astype(dtype); b = rng.uniform(-1, 1, calls * k * n).astype(dtype); c = rng.uniform(-1, 1, calls * m * n).astype(dtype)
    ref = c.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, m * k, k * n, m * n, calls)
    fn = _dispatch(xs, dtype, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    torch.cuda.synchronize()
    before = xs.lib().libxsmm_amd_launch_count()
    i = np.arange(calls)
    _run(loop, fn, da, db, dc, i * m * k, i * k * n, i * m * n, a.itemsize)
    got = dc.cpu().numpy()  # a copy on the same stream: ordered behind the calls, no call into the library in between
    assert np.array_equal(got, ref)
    launches = xs.lib().libxsmm_amd_launch_count() - before
    assert launches <= max(4, calls // 50), (launches, calls)  # recorded, not launched one by one
    assert xs.last_kernel() == "smm_deferred_calls"


def test_repeated_c_is_summed_in_call_order(xs, orc, torch_gpu, loop, scalar_kernels):
    """runs of calls with one C (the CP2K pattern, samples/cp2k/cp2k.cpp:341-346): the sequential chain, bit for bit"""
    torch = torch_gpu
    m = n = k = 13
    rng = np.random.default_rng(7)
    nc, calls = 300, 6000
    cid = np.sort(rng.integers(0, nc, calls))
    a = rng.uniform(-1, 1, calls * m * k); b = rng.uniform(-1, 1, calls * k * n); c = rng.uniform(-1, 1, nc * m * n)
    i = np.arange(calls)
    sa, sb, sc = (i * m * k).astype(np.int32), (i * k * n).astype(np.int32), (cid * m * n).astype(np.int32)
    ref = c.copy(); orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, calls)
    fn = _dispatch(xs, np.float64, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    _run(loop, fn, da, db, dc, sa, sb, sc, 8)
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("pattern", ["chain", "pingpong", "shuffled", "descending"])
def test_calls_that_depend_on_each_other(xs, orc, torch_gpu, loop, scalar_kernels, pattern):
    """operands in one pool: a call may read what an earlier call wrote, or write the same block again later --
    the result is the sequential loop's whatever is recorded side by side"""
    torch = torch_gpu
    m = n = k = 8
    sz = m * n
    rng = np.random.default_rng(11)
    nblk, calls = 512, 1500
    pool = rng.uniform(-0.5, 0.5, nblk * sz)
    if pattern == "chain":  # C of call i is A of call i + 1
        ia = np.arange(calls) % (nblk - 2); ic = ia + 1; ib = np.full(calls, nblk - 1)
    elif pattern == "pingpong":  # C blocks 0, 1, 0, 1, ... (repeats that are not consecutive)
        ic = np.arange(calls) % 2; ia = 2 + (np.arange(calls) % 100); ib = 102 + (np.arange(calls) % 100)
    elif pattern == "shuffled":  # C blocks in random order with repeats; A and B from a part of the pool that is never written
        ic = rng.integers(0, 200, calls); ia = rng.integers(200, nblk, calls); ib = rng.integers(200, nblk, calls)
    else:  # the loop walks downwards
        calls = 400
        ic = np.arange(calls)[::-1].copy(); ia = 400 + (np.arange(calls) % 50); ib = 450 + (np.arange(calls) % 50)
    sa, sb, sc = (ia * sz).astype(np.int32), (ib * sz).astype(np.int32), (ic * sz).astype(np.int32)
    ref = pool.copy(); orc.gemm_batch_idx(orc.FMA, 0, m, n, k, m, k, m, ref, ref, ref, 0, sa, sb, sc, calls)
    fn = _dispatch(xs, np.float64, m, n, k)
    dp = torch.from_numpy(pool).cuda()
    _run(loop, fn, dp, dp, dp, sa, sb, sc, 8)
    torch.cuda.synchronize()
    assert np.array_equal(dp.cpu().numpy(), ref)


def test_kernels_taking_turns(xs, orc, torch_gpu, loop, scalar_kernels):
    """two kernels called alternately (period 1 and period 37) on the same C blocks"""
    torch = torch_gpu
    rng = np.random.default_rng(3)
    m, n, k0, k1 = 16, 16, 16, 8
    calls, nc = 2000, 100
    a = rng.uniform(-1, 1, calls * m * k0); b = rng.uniform(-1, 1, calls * k0 * n); c0 = rng.uniform(-1, 1, nc * m * n)
    f0 = _dispatch(xs, np.float64, m, n, k0); f1 = _dispatch(xs, np.float64, m, n, k1)
    i = np.arange(calls)
    ia, ib, ic = _idx(i * m * k0), _idx(i * k0 * n), _idx((i % nc) * m * n)
    for period in (1, 37):
        ref = c0.copy()
        for j in range(calls):
            kk = k1 if (j // period) & 1 else k0
            blk = ref[ic[j]:ic[j] + m * n]
            orc.smm(orc.FMA, 0, m, n, kk, m, kk, m, a[ia[j]:ia[j] + m * kk], b[ib[j]:ib[j] + kk * n], blk)
        da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c0))
        loop.call_loop2(f0, f1, period, da.data_ptr(), db.data_ptr(), dc.data_ptr(), ia.ctypes.data, ib.ctypes.data, ic.ctypes.data, calls, 8)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy(), ref), period


def test_overwriting_kernel_keeps_the_last_call(xs, orc, torch_gpu, loop, scalar_kernels):
    """beta = 0: calls that write one C twice in a row are not a run -- the later product alone remains"""
    torch = torch_gpu
    m = n = k = 12
    rng = np.random.default_rng(5)
    calls = 1000
    a = rng.uniform(-1, 1, calls * m * k); b = rng.uniform(-1, 1, calls * k * n)
    c = np.full((calls // 2) * m * n, np.nan)
    i = np.arange(calls)
    sa, sb, sc = (i * m * k).astype(np.int32), (i * k * n).astype(np.int32), ((i // 2) * m * n).astype(np.int32)
    ref = c.copy(); orc.gemm_batch_idx(orc.FMA, xs.FLAG_BETA_0, m, n, k, m, k, m, a, b, ref, 0, sa, sb, sc, calls)
    fn = _dispatch(xs, np.float64, m, n, k, flags=xs.FLAG_BETA_0)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    _run(loop, fn, da, db, dc, sa, sb, sc, 8)
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)


def test_library_calls_between_bursts_keep_the_order(xs, orc, torch_gpu, loop, scalar_kernels):
    """a burst, a batch call on the same C, another burst, an explicit flush: stream order throughout"""
    torch = torch_gpu
    m = n = k = 10
    rng = np.random.default_rng(9)
    calls = 500
    a = rng.uniform(-1, 1, calls * m * k); b = rng.uniform(-1, 1, calls * k * n); c = rng.uniform(-1, 1, calls * m * n)
    ref = c.copy()
    for _ in range(3):
        orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, m * k, k * n, m * n, calls)
    fn = _dispatch(xs, np.float64, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    i = np.arange(calls)
    _run(loop, fn, da, db, dc, i * m * k, i * k * n, i * m * n, 8)
    blob, desc = xs.descriptor(xs.F64, m, n, k)
    assert 0 == xs.lib().libxsmm_amd_gemm_batch_strided(desc, xs.dptr(da), xs.dptr(db), xs.dptr(dc), m * k, k * n, m * n, calls)
    _run(loop, fn, da, db, dc, i * m * k, i * k * n, i * m * n, 8)
    xs.lib().libxsmm_amd_flush()
    assert 0 == xs.lib().libxsmm_amd_synchronize()
    assert np.array_equal(dc.cpu().numpy(), ref)


def test_calls_during_stream_capture_are_launched(xs, orc, torch_gpu, scalar_kernels):
    """nothing is deferred while a stream is being captured: the graph holds the kernel itself and replays it -- also when
    a burst on that stream was still open when the capture began"""
    torch = torch_gpu
    m = n = k = 9
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, m * k); b = rng.uniform(-1, 1, k * n); c = rng.uniform(-1, 1, m * n)
    ref = c.copy()
    for _ in range(3):
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, a, b, ref)
    fn = _dispatch(xs, np.float64, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    xs.lib().libxsmm_amd_set_stream(C.c_void_p(stream.cuda_stream))
    try:
        xs.call_kernel(fn, da, db, dc)  # opens a burst right in front of the capture (one product outside the graph)
        with torch.cuda.graph(graph, stream=stream):
            xs.call_kernel(fn, da, db, dc)
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
    finally:
        xs.lib().libxsmm_amd_set_stream(None)
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_operator_applied_panel_by_panel(xs, orc, torch_gpu, loop, dtype, beta):
    """libxsmm_?fsspmdm_execute once per 48-column panel (the PyFR driver's loop): the calls that walk along the rows are recorded
    into a burst (one operator launch for all of them); a walk in the other direction, and a second operator in between, fall
    back to a burst per call. Results equal the oracle's per-element chain, without a library call before the copy back."""
    torch = torch_gpu
    L = xs.lib()
    M, K, N, panels = 35, 35, 48, 300
    rng = np.random.default_rng(17)
    A = np.ascontiguousarray(np.where(rng.random((M, K)) < 0.15, rng.uniform(-1, 1, (M, K)), 0.0).astype(dtype))
    A[4, :] = 0.0
    ntot = N * panels
    B = rng.uniform(-1, 1, (K, ntot)).astype(dtype); Cin = rng.uniform(-1, 1, (M, ntot)).astype(dtype)
    ref = Cin.copy()
    h = orc.Fsspmdm(A, M, N, K, K, ntot, ntot, 1.0, beta, have_avx512=True)
    for p in range(panels):
        h.execute(B.reshape(-1)[p * N:], ref.reshape(-1)[p * N:])
    h.close()
    create = L.libxsmm_dfsspmdm_create if dtype == np.float64 else L.libxsmm_sfsspmdm_create
    execute = L.libxsmm_dfsspmdm_execute if dtype == np.float64 else L.libxsmm_sfsspmdm_execute
    destroy = L.libxsmm_dfsspmdm_destroy if dtype == np.float64 else L.libxsmm_sfsspmdm_destroy
    hd = create(M, N, K, K, ntot, ntot, 1.0, beta, xs.dptr(A))
    assert hd
    fexec = C.cast(execute, C.c_void_p)
    rows = [r for r in range(M) if r != 4]  # (the empty row: see test_sparse_gpu.py::test_fsspmdm_synthetic)
    dB = torch.from_numpy(B).cuda()
    try:
        for backwards in (0, 1):
            dC = torch.from_numpy(Cin).cuda()
            torch.cuda.synchronize()
            before = L.libxsmm_amd_launch_count()
            loop.panel_loop(fexec, hd, dB.data_ptr(), dC.data_ptr(), ntot, N, B.itemsize, backwards)
            got = dC.cpu().numpy()  # ordered behind the calls on the stream
            launches = L.libxsmm_amd_launch_count() - before
            assert np.array_equal(got[rows], ref[rows]), backwards
            if not backwards:
                assert launches <= 8, launches  # recorded: a handful of operator launches for 300 panels
                assert xs.last_kernel().endswith("_jit_operator_deferred")
    finally:
        destroy(hd)


def test_threads_with_their_own_streams_and_bursts(xs, orc, torch_gpu, loop, scalar_kernels):
    """four threads, each with its own stream, loop over their own products at the same time: every thread has its own ring of
    bursts (ctypes releases the GIL inside the C loop, so the calls really interleave)"""
    import threading
    torch = torch_gpu
    L = xs.lib()
    m = n = k = 11
    calls, nthreads = 12000, 4
    fn = _dispatch(xs, np.float64, m, n, k)
    rng = np.random.default_rng(23)
    data, errors = [], []
    for t in range(nthreads):
        a = rng.uniform(-1, 1, calls * m * k); b = rng.uniform(-1, 1, calls * k * n); c = rng.uniform(-1, 1, calls * m * n)
        ref = c.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, m * k, k * n, m * n, calls)
        data.append((a, b, c, ref))
    dev = [tuple(torch.from_numpy(x).cuda() for x in d[:3]) for d in data]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    torch.cuda.synchronize()
    i = np.arange(calls)
    out = [None] * nthreads

    def work(t):
        try:
            L.libxsmm_amd_set_stream(C.c_void_p(streams[t].cuda_stream))
            L.libxsmm_amd_defer_begin()  # (the bracket is a per-thread setting)
            da, db, dc = dev[t]
            for _ in range(3):
                _run(loop, fn, da, db, dc, i * m * k, i * k * n, i * m * n, 8)
            streams[t].synchronize()  # the thread's own wait, no library call
            out[t] = dc.cpu().numpy()
            L.libxsmm_amd_defer_end()
            L.libxsmm_amd_set_stream(None)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(nthreads):
        a, b, c, ref = data[t]
        want = c.copy()
        for _ in range(3):
            orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, want, m * k, k * n, m * n, calls)
        assert np.array_equal(out[t], want), t


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_call_sequences_over_one_pool(xs, orc, torch_gpu, loop, scalar_kernels, seed):
    """random sequences of calls of three kernels whose operands are blocks of one pool: a call may read blocks earlier calls wrote
    and write blocks again that were written or read before -- whatever is recorded side by side, the pool must end up as the
    sequential loop leaves it (stretches of independent calls alternate with stretches full of dependences)"""
    torch = torch_gpu
    rng = np.random.default_rng(seed)
    nblk, sz, calls = 96, 256, 3000
    pool = rng.uniform(-0.05, 0.05, nblk * sz)  # (small: results feed later products, larger values grow without bound)
    ks = (4, 8, 16)
    fns = [_dispatch(xs, np.float64, 16, 16, kk) for kk in ks]
    which = np.zeros(calls, dtype=np.int64); ia = np.zeros(calls, dtype=np.int64); ib = np.zeros(calls, dtype=np.int64); ic = np.zeros(calls, dtype=np.int64)
    i = 0
    while i < calls:
        stretch = int(rng.integers(20, 200)); kern = int(rng.integers(0, 3)); tangled = rng.random() < 0.4
        for j in range(min(stretch, calls - i)):
            if rng.random() < 0.1:
                kern = int(rng.integers(0, 3))
            if tangled:
                c = int(rng.integers(0, nblk)); a = int(rng.integers(0, nblk)); b = int(rng.integers(0, nblk))
                while a == c:
                    a = int(rng.integers(0, nblk))
                while b == c:
                    b = int(rng.integers(0, nblk))
            else:  # inputs from the upper half, outputs walk through the lower half
                c = (i + j) % (nblk // 2); a = nblk // 2 + int(rng.integers(0, nblk // 2)); b = nblk // 2 + int(rng.integers(0, nblk // 2))
            which[i + j], ia[i + j], ib[i + j], ic[i + j] = kern, a, b, c
        i += stretch
    ref = pool.copy()
    for j in range(calls):
        kk = ks[which[j]]
        orc.smm(orc.FMA, 0, 16, 16, kk, 16, kk, 16, ref[ia[j] * sz:ia[j] * sz + 16 * kk].copy(), ref[ib[j] * sz:ib[j] * sz + kk * 16].copy(), ref[ic[j] * sz:(ic[j] + 1) * sz])
    dp = torch.from_numpy(pool).cuda()
    # runs of one kernel go through the C loop (bursts), the kernel changes between them
    j = 0
    while j < calls:
        e = j
        while e < calls and which[e] == which[j]:
            e += 1
        _run(loop, fns[which[j]], dp, dp, dp, ia[j:e] * sz, ib[j:e] * sz, ic[j:e] * sz, 8)
        j = e
    torch.cuda.synchronize()
    assert np.all(np.isfinite(ref))
    assert np.array_equal(dp.cpu().numpy().view(np.uint64), ref.view(np.uint64))
