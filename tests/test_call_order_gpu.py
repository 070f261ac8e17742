"""Stream order at the boundary: a dispatched kernel called per product (samples/smm/specialized.cpp:172-190,
samples/cp2k/cp2k.cpp:341-346) by a caller who ALSO queues work of its own on the same stream between two calls -- an
asynchronous copy that rewrites A, a memset of C, a copy that reads C, a torch operation. The reference's kernel is
synchronous, so every interleaving is legal there; here the results must equal the oracle's sequential program bit for
bit.

Two modes are covered:
  * default (no bracket): every call is a launch of its own -- nothing the library does may move a call across the
    caller's work, however fast the calls follow each other (the loops below are C, compiled here: no pause between a
    call, the foreign operation and the next call);
  * the opt-in bracket libxsmm_amd_defer_begin/end with the documented libxsmm_amd_flush() in front of the caller's own
    work (include/libxsmm_amd.h).
"""
import ctypes as C
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ORDER_C = r"""
#include <hip/hip_runtime_api.h>
typedef void (*fn3)(const void*, const void*, void*);
typedef void (*vfn)(void);
/* round r: kernel(a, b, c[2r]); copy a <- alt[r] on the stream; kernel(a, b, c[2r+1]) */
int rewrite_a(fn3 f, vfn flush, void* stream, char* a, const char* alt, const char* b, char* c, long long rounds, size_t abytes, size_t cbytes)
{
  long long r; int e = 0;
  for (r = 0; r < rounds; ++r) {
    f(a, b, c + (2 * r) * cbytes);
    if (flush) flush();
    e |= (int)hipMemcpyAsync(a, alt + r * abytes, abytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    f(a, b, c + (2 * r + 1) * cbytes);
  }
  return e;
}
/* round r: kernel(a[r], b[r], c[r]); memset c[r] on the stream; kernel(a[r], b[r], c[r])  ->  c[r] = a[r] * b[r] from zero */
int clear_c(fn3 f, vfn flush, void* stream, const char* a, const char* b, char* c, long long rounds, size_t abytes, size_t bbytes, size_t cbytes)
{
  long long r; int e = 0;
  for (r = 0; r < rounds; ++r) {
    f(a + r * abytes, b + r * bbytes, c + r * cbytes);
    if (flush) flush();
    e |= (int)hipMemsetAsync(c + r * cbytes, 0, cbytes, (hipStream_t)stream);
    f(a + r * abytes, b + r * bbytes, c + r * cbytes);
  }
  return e;
}
/* round r: kernel(a[r], b[r], c[r]); snap[r] <- c[r] on the stream; kernel(a[r], b[r], c[r])  ->  snap holds the first sum only */
int read_c(fn3 f, vfn flush, void* stream, const char* a, const char* b, char* c, char* snap, long long rounds, size_t abytes, size_t bbytes, size_t cbytes)
{
  long long r; int e = 0;
  for (r = 0; r < rounds; ++r) {
    f(a + r * abytes, b + r * bbytes, c + r * cbytes);
    if (flush) flush();
    e |= (int)hipMemcpyAsync(snap + r * cbytes, c + r * cbytes, cbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    f(a + r * abytes, b + r * bbytes, c + r * cbytes);
  }
  return e;
}
/* panels of a fixed operator (libxsmm_?fsspmdm_execute, samples/pyfr/pyfr_driver_asp_reg.c:300-308) with B's next panel
 * rewritten on the stream between two calls */
typedef void (*fnx)(const void*, const void*, void*);
int panels_rewrite_b(fnx execute, vfn flush, void* stream, const void* handle, char* b, const char* balt, char* c, long long npanels, long long panel_cols,
                     long long rows_b, long long ldb, int ts)
{
  long long p, r; int e = 0;
  for (p = 0; p < npanels; ++p) {
    execute(handle, b + p * panel_cols * ts, c + p * panel_cols * ts);
    if (p + 1 < npanels) {
      if (flush) flush();
      for (r = 0; r < rows_b; ++r) e |= (int)hipMemcpyAsync(b + (r * ldb + (p + 1) * panel_cols) * ts, balt + (r * ldb + (p + 1) * panel_cols) * ts, panel_cols * ts,
                                                           hipMemcpyDeviceToDevice, (hipStream_t)stream);
    }
  }
  return e;
}
"""


@pytest.fixture(scope="module")
def order(tmp_path_factory):
    d = tmp_path_factory.mktemp("order")
    src = d / "order.c"
    src.write_text(ORDER_C)
    so = d / "order.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src), "-o", str(so),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    lib = C.CDLL(str(so))
    vp, ll, sz = C.c_void_p, C.c_longlong, C.c_size_t
    lib.rewrite_a.argtypes = [vp, vp, vp, vp, vp, vp, vp, ll, sz, sz]; lib.rewrite_a.restype = C.c_int
    lib.clear_c.argtypes = [vp, vp, vp, vp, vp, vp, ll, sz, sz, sz]; lib.clear_c.restype = C.c_int
    lib.read_c.argtypes = [vp, vp, vp, vp, vp, vp, vp, ll, sz, sz, sz]; lib.read_c.restype = C.c_int
    lib.panels_rewrite_b.argtypes = [vp, vp, vp, vp, vp, vp, vp, ll, ll, ll, ll, C.c_int]; lib.panels_rewrite_b.restype = C.c_int
    return lib


def _dispatch(xs, dtype, m, n, k):
    f = xs.lib().libxsmm_dmmdispatch if dtype == np.float64 else xs.lib().libxsmm_smmdispatch
    fn = f(m, n, k, None, None, None, None, None, None, None)
    assert fn
    return fn


@pytest.fixture()
def scalar_kernels(xs):
    old = xs.lib().libxsmm_amd_set_mfma(0)  # bit-exact against the oracle's fma chain
    yield
    xs.lib().libxsmm_amd_set_mfma(old)


class Mode:
    """default: no bracket, no flush. bracket: libxsmm_amd_defer_begin/end around the loop, libxsmm_amd_flush in front of the
    caller's own stream work. Either on the null stream or on a stream of the caller's."""

    def __init__(self, xs, torch, bracket, own_stream):
        self.L = xs.lib(); self.torch = torch; self.bracket = bracket
        self.stream = torch.cuda.Stream() if own_stream else None
        self.flush = C.cast(self.L.libxsmm_amd_flush, C.c_void_p) if bracket else None
        self.handle = C.c_void_p(self.stream.cuda_stream) if own_stream else None

    def __enter__(self):
        self.torch.cuda.synchronize()
        self.L.libxsmm_amd_set_stream(self.handle)
        if self.bracket:
            self.L.libxsmm_amd_defer_begin()
        return self

    def __exit__(self, *exc):
        if self.bracket:
            self.L.libxsmm_amd_defer_end()
        self.L.libxsmm_amd_set_stream(None)
        self.torch.cuda.synchronize()
        return False


MODES = [pytest.param((False, False), id="default-nullstream"), pytest.param((False, True), id="default-ownstream"),
         pytest.param((True, False), id="bracket+flush-nullstream"), pytest.param((True, True), id="bracket+flush-ownstream")]


def test_deferral_is_off_unless_asked_for(xs, orc, torch_gpu, scalar_kernels):
    """an unchanged caller gets a launch per call: the launch counter moves with every call and no burst kernel appears"""
    torch = torch_gpu
    L = xs.lib()
    assert 0 == L.libxsmm_amd_defer_active()
    m = n = k = 16
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, m * k); b = rng.uniform(-1, 1, k * n); c = rng.uniform(-1, 1, m * n)
    ref = c.copy()
    for _ in range(5):
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, a, b, ref)
    fn = _dispatch(xs, np.float64, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    before = L.libxsmm_amd_launch_count()
    for _ in range(5):
        xs.call_kernel(fn, da, db, dc)
    assert 5 == L.libxsmm_amd_launch_count() - before
    assert xs.last_kernel() != "smm_deferred_calls"
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("dtype,shape", [(np.float64, (23, 23, 23)), (np.float32, (32, 32, 32)), (np.float64, (5, 7, 3))])
def test_caller_rewrites_a_between_two_calls(xs, orc, torch_gpu, order, scalar_kernels, mode, dtype, shape):
    """kernel(a, b, c1); hipMemcpyAsync(a <- a2, stream); kernel(a, b, c2) back to back, 300 rounds"""
    torch = torch_gpu
    m, n, k = shape
    rounds = 300
    rng = np.random.default_rng(rounds + m)
    a0 = rng.uniform(-1, 1, m * k).astype(dtype); alt = rng.uniform(-1, 1, rounds * m * k).astype(dtype)
    b = rng.uniform(-1, 1, k * n).astype(dtype); c = rng.uniform(-1, 1, 2 * rounds * m * n).astype(dtype)
    ref = c.copy(); cur = a0
    for r in range(rounds):
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, cur, b, ref[(2 * r) * m * n:(2 * r + 1) * m * n])
        cur = alt[r * m * k:(r + 1) * m * k]
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, cur, b, ref[(2 * r + 1) * m * n:(2 * r + 2) * m * n])
    fn = _dispatch(xs, dtype, m, n, k)
    da, dalt, db, dc = (torch.from_numpy(x).cuda() for x in (a0, alt, b, c))
    with Mode(xs, torch, *mode) as md:
        assert 0 == order.rewrite_a(fn, md.flush, md.handle, da.data_ptr(), dalt.data_ptr(), db.data_ptr(), dc.data_ptr(), rounds, a0.nbytes, m * n * c.itemsize)
    assert np.array_equal(dc.cpu().numpy(), ref)
    assert np.array_equal(da.cpu().numpy(), alt[(rounds - 1) * m * k:])


@pytest.mark.parametrize("mode", MODES)
def test_caller_clears_c_between_two_calls(xs, orc, torch_gpu, order, scalar_kernels, mode):
    """kernel(a, b, c); hipMemsetAsync(c, 0); kernel(a, b, c): C holds exactly one product, summed from zero"""
    torch = torch_gpu
    m, n, k = 13, 13, 13
    rounds = 400
    rng = np.random.default_rng(4)
    a = rng.uniform(-1, 1, rounds * m * k); b = rng.uniform(-1, 1, rounds * k * n); c = rng.uniform(-1, 1, rounds * m * n)
    ref = np.zeros_like(c)
    orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, ref, m * k, k * n, m * n, rounds)
    fn = _dispatch(xs, np.float64, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    with Mode(xs, torch, *mode) as md:
        assert 0 == order.clear_c(fn, md.flush, md.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), rounds, m * k * 8, k * n * 8, m * n * 8)
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", MODES)
def test_caller_reads_c_between_two_calls(xs, orc, torch_gpu, order, scalar_kernels, mode):
    """kernel(a, b, c); hipMemcpyAsync(snapshot <- c); kernel(a, b, c): the snapshot holds the first sum, C both"""
    torch = torch_gpu
    m, n, k = 16, 8, 24
    rounds = 400
    rng = np.random.default_rng(6)
    a = rng.uniform(-1, 1, rounds * m * k).astype(np.float32); b = rng.uniform(-1, 1, rounds * k * n).astype(np.float32)
    c = rng.uniform(-1, 1, rounds * m * n).astype(np.float32)
    one = c.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, one, m * k, k * n, m * n, rounds)
    two = one.copy(); orc.gemm_batch_strided(orc.FMA, 0, m, n, k, m, k, m, a, b, two, m * k, k * n, m * n, rounds)
    fn = _dispatch(xs, np.float32, m, n, k)
    da, db, dc = (torch.from_numpy(x).cuda() for x in (a, b, c))
    snap = torch.full_like(dc, float("nan"))
    with Mode(xs, torch, *mode) as md:
        assert 0 == order.read_c(fn, md.flush, md.handle, da.data_ptr(), db.data_ptr(), dc.data_ptr(), snap.data_ptr(), rounds, m * k * 4, k * n * 4, m * n * 4)
    assert np.array_equal(snap.cpu().numpy(), one)
    assert np.array_equal(dc.cpu().numpy(), two)


@pytest.mark.parametrize("bracket", [False, True])
def test_torch_operations_between_two_calls(xs, orc, torch_gpu, scalar_kernels, bracket):
    """the same through torch on torch's current stream: kernel; a.copy_(a2); kernel; c.mul_(2); kernel"""
    torch = torch_gpu
    L = xs.lib()
    m = n = k = 20
    rounds = 50
    rng = np.random.default_rng(8)
    a1 = rng.uniform(-1, 1, m * k); a2 = rng.uniform(-1, 1, m * k); b = rng.uniform(-1, 1, k * n); c = rng.uniform(-1, 1, m * n)
    ref = c.copy()
    for _ in range(rounds):
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, a1, b, ref)
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, a2, b, ref)
        ref *= 0.5
        orc.smm(orc.FMA, 0, m, n, k, m, k, m, a2, b, ref)
    fn = _dispatch(xs, np.float64, m, n, k)
    d1, d2, db, dc = (torch.from_numpy(x).cuda() for x in (a1, a2, b, c))
    da = torch.empty_like(d1)
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    L.libxsmm_amd_set_stream(C.c_void_p(stream.cuda_stream))
    try:
        with torch.cuda.stream(stream):
            if bracket:
                L.libxsmm_amd_defer_begin()
            for _ in range(rounds):
                da.copy_(d1)
                xs.call_kernel(fn, da, db, dc)
                if bracket:
                    L.libxsmm_amd_flush()
                da.copy_(d2)
                xs.call_kernel(fn, da, db, dc)
                if bracket:
                    L.libxsmm_amd_flush()
                dc.mul_(0.5)
                xs.call_kernel(fn, da, db, dc)
                if bracket:
                    L.libxsmm_amd_flush()
            if bracket:
                L.libxsmm_amd_defer_end()
        stream.synchronize()
    finally:
        L.libxsmm_amd_set_stream(None)
    assert np.array_equal(dc.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", MODES)
def test_caller_rewrites_the_next_panel_between_operator_calls(xs, orc, torch_gpu, order, mode):
    """libxsmm_dfsspmdm_execute once per panel (the PyFR loop) with the caller refreshing B's next panel on the stream in between"""
    torch = torch_gpu
    L = xs.lib()
    M, K, N, panels = 35, 35, 48, 60
    rng = np.random.default_rng(12)
    A = np.ascontiguousarray(np.where(rng.random((M, K)) < 0.15, rng.uniform(-1, 1, (M, K)), 0.0))
    for r in range(M):  # no empty row (the beta = 0 quirk is not what this test is about)
        if not A[r].any():
            A[r, r] = 0.5
    ntot = N * panels
    B0 = rng.uniform(-1, 1, (K, ntot)); Balt = rng.uniform(-1, 1, (K, ntot)); Cin = rng.uniform(-1, 1, (M, ntot))
    Bseen = Balt.copy(); Bseen[:, :N] = B0[:, :N]  # panel 0 is read as it was, every later panel after its refresh
    ref = Cin.copy()
    h = orc.Fsspmdm(A, M, N, K, K, ntot, ntot, 1.0, 1.0, have_avx512=True)
    for p in range(panels):
        h.execute(Bseen.reshape(-1)[p * N:], ref.reshape(-1)[p * N:])
    h.close()
    hd = L.libxsmm_dfsspmdm_create(M, N, K, K, ntot, ntot, 1.0, 1.0, xs.dptr(A))
    assert hd
    dB, dBalt, dC = (torch.from_numpy(x).cuda() for x in (B0, Balt, Cin))
    try:
        with Mode(xs, torch, *mode) as md:
            assert 0 == order.panels_rewrite_b(C.cast(L.libxsmm_dfsspmdm_execute, C.c_void_p), md.flush, md.handle, hd, dB.data_ptr(), dBalt.data_ptr(), dC.data_ptr(),
                                               panels, N, K, ntot, 8)
        assert np.array_equal(dC.cpu().numpy(), ref)
    finally:
        L.libxsmm_dfsspmdm_destroy(hd)
