"""C-ABI and host-logic tests that run without a GPU: the library loads, exports every symbol include/*.h declares,
descriptors / dispatch / registry / helpers behave like the reference, and compute entry points fail loudly (no CPU
fallback) when no HIP device is present.

Reference tests mirrored: tests/threadsafety.c (parallel dispatch, same pointer on re-dispatch :94-128, release :175-183),
tests/gemmflags.c (LIBXSMM_GEMM_PFLAGS handling of 'N'/'T'/'C' and NULL :38-71), tests/headeronly.c (one registry).
"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names, data = set(), set()
    for hdr in ("libxsmm.h", "libxsmm_amd.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"LIBXSMM_API(?:EXT)?\s+[^;{]*?\b(libxsmm_\w+)\s*\(", text):
            names.add(m.group(1))
        for m in re.finditer(r"LIBXSMM_APIVAR\(\s*[^)]*?\b(libxsmm_\w+)\s*\)", text):
            data.add(m.group(1))
    return names, data


def test_library_exports_every_declared_symbol(xs):
    names, data = declared_symbols()
    assert len(names) > 90 and data == {"libxsmm_ninit", "libxsmm_verbosity"}
    out = subprocess.run(["nm", "-D", "--defined-only", xs.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = sorted((names | data) - exported)
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    L = xs.lib()
    for n in names:
        assert getattr(L, n) is not None
    assert C.c_int.in_dll(L, "libxsmm_verbosity").value == L.libxsmm_get_verbosity()
    assert C.c_uint.in_dll(L, "libxsmm_ninit").value >= 1  # constructor ran libxsmm_init


def test_header_compiles_as_c89_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include <libxsmm.h>\nint main(void) { libxsmm_descriptor_blob b; libxsmm_spmdm_handle h; (void)b; (void)h;\n'
                   '  return (int)(sizeof(libxsmm_descriptor_blob) != 64 || sizeof(libxsmm_gemm_blob) != 128 || LIBXSMM_GEMM_FLAGS(\'N\', \'T\') != 2); }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        exe = tmp_path / ("t_" + cc)
        subprocess.run([cc, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c" if cc == "gcc" else "c++", str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_c_caller_links_and_uses_dispatch(xs, tmp_path):
    """A plain C translation unit written against the reference API links against libxsmm.so unchanged."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <libxsmm.h>
#include <string.h>
int main(void) {
  libxsmm_mmkernel_info info; libxsmm_registry_info reg; libxsmm_descriptor_blob blob;
  const int m = 23, n = 23, k = 23; const double alpha = 1, beta = 1, two = 2;
  libxsmm_dmmfunction f, g; libxsmm_xmmfunction x;
  libxsmm_init();
  f = libxsmm_dmmdispatch(m, n, k, NULL, NULL, NULL, &alpha, &beta, NULL, NULL);
  g = libxsmm_dmmdispatch(m, n, k, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
  if (NULL == f || f != g) return 1;
  if (NULL != libxsmm_dmmdispatch(m, n, k, NULL, NULL, NULL, &two, &beta, NULL, NULL)) return 2; /* alpha != 1 */
  x.dmm = f;
  if (EXIT_SUCCESS != libxsmm_get_mmkernel_info(x, &info, NULL)) return 3;
  if (info.m != 23 || info.lda != 23 || info.iprecision != LIBXSMM_GEMM_PRECISION_F64 || 0 != (info.flags & LIBXSMM_GEMM_FLAG_BETA_0)) return 4;
  if (NULL == libxsmm_gemm_descriptor_dinit(&blob, LIBXSMM_GEMM_PRECISION_F32, 4, 4, 4, 4, 4, 4, 1.0, 0.0, LIBXSMM_GEMM_FLAGS('N', 'N'), 0)) return 5;
  if (EXIT_SUCCESS != libxsmm_get_registry_info(&reg) || reg.size < 1) return 6;
  { double a[4]; LIBXSMM_MATINIT(double, 42, a, 2, 2, 2, 1.0); if (a[0] != 43.0 || a[3] != 43.0 / 4) return 7; }
  libxsmm_finalize();
  return 0;
}''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(xs.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lxsmm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_custom_allocator_and_gemm_print(xs, tmp_path):
    """libxsmm_set_default_allocator (include/libxsmm_malloc.h:53-64): plain and context forms, buffers released by the
    allocator that made them; libxsmm_gemm_print's two text forms (src/libxsmm_gemm.c:608-622)."""
    src = tmp_path / "alloc.c"
    src.write_text(r'''
#include <libxsmm.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>
static int n_malloc = 0, n_free = 0, n_ctx = 0;
static void* my_malloc(size_t size) { ++n_malloc; return malloc(size); }
static void my_free(void* p) { ++n_free; free(p); }
static void* ctx_malloc(void* ctx, size_t size) { ++*(int*)ctx; return malloc(size); }
static void ctx_free(void* ctx, void* p) { --*(int*)ctx; free(p); }
int main(int argc, char* argv[]) {
  libxsmm_malloc_function mf; libxsmm_free_function ff; void* ctx = NULL; void *p, *q;
  const libxsmm_blasint m = 23, n = 24, k = 25; const double alpha = 1, beta = 0; double a[1], b[1], c[1];
  FILE* out;
  mf.function = my_malloc; ff.function = my_free;
  if (EXIT_SUCCESS != libxsmm_set_default_allocator(NULL, mf, ff)) return 1;
  p = libxsmm_aligned_malloc(1000, 256);
  if (NULL == p || 0 != ((uintptr_t)p % 256) || 1 != n_malloc) return 2;
  memset(p, 7, 1000);
  mf.ctx_form = ctx_malloc; ff.ctx_form = ctx_free;
  if (EXIT_SUCCESS != libxsmm_set_default_allocator(&n_ctx, mf, ff)) return 3; /* switch while p is pending */
  q = libxsmm_malloc(64);
  if (NULL == q || 1 != n_ctx) return 4;
  libxsmm_free(p); if (1 != n_free) return 5;   /* p goes back through my_free */
  libxsmm_free(q); if (0 != n_ctx) return 6;
  if (EXIT_SUCCESS != libxsmm_get_default_allocator(&ctx, &mf, &ff) || ctx != &n_ctx || mf.ctx_form != ctx_malloc) return 7;
  mf.function = my_malloc; ff.function = NULL;
  if (EXIT_SUCCESS == libxsmm_set_default_allocator(NULL, mf, ff)) return 8; /* not a pair */
  mf.function = NULL;
  if (EXIT_SUCCESS != libxsmm_set_default_allocator(NULL, mf, ff)) return 9; /* back to the built-in allocator */
  p = libxsmm_malloc(128); if (NULL == p || 1 != n_malloc) return 10; libxsmm_free(p);
  out = fopen(argv[1], "w"); if (NULL == out || argc < 2) return 11;
  libxsmm_gemm_print(out, LIBXSMM_GEMM_PRECISION_F64, "N", "T", &m, &n, &k, &alpha, NULL, NULL, NULL, NULL, &beta, NULL, NULL);
  fprintf(out, "\n");
  libxsmm_gemm_print(out, LIBXSMM_GEMM_PRECISION_F64, "N", "N", &m, &n, &k, &alpha, a, NULL, b, NULL, &beta, c, NULL);
  fclose(out);
  return 0;
}''')
    exe = tmp_path / "alloc"
    libdir = os.path.dirname(xs.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lxsmm",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    txt = tmp_path / "print.txt"
    assert subprocess.run([str(exe), str(txt)]).returncode == 0
    lines = txt.read_text().splitlines()
    assert lines[0] == "dgemm(trans=NT mnk=23,24,25 ldx=23,24,23 a,b=1,0)"
    assert lines[1].startswith("dgemm('N', 'N', 23/*m*/, 24/*n*/, 25/*k*/,") and "/*ldb*/" in lines[3] and lines[4].endswith("23/*ldc*/)")


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "samples")), reason="the reference tree is only mounted in the build container")
def test_reference_sample_programs_compile_and_link_unchanged(xs, tmp_path):
    """The drop-in claim at the source level: the reference's own sample callers (read in place, never copied, never run)
    compile against include/libxsmm.h and link against libxsmm.so without a change -- samples/smm (specialized, dispatched),
    samples/cp2k, samples/spmdm, samples/blocked_gemm (compile only: its gold needs a Fortran BLAS), samples/edge,
    samples/utilities/wrap (dgemm.c relinked with --wrap=dgemm_, autobatch.c) and samples/xgemm (kernel.c, xgemm.c)."""
    libdir = os.path.dirname(xs.LIB_PATH)
    inc = ["-I", os.path.join(ROOT, "include")]
    link = ["-L", libdir, "-lxsmm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    S = os.path.join(REFERENCE, "samples")
    jobs = [(["g++", "-std=c++11"], ["smm/specialized.cpp"], True), (["g++", "-std=c++11"], ["smm/dispatched.cpp"], True),
            (["g++", "-std=c++11"], ["cp2k/cp2k.cpp"], True), (["gcc", "-std=gnu99"], ["spmdm/spmdm.c"], True),
            (["gcc", "-std=gnu99"], ["blocked_gemm/blocked_gemm.c"], False),
            # BLAS call wrapper (relinked with --wrap), auto-batching of intercepted calls, the kernel harness incl. low precision
            (["gcc", "-std=gnu99", "-Wl,--wrap=dgemm_,--wrap=sgemm_"], ["utilities/wrap/dgemm.c"], True),
            (["gcc", "-std=gnu99"], ["utilities/wrap/autobatch.c"], True),
            (["gcc", "-std=gnu99", "-Werror=implicit-function-declaration"], ["xgemm/kernel.c"], True),
            (["gcc", "-std=gnu99", "-Werror=implicit-function-declaration", "-Wl,--wrap=dgemm_,--wrap=sgemm_"], ["xgemm/xgemm.c"], True)]  # (its gold calls dgemm_)
    for name in ("asparse_srsoa", "bsparse_srsoa", "bsparse_scsoa", "dense_rmacsoa", "dense_rmbcsoa"):
        jobs.append((["gcc", "-std=gnu99", "-I", os.path.join(S, "edge")], ["edge/%s.c" % name, "edge/edge_proxy_common.c"], True))
    for cc, srcs, do_link in jobs:
        out = tmp_path / os.path.basename(srcs[0]).split(".")[0]
        cmd = cc + ["-O0", "-fopenmp"] + inc + [os.path.join(S, f) for f in srcs]
        cmd += (["-o", str(out)] + link) if do_link else ["-fsyntax-only"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, (srcs, res.stderr[-3000:])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is only present in the build container")
def test_reference_unit_tests_pass_against_this_library(xs, tmp_path):
    """The reference's own unit tests for the parts of the interface that need no device -- tests/gemmflags.c (transpose flag
    macros), tests/vla.c (array views), tests/rng.c (distribution of libxsmm_rng_*), tests/matdiff.c, tests/threadsafety.c
    (concurrent dispatch, registry, release) and tests/headeronly.c (+_aux: one registry across translation units) -- are
    compiled from where they lie against include/ and libxsmm.so and RUN here; tests/gemm.c (needs the GPU to run) is compiled
    and linked with the BLAS wrapper. Nothing of the reference is copied or modified."""
    libdir = os.path.dirname(xs.LIB_PATH)
    T = os.path.join(REFERENCE, "tests")
    base = ["gcc", "-std=gnu99", "-O0", "-fopenmp", "-I", os.path.join(ROOT, "include")]
    link = ["-L", libdir, "-lxsmm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    for name, extra, run in (("gemmflags", [], True), ("vla", [], True), ("rng", [], True), ("matdiff", [], True), ("threadsafety", [], True),
                             ("headeronly", [os.path.join(T, "headeronly_aux.c")], True),
                             ("gemm", ["-Wl,--wrap=dgemm_,--wrap=sgemm_"], False)):
        exe = tmp_path / name
        res = subprocess.run(base + [os.path.join(T, name + ".c")] + extra + ["-o", str(exe)] + link, capture_output=True, text=True)
        assert res.returncode == 0, (name, res.stderr[-3000:])
        if run:
            res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
            assert res.returncode == 0, (name, res.returncode, res.stdout[-1000:], res.stderr[-1000:])


def test_descriptor_rules(xs):
    """include/libxsmm_generator.h:36-39: NULL unless alpha == 1, beta in {0,1}, no TRANS_A; beta == 0 sets FLAG_BETA_0."""
    L = xs.lib()
    assert xs.descriptor(xs.F64, 23, 23, 23)[1]
    assert not xs.descriptor(xs.F64, 23, 23, 23, alpha=2.0)[1]
    assert not xs.descriptor(xs.F64, 23, 23, 23, beta=0.5)[1]
    assert not xs.descriptor(xs.F64, 23, 23, 23, flags=xs.FLAG_TRANS_A)[1]
    blob, d = xs.descriptor(xs.F32, 5, 6, 7, 8, 9, 10, 1.0, 0.0, xs.FLAG_TRANS_B, 0)
    raw = C.string_at(C.byref(blob), 28)
    assert raw[0] == xs.F32                                        # datatype
    assert int.from_bytes(raw[1:3], "little") == (xs.FLAG_TRANS_B | xs.FLAG_BETA_0)
    assert [int.from_bytes(raw[3 + 4 * i:7 + 4 * i], "little") for i in range(6)] == [5, 6, 7, 8, 9, 10]  # packed m,n,k,lda,ldb,ldc
    # generic init reads alpha/beta in the input precision; NULL selects LIBXSMM_ALPHA/BETA = 1
    b2 = xs.DescriptorBlob()
    assert L.libxsmm_gemm_descriptor_init(C.byref(b2), xs.F32, 4, 4, 4, 4, 4, 4, None, None, 0, 0)
    half = C.c_float(0.5)
    assert not L.libxsmm_gemm_descriptor_init(C.byref(b2), xs.F32, 4, 4, 4, 4, 4, 4, C.byref(half), None, 0, 0)


def test_dispatch_rules_and_kernel_info(xs):
    L = xs.lib()
    f1 = L.libxsmm_smmdispatch(32, 32, 32, None, None, None, None, None, None, None)
    f2 = L.libxsmm_smmdispatch(32, 32, 32, xs.iptr(32), xs.iptr(32), xs.iptr(32), None, None, None, None)
    f3 = L.libxsmm_smmdispatch(32, 32, 32, xs.iptr(40), None, None, None, None, None, None)
    assert f1 and f1 == f2 and f3 and f3 != f1
    assert not L.libxsmm_smmdispatch(32, 32, 32, xs.iptr(16), None, None, None, None, None, None)  # lda < m (generator_gemm.c:211)
    assert not L.libxsmm_smmdispatch(32, 32, 32, None, xs.iptr(8), None, None, None, None, None)   # ldb < k
    assert not L.libxsmm_smmdispatch(32, 32, 32, None, None, xs.iptr(31), None, None, None, None)  # ldc < m
    assert not L.libxsmm_wimmdispatch(8, 8, 7, None, None, None, None, None, None, None)           # low precision: k must be even (tests/test_lowp.py)
    assert not L.libxsmm_xmmdispatch(None)
    # TRANS_B: ldb is checked against n
    assert L.libxsmm_dmmdispatch(8, 16, 4, None, xs.iptr(16), None, None, None, xs.iptr(xs.FLAG_TRANS_B), None)
    assert not L.libxsmm_dmmdispatch(8, 16, 4, None, xs.iptr(4), None, None, None, xs.iptr(xs.FLAG_TRANS_B), None)
    r = L.libxsmm_dmmdispatch_reducebatch(13, 13, 13, None, None, None, None, None, None, None)
    d = L.libxsmm_dmmdispatch(13, 13, 13, None, None, None, None, None, None, None)
    assert r and d and r != d
    info = xs.MMKernelInfo(); size = C.c_size_t()
    assert 0 == L.libxsmm_get_mmkernel_info(r, C.byref(info), C.byref(size))
    assert (info.m, info.n, info.k, info.lda, info.ldb, info.ldc) == (13, 13, 13, 13, 13, 13) and (info.flags & xs.FLAG_BATCH_REDUCE) and size.value > 0
    kind = C.c_int(-1)
    assert 0 == L.libxsmm_get_kernel_kind(d, C.byref(kind)) and kind.value == 0
    junk = (C.c_char * 8)()
    assert 0 != L.libxsmm_get_kernel_kind(C.cast(junk, C.c_void_p), C.byref(kind)) and kind.value == 7
    reg = xs.RegistryInfo()
    assert 0 == L.libxsmm_get_registry_info(C.byref(reg)) and reg.size >= 4 and reg.capacity >= reg.size
    # LIBXSMM_TARGET=generic disables JIT => dispatch returns NULL (reference behaviour noted in SURVEY 8(c))
    L.libxsmm_set_target_arch(b"generic")
    assert L.libxsmm_get_target_arch() == b"generic"
    assert not L.libxsmm_dmmdispatch(7, 7, 7, None, None, None, None, None, None, None)
    L.libxsmm_set_target_arch(b"0")
    assert L.libxsmm_get_target_arch() == b"gfx950"
    assert L.libxsmm_dmmdispatch(7, 7, 7, None, None, None, None, None, None, None)


def test_threadsafe_dispatch(xs):
    """tests/threadsafety.c:94-128: many random dispatches from concurrent threads; re-dispatch yields the same pointer."""
    L = xs.lib()
    rng = np.random.default_rng(0)
    shapes = [tuple(int(v) for v in rng.integers(1, 65, 3)) for _ in range(800)]
    results = [[None] * len(shapes) for _ in range(8)]

    def worker(t):
        for i, (m, n, k) in enumerate(shapes):
            results[t][i] = L.libxsmm_dmmdispatch(m, n, k, None, None, None, None, None, None, None)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in threads]; [t.join() for t in threads]
    for i in range(len(shapes)):
        assert results[0][i] and all(results[t][i] == results[0][i] for t in range(8))
    distinct = {}
    for s, f in zip(shapes, results[0]):
        assert distinct.setdefault(s, f) == f
    assert len(set(distinct.values())) == len(distinct)  # one kernel (thunk) per descriptor
    L.libxsmm_release_kernel(results[0][0])               # registered kernels stay valid (reference: warning only)
    assert L.libxsmm_dmmdispatch(*shapes[0], None, None, None, None, None, None, None) == results[0][0]


def test_helpers(xs):
    L = xs.lib()
    libc = C.CDLL(None); libc.drand48.restype = C.c_double
    L.libxsmm_rng_set_seed(1); libc.srand48(C.c_long(1))
    assert [L.libxsmm_rng_f64() for _ in range(16)] == [libc.drand48() for _ in range(16)]  # src/libxsmm_rng.c:256
    assert all(L.libxsmm_rng_u32(10) < 10 for _ in range(100))
    for x in (0, 1, 2, 3, 4, 15, 16, 17, 10**12, 2**63 - 1, 155344 * 160 // 240):
        r = L.libxsmm_isqrt_u64(x)
        assert r * r <= x < (r + 1) * (r + 1)
    for n in (2, 3, 10, 1024, 4232, 65536):
        s = L.libxsmm_shuffle(n)
        assert 0 < s < n and np.gcd(s, n) == 1
    assert [L.libxsmm_typesize(t) for t in (0, 1, 2, 4, 5, 6)] == [8, 4, 2, 4, 2, 1]
    t0 = L.libxsmm_timer_tick(); t1 = L.libxsmm_timer_tick()
    assert L.libxsmm_timer_duration(t0, t1) >= 0
    p = L.libxsmm_aligned_malloc(1000, 64)
    assert p and p % 64 == 0
    C.memset(p, 1, 1000); L.libxsmm_free(p)
    # matdiff: same definition as the oracle's restatement of src/template/libxsmm_matdiff.tpl.c
    ref = np.arange(12, dtype=np.float64).reshape(3, 4); tst = ref.copy(); tst[1, 2] += 0.5
    info = xs.MatdiffInfo()
    assert 0 == L.libxsmm_matdiff(C.byref(info), xs.F64, 4, 3, xs.dptr(ref), xs.dptr(tst), None, None)
    assert info.linf_abs == 0.5 and (info.m, info.n) == (2, 1)
    assert abs(info.normf_rel - np.sqrt(0.25 / np.sum(ref * ref))) < 1e-15
    tst[0, 0] = np.nan
    assert 0 == L.libxsmm_matdiff(C.byref(info), xs.F64, 4, 3, xs.dptr(ref), xs.dptr(tst), None, None) and np.isinf(info.linf_abs)


def test_compute_fails_loudly_without_gpu(xs, capfd):
    """No CPU fallback: on a box without a HIP device every compute entry point reports failure and says why."""
    L = xs.lib()
    if L.libxsmm_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    a = np.ones(32 * 32, dtype=np.float32); b = a.copy(); c = a.copy()
    blob, desc = xs.descriptor(xs.F32, 32, 32, 32)
    assert 0 != L.libxsmm_amd_gemm_batch_strided(desc, xs.dptr(a), xs.dptr(b), xs.dptr(c), 0, 0, 0, 1)
    fn = L.libxsmm_smmdispatch(32, 32, 32, None, None, None, None, None, None, None)
    xs.call_kernel(fn, a, b, c)  # must not crash, must not compute
    assert np.all(c == 1.0)
    assert not L.libxsmm_dfsspmdm_create(16, 16, 16, 16, 16, 16, 1.0, 1.0, xs.dptr(np.eye(16)))
    assert not L.libxsmm_amd_spmdm_batch_create(64, 48, 64, 4)
    err = capfd.readouterr().err
    assert "no CPU fallback" in err


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under libxsmm-1_amd/ or include/ may include, link or call it."""
    bad = []
    for base in ("libxsmm-1_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"xsmm_oracle|liboracle|oracle_binding|xo_[a-z]+\(", text):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["nm", "-D", os.path.join(ROOT, "libxsmm-1_amd", "lib", "libxsmm.so")], capture_output=True, text=True).stdout
    assert " xo_" not in out


def test_illegal_leading_dimensions_are_an_error_not_a_launch(xs):
    """What the reference hands to BLAS (general products, batches of them) BLAS rejects with xerbla when a leading dimension is
    smaller than the rows it has to hold; here the call fails before anything is launched (a kernel would read beyond the operands)."""
    L = xs.lib()
    m, n, k, batch = 40, 8, 12, 3
    a = np.zeros(batch * 19 * k); b = np.zeros(batch * k * n); c = np.zeros(batch * m * n)
    lda, ldb, ldc = C.c_int(19), C.c_int(k), C.c_int(m)  # lda < m
    alpha, beta = C.c_double(1.0), C.c_double(1.0)
    sa = (np.arange(batch) * 19 * k).astype(np.int32); sb = (np.arange(batch) * k * n).astype(np.int32); sc = (np.arange(batch) * m * n).astype(np.int32)
    L.libxsmm_mmbatch_blas.restype = C.c_int
    L.libxsmm_mmbatch_blas.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rc = L.libxsmm_mmbatch_blas(xs.F64, xs.F64, b"N", b"N", m, n, k, C.addressof(alpha), a.ctypes.data, C.addressof(lda), b.ctypes.data, C.addressof(ldb),
                                C.addressof(beta), c.ctypes.data, C.addressof(ldc), 0, 4, sa.ctypes.data, sb.ctypes.data, sc.ctypes.data, batch)
    assert rc != 0
    assert not np.any(c)
