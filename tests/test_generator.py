"""Text generators (libxsmm_generator_spgemm*, libxsmm_generator_gemm_*) and the executable form of the sparse text
kernels.

Reference material: samples/generator/{left_sparse_test_csr,left_sparse_test_csc,right_sparse_test_csc}.mtx with the
shapes of samples/generator/test_xGEMM.sh (left sparse M=84,N=9,K=84; right sparse M=20,N=9,K=9), the self-check of
samples/generator/validation.c:203-209, error codes of src/generator_common.h:267-320 and the mux rules of
src/generator_spgemm.c:55-145. CPU part: every generated text is valid gfx950 code (hiprtc / hipcc, no device); GPU part:
the compiled kernels reproduce the oracle's statement-by-statement arithmetic bit for bit, in both flavours (fused
multiply-add, and multiply-then-add as the C text is written).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tests", "golden", "mtx", "generator")


def sparse_desc(xs, prec, m, n, k, lda, ldb, ldc, beta=1.0):
    return xs.descriptor(prec, m, n, k, lda, ldb, ldc, 1.0, beta, 0, 0)


def cases(orc):
    """(name, is_csr, m, n, k, lda, ldb, ldc, ptr, idx, vals) for the three fixture kernels."""
    rowptr, colidx, vals, M, K, _ = orc.read_csr(os.path.join(GEN, "left_sparse_test_csr.mtx"))
    yield ("csr_asparse", 1, M, 9, K, 0, 9, 9, rowptr, colidx, vals)
    yield ("csr_asparse_ld", 1, M, 9, K, 0, 12, 11, rowptr, colidx, vals)       # padded rows: beta == 0 clears ldc entries
    yield ("csr_asparse_kcut", 1, M, 9, 40, 0, 9, 9, rowptr, colidx, vals)       # entries with col >= k are dropped (:136)
    colptr, rowidx, vals, M, K, _ = orc.read_csc(os.path.join(GEN, "left_sparse_test_csc.mtx"))
    yield ("csc_asparse", 0, M, 9, K, 0, K, M, colptr, rowidx, vals)
    yield ("csc_asparse_ld", 0, M, 9, K, 0, K + 3, M + 5, colptr, rowidx, vals)
    colptr, rowidx, vals, K, N, _ = orc.read_csc(os.path.join(GEN, "right_sparse_test_csc.mtx"))
    yield ("csc_bsparse", 0, 20, N, K, 20, 0, 20, colptr, rowidx, vals)
    yield ("csc_bsparse_ld", 0, 20, N, K, 24, 0, 22, colptr, rowidx, vals)


def args_for(is_csr, ptr, idx):
    """(row_idx, column_idx) in the reference's argument order."""
    return (ptr, idx) if is_csr else (idx, ptr)


def test_sparse_text_compiles_for_gfx950(xs, orc):
    L = xs.lib()
    buf = C.create_string_buffer(1 << 20)
    for (name, is_csr, m, n, k, lda, ldb, ldc, ptr, idx, vals) in cases(orc):
        for prec in (xs.F64, xs.F32):
            for beta in (1.0, 0.0):
                blob, d = sparse_desc(xs, prec, m, n, k, lda, ldb, ldc, beta)
                assert d
                ri, ci = args_for(is_csr, ptr, idx)
                for fma in (1, 0):
                    rc = L.libxsmm_amd_spgemm_source(d, is_csr, xs.dptr(ri), xs.dptr(ci), fma, buf, len(buf), 1)
                    if rc == -1:
                        pytest.skip("libhiprtc is not available here")
                    assert rc == 0, (name, prec, beta, fma)
                src = buf.value.decode()
                # one statement per non-zero that survives the k/m cut
                limit = k if name.startswith(("csr_asparse", "csc_bsparse")) else m
                assert src.count("= XACC(") == int(np.sum(idx < limit)), name
                assert ("contract(off)" in src) and "xsmm_spgemm_op" in src


def test_generator_entry_points_and_errors(xs, orc):
    L = xs.lib()
    rowptr, colidx, vals, M, K, _ = orc.read_csr(os.path.join(GEN, "left_sparse_test_csr.mtx"))
    code = xs.GeneratedCode()
    blob, d = sparse_desc(xs, xs.F64, M, 9, K, 0, 9, 9)
    L.libxsmm_generator_spgemm_csr_kernel(C.byref(code), d, b"gfx950", xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(vals))
    assert code.last_error == 0 and code.code_size > 0 and code.buffer_size == code.code_size + 1 and code.code_type == 0
    first = code.text()
    assert first.count("= XACC(") == len(vals) and "c[" in first
    # appending keeps what is there (string-buffer semantics of libxsmm_append_code_as_string)
    L.libxsmm_generator_spgemm_csr_kernel(C.byref(code), d, b"gfx950", xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(vals))
    assert code.text() == first + first
    code.release()
    # leading-dimension checks and the "which operand is sparse" rule (src/generator_spgemm.c:55-145)
    for (lda, ldb, ldc, csr, expect) in ((0, 8, 9, True, 90008), (0, 9, 8, True, 90009), (5, 9, 9, True, 90010), (0, 0, 9, True, 90010),
                                          (84, 0, 9, True, 90010),   # B sparse in CSR form: not available
                                          (0, 80, 84, False, 90008), (0, 84, 80, False, 90009), (80, 0, 84, False, 90007), (84, 0, 80, False, 90009)):
        code = xs.GeneratedCode()
        blob, d = sparse_desc(xs, xs.F64, M, 9, K, lda, ldb, ldc)
        f = L.libxsmm_generator_spgemm_csr_kernel if csr else L.libxsmm_generator_spgemm_csc_kernel
        f(C.byref(code), d, b"gfx950", xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(vals))
        assert code.last_error == expect and not code.generated_code, (lda, ldb, ldc, csr)
        assert ("#%d" % expect) in L.libxsmm_strerror(expect).decode()
    code = xs.GeneratedCode(); code.code_type = 2  # binary buffers cannot take text
    blob, d = sparse_desc(xs, xs.F64, M, 9, K, 0, 9, 9)
    L.libxsmm_generator_spgemm_csr_kernel(C.byref(code), d, b"gfx950", xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(vals))
    assert code.last_error == 90003
    # csr_reg: values baked in, a complete translation unit
    code = xs.GeneratedCode()
    blob, d = sparse_desc(xs, xs.F64, M, 8, K, 0, 8, 8)
    L.libxsmm_generator_spgemm_csr_reg_kernel(C.byref(code), d, b"gfx950", xs.dptr(rowptr), xs.dptr(colidx), xs.dptr(vals))
    assert code.last_error == 0 and "xsmm_csr_op" in code.text()
    code.release()
    # dense: specialised template for tight shapes, plain form otherwise
    for (m, n, k, lda, ldb, ldc, marker) in ((23, 23, 23, 23, 23, 23, "#define XM 23"), (23, 23, 23, 32, 32, 32, "plain form"), (80, 70, 90, 80, 90, 80, "plain form")):
        code = xs.GeneratedCode()
        blob, d = xs.descriptor(xs.F64, m, n, k, lda, ldb, ldc)
        L.libxsmm_generator_gemm_kernel(C.byref(code), d, b"gfx950")
        assert code.last_error == 0 and marker in code.text() and "xsmm_smm_op" in code.text()
        code.release()
    code = xs.GeneratedCode()
    blob, d = xs.descriptor(xs.F64, 23, 23, 23, 20, 23, 23)
    L.libxsmm_generator_gemm_kernel(C.byref(code), d, b"gfx950")
    assert code.last_error == 90007


def parse_mtx(path):
    """independent reading of a MatrixMarket coordinate file: (rows, cols, [(row, col, value)] 0-based, in file order)"""
    rows = cols = nnz = None
    ent = []
    with open(path) as f:
        for line in f:
            if line.startswith("%"):
                continue
            t = line.split()
            if rows is None:
                rows, cols, nnz = int(t[0]), int(t[1]), int(t[2])
            else:
                ent.append((int(t[0]) - 1, int(t[1]) - 1, float(t[2])))
    assert len(ent) == nnz
    return rows, cols, ent


def all_fixture_files():
    out = []
    for root, _, files in os.walk(os.path.join(ROOT, "tests", "golden", "mtx")):
        out += [os.path.join(root, f) for f in sorted(files) if f.endswith(".mtx")]
    return sorted(f for f in out if "coordinate" in open(f).readline())  # (the *-de.mtx files are dense "array" files)


def test_product_reader_known_answers(xs, orc):
    """The library's own MatrixMarket reader (what libxsmm_generator_spgemm reads its input with; reference
    src/generator_spgemm_csr_reader.c:46-170, generator_spgemm_csc_reader.c) against known answers: every fixture the
    reference ships for this path (samples/generator/*.mtx, samples/pyfr/mats/**, samples/edge/mats) read as CSR and as
    CSC where the file's grouping allows it, compared with an independent parse of the text and with the oracle's reader."""
    files = all_fixture_files()
    assert len(files) >= 19
    checked = 0
    for path in files:
        rows, cols, ent = parse_mtx(path)
        for is_csr in (True, False):
            major = [e[0] if is_csr else e[1] for e in ent]
            if major != sorted(major):
                continue  # the file is grouped the other way (the reader assumes grouping, :147-149)
            rc, ptr, idx, val, r, c, z = xs.read_mtx(path, is_csr)
            assert rc == 0 and (r, c, z) == (rows, cols, len(ent)), path
            counts = np.bincount(major, minlength=rows if is_csr else cols)
            assert np.array_equal(ptr, np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)), path   # empty majors back-filled
            assert np.array_equal(idx, np.array([e[1] if is_csr else e[0] for e in ent], dtype=np.uint32)), path
            assert np.array_equal(val, np.array([e[2] for e in ent])), path                                   # bit-equal doubles
            optr, oidx, oval, o_r, o_c, o_z = (orc.read_csr if is_csr else orc.read_csc)(path)
            assert np.array_equal(ptr, optr) and np.array_equal(idx, oidx) and np.array_equal(val, oval) and (o_r, o_c, o_z) == (r, c, z)
            checked += 1
    assert checked >= 19
    # first lines of samples/generator/left_sparse_test_csr.mtx: "1 2 2", "1 6 1"
    rc, ptr, idx, val, r, c, z = xs.read_mtx(os.path.join(GEN, "left_sparse_test_csr.mtx"), True)
    assert (r, c, z) == (84, 84, 686) and (idx[0], val[0], idx[1], val[1]) == (1, 2.0, 5, 1.0)
    rc, ptr, idx, val, r, c, z = xs.read_mtx(os.path.join(GEN, "right_sparse_test_csc.mtx"), False)
    assert (r, c, z) == (9, 9, 24)


def test_product_reader_malformed_files(xs, tmp_path):
    """error codes of the reference's readers (src/generator_common.h:278-305): missing file, over-long line, header that does
    not parse or holds a zero, element line that does not parse, fewer / more elements than announced; comments and rows
    without entries are fine."""
    def write(name, text):
        f = tmp_path / name
        f.write_text(text)
        return str(f)
    cases = [
        (str(tmp_path / "missing.mtx"), 90035, 90011),
        (write("long.mtx", "3 3 1\n1 1 " + "1" * 600 + "\n"), 90036, 90012),
        (write("zero.mtx", "% c\n0 3 1\n1 1 1.0\n"), 90037, 90013),
        (write("header.mtx", "three 3 1\n1 1 1.0\n"), 90037, 90013),
        (write("elem.mtx", "3 3 2\n1 1 1.0\n2 x 1.0\n"), 90038, 90014),
        (write("short.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n"), 90039, 90015),
        (write("empty.mtx", "% nothing but comments\n"), 90039, 90015),
        (write("many.mtx", "3 3 1\n1 1 1.0\n2 2 2.0\n"), 90038, 90014),   # (the reference would write past its arrays here)
        (write("range.mtx", "3 3 1\n4 4 1.0\n"), 90038, 90014),             # (likewise: an index beyond the header's shape)
    ]
    for path, csr_code, csc_code in cases:
        assert xs.read_mtx(path, True)[0] == csr_code, path
        assert xs.read_mtx(path, False)[0] == csc_code, path
        assert ("#%d" % csr_code) in xs.lib().libxsmm_strerror(csr_code).decode()
    gap = write("gap.mtx", "% comment\n%another\n4 3 2\n1 2 5.0\n4 1 -1.5\n")
    rc, ptr, idx, val, r, c, z = xs.read_mtx(gap, True)
    assert rc == 0 and list(ptr) == [0, 1, 1, 1, 2] and list(idx) == [1, 0] and list(val) == [5.0, -1.5]  # empty rows back-filled (:158-163)
    gapc = write("gapc.mtx", "3 4 2\n2 1 5.0\n1 4 -1.5\n")
    rc, ptr, idx, val, r, c, z = xs.read_mtx(gapc, False)
    assert rc == 0 and list(ptr) == [0, 1, 1, 1, 2] and list(idx) == [1, 0]


def test_file_front_door_text_equals_text_from_arrays(xs, tmp_path):
    """libxsmm_generator_spgemm(file): the kernel text written for a file is the text the array entry points
    (libxsmm_generator_spgemm_{csr,csc,csr_reg}_kernel) emit for an independent parse of that file -- pattern (unrolled
    kernels) and values (csr_reg kernel: values baked in) both go through the library's reader."""
    L = xs.lib()
    jobs = [("left_sparse_test_csr.mtx", 1, xs.F64, (84, 9, 84, 0, 9, 9)), ("left_sparse_test_csc.mtx", 0, xs.F32, (84, 9, 84, 0, 84, 84)),
            ("right_sparse_test_csc.mtx", 0, xs.F64, (20, 9, 9, 20, 0, 20)), ("left_sparse_test_csr.mtx", 3, xs.F64, (84, 8, 84, 0, 8, 8))]
    for fname, mode, prec, (m, n, k, lda, ldb, ldc) in jobs:
        path = os.path.join(GEN, fname)
        rows, cols, ent = parse_mtx(path)
        is_csr = mode in (1, 3)
        major = [e[0] if is_csr else e[1] for e in ent]
        ptr = np.concatenate([[0], np.cumsum(np.bincount(major, minlength=rows if is_csr else cols))]).astype(np.uint32)
        idx = np.array([e[1] if is_csr else e[0] for e in ent], dtype=np.uint32)
        val = np.array([e[2] for e in ent])
        blob, d = sparse_desc(xs, prec, m, n, k, lda, ldb, ldc)
        code = xs.GeneratedCode()
        if mode == 1:
            L.libxsmm_generator_spgemm_csr_kernel(C.byref(code), d, b"gfx950", xs.dptr(ptr), xs.dptr(idx), xs.dptr(val))
        elif mode == 3:
            L.libxsmm_generator_spgemm_csr_reg_kernel(C.byref(code), d, b"gfx950", xs.dptr(ptr), xs.dptr(idx), xs.dptr(val))
        else:
            L.libxsmm_generator_spgemm_csc_kernel(C.byref(code), d, b"gfx950", xs.dptr(idx), xs.dptr(ptr), xs.dptr(val))
        assert code.last_error == 0
        from_arrays = code.text()
        code.release()
        out = tmp_path / ("k%d_%s.hip" % (mode, fname))
        L.libxsmm_generator_spgemm(str(out).encode(), b"routine", d, b"gfx950", path.encode(), mode)
        from_file = out.read_text()
        if mode == 3:
            # (a file gets the runtime header included once; otherwise the same text)
            assert from_file.replace("#include <hip/hip_runtime.h>\n", "", 1) == from_arrays.replace("xsmm_csr_op", "routine")
        else:
            assert from_arrays in from_file and from_file.count("= XACC(") == from_arrays.count("= XACC(") > 0


@pytest.mark.gpu
def test_kernels_built_from_the_product_reader(xs, orc, torch_gpu):
    """A kernel whose pattern and values come from the library's own reader (not the oracle's) against the oracle's arithmetic
    on an independent parse: the csr_asparse fixture as an executable text kernel, bit for bit."""
    torch = torch_gpu
    L = xs.lib()
    path = os.path.join(GEN, "left_sparse_test_csr.mtx")
    rc, ptr, idx, val, M, K, nnz = xs.read_mtx(path, True)
    assert rc == 0
    n, batch = 9, 37
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, batch * K * n); c = rng.uniform(-1, 1, batch * M * n)
    rows, cols, ent = parse_mtx(path)
    optr = np.concatenate([[0], np.cumsum(np.bincount([e[0] for e in ent], minlength=rows))]).astype(np.uint32)
    oidx = np.array([e[1] for e in ent], dtype=np.uint32); oval = np.array([e[2] for e in ent])
    ref = c.copy()
    for i in range(batch):
        orc.csr_asparse(orc.FMA, 0, M, n, K, n, n, optr, oidx, oval, b[i * K * n:(i + 1) * K * n], ref[i * M * n:(i + 1) * M * n])
    blob, d = sparse_desc(xs, xs.F64, M, n, K, 0, n, n)
    h = L.libxsmm_amd_spgemm_create(d, 1, xs.dptr(ptr), xs.dptr(idx), 1)
    assert h
    dv, db, dc = torch.from_numpy(val).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda()
    assert 0 == L.libxsmm_amd_spgemm_execute_batch(h, xs.dptr(dv), xs.dptr(db), xs.dptr(dc), K * n, M * n, batch)
    torch.cuda.synchronize()
    L.libxsmm_amd_spgemm_destroy(h)
    assert np.array_equal(dc.cpu().numpy(), ref)


def hipcc_compiles(path):
    res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-x", "hip", "-c", str(path), "-o", str(path) + ".o"],
                         capture_output=True, text=True)
    return res.returncode == 0, res.stderr[-2000:]


def test_file_front_doors(xs, tmp_path):
    """libxsmm_generator_spgemm / libxsmm_generator_gemm_inlineasm append kernels to a source file (the generator driver's
    job, src/libxsmm_generator_gemm_driver.c); the file as a whole must be valid HIP. Errors terminate the process, as in
    the reference (src/generator_spgemm.c:424-446) -- exercised in a child process."""
    out = tmp_path / "kernels.hip"
    script = r'''
import ctypes as C, importlib, sys
sys.path.insert(0, %r)
xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
out, gen, mode = sys.argv[1], sys.argv[2], sys.argv[3]
if mode == "ok":
    blob, d = xs.descriptor(xs.F64, 84, 9, 84, 0, 9, 9, 1.0, 1.0, 0, 0)
    L.libxsmm_generator_spgemm(out.encode(), b"left_csr", d, b"gfx950", (gen + "/left_sparse_test_csr.mtx").encode(), 1)
    blob, d = xs.descriptor(xs.F32, 84, 9, 84, 0, 84, 84, 1.0, 0.0, 0, 0)
    L.libxsmm_generator_spgemm(out.encode(), b"left_csc", d, b"gfx950", (gen + "/left_sparse_test_csc.mtx").encode(), 0)
    blob, d = xs.descriptor(xs.F64, 84, 8, 84, 0, 8, 8, 1.0, 1.0, 0, 0)
    L.libxsmm_generator_spgemm(out.encode(), b"left_reg", d, b"gfx950", (gen + "/left_sparse_test_csr.mtx").encode(), 3)
elif mode == "dense":
    blob, d = xs.descriptor(xs.F64, 23, 23, 23, 32, 32, 32)
    L.libxsmm_generator_gemm_inlineasm(out.encode(), b"dense_plain", d, b"gfx950")
elif mode == "dense_tight":
    blob, d = xs.descriptor(xs.F32, 13, 23, 32)
    L.libxsmm_generator_gemm_directasm(out.encode(), b"dense_13_23_32", d, b"gfx950")
elif mode == "missing":
    blob, d = xs.descriptor(xs.F64, 84, 9, 84, 0, 9, 9, 1.0, 1.0, 0, 0)
    L.libxsmm_generator_spgemm(out.encode(), b"k", d, b"gfx950", b"/nonexistent.mtx", 1)
elif mode == "badld":
    blob, d = xs.descriptor(xs.F64, 84, 9, 84, 0, 4, 9, 1.0, 1.0, 0, 0)
    L.libxsmm_generator_spgemm(out.encode(), b"k", d, b"gfx950", (gen + "/left_sparse_test_csr.mtx").encode(), 1)
print("survived")
''' % ROOT
    def run(mode, target):
        return subprocess.run([sys.executable, "-c", script, str(target), GEN, mode], capture_output=True, text=True)
    res = run("ok", out)
    assert res.returncode == 0 and "survived" in res.stdout, res.stderr
    text = out.read_text()
    # the two unrolled kernels define T/XACC once each: keep them in separate files for compilation
    assert text.count("void left_csr(") == 1 and text.count("void left_csc(") == 1 and "left_reg" in text
    parts = text.split("// generated by libxsmm-amd")
    assert len(parts) == 4
    for i, part in enumerate(parts[1:]):
        f = tmp_path / ("part%d.hip" % i)
        f.write_text("// generated by libxsmm-amd" + part)
        ok, err = hipcc_compiles(f)
        assert ok, err
    for mode in ("dense", "dense_tight"):
        f = tmp_path / (mode + ".hip")
        res = run(mode, f)
        assert res.returncode == 0, res.stderr
        assert ("void dense_plain(" in f.read_text()) if mode == "dense" else ("dense_13_23_32" in f.read_text() and "xsmm_smm_op" not in f.read_text())
        ok, err = hipcc_compiles(f)
        assert ok, err
    for mode, code in (("missing", "90035"), ("badld", "90008")):
        res = run(mode, tmp_path / "never.hip")
        assert res.returncode != 0 and "survived" not in res.stdout and code in res.stderr
    assert not (tmp_path / "never.hip").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_sparse_text_kernels_bitexact(xs, orc, torch_gpu, dtype, beta):
    torch = torch_gpu
    L = xs.lib()
    prec = xs.F64 if dtype == np.float64 else xs.F32
    oflags = orc.FLAG_BETA_0 if beta == 0.0 else 0
    batch = 37
    rng = np.random.default_rng(12)
    for (name, is_csr, m, n, k, lda, ldb, ldc, ptr, idx, vals64) in cases(orc):
        vals = vals64.astype(dtype)
        blob, d = sparse_desc(xs, prec, m, n, k, lda, ldb, ldc, beta)
        ri, ci = args_for(is_csr, ptr, idx)
        if name.startswith("csr_asparse"):   # row-major B (K x ldb), C (M x ldc)
            dense_shape, c_shape = (k, ldb), (m, ldc)
        elif name.startswith("csc_asparse"):  # col-major B (ldb x N), C (ldc x N)
            dense_shape, c_shape = (n, ldb), (n, ldc)
        else:                                 # col-major A (lda x K), C (ldc x N)
            dense_shape, c_shape = (k, lda), (n, ldc)
        dense = rng.uniform(-1, 1, (batch,) + dense_shape).astype(dtype)
        cin = rng.uniform(-1, 1, (batch,) + c_shape).astype(dtype)
        sd, sc = int(np.prod(dense_shape)), int(np.prod(c_shape))
        for fma in (1, 0):
            arith = orc.FMA if fma else orc.MULADD
            ref = cin.copy()
            for i in range(batch):
                if name.startswith("csr_asparse"):
                    orc.csr_asparse(arith, oflags, m, n, k, ldb, ldc, ptr, idx, vals, dense[i], ref[i])
                elif name.startswith("csc_asparse"):
                    orc.csc_asparse(arith, oflags, m, n, k, ldb, ldc, ptr, idx, vals, dense[i], ref[i])
                else:
                    orc.csc_bsparse(arith, oflags, m, n, k, lda, ldc, ptr, idx, dense[i], vals, ref[i])
            h = L.libxsmm_amd_spgemm_create(d, is_csr, xs.dptr(ri), xs.dptr(ci), fma)
            assert h, name
            try:
                dv, dd, dc = torch.from_numpy(vals).cuda(), torch.from_numpy(dense).cuda(), torch.from_numpy(cin).cuda()
                assert 0 == L.libxsmm_amd_spgemm_execute_batch(h, xs.dptr(dv), xs.dptr(dd), xs.dptr(dc), sd, sc, batch)
                torch.cuda.synchronize()
                assert xs.last_kernel() == "spgemm_" + name.split("_ld")[0].split("_kcut")[0] + "_text"
                got = dc.cpu().numpy()
                # (padding of C: the row-major kernel clears it for beta == 0, the column-major ones leave it alone --
                # the oracle does the same, so the whole array is compared)
                assert np.array_equal(got, ref), (name, fma)
                # host operands are staged
                hc = cin.copy()
                assert 0 == L.libxsmm_amd_spgemm_execute_batch(h, xs.dptr(vals), xs.dptr(dense), xs.dptr(hc), sd, sc, batch)
                assert np.array_equal(hc, ref), (name, fma, "host")
            finally:
                L.libxsmm_amd_spgemm_destroy(h)
    assert not L.libxsmm_amd_spgemm_create(None, 1, None, None, 1)
