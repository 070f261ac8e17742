"""SOA kernels for fused runs: libxsmm_create_{xcsr,xcsc,rm_ac,rm_bc}_soa, their batch form and their text generators.

Reference material: samples/edge/{asparse_srsoa,bsparse_srsoa,bsparse_scsoa,dense_rmacsoa,dense_rmbcsoa}.c (call
signatures, descriptors and the dense gold loops :120-135 / :143-158), the EDGE operator files samples/edge/mats (a few
copied as data to tests/golden/mtx/edge; shapes per samples/edge/test_matops.sh: 9 quantities, degree-3 basis K=20,
N=10), src/libxsmm_main.c:2423-2520. Operands are [row][col][v], v = 8 (fp64) / 16 (fp32) innermost.
CPU part: the oracle's restatement against the samples' gold loops, the generated text compiles for gfx950. GPU part:
bit-exact against the oracle (one call through the kernel pointer, and the batch form).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDGE = os.path.join(ROOT, "tests", "golden", "mtx", "edge")
CALL = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)


def dense_of(ptr, idx, vals, rows, cols, csr):
    d = np.zeros((rows, cols))
    for major in range(len(ptr) - 1):
        for p in range(ptr[major], ptr[major + 1]):
            if csr:
                d[major, idx[p]] = vals[p]
            else:
                d[idx[p], major] = vals[p]
    return d


def soa_cases(orc):
    """(name, form, m, n, k, lda, ldb, ldc, ptr, idx, vals, dense operator) -- forms: asparse (CSR), bsparse_csr, bsparse_csc"""
    for f in ("tet4_starMatrix_csr.mtx", "tet4_fluxMatrix_csr_sp.mtx"):   # 9 x 9 operators acting on [9][K][v]
        ptr, idx, vals, r, c, _ = orc.read_csr(os.path.join(EDGE, f))
        yield ("asparse_" + f[5:9], "asparse", r, 20, c, 0, 20, 20, ptr, idx, vals, dense_of(ptr, idx, vals, r, c, True))
        yield ("asparse_ld_" + f[5:9], "asparse", r, 20, c, 0, 23, 22, ptr, idx, vals, dense_of(ptr, idx, vals, r, c, True))
    for f in ("tet4_3_fluxN_0", "tet4_3_fluxT_0", "tet4_3_stiffT_0"):       # B sparse (K x N), A is [9][K][v]
        ptr, idx, vals, r, c, _ = orc.read_csr(os.path.join(EDGE, f + "_csr.mtx"))
        yield ("bcsr_" + f[7:], "bsparse_csr", 9, c, r, r, 0, c, ptr, idx, vals, dense_of(ptr, idx, vals, r, c, True))
        ptr, idx, vals, r, c, _ = orc.read_csc(os.path.join(EDGE, f + "_csc.mtx"))
        yield ("bcsc_" + f[7:], "bsparse_csc", 9, c, r, r + 2, 0, c + 1, ptr, idx, vals, dense_of(ptr, idx, vals, r, c, False))


def test_oracle_soa_against_sample_gold_loops(orc):
    rng = np.random.default_rng(2)
    for dtype, v in ((np.float64, 8), (np.float32, 16)):
        tol = 1e-12 if dtype == np.float64 else 1e-5
        for (name, form, m, n, k, lda, ldb, ldc, ptr, idx, vals, dense) in soa_cases(orc):
            sv = vals.astype(dtype)
            if form == "asparse":
                b = rng.uniform(-1, 1, (k, ldb, v)).astype(dtype); c = rng.uniform(-1, 1, (m, ldc, v)).astype(dtype)
                ref = c.copy(); orc.soa_csr_asparse(0, m, n, k, ldb, ldc, v, ptr, idx, sv, b, ref)
                gold = c.astype(np.float64); gold[:, :n, :] += np.einsum("mk,knv->mnv", dense, b[:, :n, :].astype(np.float64))
            else:
                a = rng.uniform(-1, 1, (m, lda, v)).astype(dtype); c = rng.uniform(-1, 1, (m, ldc, v)).astype(dtype)
                ref = c.copy(); orc.soa_bsparse(0, form.endswith("csr"), m, n, k, lda, ldc, v, ptr, idx, a, sv, ref)
                gold = c.astype(np.float64); gold[:, :n, :] += np.einsum("mkv,kn->mnv", a[:, :k, :].astype(np.float64), dense)
            assert np.max(np.abs(ref - gold)) <= tol * max(1.0, np.max(np.abs(gold))), name
            assert np.array_equal(ref[:, n:, :], c[:, n:, :])  # padding columns of C are not touched
        # dense forms (samples/edge/dense_rmacsoa.c:60-80)
        m, n, k = 9, 10, 20
        a = rng.uniform(-1, 1, (m, k, v)).astype(dtype); b = rng.uniform(-1, 1, (k, n)).astype(dtype); c = rng.uniform(-1, 1, (m, n, v)).astype(dtype)
        ref = c.copy(); orc.soa_rm_ac(0, m, n, k, k, n, n, v, a, b, ref)
        gold = c.astype(np.float64) + np.einsum("mkv,kn->mnv", a.astype(np.float64), b.astype(np.float64))
        assert np.max(np.abs(ref - gold)) <= tol * np.max(np.abs(gold))
        a = rng.uniform(-1, 1, (m, k)).astype(dtype); b = rng.uniform(-1, 1, (k, n, v)).astype(dtype)
        ref = c.copy(); orc.soa_rm_bc(orc.FLAG_BETA_0, m, n, k, k, n, n, v, a, b, ref)
        gold = np.einsum("mk,knv->mnv", a.astype(np.float64), b.astype(np.float64))
        assert np.max(np.abs(ref - gold)) <= tol * np.max(np.abs(gold))


def test_soa_text_and_front_door(xs, orc, tmp_path):
    """The text entry points (libxsmm_generator_spgemm_{csr,csc}_soa_kernel) and the file front door with the SOA kinds
    (i_is_csr = 2: CSR, > 9: CSC; src/generator_spgemm.c:271,400-409); the written file must be valid HIP."""
    L = xs.lib()
    assert L.libxsmm_amd_soa_width(xs.F64) == 8 and L.libxsmm_amd_soa_width(xs.F32) == 16
    ptr, idx, vals, r, c, _ = orc.read_csr(os.path.join(EDGE, "tet4_starMatrix_csr.mtx"))
    code = xs.GeneratedCode()
    blob, d = xs.descriptor(xs.F64, 9, 20, 9, 0, 20, 20, 1.0, 1.0, 0, 0)
    L.libxsmm_generator_spgemm_csr_soa_kernel(C.byref(code), d, b"gfx950", xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
    assert code.last_error == 0 and code.text().count("= XACC(") == len(vals)
    code.release()
    for (lda, ldb, ldc, expect) in ((0, 19, 20, 90008), (0, 20, 19, 90009), (0, 0, 20, 90010), (8, 0, 20, 90007)):
        code = xs.GeneratedCode()
        blob, d = xs.descriptor(xs.F64, 9, 20, 9, lda, ldb, ldc, 1.0, 1.0, 0, 0)
        L.libxsmm_generator_spgemm_csr_soa_kernel(C.byref(code), d, b"gfx950", xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
        assert code.last_error == expect, (lda, ldb, ldc)
    code = xs.GeneratedCode()  # CSC: only B may be sparse
    blob, d = xs.descriptor(xs.F64, 9, 20, 9, 0, 20, 20, 1.0, 1.0, 0, 0)
    L.libxsmm_generator_spgemm_csc_soa_kernel(C.byref(code), d, b"gfx950", xs.dptr(idx), xs.dptr(ptr), xs.dptr(vals))
    assert code.last_error == 90010
    out = tmp_path / "soa.hip"
    script = r'''
import importlib, sys
sys.path.insert(0, %r)
xs = importlib.import_module("libxsmm-1_amd")
L = xs.lib()
out, edge = sys.argv[1], sys.argv[2]
blob, d = xs.descriptor(xs.F64, 9, 20, 9, 0, 20, 20, 1.0, 1.0, 0, 0)
L.libxsmm_generator_spgemm(out.encode(), b"star_soa", d, b"gfx950", (edge + "/tet4_starMatrix_csr.mtx").encode(), 2)
blob, d = xs.descriptor(xs.F32, 9, 10, 20, 20, 0, 10, 1.0, 0.0, 0, 0)
L.libxsmm_generator_spgemm((out + ".2").encode(), b"fluxn_soa", d, b"gfx950", (edge + "/tet4_3_fluxN_0_csc.mtx").encode(), 10)
''' % ROOT
    res = subprocess.run([sys.executable, "-c", script, str(out), EDGE], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    for f, name in ((out, "void star_soa("), (tmp_path / "soa.hip.2", "void fluxn_soa(")):
        text = open(f).read()
        assert name in text and "__builtin_fma" in text
        hip = str(f) + ".hip"
        os.rename(f, hip)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-x", "hip", "-c", hip, "-o", hip + ".o"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_soa_kernels_bitexact(xs, orc, torch_gpu, dtype, beta):
    torch = torch_gpu
    L = xs.lib()
    prec = xs.F64 if dtype == np.float64 else xs.F32
    v = L.libxsmm_amd_soa_width(prec)
    oflags = orc.FLAG_BETA_0 if beta == 0.0 else 0
    batch = 29
    rng = np.random.default_rng(6)
    for (name, form, m, n, k, lda, ldb, ldc, ptr, idx, vals64, dense) in soa_cases(orc):
        vals = vals64.astype(dtype)
        blob, d = xs.descriptor(prec, m, n, k, lda, ldb, ldc, 1.0, beta, 0, 0)
        if form == "asparse":
            fn = L.libxsmm_create_xcsr_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
            dense_shape = (k, ldb, v)
        elif form == "bsparse_csr":
            fn = L.libxsmm_create_xcsr_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
            dense_shape = (m, lda, v)
        else:
            fn = L.libxsmm_create_xcsc_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
            dense_shape = (m, lda, v)
        assert fn, name
        try:
            x = rng.uniform(-1, 1, (batch,) + dense_shape).astype(dtype)
            cin = rng.uniform(-1, 1, (batch, m, ldc, v)).astype(dtype)
            ref = cin.copy()
            for i in range(batch):
                if form == "asparse":
                    orc.soa_csr_asparse(oflags, m, n, k, ldb, ldc, v, ptr, idx, vals, x[i], ref[i])
                else:
                    orc.soa_bsparse(oflags, form.endswith("csr"), m, n, k, lda, ldc, v, ptr, idx, x[i], vals, ref[i])
            dv, dx, dc = torch.from_numpy(vals).cuda(), torch.from_numpy(x).cuda(), torch.from_numpy(cin).cuda()
            sd, sc = int(np.prod(dense_shape)), m * ldc * v
            a_arg, b_arg = (dv, dx) if form == "asparse" else (dx, dv)
            assert 0 == L.libxsmm_amd_kernel_execute_batch(fn, xs.dptr(a_arg), xs.dptr(b_arg), xs.dptr(dc), sd, sc, batch)
            torch.cuda.synchronize()
            assert xs.last_kernel() == ("soa_asparse_text" if form == "asparse" else "soa_bsparse_text")
            assert np.array_equal(dc.cpu().numpy(), ref), name
            # one product through the bare kernel pointer, host operands (the way samples/edge call it)
            hc = cin[0].copy()
            ha, hb = (vals, x[0]) if form == "asparse" else (x[0], vals)
            CALL(fn)(xs.dptr(ha), xs.dptr(hb), xs.dptr(hc))
            assert np.array_equal(hc, ref[0]), name
        finally:
            L.libxsmm_release_kernel(fn)
    # dense forms
    m, n, k, lda, ldb, ldc = 9, 10, 20, 21, 12, 11
    blob, d = xs.descriptor(prec, m, n, k, lda, ldb, ldc, 1.0, beta, 0, 0)
    for which in ("rm_ac", "rm_bc"):
        fn = getattr(L, "libxsmm_create_%s_soa" % which)(d)
        assert fn
        try:
            if which == "rm_ac":
                x = rng.uniform(-1, 1, (batch, m, lda, v)).astype(dtype); plain = rng.uniform(-1, 1, (k, ldb)).astype(dtype)
            else:
                x = rng.uniform(-1, 1, (batch, k, ldb, v)).astype(dtype); plain = rng.uniform(-1, 1, (m, lda)).astype(dtype)
            cin = rng.uniform(-1, 1, (batch, m, ldc, v)).astype(dtype)
            ref = cin.copy()
            for i in range(batch):
                if which == "rm_ac":
                    orc.soa_rm_ac(oflags, m, n, k, lda, ldb, ldc, v, x[i], plain, ref[i])
                else:
                    orc.soa_rm_bc(oflags, m, n, k, lda, ldb, ldc, v, plain, x[i], ref[i])
            dp, dx, dc = torch.from_numpy(plain).cuda(), torch.from_numpy(x).cuda(), torch.from_numpy(cin).cuda()
            a_arg, b_arg = (dx, dp) if which == "rm_ac" else (dp, dx)
            assert 0 == L.libxsmm_amd_kernel_execute_batch(fn, xs.dptr(a_arg), xs.dptr(b_arg), xs.dptr(dc), int(np.prod(x.shape[1:])), m * ldc * v, batch)
            torch.cuda.synchronize()
            assert np.array_equal(dc.cpu().numpy(), ref), which
        finally:
            L.libxsmm_release_kernel(fn)
    # rejected descriptors: neither operand sparse / leading dimension too small
    blob, d = xs.descriptor(prec, 9, 20, 9, 9, 20, 20, 1.0, 1.0, 0, 0)
    ptr, idx, vals, r, c, _ = orc.read_csr(os.path.join(EDGE, "tet4_starMatrix_csr.mtx"))
    assert not L.libxsmm_create_xcsr_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
    blob, d = xs.descriptor(prec, 9, 20, 9, 0, 10, 20, 1.0, 1.0, 0, 0)
    assert not L.libxsmm_create_xcsr_soa(d, xs.dptr(ptr), xs.dptr(idx), xs.dptr(vals))
