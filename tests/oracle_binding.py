"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
The product (libxsmm-1_amd/) never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

MULADD, FMA = 0, 1
FLAG_TRANS_B, FLAG_BETA_0 = 2, 16


def build():
    res = subprocess.run(["make", "-C", ORACLE_DIR], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)
    return ORACLE_LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build()
        _lib = C.CDLL(ORACLE_LIB)
        _lib.xo_rng_f64.restype = C.c_double
        _lib.xo_csr_reg_unique.restype = C.c_int
        _lib.xo_fsspmdm_create.restype = C.c_void_p
    return _lib


def p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _ts(a):
    return 8 if a.dtype == np.float64 else 4


def smm(arith, flags, m, n, k, lda, ldb, ldc, a, b, c):
    """one dense SMM, in place on c (numpy arrays, flat column-major storage)"""
    f = lib().xo_dsmm if a.dtype == np.float64 else lib().xo_ssmm
    f(arith, flags, m, n, k, lda, ldb, ldc, p(a), p(b), p(c))


def gemm_batch_ptr(arith, typesize, flags, m, n, k, lda, ldb, ldc, pa, pb, pc, batchsize, da=8, db=8, dc=8):
    """pointer-array mode: pa/pb/pc are uint64 numpy arrays of addresses"""
    ia = None if da is None else np.array([da], dtype=np.int32)
    ib = None if db is None else np.array([db], dtype=np.int32)
    ic = None if dc is None else np.array([dc], dtype=np.int32)
    return lib().xo_gemm_batch(arith, typesize, flags, m, n, k, lda, ldb, ldc, p(pa), p(pb), p(pc), 0, 0, p(ia), p(ib), p(ic), batchsize)


def gemm_batch_idx(arith, flags, m, n, k, lda, ldb, ldc, a, b, c, index_base, sa, sb, sc, batchsize, index_stride=4):
    return lib().xo_gemm_batch(arith, _ts(a), flags, m, n, k, lda, ldb, ldc, p(a), p(b), p(c), index_base, index_stride,
                               p(sa), p(sb), p(sc), batchsize)


def gemm_batch_strided(arith, flags, m, n, k, lda, ldb, ldc, a, b, c, sa, sb, sc, batch, nthreads=1):
    lib().xo_gemm_batch_strided(arith, _ts(a), flags, m, n, k, lda, ldb, ldc, p(a), p(b), p(c),
                                C.c_longlong(sa), C.c_longlong(sb), C.c_longlong(sc), C.c_longlong(batch), nthreads)


def smm_reduce(arith, flags, m, n, k, lda, ldb, ldc, a_list, b_list, c):
    n_items = len(a_list)
    pa = (C.c_void_p * n_items)(*[x.ctypes.data for x in a_list])
    pb = (C.c_void_p * n_items)(*[x.ctypes.data for x in b_list])
    f = lib().xo_dsmm_reduce if c.dtype == np.float64 else lib().xo_ssmm_reduce
    f(arith, flags, m, n, k, lda, ldb, ldc, pa, pb, p(c), C.c_ulonglong(n_items))


def read_csr(path):
    rp, ci, va = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)(), C.POINTER(C.c_double)()
    r, c, z = C.c_uint(), C.c_uint(), C.c_uint()
    rc = lib().xo_csr_reader(path.encode(), C.byref(rp), C.byref(ci), C.byref(va), C.byref(r), C.byref(c), C.byref(z))
    if rc != 0:
        raise IOError("xo_csr_reader failed for " + path)
    out = (np.ctypeslib.as_array(rp, (r.value + 1,)).copy(), np.ctypeslib.as_array(ci, (z.value,)).copy(),
           np.ctypeslib.as_array(va, (z.value,)).copy(), r.value, c.value, z.value)
    for q in (rp, ci, va):
        lib().xo_free(q)
    return out


def read_csc(path):
    ri, cp, va = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)(), C.POINTER(C.c_double)()
    r, c, z = C.c_uint(), C.c_uint(), C.c_uint()
    rc = lib().xo_csc_reader(path.encode(), C.byref(ri), C.byref(cp), C.byref(va), C.byref(r), C.byref(c), C.byref(z))
    if rc != 0:
        raise IOError("xo_csc_reader failed for " + path)
    out = (np.ctypeslib.as_array(cp, (c.value + 1,)).copy(), np.ctypeslib.as_array(ri, (z.value,)).copy(),
           np.ctypeslib.as_array(va, (z.value,)).copy(), r.value, c.value, z.value)
    for q in (ri, cp, va):
        lib().xo_free(q)
    return out


def read_dense_mtx(path):
    d = C.POINTER(C.c_double)()
    r, c = C.c_uint(), C.c_uint()
    rc = lib().xo_dense_mtx_reader(path.encode(), C.byref(d), C.byref(r), C.byref(c))
    if rc != 0:
        raise IOError("xo_dense_mtx_reader failed for " + path)
    out = np.ctypeslib.as_array(d, (r.value, c.value)).copy()
    lib().xo_free(d)
    return out


def csr_asparse(arith, flags, m, n, k, ldb, ldc, rowptr, colidx, a_vals, b, c):
    f = lib().xo_dcsr_asparse if b.dtype == np.float64 else lib().xo_scsr_asparse
    f(arith, flags, m, n, k, ldb, ldc, p(rowptr), p(colidx), p(a_vals), p(b), p(c))


def csc_bsparse(arith, flags, m, n, k, lda, ldc, colptr, rowidx, a, b_vals, c):
    f = lib().xo_dcsc_bsparse if a.dtype == np.float64 else lib().xo_scsc_bsparse
    f(arith, flags, m, n, k, lda, ldc, p(colptr), p(rowidx), p(a), p(b_vals), p(c))


def csc_asparse(arith, flags, m, n, k, ldb, ldc, colptr, rowidx, a_vals, b, c):
    f = lib().xo_dcsc_asparse if b.dtype == np.float64 else lib().xo_scsc_asparse
    f(arith, flags, m, n, k, ldb, ldc, p(colptr), p(rowidx), p(a_vals), p(b), p(c))


def _sfx(a):
    return "f64" if a.dtype == np.float64 else "f32"


def soa_csr_asparse(flags, m, n, k, ldb, ldc, v, rowptr, colidx, a_vals, b, c):
    getattr(lib(), "xo_soa_csr_asparse_" + _sfx(c))(flags, m, n, k, ldb, ldc, v, p(rowptr), p(colidx), p(a_vals), p(b), p(c))


def soa_bsparse(flags, csr, m, n, k, lda, ldc, v, ptr, idx, a, b_vals, c):
    getattr(lib(), "xo_soa_bsparse_" + _sfx(c))(flags, 1 if csr else 0, m, n, k, lda, ldc, v, p(ptr), p(idx), p(a), p(b_vals), p(c))


def soa_rm_ac(flags, m, n, k, lda, ldb, ldc, v, a, b, c):
    getattr(lib(), "xo_soa_rm_ac_" + _sfx(c))(flags, m, n, k, lda, ldb, ldc, v, p(a), p(b), p(c))


def soa_rm_bc(flags, m, n, k, lda, ldb, ldc, v, a, b, c):
    getattr(lib(), "xo_soa_rm_bc_" + _sfx(c))(flags, m, n, k, lda, ldb, ldc, v, p(a), p(b), p(c))


def csr_reg(flags, m, n, k, ldb, ldc, rowptr, colidx, values, b, c):
    f = lib().xo_dcsr_reg if b.dtype == np.float64 else lib().xo_scsr_reg
    return f(flags, m, n, k, ldb, ldc, p(rowptr), p(colidx), p(values), p(b), p(c))


class Fsspmdm(object):
    def __init__(self, a_dense, M, N, K, lda, ldb, ldc, alpha, beta, have_avx512):
        self.h = lib().xo_fsspmdm_create(_ts(a_dense), M, N, K, lda, ldb, ldc, C.c_double(alpha), C.c_double(beta),
                                         p(a_dense), int(have_avx512))
        if not self.h:
            raise ValueError("xo_fsspmdm_create rejected the arguments")

    def execute(self, b, c):
        lib().xo_fsspmdm_execute(C.c_void_p(self.h), p(b), p(c))

    def sparse(self):
        # struct xo_fsspmdm: 8 ints then `int sparse`
        return C.cast(self.h, C.POINTER(C.c_int))[8]

    def close(self):
        if self.h:
            lib().xo_fsspmdm_destroy(C.c_void_p(self.h))
            self.h = None


class SpmdmHandle(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("m", "n", "k", "bm", "bn", "bk", "mb", "nb", "kb")]


def spmdm_init(M, N, K, max_threads, bn_isa):
    h = SpmdmHandle()
    lib().xo_spmdm_init(M, N, K, max_threads, bn_isa, C.byref(h))
    return h


def spmdm_exec(arith, M, N, K, bn_isa, transa, transb, transc, beta, a, b, c):
    lib().xo_spmdm_exec(arith, M, N, K, bn_isa, C.c_char(transa.encode()), C.c_char(transb.encode()), C.c_char(transc.encode()),
                        C.c_float(beta), p(a), p(b), p(c))


def spmdm_exec_bf16(arith, M, N, K, bn_isa, transa, transb, transc, beta_bits, a, b, c):
    lib().xo_spmdm_exec_bf16(arith, M, N, K, bn_isa, C.c_char(transa.encode()), C.c_char(transb.encode()), C.c_char(transc.encode()),
                             C.c_ushort(beta_bits), p(a), p(b), p(c))


def spmdm_exec_batch(arith, M, N, K, bn_isa, transa, transb, transc, beta, a, b, c, batch, nthreads=1):
    lib().xo_spmdm_exec_batch(arith, M, N, K, bn_isa, C.c_char(transa.encode()), C.c_char(transb.encode()),
                              C.c_char(transc.encode()), C.c_float(beta), p(a), p(b), p(c), C.c_longlong(batch), nthreads)


def spmdm_geometry(M, N, K, bm, bn, bk):
    """a handle with a given block geometry (results do not depend on it; slices do)"""
    h = SpmdmHandle()
    h.m, h.n, h.k, h.bm, h.bn, h.bk = M, N, K, bm, bn, bk
    h.mb, h.nb, h.kb = (M + bm - 1) // bm, (N + bn - 1) // bn, (K + bk - 1) // bk
    return h


class _Slice(C.Structure):
    _fields_ = [("rowidx", C.POINTER(C.c_uint16)), ("colidx", C.POINTER(C.c_uint16)), ("values", C.POINTER(C.c_float))]


def spmdm_slices(M, N, K, bn_isa, transa, a, max_threads=1, handle=None):
    """createSparseSlice on every block; returns (handle, [(rowidx, colidx, values)] indexed kb*mb_count+mb)"""
    h = handle if handle is not None else spmdm_init(M, N, K, max_threads, bn_isa)
    lib().xo_spmdm_alloc_slices.restype = C.POINTER(_Slice)
    s = lib().xo_spmdm_alloc_slices(C.byref(h))
    out = []
    for blk in range(h.mb * h.kb):
        lib().xo_spmdm_create_slice(C.byref(h), C.c_char(transa.encode()), p(a), s, blk)
    for blk in range(h.mb * h.kb):
        mb = blk % h.mb
        nrows = min(h.bm, h.m - mb * h.bm)
        ri = np.ctypeslib.as_array(s[blk].rowidx, (nrows + 1,)).copy()
        nnz = int(ri[nrows])
        ci = np.ctypeslib.as_array(s[blk].colidx, (max(nnz, 1),)).copy()[:nnz]
        va = np.ctypeslib.as_array(s[blk].values, (max(nnz, 1),)).copy()[:nnz]
        out.append((ri, ci, va))
    lib().xo_spmdm_free_slices(C.byref(h), s)
    return h, out


def spmdm_compute_blocks(arith, h, transa, transb, transc, beta, a, b, c, blocks):
    """slices of `a` under geometry `h`, then xo_spmdm_compute for the listed compute block ids only (in place on c)"""
    lib().xo_spmdm_alloc_slices.restype = C.POINTER(_Slice)
    s = lib().xo_spmdm_alloc_slices(C.byref(h))
    for blk in range(h.mb * h.kb):
        lib().xo_spmdm_create_slice(C.byref(h), C.c_char(transa.encode()), p(a), s, blk)
    alpha, be = C.c_float(1.0), C.c_float(beta)
    for blk in blocks:
        lib().xo_spmdm_compute(arith, C.byref(h), C.c_char(transa.encode()), C.c_char(transb.encode()), C.byref(alpha), s, p(b),
                               C.c_char(transc.encode()), C.byref(be), p(c), blk)
    lib().xo_spmdm_free_slices(C.byref(h), s)


class Bgemm(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("typesize", "m", "n", "k", "bm", "bn", "bk", "mb", "nb", "kb",
                                       "b_m1", "b_n1", "b_k1", "b_k2", "order", "flags")]


def bgemm_init(typesize, m, n, k, bm, bn, bk, b_m1=1, b_n1=1, b_k1=1, b_k2=1, alpha=1.0, beta=1.0, order=0):
    h = Bgemm()
    rc = lib().xo_bgemm_init(C.byref(h), typesize, m, n, k, bm, bn, bk, b_m1, b_n1, b_k1, b_k2, C.c_double(alpha), C.c_double(beta), order)
    return h if rc == 0 else None


def bgemm_copy(h, which, src, ld, dst):
    f = {"a": lib().xo_bgemm_copyin_a, "b": lib().xo_bgemm_copyin_b, "c": lib().xo_bgemm_copyin_c, "out": lib().xo_bgemm_copyout_c}[which]
    f(C.byref(h), p(src), ld, p(dst))


def bgemm_permute(h, which, src, dst):
    f = {"convert_b_to_a": lib().xo_bgemm_convert_b_to_a, "transpose_b": lib().xo_bgemm_transpose_b}[which]
    f(C.byref(h), p(src), p(dst))


def bgemm_st(arith, h, a, b, c):
    lib().xo_bgemm_st(arith, C.byref(h), p(a), p(b), p(c))


def matinit(seed, nrows, ncols, ld, scale, dtype):
    out = np.zeros(ncols * ld, dtype=dtype)
    f = lib().xo_matinit_f64 if dtype == np.float64 else lib().xo_matinit_f32
    f(seed, p(out), nrows, ncols, ld, C.c_double(scale))
    return out


def rng_seed(seed):
    lib().xo_rng_seed(C.c_uint(seed))


def rng_f64():
    return lib().xo_rng_f64()


def gemm_lowp(kind, beta0, m, n, k, lda, ldb, ldc, a, b, c, scf=1.0):
    """low-precision gold loops (samples/xgemm/kernel.c); kind 0 i16->i32, 1 i16->f32, 2 bf16->f32, 3 bf16->bf16; a, b uint16/int16 arrays"""
    f = lib().xo_gemm_lowp
    f.argtypes = [C.c_int] * 8 + [C.c_void_p] * 3 + [C.c_float]
    f.restype = C.c_int
    return f(kind, beta0, m, n, k, lda, ldb, ldc, a.ctypes.data, b.ctypes.data, c.ctypes.data, scf)
