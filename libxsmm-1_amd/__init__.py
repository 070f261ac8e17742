"""libxsmm-1_amd -- Python-side plumbing for the MI355X-native LIBXSMM engine.

The product is the C-ABI shared library ``lib/libxsmm.so`` (sources in ``csrc/``, interface in
``/include/libxsmm.h``). This module only loads it through ``ctypes`` and declares argument
types, so that tests and ``bench.py`` can call the *same* entry points a C caller binds
(reference interface: ``src/template/libxsmm.h:73-414``). There is no Python or CPU compute path
here: if the library is missing, or no HIP device is usable, calls fail loudly.

Import with ``importlib.import_module("libxsmm-1_amd")`` (the directory name is not an identifier).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIBXSMM_AMD_LIBRARY", os.path.join(_HERE, "lib", "libxsmm.so"))  # override: A/B builds in tools/
CSRC = os.path.join(_HERE, "csrc")

# enum values (include/libxsmm.h; reference include/libxsmm_typedefs.h:158-213)
F64, F32, BF16, I32, I16 = 0, 1, 2, 4, 5  # libxsmm_gemm_precision (include/libxsmm.h)
FLAG_TRANS_A, FLAG_TRANS_B, FLAG_BETA_0, FLAG_BATCH_REDUCE = 1, 2, 16, 256

c_int_p = C.POINTER(C.c_int)


class DescriptorBlob(C.Structure):
    _fields_ = [("data", C.c_char * 64)]


class MMKernelInfo(C.Structure):  # libxsmm_mmkernel_info
    _fields_ = [("iprecision", C.c_int), ("oprecision", C.c_int), ("prefetch", C.c_int),
                ("lda", C.c_uint), ("ldb", C.c_uint), ("ldc", C.c_uint),
                ("m", C.c_uint), ("n", C.c_uint), ("k", C.c_uint), ("flags", C.c_int)]


class RegistryInfo(C.Structure):
    _fields_ = [("capacity", C.c_size_t), ("size", C.c_size_t), ("nbytes", C.c_size_t),
                ("nstatic", C.c_size_t), ("ncache", C.c_size_t)]


class SpmdmHandle(C.Structure):  # libxsmm_spmdm_handle (reference include/libxsmm_spmdm.h:42-61)
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("bm", C.c_int), ("bn", C.c_int), ("bk", C.c_int),
                ("mb", C.c_int), ("nb", C.c_int), ("kb", C.c_int), ("datatype", C.c_int),
                ("base_ptr_scratch_A", C.c_void_p), ("base_ptr_scratch_B_scratch_C", C.c_void_p),
                ("memory_for_scratch_per_thread", C.c_int)]


class CSRSlice(C.Structure):  # libxsmm_CSR_sparseslice
    _fields_ = [("rowidx", C.c_void_p), ("colidx", C.c_void_p), ("values", C.c_void_p)]


class GeneratedCode(C.Structure):  # libxsmm_generated_code
    _fields_ = [("generated_code", C.c_void_p), ("buffer_size", C.c_uint), ("code_size", C.c_uint), ("code_type", C.c_uint),
                ("last_error", C.c_uint)]

    def text(self):
        return C.string_at(self.generated_code, self.code_size).decode() if self.generated_code else ""

    def release(self):
        if self.generated_code:
            C.CDLL(None).free(C.c_void_p(self.generated_code))
            self.generated_code = None


class MatdiffInfo(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "norm1_abs", "norm1_rel", "normi_abs", "normi_rel", "normf_rel", "linf_abs", "linf_rel", "l2_abs", "l2_rel",
        "l1_ref", "min_ref", "max_ref", "avg_ref", "var_ref", "l1_tst", "min_tst", "max_tst", "avg_tst", "var_tst")] + \
        [("m", C.c_int), ("n", C.c_int)]


def build(verbose=False):
    """Compile csrc/ into lib/libxsmm.so for gfx950 (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libxsmm.so failed")
    return LIB_PATH


# shapes whose code objects build() leaves in lib/jit_cache (they travel with the library): the BASELINE configurations
PREBUILD_F64 = [(m, n, k) for m in (13, 23, 32) for n in (13, 23, 32) for k in (13, 23, 32)]  # config 1 and 5 (CP2K stacks: also grouped)
PREBUILD_F64 += [(40, 40, 40), (48, 48, 48), (56, 56, 56), (64, 64, 64)]  # shapes beyond 32: matrix-core forms
PREBUILD_F32 = [(32, 32, 32), (23, 23, 23), (13, 13, 13), (40, 40, 40), (48, 48, 48), (56, 56, 56), (64, 64, 32)]


def prebuild_kernels(verbose=False):
    """hiprtc ahead of time (no device needed): see libxsmm_amd_jit_prebuild"""
    L = lib()
    total = 0
    for prec, shapes, grouped in ((F64, PREBUILD_F64, 1), (F32, PREBUILD_F32, 0)):
        keep, arr = [], (C.c_void_p * len(shapes))()
        for idx, (m, n, k) in enumerate(shapes):
            blob, d = descriptor(prec, m, n, k)
            keep.append(blob); arr[idx] = C.cast(d, C.c_void_p)
        rc = L.libxsmm_amd_jit_prebuild(arr, len(shapes), grouped)
        if verbose:
            print("prebuild: precision %d, %d shapes -> %d" % (prec, len(shapes), rc))
        if rc < 0:
            raise RuntimeError("libxsmm_amd_jit_prebuild failed for %d code objects" % -rc)
        total += rc
    return total


_lib = None


def lib():
    """The loaded C-ABI library (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run __graft_entry__.build() (there is no non-HIP fallback)" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's). If
        # libxsmm.so were loaded first, a later `import torch` would map a second runtime and neither would see the
        # GPU reliably. Importing torch first makes libxsmm.so bind to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
        import atexit
        atexit.register(L.libxsmm_amd_jit_drain)  # before the interpreter goes: see include/libxsmm_amd.h
    return _lib


def _declare(L):
    vp, i, ll = C.c_void_p, C.c_int, C.c_longlong
    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)
    sig("libxsmm_init", None)
    sig("libxsmm_finalize", None)
    sig("libxsmm_get_verbosity", i)
    sig("libxsmm_set_verbosity", None, i)
    sig("libxsmm_get_target_arch", C.c_char_p)
    sig("libxsmm_set_target_arch", None, C.c_char_p)
    sig("libxsmm_get_target_archid", i)
    sig("libxsmm_set_target_archid", None, i)
    sig("libxsmm_dgemm_descriptor_init", vp, C.POINTER(DescriptorBlob), i, i, i, i, i, i, C.c_double, C.c_double, i, i)
    sig("libxsmm_sgemm_descriptor_init", vp, C.POINTER(DescriptorBlob), i, i, i, i, i, i, C.c_float, C.c_float, i, i)
    sig("libxsmm_gemm_descriptor_dinit", vp, C.POINTER(DescriptorBlob), i, i, i, i, i, i, i, C.c_double, C.c_double, i, i)
    sig("libxsmm_gemm_descriptor_init", vp, C.POINTER(DescriptorBlob), i, i, i, i, i, i, i, vp, vp, i, i)
    sig("libxsmm_xmmdispatch", vp, vp)
    sig("libxsmm_dmmdispatch", vp, i, i, i, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_double), C.POINTER(C.c_double), c_int_p, c_int_p)
    sig("libxsmm_smmdispatch", vp, i, i, i, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_float), C.POINTER(C.c_float), c_int_p, c_int_p)
    sig("libxsmm_wimmdispatch", vp, i, i, i, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p)
    for name in ("libxsmm_wsmmdispatch", "libxsmm_bsmmdispatch", "libxsmm_bmmdispatch"):
        sig(name, vp, i, i, i, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_float), C.POINTER(C.c_float), c_int_p, c_int_p)
    sig("libxsmm_dmmdispatch_reducebatch", vp, i, i, i, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_double), C.POINTER(C.c_double), c_int_p, c_int_p)
    sig("libxsmm_smmdispatch_reducebatch", vp, i, i, i, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_float), C.POINTER(C.c_float), c_int_p, c_int_p)
    sig("libxsmm_wimmdispatch", vp, i, i, i, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p)
    sig("libxsmm_release_kernel", None, vp)
    sig("libxsmm_get_kernel_kind", i, vp, c_int_p)
    sig("libxsmm_get_mmkernel_info", i, vp, C.POINTER(MMKernelInfo), C.POINTER(C.c_size_t))
    sig("libxsmm_get_registry_info", i, C.POINTER(RegistryInfo))
    sig("libxsmm_create_dcsr_reg", vp, vp, vp, vp, vp)
    sig("libxsmm_create_scsr_reg", vp, vp, vp, vp, vp)
    batch_args = [i, i, vp, vp, i, i, i, vp, vp, c_int_p, vp, c_int_p, vp, vp, c_int_p, i, i, vp, vp, vp, i]
    sig("libxsmm_gemm_batch", None, *batch_args)
    sig("libxsmm_gemm_batch_omp", None, *batch_args)
    sig("libxsmm_mmbatch", None, *(batch_args + [i, i]))
    sig("libxsmm_mmbatch_kernel", i, vp, i, i, vp, vp, vp, vp, vp, vp, i, i, i, C.c_ubyte, C.c_ubyte, i)
    sig("libxsmm_mmbatch_blas", i, *batch_args)
    grp = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    sig("libxsmm_dgemm_batch", None, *grp)
    sig("libxsmm_sgemm_batch", None, *grp)
    sig("libxsmm_mmbatch_begin", None, i, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, vp, vp)
    sig("libxsmm_mmbatch_end", None)
    gemm = [C.c_char_p, C.c_char_p, c_int_p, c_int_p, c_int_p, vp, vp, c_int_p, vp, c_int_p, vp, vp, c_int_p]
    sig("libxsmm_dgemm", None, *gemm)
    sig("libxsmm_sgemm", None, *gemm)
    sig("libxsmm_dfsspmdm_create", vp, i, i, i, i, i, i, C.c_double, C.c_double, vp)
    sig("libxsmm_dfsspmdm_execute", None, vp, vp, vp)
    sig("libxsmm_dfsspmdm_destroy", None, vp)
    sig("libxsmm_sfsspmdm_create", vp, i, i, i, i, i, i, C.c_float, C.c_float, vp)
    sig("libxsmm_sfsspmdm_execute", None, vp, vp, vp)
    sig("libxsmm_sfsspmdm_destroy", None, vp)
    sig("libxsmm_amd_dfsspmdm_execute_batch", i, vp, vp, vp, ll)
    sig("libxsmm_amd_sfsspmdm_execute_batch", i, vp, vp, vp, ll)
    sig("libxsmm_spmdm_init", None, i, i, i, i, C.POINTER(SpmdmHandle), C.POINTER(C.POINTER(CSRSlice)))
    sig("libxsmm_spmdm_destroy", None, C.POINTER(SpmdmHandle))
    sig("libxsmm_spmdm_get_num_createSparseSlice_blocks", i, C.POINTER(SpmdmHandle))
    sig("libxsmm_spmdm_get_num_compute_blocks", i, C.POINTER(SpmdmHandle))
    sig("libxsmm_spmdm_createSparseSlice_fp32_thread", None, C.POINTER(SpmdmHandle), C.c_char, vp, C.POINTER(CSRSlice), i, i, i)
    sig("libxsmm_spmdm_compute_fp32_thread", None, C.POINTER(SpmdmHandle), C.c_char, C.c_char, vp, C.POINTER(CSRSlice), vp,
        C.c_char, vp, vp, i, i, i)
    sig("libxsmm_spmdm_createSparseSlice_bfloat16_thread", None, C.POINTER(SpmdmHandle), C.c_char, vp, C.POINTER(CSRSlice), i, i, i)
    sig("libxsmm_spmdm_compute_bfloat16_thread", None, C.POINTER(SpmdmHandle), C.c_char, C.c_char, vp, C.POINTER(CSRSlice), vp,
        C.c_char, vp, vp, i, i, i)
    sig("libxsmm_amd_memcpy_h2d", i, vp, vp, C.c_size_t)
    sig("libxsmm_amd_memcpy_d2h", i, vp, vp, C.c_size_t)
    sig("libxsmm_amd_spmdm_createSparseSlice_all", i, C.POINTER(SpmdmHandle), C.c_char, vp, C.POINTER(CSRSlice))
    sig("libxsmm_amd_spmdm_compute_all", i, C.POINTER(SpmdmHandle), C.c_char, C.c_char, vp, C.POINTER(CSRSlice), vp, C.c_char, vp, vp)
    sig("libxsmm_amd_spmdm_createSparseSlice_bfloat16_all", i, C.POINTER(SpmdmHandle), C.c_char, vp, C.POINTER(CSRSlice))
    sig("libxsmm_amd_spmdm_compute_bfloat16_all", i, C.POINTER(SpmdmHandle), C.c_char, C.c_char, vp, C.POINTER(CSRSlice), vp, C.c_char, vp, vp)
    sig("libxsmm_amd_spmdm_batch_create", vp, i, i, i, ll)
    sig("libxsmm_amd_spmdm_batch_destroy", None, vp)
    sig("libxsmm_amd_spmdm_batch_create_slices", i, vp, C.c_char, vp)
    sig("libxsmm_amd_spmdm_batch_compute", i, vp, C.c_char, vp, C.c_char, vp, vp)
    sig("libxsmm_amd_spmdm_batch_get_slice", i, vp, ll, vp, vp, vp, i)
    sig("libxsmm_blocked_gemm_handle_create", vp, i, i, i, i, i, i, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p,
        vp, vp, c_int_p, c_int_p, c_int_p)
    sig("libxsmm_blocked_gemm_handle_destroy", None, vp)
    for nm in ("copyin_a", "copyin_b", "copyin_c", "copyout_c", "convert_b_to_a", "transpose_b"):
        sig("libxsmm_blocked_gemm_" + nm, i, vp, vp, c_int_p, vp)
    sig("libxsmm_blocked_gemm_st", None, vp, vp, vp, vp, i, i)
    sig("libxsmm_blocked_gemm_omp", None, vp, vp, vp, vp, i)
    sig("libxsmm_malloc", vp, C.c_size_t)
    sig("libxsmm_aligned_malloc", vp, C.c_size_t, C.c_size_t)
    sig("libxsmm_free", None, vp)
    sig("libxsmm_typesize", C.c_ubyte, i)
    sig("libxsmm_timer_tick", C.c_ulonglong)
    sig("libxsmm_timer_duration", C.c_double, C.c_ulonglong, C.c_ulonglong)
    sig("libxsmm_rng_set_seed", None, C.c_uint)
    sig("libxsmm_rng_f64", C.c_double)
    sig("libxsmm_rng_u32", C.c_uint, C.c_uint)
    sig("libxsmm_isqrt_u64", C.c_uint, C.c_ulonglong)
    sig("libxsmm_shuffle", C.c_size_t, C.c_uint)
    sig("libxsmm_matdiff", i, C.POINTER(MatdiffInfo), i, i, i, vp, vp, c_int_p, c_int_p)
    sig("libxsmm_matdiff_clear", None, C.POINTER(MatdiffInfo))
    sig("libxsmm_matdiff_reduce", None, C.POINTER(MatdiffInfo), C.POINTER(MatdiffInfo))
    sig("libxsmm_amd_device_count", i)
    sig("libxsmm_amd_set_stream", None, vp)
    sig("libxsmm_amd_get_stream", vp)
    sig("libxsmm_amd_synchronize", i)
    sig("libxsmm_amd_set_mfma", i, i)
    sig("libxsmm_amd_get_mfma", i)
    sig("libxsmm_amd_last_kernel", C.c_char_p)
    sig("libxsmm_amd_launch_count", C.c_ulonglong)
    sig("libxsmm_amd_flush", None)
    sig("libxsmm_amd_defer_begin", None)
    sig("libxsmm_amd_defer_end", None)
    sig("libxsmm_amd_defer_active", i)
    sig("libxsmm_amd_is_device_pointer", i, vp)
    sig("libxsmm_amd_gemm_batch_strided", i, vp, vp, vp, vp, ll, ll, ll, ll)
    sig("libxsmm_amd_stream_probe", i, vp, vp, vp, ll)
    sig("libxsmm_amd_csr_kernel_source", i, i, i, i, vp, vp, vp, i, i, vp, C.c_size_t, i)
    gc = C.POINTER(GeneratedCode)
    sig("libxsmm_strerror", C.c_char_p, C.c_uint)
    sig("libxsmm_generator_gemm_kernel", None, gc, vp, C.c_char_p)
    sig("libxsmm_generator_gemm_inlineasm", None, C.c_char_p, C.c_char_p, vp, C.c_char_p)
    sig("libxsmm_generator_gemm_directasm", None, C.c_char_p, C.c_char_p, vp, C.c_char_p)
    sig("libxsmm_generator_spgemm", None, C.c_char_p, C.c_char_p, vp, C.c_char_p, C.c_char_p, i)
    for nm in ("csr", "csc", "csr_reg"):
        sig("libxsmm_generator_spgemm_%s_kernel" % nm, None, gc, vp, C.c_char_p, vp, vp, vp)
    for nm in ("csr_soa", "csc_soa"):
        sig("libxsmm_generator_spgemm_%s_kernel" % nm, None, gc, vp, C.c_char_p, vp, vp, vp)
    sig("libxsmm_create_xcsr_soa", vp, vp, vp, vp, vp)
    sig("libxsmm_create_xcsc_soa", vp, vp, vp, vp, vp)
    sig("libxsmm_create_rm_ac_soa", vp, vp)
    sig("libxsmm_create_rm_bc_soa", vp, vp)
    sig("libxsmm_amd_soa_width", i, i)
    sig("libxsmm_amd_sparse_reader", i, C.c_char_p, i, C.POINTER(C.POINTER(C.c_uint)), C.POINTER(C.POINTER(C.c_uint)), C.POINTER(C.POINTER(C.c_double)),
        C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint))
    sig("libxsmm_amd_kernel_execute_batch", i, vp, vp, vp, vp, ll, ll, ll)
    sig("libxsmm_amd_spgemm_create", vp, vp, i, vp, vp, i)
    sig("libxsmm_amd_spgemm_execute_batch", i, vp, vp, vp, vp, ll, ll, ll)
    sig("libxsmm_amd_spgemm_destroy", None, vp)
    sig("libxsmm_amd_spgemm_source", i, vp, i, vp, vp, i, vp, C.c_size_t, i)
    sig("libxsmm_amd_smm_kernel_source", i, vp, i, vp, C.c_size_t, i)
    sig("libxsmm_amd_device_malloc", vp, C.c_size_t)
    sig("libxsmm_amd_device_free", None, vp)
    sig("libxsmm_amd_smm_grouped_kernel_source", i, C.POINTER(vp), i, vp, C.c_size_t, i)
    sig("libxsmm_amd_jit_prebuild", i, C.POINTER(vp), i, i)
    sig("libxsmm_amd_jit_wait", None)
    sig("libxsmm_amd_jit_drain", None)
    sig("libxsmm_amd_gemm_batch_groups", i, i, i, i, C.c_char_p, C.c_char_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, vp, vp,
        C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), c_int_p, i)


# ---- thin helpers used by tests and bench (argument marshalling only) -------------------------------------------------
def iptr(v):
    """pointer to a C int holding v (or NULL)"""
    return None if v is None else C.byref(C.c_int(int(v)))


def dptr(t):
    """raw pointer of a torch tensor / numpy array / int"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    if hasattr(t, "ctypes"):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(int(t))


def descriptor(prec, m, n, k, lda=None, ldb=None, ldc=None, alpha=1.0, beta=1.0, flags=0, prefetch=0):
    """libxsmm_gemm_descriptor_dinit -> (blob, pointer); pointer is None when the reference would return NULL."""
    blob = DescriptorBlob()
    lda = m if lda is None else lda
    ldb = (n if (flags & FLAG_TRANS_B) else k) if ldb is None else ldb
    ldc = m if ldc is None else ldc
    p = lib().libxsmm_gemm_descriptor_dinit(C.byref(blob), prec, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch)
    return blob, p


def gemm_batch(prec, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, index_base, index_stride,
               stride_a, stride_b, stride_c, batchsize, omp=False):
    """libxsmm_gemm_batch[_omp]; alpha/beta are Python floats (or None), a/b/c/stride_* tensors, arrays or addresses."""
    ct = C.c_double if prec == F64 else C.c_float
    al = None if alpha is None else C.byref(ct(alpha))
    be = None if beta is None else C.byref(ct(beta))
    f = lib().libxsmm_gemm_batch_omp if omp else lib().libxsmm_gemm_batch
    f(prec, prec, C.c_char_p(transa.encode()) if transa else None, C.c_char_p(transb.encode()) if transb else None,
      m, n, k, al, dptr(a), iptr(lda), dptr(b), iptr(ldb), be, dptr(c), iptr(ldc), index_base, index_stride,
      dptr(stride_a), dptr(stride_b), dptr(stride_c), batchsize)


def read_mtx(path, is_csr):
    """libxsmm_amd_sparse_reader -> (error code, ptr, idx, values, rows, cols, nnz) as numpy arrays (None on error)"""
    import numpy as np
    ptr, idx, val = C.POINTER(C.c_uint)(), C.POINTER(C.c_uint)(), C.POINTER(C.c_double)()
    r, c, z = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    rc = lib().libxsmm_amd_sparse_reader(str(path).encode(), 1 if is_csr else 0, C.byref(ptr), C.byref(idx), C.byref(val), C.byref(r), C.byref(c), C.byref(z))
    if 0 != rc:
        assert not ptr and not idx and not val
        return rc, None, None, None, 0, 0, 0
    nmajor = r.value if is_csr else c.value
    out = (rc, np.ctypeslib.as_array(ptr, (nmajor + 1,)).copy(), np.ctypeslib.as_array(idx, (z.value,)).copy(),
           np.ctypeslib.as_array(val, (z.value,)).copy(), r.value, c.value, z.value)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for p in (ptr, idx, val):
        libc.free(C.cast(p, C.c_void_p))
    return out


def gemm_batch_groups(prec, shapes, a, b, c, stride_a, stride_b, stride_c, sizes, index_base=0, index_stride=4, beta=1.0, relaxed=False):
    """libxsmm_amd_gemm_batch_groups: shapes = [(m, n, k)], a/b/c/stride_* = per-group tensors / arrays, sizes = per-group batch sizes"""
    n = len(shapes)
    ints = lambda v: (C.c_int * n)(*[int(x) for x in v])
    ptrs = lambda v: (C.c_void_p * n)(*[dptr(x).value if x is not None else None for x in v])
    ct = C.c_double if prec == F64 else C.c_float
    be = ct(beta)
    return lib().libxsmm_amd_gemm_batch_groups(prec, prec, n, None, None, ints(s[0] for s in shapes), ints(s[1] for s in shapes), ints(s[2] for s in shapes),
                                               None, None, None, None, C.byref(be), ptrs(a), ptrs(b), ptrs(c), index_base, index_stride,
                                               ptrs(stride_a), ptrs(stride_b), ptrs(stride_c), ints(sizes), 1 if relaxed else 0)


def call_kernel(fn_ptr, a, b, c, x3=None):
    """Call a dispatched kernel (bare function pointer) with three (or four) pointer arguments."""
    proto = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    proto(fn_ptr)(dptr(a), dptr(b), dptr(c), dptr(x3))


def last_kernel():
    return lib().libxsmm_amd_last_kernel().decode()
