// xsmm_sparse.cpp -- host side of fsspmdm (fixed-sparsity operator) and spmdm (dense-in, sparse-compute).
//
// Reference: src/libxsmm_fsspmdm.c:48-329 and src/libxsmm_spmdm.c:103-627 with the fp32 templates.
#include "xsmm_internal.hpp"

#include <hip/hip_runtime_api.h>

#include <cassert>
#include <cstring>
#include <vector>

namespace xsmm {
int launch_spmdm_create_blocks(int M, int K, int bm, int bk, int mb, int first_slice, int slice_step, int nslices, int transa, const float* a,
                               uint16_t* rowidx, uint16_t* colidx, float* values, void* stream, const char** name);
int launch_spmdm_compute_generic(long long batch, int M, int N, int K, int bm, int bk, int mb, int kb, int transb, int transc, float beta,
                                 const uint16_t* rowidx, const uint16_t* colidx, const float* values, long long rowidx_stride, long long cap,
                                 const float* b, float* c, long long b_stride, long long c_stride,
                                 int m_begin, int m_end, int n_begin, int n_end, void* stream, const char** name);
}

using namespace xsmm;

// ---------------------------------------------------------------------------------------------------------------
// fsspmdm. The reference builds a CSR copy of the constant operator A and JITs a kernel with the pattern and
// the (<= 31 unique) values baked in, or falls back to a dense SMM on the transposed problem. Here the CSR
// arrays live in HBM and are read through the scalar cache; there is no limit on the number of unique values,
// so the sparse path always applies. What is kept from the reference: CSR order (row scan, ascending column,
// != 0 test), N % 16 == 0 and the other preconditions, and the beta == 0 behaviour for rows without nnz --
// which differs between the reference's two paths (csr_reg leaves such rows untouched, the dense fallback
// zeroes them): this engine zeroes them when beta == 0 (the dense fallback's, i.e. the mathematically
// expected, result) except for kernels created explicitly through libxsmm_create_?csr_reg.
// ---------------------------------------------------------------------------------------------------------------
struct libxsmm_dfsspmdm {
  int M, N, K, ldb, ldc, N_chunksize; // first six fields as in the reference handle (src/libxsmm_main.h:695-704)
  double* a_dense;                    // always NULL here (sparse path)
  libxsmm_dmmfunction kernel;         // always NULL here; execution goes through the fields below
  int typesize, beta0; unsigned nnz;
  unsigned* d_rowptr; unsigned* d_colidx; void* d_values;
  JitKernel* jit; int jit_vec;        // operator-specific kernel compiled at create time (NULL: generic kernel)
};
struct libxsmm_sfsspmdm {
  int M, N, K, ldb, ldc, N_chunksize;
  float* a_dense;
  libxsmm_smmfunction kernel;
  int typesize, beta0; unsigned nnz;
  unsigned* d_rowptr; unsigned* d_colidx; void* d_values;
  JitKernel* jit; int jit_vec;
};
static_assert(sizeof(libxsmm_dfsspmdm) == sizeof(libxsmm_sfsspmdm), "handles share one layout");

namespace {

template<typename T, typename H>
H* fsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K, libxsmm_blasint lda, libxsmm_blasint ldb,
                  libxsmm_blasint ldc, T alpha, T beta, const T* a_dense)
{
  // preconditions: asserts in the reference (src/libxsmm_fsspmdm.c:65-71)
  assert(N % 16 == 0); assert(N >= 16); assert(T(1) == alpha); assert(T(1) == beta || T(0) == beta);
  assert(K <= lda); assert(N <= ldc); assert(N <= ldb);
  if (nullptr == a_dense || 0 != (N % 16) || N < 16 || T(1) != alpha || (T(1) != beta && T(0) != beta) || K > lda || N > ldc || N > ldb) return nullptr;
  libxsmm_init();
  if (!device_ready()) { fail_no_device("libxsmm_?fsspmdm_create"); return nullptr; }
  H* h = static_cast<H*>(calloc(1, sizeof(H)));
  if (nullptr == h) return nullptr;
  h->M = (int)M; h->N = (int)N; h->K = (int)K; h->ldb = (int)ldb; h->ldc = (int)ldc;
  h->N_chunksize = (8 == sizeof(T) ? 8 : 16); h->typesize = (int)sizeof(T); h->beta0 = (T(0) == beta) ? 1 : 0;
  std::vector<unsigned> rowptr((size_t)M + 1), colidx; std::vector<T> values;
  for (int i = 0; i < M; ++i) { // src/libxsmm_fsspmdm.c:102-113
    rowptr[i] = (unsigned)values.size();
    for (int j = 0; j < K; ++j) {
      const T v = a_dense[(size_t)i * lda + j];
      if (v != T(0)) { values.push_back(v); colidx.push_back((unsigned)j); } // LIBXSMM_NEQ: -0 dropped, NaN kept
    }
  }
  rowptr[M] = (unsigned)values.size();
  h->nnz = (unsigned)values.size();
  h->d_rowptr = static_cast<unsigned*>(dev_alloc(sizeof(unsigned) * ((size_t)M + 1)));
  h->d_colidx = static_cast<unsigned*>(dev_alloc(sizeof(unsigned) * (h->nnz + 1)));
  h->d_values = dev_alloc(sizeof(T) * (h->nnz + 1));
  bool ok = (nullptr != h->d_rowptr && nullptr != h->d_colidx && nullptr != h->d_values);
  ok = ok && 0 == h2d(h->d_rowptr, rowptr.data(), sizeof(unsigned) * ((size_t)M + 1));
  if (0 < h->nnz) ok = ok && 0 == h2d(h->d_colidx, colidx.data(), sizeof(unsigned) * h->nnz) && 0 == h2d(h->d_values, values.data(), sizeof(T) * h->nnz);
  ok = ok && 0 == stream_sync();
  if (!ok) { dev_free(h->d_rowptr); dev_free(h->d_colidx); dev_free(h->d_values); free(h); return nullptr; }
  // operator-specific kernel (the analogue of the reference's JIT, src/libxsmm_fsspmdm.c:119-126): pattern and values
  // baked into HIP source, compiled by hiprtc. Limits keep the generated kernel within the register file.
  const char* const env_jit = getenv("LIBXSMM_AMD_JIT");
  const bool want_jit = (nullptr == env_jit || 0 != atoi(env_jit));
  if (want_jit && 0 < h->nnz && h->nnz <= 8192 && K <= 160) {
    const char* const env_vec = getenv("LIBXSMM_AMD_JIT_VEC");
    int vec = (nullptr != env_vec && 0 != *env_vec) ? atoi(env_vec) : (8 == sizeof(T) ? 1 : 2);
    while (1 < vec && (0 != (ldb % vec) || 0 != (ldc % vec) || 0 != (N % vec))) vec >>= 1;
    if (vec < 1) vec = 1;
    std::vector<double> dv(values.begin(), values.end());
    const std::string src = gen_csr_panels_source((int)sizeof(T), (int)M, (int)K, rowptr.data(), colidx.data(), dv.data(),
      h->beta0, 0/*empty rows are zeroed for beta == 0*/, vec, "xsmm_fsspmdm_op", true);
    std::string log;
    h->jit = jit_compile(src, "xsmm_fsspmdm_op", &log);
    h->jit_vec = vec;
    if (nullptr == h->jit && 0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM WARNING: fsspmdm JIT unavailable (%s); using the generic kernel\n", log.c_str());
  }
  return h;
}

template<typename T, typename H>
int fsspmdm_run(const H* h, const T* B, T* C, long long batch)
{
  if (nullptr == h || nullptr == B || nullptr == C || batch < 0) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_?fsspmdm_execute"); return EXIT_FAILURE; }
  CsrPanels p; memset(&p, 0, sizeof(p));
  p.typesize = h->typesize; p.m = h->M; p.k = h->K; p.n = h->N; p.ldb = h->ldb; p.ldc = h->ldc; p.beta0 = h->beta0;
  p.skip_empty_rows = 0;
  p.rowptr = h->d_rowptr; p.colidx = h->d_colidx; p.values = h->d_values; p.nnz = h->nnz; p.batch = batch;
  const char* name = "";
  auto launch = [&](const void* db, void* dc) -> int {
    const long long ncols = (long long)h->N * batch;
    const bool aligned = (1 == h->jit_vec) || (0 == ((reinterpret_cast<uintptr_t>(db) | reinterpret_cast<uintptr_t>(dc)) & (sizeof(T) * h->jit_vec - 1)));
    if (nullptr != h->jit && aligned && 0 == (ncols % h->jit_vec)) {
      const int e = jit_launch_panels(h->jit, db, dc, ncols, h->ldb, h->ldc, h->jit_vec, device().stream);
      note_launch(8 == h->typesize ? "fsspmdm_f64_jit_operator" : "fsspmdm_f32_jit_operator");
      return e;
    }
    p.b = db; p.c = dc;
    const int e = launch_csr_panels(p, device().stream, &name); note_launch(name);
    return e;
  };
  // one panel per call on device memory (the PyFR driver's loop, samples/pyfr/pyfr_driver_asp_reg.c:300-308): consecutive calls
  // that walk along the rows are, inside the caller's opt-in bracket (libxsmm_amd_defer_begin/end), recorded into a burst instead of
  // costing a launch each (xsmm_defer.cpp)
  if (1 == batch && nullptr != h->jit && 0 == (h->N % h->jit_vec)
    && defer_panels(h, h->jit, B, C, h->typesize, h->M, h->N, h->K, h->ldb, h->ldc, h->jit_vec)) return EXIT_SUCCESS;
  if (is_device_ptr(B) && is_device_ptr(C)) { const int e = launch(B, C); if (0 == e) settle(B, C); return 0 == e ? EXIT_SUCCESS : EXIT_FAILURE; }
  // host panels: stage rows [0,K) x columns [0, N*batch) of B and rows [0,M) of C (strided by ldb/ldc)
  const long long ncols = (long long)h->N * batch;
  const size_t eb = (size_t)(h->K - 1) * h->ldb + ncols, ec = (size_t)(h->M - 1) * h->ldc + ncols;
  char* const db = static_cast<char*>(scratch(4, eb * sizeof(T))); char* const dc = static_cast<char*>(scratch(5, ec * sizeof(T)));
  if (nullptr == db || nullptr == dc || 0 != h2d(db, B, eb * sizeof(T)) || 0 != h2d(dc, C, ec * sizeof(T))) return EXIT_FAILURE;
  if (0 != launch(db, dc)) return EXIT_FAILURE;
  return 0 == d2h(C, dc, ec * sizeof(T)) ? EXIT_SUCCESS : EXIT_FAILURE;
}

template<typename H> void fsspmdm_destroy(H* h)
{
  if (nullptr == h) return;
  if (device_ready()) (void)stream_sync();
  jit_release(h->jit);
  dev_free(h->d_rowptr); dev_free(h->d_colidx); dev_free(h->d_values);
  free(h);
}

} // namespace

LIBXSMM_API int libxsmm_amd_csr_kernel_source(int typesize, int M, int K, const unsigned int* row_ptr, const unsigned int* column_idx,
  const double* values, int beta0, int vec, char* buffer, size_t buffer_size, int compile)
{ // the HIP text an operator is specialised to (counterpart of the reference's text generators,
  // libxsmm_generator_spgemm_csr_kernel, include/libxsmm_generator.h:190-197); optional hiprtc compile check (no device needed)
  if ((4 != typesize && 8 != typesize) || M <= 0 || K <= 0 || nullptr == row_ptr || nullptr == column_idx || nullptr == values || vec < 1) return -1;
  const std::string src = gen_csr_panels_source(typesize, M, K, row_ptr, column_idx, values, beta0, 0, vec, "xsmm_fsspmdm_op");
  if (nullptr != buffer && 0 < buffer_size) {
    const size_t n = (src.size() < buffer_size - 1 ? src.size() : buffer_size - 1);
    memcpy(buffer, src.data(), n); buffer[n] = 0;
  }
  if (0 != compile) {
    std::string log;
    const int rc = jit_check_source(src, &log);
    if (0 != rc && 0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM-AMD: hiprtc: %s\n", log.c_str());
    return rc;
  }
  return (int)src.size();
}

LIBXSMM_API libxsmm_dfsspmdm* libxsmm_dfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const double alpha, const double beta, const double* a_dense)
{ return fsspmdm_create<double, libxsmm_dfsspmdm>(M, N, K, lda, ldb, ldc, alpha, beta, a_dense); }
LIBXSMM_API libxsmm_sfsspmdm* libxsmm_sfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const float alpha, const float beta, const float* a_dense)
{ return fsspmdm_create<float, libxsmm_sfsspmdm>(M, N, K, lda, ldb, ldc, alpha, beta, a_dense); }
LIBXSMM_API void libxsmm_dfsspmdm_execute(const libxsmm_dfsspmdm* handle, const double* B, double* C) { assert(nullptr != handle); (void)fsspmdm_run<double>(handle, B, C, 1); }
LIBXSMM_API void libxsmm_sfsspmdm_execute(const libxsmm_sfsspmdm* handle, const float* B, float* C) { assert(nullptr != handle); (void)fsspmdm_run<float>(handle, B, C, 1); }
LIBXSMM_API void libxsmm_dfsspmdm_destroy(libxsmm_dfsspmdm* handle) { fsspmdm_destroy(handle); }
LIBXSMM_API void libxsmm_sfsspmdm_destroy(libxsmm_sfsspmdm* handle) { fsspmdm_destroy(handle); }
LIBXSMM_API int libxsmm_amd_dfsspmdm_execute_batch(const libxsmm_dfsspmdm* handle, const double* B, double* C, long long batch)
{ return fsspmdm_run<double>(handle, B, C, batch); }
LIBXSMM_API int libxsmm_amd_sfsspmdm_execute_batch(const libxsmm_sfsspmdm* handle, const float* B, float* C, long long batch)
{ return fsspmdm_run<float>(handle, B, C, batch); }

// ---------------------------------------------------------------------------------------------------------------
// spmdm, reference API (one M x N x K problem; the caller loops over block ids, samples/spmdm/spmdm.c:99-109).
// The handle and slice structs are caller-visible (include/libxsmm_spmdm.h:42-72); the slice arrays
// (rowidx/colidx/values) are allocated in HBM, the slice descriptors in host memory.
// Block geometry: bm/bn/bk only partition the work (results do not depend on them); the reference derives them
// from the CPU ISA and thread count (src/libxsmm_spmdm.c:555-608: bm 256|512 shrunk for load balance, bn 96|48|6,
// bk 128). Here they are sized for the GPU: a block call is one launch, so blocks are as large as the uint16 slice-local
// indexes and counters allow -- bm = 512 rows, bk = 64 columns (a slice holds at most 32 768 entries; 64 is also the
// width of a wavefront: a row of a slice is one load per wave and at most one entry per lane), bn = 2048.
// A createSparseSlice block is a row block of A: block id = mb covers the slices (kb, mb) of all column blocks kb (the
// reference hands out one slice per id, mb * kb ids; a caller only sees libxsmm_spmdm_get_num_createSparseSlice_blocks
// and the ids below it) -- one launch then has kb work-groups instead of one.
// Contract kept from the reference (compute tpl :38-39, 509-558): a compute call writes the C tile of its block and
// nothing else, whatever the other calls did; a create call writes the slices of its block and nothing else. Calls are
// asynchronous for device operands. The whole problem in one launch is an explicit extension: libxsmm_amd_spmdm_*_all.
// ---------------------------------------------------------------------------------------------------------------
namespace xsmm {
int launch_spmdm_compute_tiled(int M, int N, int K, int bm, int bk, int mb, int kb, int transb, int transc, float beta,
                               const uint16_t* rowidx, const uint16_t* colidx, const float* values, long long rowidx_stride, long long cap,
                               const float* b, float* c, int mb_begin, int mb_n, int n_begin, int n_end, void* stream, const char** name);
}

LIBXSMM_API void libxsmm_spmdm_init(int M, int N, int K, int max_threads,
  libxsmm_spmdm_handle* handle, libxsmm_CSR_sparseslice** libxsmm_output_csr)
{
  (void)max_threads;
  libxsmm_init();
  if (nullptr == handle || nullptr == libxsmm_output_csr) return;
  memset(handle, 0, sizeof(*handle));
  *libxsmm_output_csr = nullptr;
  handle->m = M; handle->n = N; handle->k = K;
  handle->bm = 512;
  if (handle->bm > M && 0 < M) handle->bm = M; // no point in a block taller than the matrix (keeps scratch small)
  handle->bn = 2048; handle->bk = 64;
  handle->mb = (M + handle->bm - 1) / handle->bm;
  handle->nb = (N + handle->bn - 1) / handle->bn;
  handle->kb = (K + handle->bk - 1) / handle->bk;
  handle->datatype = LIBXSMM_SPMDM_DATATYPE_F32;
  if (!device_ready()) { fail_no_device("libxsmm_spmdm_init"); return; }
  const size_t nslices = (size_t)handle->mb * handle->kb;
  const size_t cap = (size_t)handle->bm * handle->bk, rstride = (size_t)handle->bm + 1;
  // one device block: [values | colidx | rowidx] for all slices (capacity per slice as in the reference :109-112)
  const size_t bytes = nslices * (rstride * sizeof(uint16_t) + cap * sizeof(uint16_t) + cap * sizeof(float)) + 256;
  char* const block = static_cast<char*>(dev_alloc(bytes));
  libxsmm_CSR_sparseslice* const slices = static_cast<libxsmm_CSR_sparseslice*>(calloc(nslices ? nslices : 1, sizeof(libxsmm_CSR_sparseslice)));
  if (nullptr == block || nullptr == slices) {
    if (0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM ERROR: SPMDM CSR scratch memory allocation failed!\n");
    dev_free(block); free(slices); return;
  }
  float* const values = reinterpret_cast<float*>(block);
  uint16_t* const colidx = reinterpret_cast<uint16_t*>(block + nslices * cap * sizeof(float));
  uint16_t* const rowidx = colidx + nslices * cap;
  for (size_t i = 0; i < nslices; ++i) {
    slices[i].rowidx = rowidx + i * rstride; slices[i].colidx = colidx + i * cap; slices[i].values = values + i * cap;
  }
  handle->base_ptr_scratch_A = block;                                         // device memory, owned by the library
  handle->base_ptr_scratch_B_scratch_C = reinterpret_cast<char*>(slices);     // host array of slice descriptors
  handle->memory_for_scratch_per_thread = 0;
  *libxsmm_output_csr = slices;
}

LIBXSMM_API void libxsmm_spmdm_destroy(libxsmm_spmdm_handle* handle)
{
  if (nullptr == handle) return;
  spmdm_flush_record();
  if (device_ready()) (void)stream_sync();
  dev_free(handle->base_ptr_scratch_A); handle->base_ptr_scratch_A = nullptr;
  free(handle->base_ptr_scratch_B_scratch_C); handle->base_ptr_scratch_B_scratch_C = nullptr;
}

LIBXSMM_API int libxsmm_spmdm_get_num_createSparseSlice_blocks(const libxsmm_spmdm_handle* handle) { return handle->mb; }
LIBXSMM_API int libxsmm_spmdm_get_num_compute_blocks(const libxsmm_spmdm_handle* handle) { return handle->mb * handle->nb; }

namespace {
// Dense operands of the single-problem API may be host memory (an unchanged caller). They are mirrored in device scratch
// buffers that have the shape of the whole matrix, but only the part a block call needs travels: rows [r0, r0 + nr) x
// columns [c0, c0 + nc) of a row-major matrix with leading dimension ld (one strided copy).
int copy_window(float* dev, float* host, int ld, int r0, int nr, int c0, int nc, bool to_device)
{
  if (0 >= nr || 0 >= nc) return 0;
  float* const d = dev + (size_t)r0 * ld + c0; float* const h = host + (size_t)r0 * ld + c0;
  const size_t pitch = (size_t)ld * sizeof(float), width = (size_t)nc * sizeof(float);
  hipStream_t st = (hipStream_t)device().stream;
  const hipError_t e = to_device ? hipMemcpy2DAsync(d, pitch, h, pitch, width, (size_t)nr, hipMemcpyHostToDevice, st)
                                 : hipMemcpy2DAsync(h, pitch, d, pitch, width, (size_t)nr, hipMemcpyDeviceToHost, st);
  return (int)e;
}

// bfloat16 operands (reference: libxsmm_bfloat16 = the upper half of an IEEE float, src/libxsmm_spmdm_begin.h:69-75) are
// widened to float in a device scratch buffer; from there on the fp32 kernels run unchanged -- exactly what the reference
// templates do element by element (EXPAND_BFLOAT16 before the compare / the copy into scratch_B).
// (first, count): the elements a block call needs when they are one contiguous piece of the matrix -- rows of A, rows of a transposed
// B -- else the whole matrix; the scratch images keep the shape of the whole matrix, so the kernels address them as usual
const float* widen_bf16(const libxsmm_bfloat16* p, size_t elems, int slot_raw, int slot_wide, bool* ok, size_t first = 0, size_t count = 0)
{
  if (0 == count || first + count > elems) { first = 0; count = elems; }
  const libxsmm_bfloat16* src = p;
  if (!is_device_ptr(p)) {
    char* d = static_cast<char*>(scratch(slot_raw, elems * sizeof(libxsmm_bfloat16)));
    if (nullptr == d || 0 != h2d(d + first * sizeof(libxsmm_bfloat16), p + first, count * sizeof(libxsmm_bfloat16))) { *ok = false; return nullptr; }
    src = reinterpret_cast<const libxsmm_bfloat16*>(d);
  }
  float* const wide = static_cast<float*>(scratch(slot_wide, elems * sizeof(float)));
  if (nullptr == wide || 0 != launch_bf16_widen(src + first, wide + first, (long long)count, device().stream)) { *ok = false; return nullptr; }
  note_launch("bf16_widen");
  return wide;
}

struct SliceArrays { float* values; uint16_t* colidx; uint16_t* rowidx; size_t cap, rstride; };
SliceArrays slice_arrays(const libxsmm_spmdm_handle* handle)
{
  SliceArrays s;
  const size_t nslices = (size_t)handle->mb * handle->kb;
  s.cap = (size_t)handle->bm * handle->bk; s.rstride = (size_t)handle->bm + 1;
  s.values = reinterpret_cast<float*>(handle->base_ptr_scratch_A);
  s.colidx = reinterpret_cast<uint16_t*>(handle->base_ptr_scratch_A + nslices * s.cap * sizeof(float));
  s.rowidx = s.colidx + nslices * s.cap;
  return s;
}

// the slices of the row blocks [mb0, mb0 + mbn) of the device-resident M x K (or K x M) fp32 matrix da
int spmdm_create_slices(const libxsmm_spmdm_handle* handle, int ta, const float* da, int mb0, int mbn)
{
  const SliceArrays s = slice_arrays(handle);
  const char* name = "";
  // one row block: slices mb0 + kb * mb; all row blocks: every slice
  const bool all = (0 == mb0 && mbn == handle->mb);
  const int e = launch_spmdm_create_blocks(handle->m, handle->k, handle->bm, handle->bk, handle->mb, all ? 0 : mb0, all ? 1 : handle->mb,
    all ? handle->mb * handle->kb : handle->kb, ta, da, s.rowidx, s.colidx, s.values, device().stream, &name);
  note_launch(name);
  if (0 != e) fprintf(stderr, "LIBXSMM-AMD ERROR: kernel launch failed (%s, hip error %d)\n", name, e);
  return e;
}

// C tiles of the row blocks [mb0, mb0 + mbn) x columns [n0, n1): device-resident B and C
int spmdm_compute_tiles(const libxsmm_spmdm_handle* handle, int tb, int tc, float beta, const float* db, float* dc, int mb0, int mbn, int n0, int n1)
{
  const SliceArrays s = slice_arrays(handle);
  const char* name = "";
  int e = launch_spmdm_compute_tiled(handle->m, handle->n, handle->k, handle->bm, handle->bk, handle->mb, handle->kb, tb, tc, beta,
    s.rowidx, s.colidx, s.values, (long long)s.rstride, (long long)s.cap, db, dc, mb0, mbn, n0, n1, device().stream, &name);
  if (e < 0) { // a geometry the tiled kernel does not take (handles not made by libxsmm_spmdm_init): a thread per C element
    const int m0 = mb0 * handle->bm, m1 = LIBXSMM_MIN(m0 + mbn * handle->bm, handle->m);
    e = launch_spmdm_compute_generic(1, handle->m, handle->n, handle->k, handle->bm, handle->bk, handle->mb, handle->kb, tb, tc, beta,
      s.rowidx, s.colidx, s.values, (long long)s.rstride, (long long)s.cap, db, dc, 0, 0, m0, m1, n0, n1, device().stream, &name);
  }
  note_launch(name);
  if (0 != e) fprintf(stderr, "LIBXSMM-AMD ERROR: kernel launch failed (%s, hip error %d)\n", name, e);
  return e;
}

bool is_trans(char t) { return 'T' == t || 't' == t; }

// one create call: row block mb0 (mbn == 1) or all of them; a: float (widened already) or the caller's matrix
// (staged_input: `a` is the engine's widened copy; sync_after: the caller's matrix came from host memory -- the call returns when it is
// done with, as an unchanged CPU caller expects; device inputs stay asynchronous: the scratch copies are reused in stream order)
// returns 0, or the error of the failed allocation / copy / launch
int spmdm_create(const libxsmm_spmdm_handle* handle, char transa, const float* a, int mb0, int mbn, bool staged_input, bool sync_after = true)
{
  const int ta = is_trans(transa) ? 1 : 0;
  const float* da = a;
  if (!staged_input && !is_device_ptr(a)) { // host matrix: only the rows of these blocks travel
    float* const d = static_cast<float*>(scratch(3, (size_t)handle->m * handle->k * sizeof(float)));
    if (nullptr == d) return -1;
    const int r0 = mb0 * handle->bm, nr = LIBXSMM_MIN(mbn * handle->bm, handle->m - r0);
    const int e = ta ? copy_window(d, const_cast<float*>(a), handle->m, 0, handle->k, r0, nr, true) : copy_window(d, const_cast<float*>(a), handle->k, r0, nr, 0, handle->k, true);
    if (0 != e) return e;
    da = d; staged_input = true;
  }
  const int e = spmdm_create_slices(handle, ta, da, mb0, mbn);
  if (0 != e) return e;
  if (staged_input) { if (sync_after) return stream_sync(); }
  else settle(a);
  return 0;
}

// one compute call: row blocks [mb0, mb0 + mbn) x columns [n0, n1) of C
int spmdm_compute(const libxsmm_spmdm_handle* handle, char transb, const float* b, bool staged_b, char transc, float beta, float* c,
                  int mb0, int mbn, int n0, int n1, bool sync_after = true)
{
  const int tb = is_trans(transb) ? 1 : 0, tc = is_trans(transc) ? 1 : 0;
  const float* db = b;
  if (!staged_b && !is_device_ptr(b)) {
    float* const d = static_cast<float*>(scratch(4, (size_t)handle->k * handle->n * sizeof(float)));
    if (nullptr == d) return -1;
    // columns [n0, n1) of B (all k): B[k][n] resp. rows [n0, n1) of B[n][k]
    const int e = tb ? copy_window(d, const_cast<float*>(b), handle->k, n0, n1 - n0, 0, handle->k, true) : copy_window(d, const_cast<float*>(b), handle->n, 0, handle->k, n0, n1 - n0, true);
    if (0 != e) return e;
    db = d; staged_b = true;
  }
  const int m0 = mb0 * handle->bm, m1 = LIBXSMM_MIN(m0 + mbn * handle->bm, handle->m);
  float* dc = c;
  const bool c_host = !is_device_ptr(c);
  if (c_host) { // the tile travels in (beta != 0) and out; nothing else of C is touched
    dc = static_cast<float*>(scratch(5, (size_t)handle->m * handle->n * sizeof(float)));
    if (nullptr == dc) return -1;
    if (0.f != beta) {
      const int e = tc ? copy_window(dc, c, handle->m, n0, n1 - n0, m0, m1 - m0, true) : copy_window(dc, c, handle->n, m0, m1 - m0, n0, n1 - n0, true);
      if (0 != e) return e;
    }
  }
  const int e = spmdm_compute_tiles(handle, tb, tc, beta, db, dc, mb0, mbn, n0, n1);
  if (0 != e) return e;
  if (c_host) {
    const int e2 = tc ? copy_window(dc, c, handle->m, n0, n1 - n0, m0, m1 - m0, false) : copy_window(dc, c, handle->n, m0, m1 - m0, n0, n1 - n0, false);
    const int e3 = stream_sync();
    return 0 != e2 ? e2 : e3;
  }
  if (staged_b) { if (sync_after) return stream_sync(); } // (host inputs: done with on return; device inputs: stream order)
  else settle(b, c);
  return 0;
}

bool spmdm_block_ok(const libxsmm_spmdm_handle* handle, int block_id, int nblocks, const char* what)
{
  if (0 <= block_id && block_id < nblocks) return true;
  static int error_once = 0;
  if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: %s: block id %d out of range (handle has %d blocks)!\n", what, block_id, nblocks);
  (void)handle;
  return false;
}

// ---- a block call's work, for the row blocks [mb0, mb0 + mbn) resp. the C tiles [mb0, mb0 + mbn) x [n0, n1) ------------------------
int create_blocks_f32(const libxsmm_spmdm_handle* handle, char transa, const float* a, int mb0, int mbn)
{
  return spmdm_create(handle, transa, a, mb0, mbn, false);
}

int create_blocks_bf16(const libxsmm_spmdm_handle* handle, char transa, const libxsmm_bfloat16* a, int mb0, int mbn)
{ // only the rows of these blocks are widened (A as stored: they are contiguous; a transposed A: the whole matrix)
  bool ok = true;
  const size_t r0 = (size_t)mb0 * handle->bm, nr = (size_t)LIBXSMM_MIN(mbn * handle->bm, handle->m - (int)r0);
  const bool rows_contiguous = !is_trans(transa);
  const float* const da = widen_bf16(a, (size_t)handle->m * handle->k, 6, 3, &ok, rows_contiguous ? r0 * handle->k : 0, rows_contiguous ? nr * handle->k : 0);
  if (!ok) return -1;
  return spmdm_create(handle, transa, da, mb0, mbn, true, !is_device_ptr(a));
}

int compute_tiles_bf16(const libxsmm_spmdm_handle* handle, char transb, const libxsmm_bfloat16* b, char transc, float beta, float* c,
                       int mb0, int mbn, int n0, int n1)
{ // only the columns [n0, n1) of B are widened where they are one piece (a transposed B: rows of B^T; else: all of B)
  bool ok = true;
  const bool piece = is_trans(transb);
  const float* const db = widen_bf16(b, (size_t)handle->k * handle->n, 6, 4, &ok, piece ? (size_t)n0 * handle->k : 0, piece ? (size_t)(n1 - n0) * handle->k : 0);
  if (!ok) return -1;
  return spmdm_compute(handle, transb, db, true, transc, beta, c, mb0, mbn, n0, n1, !is_device_ptr(b));
}

// ---- block calls inside a libxsmm_amd_defer_begin/end bracket ----------------------------------------------------------------------
// The reference's caller walks the block ids (samples/spmdm/spmdm.c:99-109), one call per block; a call is a launch that covers a
// fraction of the chip (2048^3: four of each kind, a quarter of the tiles each). Inside the bracket -- the caller's promise that it
// queues nothing of its own on the stream that touches the operands before the bracket ends or libxsmm_amd_flush() is called -- calls of
// one kind on one handle with the same operands (pure device memory) are only recorded: the blocks merge into rectangles of blocks, and
// whatever ends the record (another kind of call, other operands, any other entry point of the library on this thread,
// libxsmm_amd_flush / libxsmm_amd_defer_end) launches one kernel per rectangle -- one for a full sweep. Every block still touches only
// its own slices / its own C tile; results are the same bits (the kernels are the ones a block call launches, on a wider range).
// Outside a bracket, and for operands the CPU addresses, a call is a launch (and a copy) of its own, as before.
struct SpmdmRect { int mb0, mbn, n0, n1; };
struct SpmdmRecord {
  int kind = 0;                         // 0: none, 1: createSparseSlice, 2: compute
  libxsmm_spmdm_handle handle;          // (a copy: the caller's struct may be a temporary; the scratch pointers identify it)
  const void* src = nullptr; bool bf16 = false; char trans = 'N', transc = 'N'; float beta = 0.f; float* c = nullptr;
  std::vector<SpmdmRect> rects;
};
thread_local SpmdmRecord tl_record;

bool pure_device(const void* p) { return is_device_ptr(p) && !is_host_visible(p); }

// the new rectangle joins the last one when the two form a rectangle again (block ids ascend: row-major over (mb, nb))
void add_rect(std::vector<SpmdmRect>& rects, SpmdmRect r)
{
  rects.push_back(r);
  while (rects.size() >= 2) {
    SpmdmRect& u = rects[rects.size() - 2]; const SpmdmRect& v = rects.back();
    if (u.mb0 == v.mb0 && u.mbn == v.mbn && u.n1 == v.n0) u.n1 = v.n1;                 // next to it
    else if (u.n0 == v.n0 && u.n1 == v.n1 && u.mb0 + u.mbn == v.mb0) u.mbn += v.mbn;   // below it
    else break;
    rects.pop_back();
  }
}

bool record_block(int kind, const libxsmm_spmdm_handle* handle, const void* src, bool bf16, char trans, char transc, float beta, float* c, SpmdmRect r)
{
  if (!defer_bracket_open() || !pure_device(src) || (2 == kind && !pure_device(c))) return false;
  if (tl_defer_open) defer_flush(); // an open burst of per-call kernels is sealed: later calls of that kernel must not run ahead of this block
  SpmdmRecord& p = tl_record;
  if (tl_spmdm_open && !(p.kind == kind && p.handle.base_ptr_scratch_A == handle->base_ptr_scratch_A && p.src == src && p.bf16 == bf16
      && is_trans(p.trans) == is_trans(trans) && is_trans(p.transc) == is_trans(transc) && p.beta == beta && p.c == c)) spmdm_flush_record();
  if (!tl_spmdm_open) {
    p.kind = kind; p.handle = *handle; p.src = src; p.bf16 = bf16; p.trans = trans; p.transc = transc; p.beta = beta; p.c = c; p.rects.clear();
    tl_spmdm_open = true;
  }
  for (const SpmdmRect& q : p.rects) { // a block that is recorded already (beta != 0: it must run twice): launch what is there first
    if (q.mb0 < r.mb0 + r.mbn && r.mb0 < q.mb0 + q.mbn && q.n0 < r.n1 && r.n0 < q.n1) { spmdm_flush_record(); return record_block(kind, handle, src, bf16, trans, transc, beta, c, r); }
  }
  add_rect(p.rects, r);
  return true;
}
}

namespace xsmm {
thread_local bool tl_spmdm_open = false;
void spmdm_flush_record()
{
  if (!tl_spmdm_open) return;
  tl_spmdm_open = false; // (first: the launches below ask for the stream, which flushes what is open)
  SpmdmRecord& p = tl_record;
  const libxsmm_spmdm_handle* const h = &p.handle;
  if (1 == p.kind) {
    for (const SpmdmRect& r : p.rects) {
      if (0 == r.mb0 && r.mbn == h->mb) { // every row block: one launch
        (void)(p.bf16 ? create_blocks_bf16(h, p.trans, static_cast<const libxsmm_bfloat16*>(p.src), 0, h->mb) : create_blocks_f32(h, p.trans, static_cast<const float*>(p.src), 0, h->mb));
      }
      else for (int mb = r.mb0; mb < r.mb0 + r.mbn; ++mb) { // (a slice range that is not every slice: a launch per row block)
        (void)(p.bf16 ? create_blocks_bf16(h, p.trans, static_cast<const libxsmm_bfloat16*>(p.src), mb, 1) : create_blocks_f32(h, p.trans, static_cast<const float*>(p.src), mb, 1));
      }
    }
  }
  else if (2 == p.kind) {
    const float* wide = nullptr;
    if (p.bf16 && 1 < p.rects.size()) { // several rectangles: B is widened once
      bool ok = true;
      wide = widen_bf16(static_cast<const libxsmm_bfloat16*>(p.src), (size_t)h->k * h->n, 6, 4, &ok);
      if (!ok) { p.rects.clear(); p.kind = 0; return; }
    }
    for (const SpmdmRect& r : p.rects) {
      if (!p.bf16) (void)spmdm_compute(h, p.trans, static_cast<const float*>(p.src), false, p.transc, p.beta, p.c, r.mb0, r.mbn, r.n0, r.n1);
      else if (nullptr != wide) (void)spmdm_compute(h, p.trans, wide, true, p.transc, p.beta, p.c, r.mb0, r.mbn, r.n0, r.n1, false);
      else (void)compute_tiles_bf16(h, p.trans, static_cast<const libxsmm_bfloat16*>(p.src), p.transc, p.beta, p.c, r.mb0, r.mbn, r.n0, r.n1);
    }
  }
  p.rects.clear(); p.kind = 0;
}
}

LIBXSMM_API void libxsmm_spmdm_createSparseSlice_fp32_thread(const libxsmm_spmdm_handle* handle, char transa,
  const float* a, libxsmm_CSR_sparseslice* libxsmm_output_csr_a, int block_id, int tid, int nthreads)
{
  (void)tid; (void)nthreads;
  if (nullptr == handle || nullptr == a || nullptr == libxsmm_output_csr_a || nullptr == handle->base_ptr_scratch_A) return;
  if (!device_ready()) { fail_no_device("libxsmm_spmdm_createSparseSlice_fp32_thread"); return; }
  if (!spmdm_block_ok(handle, block_id, handle->mb, "libxsmm_spmdm_createSparseSlice_fp32_thread")) return;
  if (record_block(1, handle, a, false, transa, 'N', 0.f, nullptr, SpmdmRect{ block_id, 1, 0, handle->n })) return;
  (void)create_blocks_f32(handle, transa, a, block_id, 1);
}

LIBXSMM_API void libxsmm_spmdm_createSparseSlice_bfloat16_thread(const libxsmm_spmdm_handle* handle, char transa,
  const libxsmm_bfloat16* a, libxsmm_CSR_sparseslice* libxsmm_output_csr_a, int block_id, int tid, int nthreads)
{ // src/template/libxsmm_spmdm_createSparseSlice_bfloat16_thread.tpl.c:47-144: widen, keep v != 0, store as float
  (void)tid; (void)nthreads;
  if (nullptr == handle || nullptr == a || nullptr == libxsmm_output_csr_a || nullptr == handle->base_ptr_scratch_A) return;
  if (!device_ready()) { fail_no_device("libxsmm_spmdm_createSparseSlice_bfloat16_thread"); return; }
  if (!spmdm_block_ok(handle, block_id, handle->mb, "libxsmm_spmdm_createSparseSlice_bfloat16_thread")) return;
  if (record_block(1, handle, a, true, transa, 'N', 0.f, nullptr, SpmdmRect{ block_id, 1, 0, handle->n })) return;
  (void)create_blocks_bf16(handle, transa, a, block_id, 1);
}

LIBXSMM_API void libxsmm_spmdm_compute_fp32_thread(const libxsmm_spmdm_handle* handle, char transa, char transb,
  const float* alpha, libxsmm_CSR_sparseslice* a_sparse, const float* b, char transc, const float* beta, float* c,
  int block_id, int tid, int nthreads)
{
  (void)transa; (void)alpha; (void)tid; (void)nthreads; // alpha is ignored by the reference (include/libxsmm_spmdm.h:104)
  if (nullptr == handle || nullptr == a_sparse || nullptr == b || nullptr == c || nullptr == beta) return;
  if (!device_ready()) { fail_no_device("libxsmm_spmdm_compute_fp32_thread"); return; }
  if (!spmdm_block_ok(handle, block_id, handle->mb * handle->nb, "libxsmm_spmdm_compute_fp32_thread")) return;
  const int mb = block_id / handle->nb, nb = block_id % handle->nb; // compute tpl :38-39
  const int n0 = nb * handle->bn, n1 = LIBXSMM_MIN((nb + 1) * handle->bn, handle->n);
  if (record_block(2, handle, b, false, transb, transc, *beta, c, SpmdmRect{ mb, 1, n0, n1 })) return;
  (void)spmdm_compute(handle, transb, b, false, transc, *beta, c, mb, 1, n0, n1);
}

LIBXSMM_API void libxsmm_spmdm_compute_bfloat16_thread(const libxsmm_spmdm_handle* handle, char transa, char transb,
  const libxsmm_bfloat16* alpha, libxsmm_CSR_sparseslice* a_sparse, const libxsmm_bfloat16* b, char transc,
  const libxsmm_bfloat16* beta, float* c, int block_id, int tid, int nthreads)
{ // src/template/libxsmm_spmdm_compute_bfloat16_thread.tpl.c: B is widened while it is copied, C and the sums are float.
  // NOTE the reference reads `*beta` as a number without widening it (:91,113,164): the 16-bit pattern itself is the
  // factor (pattern 0 -> beta 0, pattern 1 -> beta 1, bf16(1.0) = 0x3F80 -> 16256). Reproduced as is.
  (void)transa; (void)alpha; (void)tid; (void)nthreads;
  if (nullptr == handle || nullptr == a_sparse || nullptr == b || nullptr == c || nullptr == beta) return;
  if (!device_ready()) { fail_no_device("libxsmm_spmdm_compute_bfloat16_thread"); return; }
  if (!spmdm_block_ok(handle, block_id, handle->mb * handle->nb, "libxsmm_spmdm_compute_bfloat16_thread")) return;
  const int mb = block_id / handle->nb, nb = block_id % handle->nb;
  const int n0 = nb * handle->bn, n1 = LIBXSMM_MIN((nb + 1) * handle->bn, handle->n);
  if (record_block(2, handle, b, true, transb, transc, (float)(*beta), c, SpmdmRect{ mb, 1, n0, n1 })) return;
  (void)compute_tiles_bf16(handle, transb, b, transc, (float)(*beta), c, mb, 1, n0, n1);
}

// ---- the whole problem in one call (extension) -------------------------------------------------------------------------
// Equivalent to calling the *_thread function for every block id, as one launch that fills the chip: what a caller that
// owns the whole loop (samples/spmdm/spmdm.c:74-112) uses instead of the loop.
LIBXSMM_API int libxsmm_amd_spmdm_createSparseSlice_all(const libxsmm_spmdm_handle* handle, char transa, const float* a,
  libxsmm_CSR_sparseslice* libxsmm_output_csr_a)
{
  if (nullptr == handle || nullptr == a || nullptr == libxsmm_output_csr_a || nullptr == handle->base_ptr_scratch_A) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_amd_spmdm_createSparseSlice_all"); return EXIT_FAILURE; }
  return 0 == spmdm_create(handle, transa, a, 0, handle->mb, false) ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_amd_spmdm_compute_all(const libxsmm_spmdm_handle* handle, char transa, char transb, const float* alpha,
  libxsmm_CSR_sparseslice* a_sparse, const float* b, char transc, const float* beta, float* c)
{
  (void)transa; (void)alpha;
  if (nullptr == handle || nullptr == a_sparse || nullptr == b || nullptr == c || nullptr == beta) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_amd_spmdm_compute_all"); return EXIT_FAILURE; }
  return 0 == spmdm_compute(handle, transb, b, false, transc, *beta, c, 0, handle->mb, 0, handle->n) ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_amd_spmdm_createSparseSlice_bfloat16_all(const libxsmm_spmdm_handle* handle, char transa, const libxsmm_bfloat16* a,
  libxsmm_CSR_sparseslice* libxsmm_output_csr_a)
{
  if (nullptr == handle || nullptr == a || nullptr == libxsmm_output_csr_a || nullptr == handle->base_ptr_scratch_A) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_amd_spmdm_createSparseSlice_bfloat16_all"); return EXIT_FAILURE; }
  bool ok = true;
  const float* const da = widen_bf16(a, (size_t)handle->m * handle->k, 6, 3, &ok);
  if (!ok) return EXIT_FAILURE;
  return 0 == spmdm_create(handle, transa, da, 0, handle->mb, true, !is_device_ptr(a)) ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_amd_spmdm_compute_bfloat16_all(const libxsmm_spmdm_handle* handle, char transa, char transb, const libxsmm_bfloat16* alpha,
  libxsmm_CSR_sparseslice* a_sparse, const libxsmm_bfloat16* b, char transc, const libxsmm_bfloat16* beta, float* c)
{
  (void)transa; (void)alpha;
  if (nullptr == handle || nullptr == a_sparse || nullptr == b || nullptr == c || nullptr == beta) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_amd_spmdm_compute_bfloat16_all"); return EXIT_FAILURE; }
  bool ok = true;
  const float* const db = widen_bf16(b, (size_t)handle->k * handle->n, 6, 4, &ok);
  if (!ok) return EXIT_FAILURE;
  return 0 == spmdm_compute(handle, transb, db, true, transc, (float)(*beta), c, 0, handle->mb, 0, handle->n, !is_device_ptr(b)) ? EXIT_SUCCESS : EXIT_FAILURE;
}

// ---------------------------------------------------------------------------------------------------------------
// spmdm batch extension: `batch` independent problems, one slice per item (bm = M, bk = K).
// HBM layout: rowidx[batch][M+1] (u16), colidx[batch][M*K] (u16), values[batch][M*K] (f32): fixed-capacity slots,
// only the first nnz entries of a slot are ever touched.
// ---------------------------------------------------------------------------------------------------------------
struct libxsmm_amd_spmdm_batch {
  SpmdmGeom g;
  uint16_t* rowidx; uint16_t* colidx; float* values;
};

LIBXSMM_API libxsmm_amd_spmdm_batch* libxsmm_amd_spmdm_batch_create(int M, int N, int K, long long batch)
{
  libxsmm_init();
  if (M <= 0 || N <= 0 || K <= 0 || batch < 0 || (long long)M * K > 65535 || K > 65535) return nullptr; // uint16 counters/indexes
  if (!device_ready()) { fail_no_device("libxsmm_amd_spmdm_batch_create"); return nullptr; }
  libxsmm_amd_spmdm_batch* sb = static_cast<libxsmm_amd_spmdm_batch*>(calloc(1, sizeof(*sb)));
  if (nullptr == sb) return nullptr;
  // slot sizes: capacity rounded up to 8 entries and rowidx to an even count, so that every item's arrays start on a
  // 16-byte boundary (vector loads in the compute kernel); a few entries of slack follow the last slot
  sb->g.m = M; sb->g.n = N; sb->g.k = K; sb->g.batch = batch; sb->g.cap = (M * K + 7) & ~7; sb->g.rstride = (M + 2) & ~1;
  const size_t nb = (size_t)(batch ? batch : 1);
  sb->rowidx = static_cast<uint16_t*>(dev_alloc((nb * (size_t)sb->g.rstride + 64) * sizeof(uint16_t)));
  sb->colidx = static_cast<uint16_t*>(dev_alloc((nb * (size_t)sb->g.cap + 64) * sizeof(uint16_t)));
  sb->values = static_cast<float*>(dev_alloc((nb * (size_t)sb->g.cap + 64) * sizeof(float)));
  if (nullptr == sb->rowidx || nullptr == sb->colidx || nullptr == sb->values) { libxsmm_amd_spmdm_batch_destroy(sb); return nullptr; }
  return sb;
}

LIBXSMM_API void libxsmm_amd_spmdm_batch_destroy(libxsmm_amd_spmdm_batch* sb)
{
  if (nullptr == sb) return;
  if (device_ready()) (void)stream_sync();
  dev_free(sb->rowidx); dev_free(sb->colidx); dev_free(sb->values);
  free(sb);
}

LIBXSMM_API int libxsmm_amd_spmdm_batch_create_slices(libxsmm_amd_spmdm_batch* sb, char transa, const float* a)
{
  if (nullptr == sb || nullptr == a || !is_device_ptr(a)) return EXIT_FAILURE;
  const char* name = "";
  const int e = launch_spmdm_create(sb->g, ('T' == transa || 't' == transa) ? 1 : 0, a, sb->rowidx, sb->colidx, sb->values, device().stream, &name);
  note_launch(name);
  if (0 == e) settle(a);
  return 0 == e ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_amd_spmdm_batch_compute(libxsmm_amd_spmdm_batch* sb, char transb, const float* b,
  char transc, const float* beta, float* c)
{
  if (nullptr == sb || nullptr == b || nullptr == c || nullptr == beta || !is_device_ptr(b) || !is_device_ptr(c)) return EXIT_FAILURE;
  const char* name = "";
  const int e = launch_spmdm_compute(sb->g, ('T' == transb || 't' == transb) ? 1 : 0, ('T' == transc || 't' == transc) ? 1 : 0, *beta,
    sb->rowidx, sb->colidx, sb->values, b, c, device().stream, &name);
  note_launch(name);
  if (0 == e) settle(b, c);
  return 0 == e ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_amd_spmdm_batch_get_slice(const libxsmm_amd_spmdm_batch* sb, long long item,
  uint16_t* rowidx, uint16_t* colidx, float* values, int capacity)
{
  if (nullptr == sb || item < 0 || item >= sb->g.batch || nullptr == rowidx) return EXIT_FAILURE;
  if (0 != d2h(rowidx, sb->rowidx + item * sb->g.rstride, ((size_t)sb->g.m + 1) * sizeof(uint16_t))) return EXIT_FAILURE;
  const int nnz = rowidx[sb->g.m];
  const int ncopy = LIBXSMM_MIN(nnz, capacity);
  if (nullptr != colidx && 0 < ncopy && 0 != d2h(colidx, sb->colidx + item * sb->g.cap, (size_t)ncopy * sizeof(uint16_t))) return EXIT_FAILURE;
  if (nullptr != values && 0 < ncopy && 0 != d2h(values, sb->values + item * sb->g.cap, (size_t)ncopy * sizeof(float))) return EXIT_FAILURE;
  return EXIT_SUCCESS;
}
