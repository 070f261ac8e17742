/*
 * libxsmm_amd.h -- engine-specific additions to the LIBXSMM C-ABI (no counterpart in the reference).
 *
 * The reference is a CPU library: a kernel call returns when C is written. On MI355X the operands
 * normally live in HBM and a call only enqueues work on a HIP stream. These entry points expose that
 * model (stream, synchronisation, device allocation) plus batch forms for the paths whose reference
 * API processes one problem per call (spmdm, fsspmdm) -- the batch axis is what fills 256 CUs.
 */
#ifndef LIBXSMM_AMD_H
#define LIBXSMM_AMD_H

#if !defined(LIBXSMM_H)
# include "libxsmm.h"
#endif

/* ---- device / stream ------------------------------------------------------------------------ */
/** Number of usable HIP devices (0: none; every compute entry point then fails loudly). */
LIBXSMM_API int libxsmm_amd_device_count(void);
/** HIP stream (hipStream_t) on which the calling thread's device-resident work is enqueued; NULL selects the default
 *  stream. The setting is per thread (every entry point may be called from any thread): independent batches can be put
 *  on different streams, by one thread switching streams between calls or by several threads. */
LIBXSMM_API void libxsmm_amd_set_stream(void* hip_stream);
LIBXSMM_API void* libxsmm_amd_get_stream(void);
/** Calls of a dispatched kernel on device memory, kernel(a, b, c) once per product (samples/smm/specialized.cpp:172-190;
 *  likewise libxsmm_?fsspmdm_execute once per panel, samples/pyfr/pyfr_driver_asp_reg.c:300-308).
 *  DEFAULT: every call is an asynchronous launch of its own on the calling thread's stream; stream order is call order,
 *  so work the caller queues on that stream between two calls (own kernels, hipMemcpyAsync, torch operations) is
 *  ordered between them exactly as it was issued.
 *  OPT-IN, no launch per call: between libxsmm_amd_defer_begin() and libxsmm_amd_defer_end() on the calling thread (the
 *  analogue of the reference's libxsmm_mmbatch_begin/end bracket, src/libxsmm_ext_gemm.c:1016-1135; brackets nest), or
 *  process-wide with LIBXSMM_AMD_DEFER=1, consecutive calls form a burst: the first call queues a gate kernel and the
 *  batch kernel behind it on the stream, the following calls only append their operands to a ring in pinned memory and
 *  RUN AT THE STREAM POSITION OF THE BURST'S FIRST CALL. The caller's side of the contract: inside the bracket, before
 *  queueing work of its own on that stream that reads or writes operands of the calls -- call libxsmm_amd_flush().
 *  Work queued or waited for after libxsmm_amd_defer_end() / libxsmm_amd_flush() / any other entry point of the library
 *  on the thread is ordered behind all recorded calls (an idle burst is also sealed by a helper thread after a few
 *  microseconds, so a caller's hipStreamSynchronize inside the bracket never hangs). Calls that depend on each other
 *  (a C read as A by a later call, a C written again later) are detected from the operand addresses and keep the call
 *  order. libxsmm_amd_defer_active(): 1 if calls of this thread are being recorded.
 *  Inside a bracket (not with the environment variable alone) the block calls of the spmdm interface on device operands,
 *  libxsmm_spmdm_createSparseSlice_*_thread / libxsmm_spmdm_compute_*_thread (samples/spmdm/spmdm.c:99-109), are recorded
 *  as well: consecutive calls of one kind on one handle with the same operands merge into rectangles of blocks, launched
 *  -- one kernel per rectangle, one for a full sweep -- by whatever ends the record (a call of another kind or with other
 *  operands, libxsmm_amd_flush / libxsmm_amd_defer_end, any other entry point of the library on the thread). They run at
 *  the stream position of THAT moment; every block still touches only its own slices / C tile. */
LIBXSMM_API void libxsmm_amd_defer_begin(void);
LIBXSMM_API void libxsmm_amd_defer_end(void);
LIBXSMM_API int libxsmm_amd_defer_active(void);
LIBXSMM_API void libxsmm_amd_flush(void);
/** Block until all work enqueued by this library on its stream has completed. Returns EXIT_SUCCESS/FAILURE. */
LIBXSMM_API int libxsmm_amd_synchronize(void);
/** Device memory (hipMalloc/hipFree) -- what libxsmm_malloc returns when a device is present is host-pinned
 *  memory (usable by unchanged callers); these return HBM. */
LIBXSMM_API void* libxsmm_amd_device_malloc(size_t size);
LIBXSMM_API void libxsmm_amd_device_free(void* ptr);
LIBXSMM_API int libxsmm_amd_memcpy_h2d(void* dst_device, const void* src_host, size_t size);
LIBXSMM_API int libxsmm_amd_memcpy_d2h(void* dst_host, const void* src_device, size_t size);
/** 1 if ptr is device-accessible memory (hipMalloc, managed, or pinned host), else 0. */
LIBXSMM_API int libxsmm_amd_is_device_pointer(const void* ptr);

/* ---- kernel selection ----------------------------------------------------------------------- */
/** MFMA policy for dense SMM: 0 = scalar-FMA kernels only (bit-identical to a k-ordered fma chain),
 *  1 = use v_mfma_* where the shape maps onto matrix-core tiles (default; env LIBXSMM_AMD_MFMA).
 *  Returns the previous value. */
LIBXSMM_API int libxsmm_amd_set_mfma(int mode);
LIBXSMM_API int libxsmm_amd_get_mfma(void);
/** Name of the device kernel variant chosen for the most recent launch of the calling thread
 *  (e.g. "smm_f32_32x32x32_mfma"); "" if nothing was launched. Used by the tests to prove that the
 *  native path ran. */
LIBXSMM_API const char* libxsmm_amd_last_kernel(void);
/** Number of device kernel launches issued by this process (monotonic). */
LIBXSMM_API unsigned long long libxsmm_amd_launch_count(void);

/** Measurement aid: c[i] += a[i] + b[i] over `bytes` bytes per operand (device memory, 16-byte aligned) -- the
 *  3-read/1-write traffic mix of a beta=1 SMM batch with no arithmetic; bench.py reports its rate as the measured
 *  streaming ceiling next to the 8 TB/s datasheet peak. */
LIBXSMM_API int libxsmm_amd_stream_probe(const void* a, const void* b, void* c, long long bytes);

/* ---- batch forms ---------------------------------------------------------------------------- */
/** Constant-stride batch: item i uses a + i*stride_a, b + i*stride_b, c + i*stride_c (strides in elements,
 *  0 = operand shared). This is the layout of samples/smm/specialized.cpp:143-146,172-190 (contiguous
 *  A[s][m*k], B[s][k*n], C[s][m*n]) without materialising index arrays. Device-resident operands only.
 *  Returns EXIT_SUCCESS, or EXIT_FAILURE if the descriptor is not supported / no device. */
LIBXSMM_API int libxsmm_amd_gemm_batch_strided(const libxsmm_gemm_descriptor* descriptor,
  const void* a, const void* b, void* c, long long stride_a, long long stride_b, long long stride_c,
  long long batchsize);

/** Several index-array batches in one call -- CP2K-style stacks: one batch per shape (samples/cp2k/cp2k.cpp:328-360;
 *  the reference's one-call form for pointer arrays is libxsmm_?gemm_batch with its groups, src/libxsmm_gemm.c:1231-1262).
 *  Equivalent to libxsmm_gemm_batch(iprec, oprec, &transa[g], &transb[g], m[g], n[g], k[g], alpha, a[g], &lda[g], b[g],
 *  &ldb[g], beta, c[g], &ldc[g], index_base, index_stride, stride_a[g], stride_b[g], stride_c[g], group_size[g]) for every
 *  g < ngroups, with the groups' launches fused: the C-ordering check of all groups is one launch and, where the
 *  shape-specialised run kernels apply (M, N <= 32, K <= 64), so is the multiplication -- the accumulation chains of all
 *  shapes are resident at the same time instead of one shape after the other. Per C block the products are added in batch
 *  order as in libxsmm_gemm_batch (relaxed != 0: any order, as libxsmm_gemm_batch_omp). The groups must be independent of
 *  each other: no C block is written by two groups and no group reads (as A or B) what another group writes -- the
 *  fused groups run side by side (libxsmm_?gemm_batch with pointer arrays checks this itself and keeps dependent groups in
 *  order). transa/transb/lda/ldb/ldc may be NULL ('N', tight leading dimensions);
 *  alpha/beta: scalars of the precision (NULL: 1), the SMM domain only (alpha = 1, beta in {0, 1}, no TRANS_A). Matrices in
 *  device memory (or libxsmm_malloc memory); index arrays in device or host memory. Returns EXIT_SUCCESS/EXIT_FAILURE. */
LIBXSMM_API int libxsmm_amd_gemm_batch_groups(libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, int ngroups,
  const char transa[], const char transb[], const libxsmm_blasint m[], const libxsmm_blasint n[], const libxsmm_blasint k[],
  const libxsmm_blasint lda[], const libxsmm_blasint ldb[], const libxsmm_blasint ldc[], const void* alpha, const void* beta,
  const void* const a[], const void* const b[], void* const c[], libxsmm_blasint index_base, libxsmm_blasint index_stride,
  const libxsmm_blasint* const stride_a[], const libxsmm_blasint* const stride_b[], const libxsmm_blasint* const stride_c[],
  const libxsmm_blasint group_size[], int relaxed);

/** Batched spmdm: `batch` independent problems of the handle's geometry (M,N,K), operands back to back
 *  (A: M*K, B: K*N, C: M*N elements per item, layouts/transposes as libxsmm_spmdm_*_thread).
 *  The CSR scratch lives in HBM and is owned by the returned object. */
typedef struct libxsmm_amd_spmdm_batch libxsmm_amd_spmdm_batch;
LIBXSMM_API libxsmm_amd_spmdm_batch* libxsmm_amd_spmdm_batch_create(int M, int N, int K, long long batch);
LIBXSMM_API void libxsmm_amd_spmdm_batch_destroy(libxsmm_amd_spmdm_batch* sb);
/** dense A (device) -> per-item CSR slices (uint16 column indexes, identical content/order to
 *  libxsmm_spmdm_createSparseSlice_fp32_thread on every item). */
LIBXSMM_API int libxsmm_amd_spmdm_batch_create_slices(libxsmm_amd_spmdm_batch* sb, char transa, const float* a);
/** C = beta*C + A_sparse*B for every item (alpha ignored as in the reference). */
LIBXSMM_API int libxsmm_amd_spmdm_batch_compute(libxsmm_amd_spmdm_batch* sb, char transb, const float* b,
  char transc, const float* beta, float* c);
/** Geometry and CSR access for tests: copies item `i`'s rowidx (M+1), and the first nnz colidx/values to host. */
LIBXSMM_API int libxsmm_amd_spmdm_batch_get_slice(const libxsmm_amd_spmdm_batch* sb, long long item,
  uint16_t* rowidx, uint16_t* colidx, float* values, int capacity);

/** The reference spmdm interface in one call per phase: what the caller's loop over block ids does
 *  (samples/spmdm/spmdm.c:74-112: libxsmm_spmdm_createSparseSlice_fp32_thread for every id below
 *  libxsmm_spmdm_get_num_createSparseSlice_blocks, then libxsmm_spmdm_compute_fp32_thread for every id below
 *  libxsmm_spmdm_get_num_compute_blocks), as one launch over the whole problem. The *_thread functions keep the reference's
 *  contract -- a call touches the slice resp. the C tile of its block id and nothing else -- and cost one launch per block;
 *  these are for callers that own the whole loop. Same operands and semantics (alpha ignored, beta == 0 never reads C);
 *  device or host operands. Returns EXIT_SUCCESS/EXIT_FAILURE. */
LIBXSMM_API int libxsmm_amd_spmdm_createSparseSlice_all(const libxsmm_spmdm_handle* handle, char transa, const float* a,
  libxsmm_CSR_sparseslice* libxsmm_output_csr_a);
LIBXSMM_API int libxsmm_amd_spmdm_compute_all(const libxsmm_spmdm_handle* handle, char transa, char transb, const float* alpha,
  libxsmm_CSR_sparseslice* a_sparse, const float* b, char transc, const float* beta, float* c);
LIBXSMM_API int libxsmm_amd_spmdm_createSparseSlice_bfloat16_all(const libxsmm_spmdm_handle* handle, char transa, const libxsmm_bfloat16* a,
  libxsmm_CSR_sparseslice* libxsmm_output_csr_a);
LIBXSMM_API int libxsmm_amd_spmdm_compute_bfloat16_all(const libxsmm_spmdm_handle* handle, char transa, char transb, const libxsmm_bfloat16* alpha,
  libxsmm_CSR_sparseslice* a_sparse, const libxsmm_bfloat16* b, char transc, const libxsmm_bfloat16* beta, float* c);

/** Batched fsspmdm: the operator of `handle` applied to `batch` column panels of width handle->N that sit
 *  side by side in B (K x ldb) / C (M x ldc): panel i = columns [i*N, (i+1)*N). Equivalent to calling
 *  libxsmm_?fsspmdm_execute(handle, B + i*N, C + i*N) for every i (samples/pyfr/pyfr_driver_asp_reg.c:295-309). */
LIBXSMM_API int libxsmm_amd_dfsspmdm_execute_batch(const libxsmm_dfsspmdm* handle, const double* B, double* C, long long batch);
LIBXSMM_API int libxsmm_amd_sfsspmdm_execute_batch(const libxsmm_sfsspmdm* handle, const float* B, float* C, long long batch);

/** Text generator: the HIP source a fixed-sparsity operator (CSR pattern + values) is specialised to -- every referenced B
 *  row is loaded once, every non-zero is one fma with an immediate. This is what libxsmm_?fsspmdm_create compiles through
 *  hiprtc (counterpart of the reference's libxsmm_generator_spgemm_csr_kernel text output). The source is copied into
 *  `buffer` (truncated to buffer_size). compile == 0: returns the source length; compile != 0: additionally compiles it for
 *  gfx950 (no device needed) and returns 0 on success, > 0 on a compile error, -1 if hiprtc is unavailable / bad arguments. */
LIBXSMM_API int libxsmm_amd_csr_kernel_source(int typesize, int M, int K, const unsigned int* row_ptr, const unsigned int* column_idx,
  const double* values, int beta0, int vec, char* buffer, size_t buffer_size, int compile);

/** Text generator for dense SMM: the HIP source a descriptor (tight leading dimensions) is specialised to when a large
 *  batch is launched (one wavefront per item, shape baked in). Same buffer/compile/return conventions as
 *  libxsmm_amd_csr_kernel_source (reference counterpart: libxsmm_generator_gemm_kernel's "noarch" C text).
 *  variant: 0 = strided batch of 16-byte aligned items (widest loads); bit 0 = element-wide accesses (index and pointer
 *  batches); bit 1 = consecutive items with one C accumulate in registers (CP2K stacks, batch-reduce). */
LIBXSMM_API int libxsmm_amd_smm_kernel_source(const libxsmm_gemm_descriptor* descriptor, int variant, char* buffer, size_t buffer_size, int compile);

/** The text a grouped launch (libxsmm_amd_gemm_batch_groups) compiles for index batches of these descriptors: the run forms
 *  of every shape, each in a namespace of its own, behind one dispatching kernel. Conventions as above. */
LIBXSMM_API int libxsmm_amd_smm_grouped_kernel_source(const libxsmm_gemm_descriptor* const descriptors[], int ndescriptors, char* buffer, size_t buffer_size, int compile);

/** Run-time specialisation off the caller's path. A batch call never waits for the compiler (hiprtc: 0.3-0.5 s per kernel,
 *  seconds for a grouped kernel of many shapes): the kernel is built on a helper thread while the call -- and the following
 *  ones -- are served by the next best kernel (pre-compiled; the same results bit for bit), and loaded from the code-object
 *  cache on disk when a previous process (or libxsmm_amd_jit_prebuild) has left it there. LIBXSMM_AMD_JIT_ASYNC=0 compiles
 *  in the calling thread instead. libxsmm_amd_jit_wait blocks until the helper thread has nothing left to do (benchmarks:
 *  after the warm-up). libxsmm_amd_jit_prebuild compiles, without needing a device, the code objects batch calls with these
 *  descriptors may ask for (strided, index and pointer batches, shared C in batch order and relaxed; grouped != 0: also the
 *  fused kernel of libxsmm_amd_gemm_batch_groups over all of them) into the cache directory (LIBXSMM_AMD_CACHE, default
 *  jit_cache/ next to the library); returns the number of code objects now present, or -(number of failures). */
LIBXSMM_API void libxsmm_amd_jit_wait(void);
/** Drops the compile jobs that have not started and waits for the one that is running. A process must not reach exit() while
 *  the helper thread is inside the compiler: the compiler's static objects, first constructed during that very job, are
 *  destroyed ahead of any exit handler this library could have registered earlier. libxsmm_finalize() calls it (the
 *  reference's callers end with libxsmm_finalize), the Python binding calls it from Python's own atexit; an exit handler of
 *  the library remains as the last line of defence. */
LIBXSMM_API void libxsmm_amd_jit_drain(void);
LIBXSMM_API int libxsmm_amd_jit_prebuild(const libxsmm_gemm_descriptor* const descriptors[], int ndescriptors, int grouped);

/** Executable form of the sparse text kernels (libxsmm_generator_spgemm_{csr,csc}_kernel): the pattern is compiled into
 *  a kernel with hiprtc, the values of the sparse operand stay a run-time argument (as for the reference's generated C
 *  functions, samples/generator/validation.c). descriptor: lda == 0 marks A sparse, ldb == 0 marks B sparse; is_csr != 0:
 *  row_idx = row pointers, column_idx = column of each entry; is_csr == 0: column_idx = column pointers, row_idx = row of
 *  each entry. fma: 1 = fused multiply-add, 0 = multiply then add (the statement as written), < 0 = default
 *  (LIBXSMM_AMD_SPGEMM_FMA, fused). One launch processes `batch` products that share the sparse operand: the dense
 *  operand and C advance by stride_dense / stride_c elements per item. Operands may live on the host (staged). */
typedef struct libxsmm_amd_spgemm libxsmm_amd_spgemm;
LIBXSMM_API libxsmm_amd_spgemm* libxsmm_amd_spgemm_create(const libxsmm_gemm_descriptor* descriptor, int is_csr,
  const unsigned int* row_idx, const unsigned int* column_idx, int fma);
LIBXSMM_API int libxsmm_amd_spgemm_execute_batch(const libxsmm_amd_spgemm* handle, const void* sparse_values, const void* dense, void* c,
  long long stride_dense, long long stride_c, long long batch);
LIBXSMM_API void libxsmm_amd_spgemm_destroy(const libxsmm_amd_spgemm* handle);
/** The text libxsmm_amd_spgemm_create compiles (buffer/compile/return conventions of libxsmm_amd_csr_kernel_source). */
LIBXSMM_API int libxsmm_amd_spgemm_source(const libxsmm_gemm_descriptor* descriptor, int is_csr, const unsigned int* row_idx,
  const unsigned int* column_idx, int fma, char* buffer, size_t buffer_size, int compile);

/** The MatrixMarket coordinate reader that libxsmm_generator_spgemm uses for its input file (reference:
 *  libxsmm_sparse_csr_reader / libxsmm_sparse_csc_reader, src/generator_spgemm_csr_reader.c:46-170 and
 *  src/generator_spgemm_csc_reader.c:46-170 -- internal to the reference's generator library). '%' lines are comments,
 *  the first other line is "rows cols nnz", then 1-based "row col value" triples grouped by row (is_csr != 0) or by
 *  column (is_csr == 0); rows / columns without an entry are back-filled. *o_ptr receives rows + 1 (columns + 1) offsets,
 *  *o_idx the 0-based column (row) of every entry, *o_values the values as double; the three arrays are the caller's, to be
 *  released with free(). Returns 0, or the reference's error code (LIBXSMM_ERR_CSR_INPUT 90035, _READ_LEN 90036,
 *  _READ_DESC 90037, _READ_ELEMS 90038, _LEN 90039; CSC: 90011 ... 90015; see libxsmm_strerror) with nothing allocated. */
LIBXSMM_API int libxsmm_amd_sparse_reader(const char* path, int is_csr, unsigned int** o_ptr, unsigned int** o_idx, double** o_values,
  unsigned int* o_row_count, unsigned int* o_column_count, unsigned int* o_element_count);

/** SOA width v of the libxsmm_create_*_soa kernels for a precision (8 for fp64, 16 for fp32; 0 if unsupported). */
LIBXSMM_API int libxsmm_amd_soa_width(libxsmm_gemm_precision precision);
/** Batch form of a kernel made by libxsmm_create_{xcsr,xcsc,rm_ac,rm_bc}_soa: `batch` products that share the operator
 *  (the sparse values / the plain dense matrix). a, b, c as in the kernel call; the SOA input and C advance by
 *  stride_dense / stride_c elements per item. */
LIBXSMM_API int libxsmm_amd_kernel_execute_batch(const void* kernel, const void* a, const void* b, void* c,
  long long stride_dense, long long stride_c, long long batch);

#endif /* LIBXSMM_AMD_H */
