/*
 * oracle/xsmm_oracle.h -- CPU restatement of the reference's SMM / sparse hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under libxsmm-1_amd/ may include, link or call
 * this. Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Parity status: pinned by the reference's own fixtures (MatrixMarket operators under
 * samples/generator and samples/pyfr/mats: reader known-answers and the sparse==dense
 * operator pairs) and by the gold loops of the reference's own self-checking samples
 * which this file restates. No outputs of a reference *binary* are available: the
 * reference cannot be compiled under this round's rules (every source includes the
 * build-generated libxsmm_config.h / libxsmm.h), see DESIGN.md "Oracle".
 *
 * All citations are relative to /root/reference.
 */
#ifndef XSMM_ORACLE_H
#define XSMM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flag values: include/libxsmm_typedefs.h:180-213 */
#define XO_FLAG_TRANS_A 1
#define XO_FLAG_TRANS_B 2
#define XO_FLAG_BETA_0 16

/* arithmetic flavour: 0 = separate multiply and add (noarch / SSE path),
 * 1 = fused multiply-add (AVX2 / AVX-512 JIT path, path of record). */
#define XO_ARITH_MULADD 0
#define XO_ARITH_FMA 1

/* ---- dense SMM (generator_gemm_noarch.c:59-84; AVX2: generator_gemm_sse3_avx_avx2_avx512.c:215-369) ---- */
void xo_dsmm(int arith, int flags, int m, int n, int k, int lda, int ldb, int ldc,
             const double* a, const double* b, double* c);
void xo_ssmm(int arith, int flags, int m, int n, int k, int lda, int ldb, int ldc,
             const float* a, const float* b, float* c);

/* ---- batch driver (libxsmm_gemm.c:1315-1608): sequential walk over the three addressing modes.
 * typesize 8 -> double, 4 -> float. index_stride != 0: stride_* are index arrays walked with byte step
 * index_stride; index_stride == 0: a,b,c are arrays of pointers and *stride_* is the byte distance between
 * consecutive pointers. NULL stride => operand shared by all items. Returns 0 on success. */
int xo_gemm_batch(int arith, int typesize, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                  const void* a, const void* b, void* c, int index_base, int index_stride,
                  const int* stride_a, const int* stride_b, const int* stride_c, int batchsize);

/* constant-stride batch (what samples/smm/specialized.cpp:172-190 does with direct kernel calls);
 * strides in elements; nthreads>1 uses OpenMP static schedule over items (items must not share C). */
void xo_gemm_batch_strided(int arith, int typesize, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                           const void* a, const void* b, void* c, long long sa, long long sb, long long sc,
                           long long batchsize, int nthreads);

/* batch-reduce kernel (include/libxsmm_typedefs.h:538-541; generator hooks
 * generator_gemm_sse3_avx_avx2_avx512.c:97-108,217-262): C (+)= sum_i A[i]*B[i], C loaded once. */
void xo_dsmm_reduce(int arith, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                    const double** a, const double** b, double* c, unsigned long long count);
void xo_ssmm_reduce(int arith, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                    const float** a, const float** b, float* c, unsigned long long count);

/* ---- sparse "text" kernels (values are run-time operands, pattern is fixed) ---- */
/* generator_spgemm_csr_asparse.c:46-151: row-major, C[m*ldc+n] += A[p]*B[col[p]*ldb+n] */
void xo_dcsr_asparse(int arith, int flags, int m, int n, int k, int ldb, int ldc,
                     const unsigned* rowptr, const unsigned* colidx, const double* a_vals, const double* b, double* c);
void xo_scsr_asparse(int arith, int flags, int m, int n, int k, int ldb, int ldc,
                     const unsigned* rowptr, const unsigned* colidx, const float* a_vals, const float* b, float* c);
/* generator_spgemm_csc_bsparse.c:85-189: col-major, C[n*ldc+m] += A[row[p]*lda+m]*B[p] */
void xo_dcsc_bsparse(int arith, int flags, int m, int n, int k, int lda, int ldc,
                     const unsigned* colptr, const unsigned* rowidx, const double* a, const double* b_vals, double* c);
void xo_scsc_bsparse(int arith, int flags, int m, int n, int k, int lda, int ldc,
                     const unsigned* colptr, const unsigned* rowidx, const float* a, const float* b_vals, float* c);
/* generator_spgemm_csc_asparse.c:223-349 (C fallback :332): col-major, C[n*ldc+row[p]] += A[p]*B[n*ldb+k] */
void xo_dcsc_asparse(int arith, int flags, int m, int n, int k, int ldb, int ldc,
                     const unsigned* colptr, const unsigned* rowidx, const double* a_vals, const double* b, double* c);
void xo_scsc_asparse(int arith, int flags, int m, int n, int k, int ldb, int ldc,
                     const unsigned* colptr, const unsigned* rowidx, const float* a_vals, const float* b, float* c);

/* generator_spgemm_csr_asparse_reg.c:80-313 -- values baked in; returns -1 (kernel creation fails)
 * when more than 31 unique values (:146) or n != vlen (:187); rows without nnz are not touched (:229,287).
 * vlen = 8 (f64) / 16 (f32). fp32: values are de-duplicated as doubles and narrowed (libxsmm_main.c:2557-2563). */
int xo_csr_reg_unique(const double* values, unsigned nnz); /* number of unique values, scan order of :125-143 */
int xo_dcsr_reg(int flags, int m, int n, int k, int ldb, int ldc,
                const unsigned* rowptr, const unsigned* colidx, const double* values, const double* b, double* c);
int xo_scsr_reg(int flags, int m, int n, int k, int ldb, int ldc,
                const unsigned* rowptr, const unsigned* colidx, const float* values, const float* b, float* c);

/* ---- MatrixMarket readers (generator_spgemm_csr_reader.c:46-170, generator_spgemm_csc_reader.c:85-215) ----
 * Return 0 on success; outputs malloc'ed (free with xo_free). */
int xo_csr_reader(const char* path, unsigned** rowptr, unsigned** colidx, double** values,
                  unsigned* rows, unsigned* cols, unsigned* nnz);
int xo_csc_reader(const char* path, unsigned** rowidx, unsigned** colptr, double** values,
                  unsigned* rows, unsigned* cols, unsigned* nnz);
/* dense "array" MatrixMarket (column-major listing) as stored in samples/pyfr/mats (-de.mtx files); returns row-major */
int xo_dense_mtx_reader(const char* path, double** rowmajor, unsigned* rows, unsigned* cols);
void xo_free(void* p);

/* ---- fsspmdm (libxsmm_fsspmdm.c:48-329) ---- */
typedef struct xo_fsspmdm {
  int M, N, K, ldb, ldc, N_chunksize, typesize, flags;
  int sparse;            /* 1: csr_reg kernel, 0: dense fallback */
  unsigned nnz;
  unsigned *rowptr, *colidx;
  double* values;        /* CSR values (as double) */
  void* a_dense;         /* tight copy (ld = K) for the dense fallback */
} xo_fsspmdm;
/* have_avx512 != 0 mimics an AVX-512 host (csr_reg attempted); 0 mimics LIBXSMM_TARGET=hsw (always dense fallback) */
xo_fsspmdm* xo_fsspmdm_create(int typesize, int M, int N, int K, int lda, int ldb, int ldc,
                              double alpha, double beta, const void* a_dense, int have_avx512);
void xo_fsspmdm_execute(const xo_fsspmdm* h, const void* B, void* C);
void xo_fsspmdm_destroy(xo_fsspmdm* h);

/* ---- spmdm (libxsmm_spmdm.c:540-627 + src/template/libxsmm_spmdm_*_fp32_thread.tpl.c) ---- */
typedef struct xo_spmdm_handle { int m, n, k, bm, bn, bk, mb, nb, kb; } xo_spmdm_handle;
typedef struct xo_csr_slice { uint16_t* rowidx; uint16_t* colidx; float* values; } xo_csr_slice;
/* bn_isa: 96 (AVX-512), 48 (AVX2), 6 (scalar) -- libxsmm_spmdm.c:555-587 */
void xo_spmdm_init(int M, int N, int K, int max_threads, int bn_isa, xo_spmdm_handle* h);
xo_csr_slice* xo_spmdm_alloc_slices(const xo_spmdm_handle* h);
void xo_spmdm_free_slices(const xo_spmdm_handle* h, xo_csr_slice* s);
void xo_spmdm_create_slice(const xo_spmdm_handle* h, char transa, const float* a, xo_csr_slice* slices, int block_id);
void xo_spmdm_compute(int arith, const xo_spmdm_handle* h, char transa, char transb, const float* alpha,
                      const xo_csr_slice* slices, const float* b, char transc, const float* beta, float* c, int block_id);
/* whole-problem convenience: init geometry (1 thread, bn_isa), create all slices, compute all blocks */
void xo_spmdm_exec(int arith, int M, int N, int K, int bn_isa, char transa, char transb, char transc,
                   float beta, const float* a, const float* b, float* c);
/* bfloat16 inputs (upper halves of floats), float C; beta_bits is the raw 16-bit pattern the caller stores behind `beta` */
void xo_spmdm_exec_bf16(int arith, int M, int N, int K, int bn_isa, char transa, char transb, char transc,
                        unsigned short beta_bits, const unsigned short* a, const unsigned short* b, float* c);
/* batch of independent problems laid out back-to-back (A: M*K, B: K*N, C: M*N per item) */
void xo_spmdm_exec_batch(int arith, int M, int N, int K, int bn_isa, char transa, char transb, char transc,
                         float beta, const float* a, const float* b, float* c, long long batch, int nthreads);

/* ---- SOA kernels (libxsmm_create_xcsr_soa/xcsc_soa/rm_ac_soa/rm_bc_soa, src/libxsmm_main.c:2423-2520): [row][col][v] ---- */
#define XO_DECLARE_SOA(SUFFIX, T) \
void xo_soa_csr_asparse_##SUFFIX(int flags, int m, int n, int k, int ldb, int ldc, int v, \
  const unsigned* rowptr, const unsigned* colidx, const T* a_vals, const T* b, T* c); \
void xo_soa_bsparse_##SUFFIX(int flags, int csr, int m, int n, int k, int lda, int ldc, int v, \
  const unsigned* ptr, const unsigned* idx, const T* a, const T* b_vals, T* c); \
void xo_soa_rm_ac_##SUFFIX(int flags, int m, int n, int k, int lda, int ldb, int ldc, int v, const T* a, const T* b, T* c); \
void xo_soa_rm_bc_##SUFFIX(int flags, int m, int n, int k, int lda, int ldb, int ldc, int v, const T* a, const T* b, T* c);
XO_DECLARE_SOA(f64, double)
XO_DECLARE_SOA(f32, float)

/* ---- blocked_gemm (libxsmm_blocked_gemm.c:47-568, template/libxsmm_blocked_gemm*.tpl.c) ---- */
typedef struct xo_bgemm { int typesize, m, n, k, bm, bn, bk, mb, nb, kb, b_m1, b_n1, b_k1, b_k2, order, flags; } xo_bgemm;
int xo_bgemm_init(xo_bgemm* h, int typesize, int m, int n, int k, int bm, int bn, int bk,
                  int b_m1, int b_n1, int b_k1, int b_k2, double alpha, double beta, int order);
void xo_bgemm_copyin_a(const xo_bgemm* h, const void* src, int ld, void* dst);
void xo_bgemm_copyin_b(const xo_bgemm* h, const void* src, int ld, void* dst);
void xo_bgemm_copyin_c(const xo_bgemm* h, const void* src, int ld, void* dst);
void xo_bgemm_copyout_c(const xo_bgemm* h, const void* src, int ld, void* dst);
void xo_bgemm_convert_b_to_a(const xo_bgemm* h, const void* src, void* dst);
void xo_bgemm_transpose_b(const xo_bgemm* h, const void* src, void* dst);
void xo_bgemm_order(int order, int w_i, int nw_i, int nw_j, int nw_k, int* i2, int* j2, int* k2);
void xo_bgemm_st(int arith, const xo_bgemm* h, const void* a, const void* b, void* c); /* nthreads = 1 */

/* ---- low-precision dense kernels (libxsmm_wimmdispatch/wsmmdispatch/bsmmdispatch/bmmdispatch, src/libxsmm_main.c:2198-2259) ----
 * The arithmetic is the gold loop the reference's own harness checks these kernels with (samples/xgemm/kernel.c: i16->i32
 * :915-927, i16->f32 with a scaling factor :1007-1021, bf16->f32 :1104-1123, bf16->bf16 :1207-1229): A is stored in pairs
 * of k ("VNNI", a[(k/2)*lda*2 + m*2 + k%2]), B column-major (b[n*ldb + k]), C column-major; k is even; per C element the
 * terms are added in ascending k. bf16 values are the upper halves of floats; a bf16 result is the upper half of the
 * float sum (truncation, as the harness does). kind: 0 = i16->i32, 1 = i16->f32 (times scf), 2 = bf16->f32, 3 = bf16->bf16. */
int xo_gemm_lowp(int kind, int beta0, int m, int n, int k, int lda, int ldb, int ldc,
                 const unsigned short* a, const unsigned short* b, void* c, float scf);

/* ---- input generators used by the reference's samples ---- */
/* LIBXSMM_MATINIT, seed != 0 branch (include/libxsmm_frontend.h:414-431) */
void xo_matinit_f64(int seed, double* dst, int nrows, int ncols, int ld, double scale);
void xo_matinit_f32(int seed, float* dst, int nrows, int ncols, int ld, double scale);
/* libxsmm_rng_set_seed / libxsmm_rng_f64 == srand48/drand48 on Linux (src/libxsmm_rng.c:131,256):
 * the POSIX 48-bit LCG, carried here explicitly so it is reproducible anywhere. */
void xo_rng_seed(unsigned seed);
double xo_rng_f64(void);

/* libxsmm_matdiff subset (src/template/libxsmm_matdiff.tpl.c): max abs diff, and Frobenius-relative */
void xo_matdiff(int typesize, int m, int n, const void* ref, const void* tst, int ldref, int ldtst,
                double* linf_abs, double* normf_rel);

#ifdef __cplusplus
}
#endif
#endif
