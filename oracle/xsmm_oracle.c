/*
 * oracle/xsmm_oracle.c -- CPU restatement of the reference's SMM / sparse hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see xsmm_oracle.h). Compile with -ffp-contract=off so that the
 * MULADD flavour really is two roundings; the FMA flavour calls fma()/fmaf() explicitly.
 * Citations are relative to /root/reference.
 */
#include "xsmm_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(_OPENMP)
# include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * dense SMM
 * generator_gemm_noarch.c:59-84 emits:  for n: [beta==0: C[n*ldc+m]=0 for all m]
 *                                       for k: for m: C[n*ldc+m] += A[k*lda+m] * B[n*ldb+k]
 * i.e. every C element is a k-ordered chain that starts from C_in (beta=1) or 0 (beta=0).
 * The AVX2/AVX-512 JIT keeps the same chain per element but with VFMADD231
 * (generator_gemm_common.c:168,188; load_C generator_gemm_sse3_avx_avx2_avx512.c:215, k-loop :270-330,
 * store_C :369). TRANS_B reads B[k*ldb+n] (generator_gemm.c:211-234 for the ld checks).
 * ------------------------------------------------------------------------------------------------ */
#define XO_DEFINE_SMM(NAME, T, FMAF)                                                               \
void NAME(int arith, int flags, int m, int n, int k, int lda, int ldb, int ldc,                    \
          const T* a, const T* b, T* c)                                                            \
{                                                                                                  \
  int im, in, ik;                                                                                  \
  const int tb = (0 != (flags & XO_FLAG_TRANS_B));                                                 \
  for (in = 0; in < n; ++in) {                                                                     \
    if (0 != (flags & XO_FLAG_BETA_0)) {                                                           \
      for (im = 0; im < m; ++im) c[(size_t)in * ldc + im] = (T)0;                                  \
    }                                                                                              \
    for (ik = 0; ik < k; ++ik) {                                                                   \
      const T bv = tb ? b[(size_t)ik * ldb + in] : b[(size_t)in * ldb + ik];                       \
      if (XO_ARITH_FMA == arith) {                                                                 \
        for (im = 0; im < m; ++im)                                                                 \
          c[(size_t)in * ldc + im] = FMAF(a[(size_t)ik * lda + im], bv, c[(size_t)in * ldc + im]); \
      }                                                                                            \
      else {                                                                                       \
        for (im = 0; im < m; ++im) {                                                               \
          const T p = a[(size_t)ik * lda + im] * bv;                                               \
          c[(size_t)in * ldc + im] = c[(size_t)in * ldc + im] + p;                                 \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
}
XO_DEFINE_SMM(xo_dsmm, double, fma)
XO_DEFINE_SMM(xo_ssmm, float, fmaf)

static void xo_smm(int arith, int typesize, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                   const void* a, const void* b, void* c)
{
  if (8 == typesize) xo_dsmm(arith, flags, m, n, k, lda, ldb, ldc, (const double*)a, (const double*)b, (double*)c);
  else xo_ssmm(arith, flags, m, n, k, lda, ldb, ldc, (const float*)a, (const float*)b, (float*)c);
}

/* batch-reduce: C is loaded once (or zeroed), then for every (A_i,B_i) the k-chain continues in the
 * same accumulator (generator_gemm_sse3_avx_avx2_avx512.c:97-108 loops the microkernel over the batch
 * between load_C :215 and store_C :369). */
#define XO_DEFINE_REDUCE(NAME, T, SMM)                                                             \
void NAME(int arith, int flags, int m, int n, int k, int lda, int ldb, int ldc,                    \
          const T** a, const T** b, T* c, unsigned long long count)                                \
{                                                                                                  \
  unsigned long long i;                                                                            \
  int f = flags;                                                                                   \
  if (0 == count && 0 != (flags & XO_FLAG_BETA_0)) {                                               \
    int im, in;                                                                                    \
    for (in = 0; in < n; ++in) for (im = 0; im < m; ++im) c[(size_t)in * ldc + im] = (T)0;         \
  }                                                                                                \
  for (i = 0; i < count; ++i) {                                                                    \
    SMM(arith, f, m, n, k, lda, ldb, ldc, a[i], b[i], c);                                          \
    f &= ~XO_FLAG_BETA_0; /* only the first product may overwrite */                               \
  }                                                                                                \
}
XO_DEFINE_REDUCE(xo_dsmm_reduce, double, xo_dsmm)
XO_DEFINE_REDUCE(xo_ssmm_reduce, float, xo_ssmm)

/* libxsmm_mmbatch_kernel (libxsmm_gemm.c:1315-1608), single task (tid=0, ntasks=1): items are
 * processed in order, so items that share a C accumulate in batch order. */
int xo_gemm_batch(int arith, int typesize, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                  const void* a, const void* b, void* c, int index_base, int index_stride,
                  const int* stride_a, const int* stride_b, const int* stride_c, int batchsize)
{
  const long long size = (batchsize < 0 ? -(long long)batchsize : batchsize);
  long long i;
  if (NULL == a || NULL == b || NULL == c) return -1;
  if (0 != index_stride) { /* stride arrays hold element indexes (:1333-1364) */
    for (i = 0; i < size; ++i) {
      const size_t off = (size_t)i * (size_t)index_stride; /* LIBXSMM_ACCESS: byte offset */
      const long long ia = (NULL != stride_a ? (*(const int*)((const char*)stride_a + off) - index_base) : 0);
      const long long ib = (NULL != stride_b ? (*(const int*)((const char*)stride_b + off) - index_base) : 0);
      const long long ic = (NULL != stride_c ? (*(const int*)((const char*)stride_c + off) - index_base) : 0);
      xo_smm(arith, typesize, flags, m, n, k, lda, ldb, ldc,
        (const char*)a + ia * typesize, (const char*)b + ib * typesize, (char*)c + ic * typesize);
    }
  }
  else { /* arrays of pointers; *stride is the byte distance between pointers (:1426-1461) */
    const long long da = (NULL != stride_a ? (*stride_a - index_base * (long long)sizeof(void*)) : 0);
    const long long db = (NULL != stride_b ? (*stride_b - index_base * (long long)sizeof(void*)) : 0);
    const long long dc = (NULL != stride_c ? (*stride_c - index_base * (long long)sizeof(void*)) : 0);
    for (i = 0; i < size; ++i) {
      const void* ai = *(const void* const*)((const char*)a + da * i);
      const void* bi = *(const void* const*)((const char*)b + db * i);
      void* ci = *(void* const*)((const char*)c + dc * i);
      xo_smm(arith, typesize, flags, m, n, k, lda, ldb, ldc, ai, bi, ci);
    }
  }
  return 0;
}

void xo_gemm_batch_strided(int arith, int typesize, int flags, int m, int n, int k, int lda, int ldb, int ldc,
                           const void* a, const void* b, void* c, long long sa, long long sb, long long sc,
                           long long batchsize, int nthreads)
{
  long long i;
  (void)nthreads;
#if defined(_OPENMP)
# pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (i = 0; i < batchsize; ++i) {
    xo_smm(arith, typesize, flags, m, n, k, lda, ldb, ldc,
      (const char*)a + i * sa * typesize, (const char*)b + i * sb * typesize, (char*)c + i * sc * typesize);
  }
}

/* ------------------------------------------------------------------------------------------------
 * sparse text kernels
 * ------------------------------------------------------------------------------------------------ */
#define XO_ACC(T, FMAF, DST, X, Y) do { \
  if (XO_ARITH_FMA == arith) (DST) = FMAF((X), (Y), (DST)); \
  else { const T xo_p_ = (X) * (Y); (DST) = (DST) + xo_p_; } } while (0)

/* generator_spgemm_csr_asparse.c: beta==0 zeroes ldc (not n) entries of each of the m rows (:79);
 * loop nest emitted is  for n: { for m: for z in row m: if col < k: C[m*ldc+n] += A[p]*B[col*ldb+n] } (:133-141) */
#define XO_DEFINE_CSR_ASPARSE(NAME, T, FMAF)                                                       \
void NAME(int arith, int flags, int m, int n, int k, int ldb, int ldc,                             \
          const unsigned* rowptr, const unsigned* colidx, const T* a_vals, const T* b, T* c)       \
{                                                                                                  \
  int im, in; unsigned p;                                                                          \
  if (0 != (flags & XO_FLAG_BETA_0)) {                                                             \
    for (im = 0; im < m; ++im) for (in = 0; in < ldc; ++in) c[(size_t)im * ldc + in] = (T)0;       \
  }                                                                                                \
  for (in = 0; in < n; ++in) {                                                                     \
    for (im = 0; im < m; ++im) {                                                                   \
      for (p = rowptr[im]; p < rowptr[im + 1]; ++p) {                                              \
        if (colidx[p] < (unsigned)k) {                                                             \
          XO_ACC(T, FMAF, c[(size_t)im * ldc + in], a_vals[p], b[(size_t)colidx[p] * ldb + in]);   \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
}
XO_DEFINE_CSR_ASPARSE(xo_dcsr_asparse, double, fma)
XO_DEFINE_CSR_ASPARSE(xo_scsr_asparse, float, fmaf)

/* generator_spgemm_csc_bsparse.c: beta==0 zeroes m entries of each of the n columns (:112-127);
 * for m: { for n: for z in column n of B: if row < k: C[n*ldc+m] += A[row*lda+m]*B[p] } (:166-181) */
#define XO_DEFINE_CSC_BSPARSE(NAME, T, FMAF)                                                       \
void NAME(int arith, int flags, int m, int n, int k, int lda, int ldc,                             \
          const unsigned* colptr, const unsigned* rowidx, const T* a, const T* b_vals, T* c)       \
{                                                                                                  \
  int im, in; unsigned p;                                                                          \
  if (0 != (flags & XO_FLAG_BETA_0)) {                                                             \
    for (in = 0; in < n; ++in) for (im = 0; im < m; ++im) c[(size_t)in * ldc + im] = (T)0;         \
  }                                                                                                \
  for (im = 0; im < m; ++im) {                                                                     \
    for (in = 0; in < n; ++in) {                                                                   \
      for (p = colptr[in]; p < colptr[in + 1]; ++p) {                                              \
        if (rowidx[p] < (unsigned)k) {                                                             \
          XO_ACC(T, FMAF, c[(size_t)in * ldc + im], a[(size_t)rowidx[p] * lda + im], b_vals[p]);   \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
}
XO_DEFINE_CSC_BSPARSE(xo_dcsc_bsparse, double, fma)
XO_DEFINE_CSC_BSPARSE(xo_scsc_bsparse, float, fmaf)

/* generator_spgemm_csc_asparse.c C fallback (:325-336): for n: [beta==0: zero m entries (:243-254)]
 * for k: for z in column k of A: if row < m: C[n*ldc+row] += A[p]*B[n*ldb+k] */
#define XO_DEFINE_CSC_ASPARSE(NAME, T, FMAF)                                                       \
void NAME(int arith, int flags, int m, int n, int k, int ldb, int ldc,                             \
          const unsigned* colptr, const unsigned* rowidx, const T* a_vals, const T* b, T* c)       \
{                                                                                                  \
  int im, in, ik; unsigned p;                                                                      \
  for (in = 0; in < n; ++in) {                                                                     \
    if (0 != (flags & XO_FLAG_BETA_0)) {                                                           \
      for (im = 0; im < m; ++im) c[(size_t)in * ldc + im] = (T)0;                                  \
    }                                                                                              \
    for (ik = 0; ik < k; ++ik) {                                                                   \
      for (p = colptr[ik]; p < colptr[ik + 1]; ++p) {                                              \
        if (rowidx[p] < (unsigned)m) {                                                             \
          XO_ACC(T, FMAF, c[(size_t)in * ldc + rowidx[p]], a_vals[p], b[(size_t)in * ldb + ik]);   \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
}
XO_DEFINE_CSC_ASPARSE(xo_dcsc_asparse, double, fma)
XO_DEFINE_CSC_ASPARSE(xo_scsc_asparse, float, fmaf)

/* unique-value scan (generator_spgemm_csr_asparse_reg.c:125-143). Equality is written as !(u<v) && !(u>v),
 * so a NaN "equals" every slot (both comparisons are false); kept as written. */
int xo_csr_reg_unique(const double* values, unsigned nnz)
{
  unsigned i, z, nunique;
  double* u;
  if (0 == nnz) return 0;
  u = (double*)malloc(sizeof(double) * nnz);
  if (NULL == u) return -1;
  nunique = 1; u[0] = values[0];
  for (i = 1; i < nnz; ++i) {
    int hit = 0;
    for (z = 0; z < nunique; ++z) {
      if (!(u[z] < values[i]) && !(u[z] > values[i])) hit = 1;
    }
    if (0 == hit) u[nunique++] = values[i];
  }
  free(u);
  return (int)nunique;
}

/* csr_asparse_reg kernel body (:227-300): per row with nnz>0: acc = beta? C : 0; acc = fma(val, B[col*ldb+n], acc)
 * for every nnz of the row in order (no k filter here); store. Rows without nnz are skipped entirely. */
#define XO_DEFINE_CSR_REG(NAME, T, FMAF, VLEN)                                                     \
int NAME(int flags, int m, int n, int k, int ldb, int ldc,                                         \
         const unsigned* rowptr, const unsigned* colidx, const T* values, const T* b, T* c)        \
{                                                                                                  \
  int im, in; unsigned p;                                                                          \
  const unsigned nnz = rowptr[m];                                                                  \
  double* dv;                                                                                      \
  int nunique;                                                                                     \
  (void)k;                                                                                         \
  if (n != (VLEN)) return -1; /* :187 */                                                           \
  if (0 == nnz) return -1; /* reference dereferences values[0]; fsspmdm never gets here with nnz==0 */ \
  dv = (double*)malloc(sizeof(double) * nnz);                                                      \
  if (NULL == dv) return -1;                                                                       \
  for (p = 0; p < nnz; ++p) dv[p] = (double)values[p]; /* libxsmm_main.c:2557-2563 widens fp32 */  \
  nunique = xo_csr_reg_unique(dv, nnz);                                                            \
  free(dv);                                                                                        \
  if (nunique < 0 || nunique > 31) return -1; /* :146 */                                           \
  for (im = 0; im < m; ++im) {                                                                     \
    if (rowptr[im + 1] > rowptr[im]) {                                                             \
      for (in = 0; in < n; ++in) {                                                                 \
        T acc = (0 != (flags & XO_FLAG_BETA_0)) ? (T)0 : c[(size_t)im * ldc + in];                 \
        for (p = rowptr[im]; p < rowptr[im + 1]; ++p) {                                            \
          acc = FMAF(values[p], b[(size_t)colidx[p] * ldb + in], acc);                             \
        }                                                                                          \
        c[(size_t)im * ldc + in] = acc;                                                            \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
  return 0;                                                                                        \
}
XO_DEFINE_CSR_REG(xo_dcsr_reg, double, fma, 8)
XO_DEFINE_CSR_REG(xo_scsr_reg, float, fmaf, 16)

/* ------------------------------------------------------------------------------------------------
 * MatrixMarket readers
 * ------------------------------------------------------------------------------------------------ */
void xo_free(void* p) { free(p); }

/* shared body: `major` entries indexed by the (row for CSR | column for CSC) coordinate.
 * generator_spgemm_csr_reader.c:46-170 / generator_spgemm_csc_reader.c:85-215: '%' lines skipped, first
 * other line "rows cols nnz" (all non-zero), then 1-based "row col value"; ptr[major+1] = running count
 * (so entries must arrive grouped by major index); untouched majors are back-filled with ptr[i+1]=ptr[i]. */
static int xo_coo_reader(const char* path, int csr, unsigned** ptr, unsigned** idx, double** values,
                         unsigned* rows, unsigned* cols, unsigned* nnz)
{
  FILE* f = fopen(path, "r");
  char line[513];
  unsigned header = 0, i = 0, nmajor = 0;
  unsigned* seen = NULL;
  *ptr = NULL; *idx = NULL; *values = NULL;
  if (NULL == f) return -1;
  while (NULL != fgets(line, 512, f)) {
    if (512 == strlen(line)) { fclose(f); goto fail; }
    if ('%' == line[0]) continue;
    if (0 == header) {
      if (3 != sscanf(line, "%u %u %u", rows, cols, nnz) || 0 == *rows || 0 == *cols || 0 == *nnz) { fclose(f); goto fail; }
      nmajor = csr ? *rows : *cols;
      *idx = (unsigned*)calloc(*nnz, sizeof(unsigned));
      *ptr = (unsigned*)calloc((size_t)nmajor + 1, sizeof(unsigned));
      *values = (double*)calloc(*nnz, sizeof(double));
      seen = (unsigned*)calloc(nmajor, sizeof(unsigned));
      if (NULL == *idx || NULL == *ptr || NULL == *values || NULL == seen) { fclose(f); goto fail; }
      for (i = 0; i <= nmajor; ++i) (*ptr)[i] = *nnz;
      (*ptr)[0] = 0; i = 0; header = 1;
    }
    else {
      unsigned r = 0, c = 0; double v = 0;
      if (3 != sscanf(line, "%u %u %lf", &r, &c, &v) || 0 == r || 0 == c || i >= *nnz) { fclose(f); goto fail; }
      --r; --c;
      if ((csr ? r : c) >= nmajor) { fclose(f); goto fail; }
      (*idx)[i] = csr ? c : r;
      (*values)[i] = v;
      ++i;
      seen[csr ? r : c] = 1;
      (*ptr)[(csr ? r : c) + 1] = i;
    }
  }
  fclose(f);
  if (0 == header || i != *nnz) goto fail;
  for (i = 0; i < nmajor; ++i) if (0 == seen[i]) (*ptr)[i + 1] = (*ptr)[i];
  free(seen);
  return 0;
fail:
  free(*ptr); free(*idx); free(*values); free(seen);
  *ptr = NULL; *idx = NULL; *values = NULL;
  return -1;
}

int xo_csr_reader(const char* path, unsigned** rowptr, unsigned** colidx, double** values,
                  unsigned* rows, unsigned* cols, unsigned* nnz)
{
  return xo_coo_reader(path, 1, rowptr, colidx, values, rows, cols, nnz);
}

int xo_csc_reader(const char* path, unsigned** rowidx, unsigned** colptr, double** values,
                  unsigned* rows, unsigned* cols, unsigned* nnz)
{
  return xo_coo_reader(path, 0, colptr, rowidx, values, rows, cols, nnz);
}

/* "%%MatrixMarket matrix array real general": header "rows cols", then rows*cols values column by column
 * (the -de.mtx files under samples/pyfr/mats). Not a reference reader (the PyFR sample reads only the -sp files,
 * samples/pyfr/pyfr_driver_asp_reg.c:95-140); provided to exploit the sp/de fixture pairs. */
int xo_dense_mtx_reader(const char* path, double** rowmajor, unsigned* rows, unsigned* cols)
{
  FILE* f = fopen(path, "r");
  char line[513];
  unsigned header = 0; size_t i = 0, total = 0;
  *rowmajor = NULL;
  if (NULL == f) return -1;
  while (NULL != fgets(line, 512, f)) {
    if ('%' == line[0]) continue;
    if (0 == header) {
      if (2 != sscanf(line, "%u %u", rows, cols) || 0 == *rows || 0 == *cols) { fclose(f); return -1; }
      total = (size_t)*rows * *cols;
      *rowmajor = (double*)calloc(total, sizeof(double));
      if (NULL == *rowmajor) { fclose(f); return -1; }
      header = 1;
    }
    else {
      double v;
      if (1 != sscanf(line, "%lf", &v) || i >= total) { fclose(f); free(*rowmajor); *rowmajor = NULL; return -1; }
      { const size_t c = i / *rows, r = i % *rows; (*rowmajor)[r * *cols + c] = v; }
      ++i;
    }
  }
  fclose(f);
  if (0 == header || i != total) { free(*rowmajor); *rowmajor = NULL; return -1; }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * fsspmdm (libxsmm_fsspmdm.c)
 * ------------------------------------------------------------------------------------------------ */
xo_fsspmdm* xo_fsspmdm_create(int typesize, int M, int N, int K, int lda, int ldb, int ldc,
                              double alpha, double beta, const void* a_dense, int have_avx512)
{
  xo_fsspmdm* h;
  int i, j; unsigned n = 0;
  /* asserts of :65-71 */
  if (0 != (N % 16) || N < 16 || 1.0 != alpha || (1.0 != beta && 0.0 != beta) || K > lda || N > ldc || N > ldb) return NULL;
  h = (xo_fsspmdm*)calloc(1, sizeof(*h));
  if (NULL == h) return NULL;
  h->M = M; h->N = N; h->K = K; h->ldb = ldb; h->ldc = ldc; h->typesize = typesize;
  h->flags = (0.0 == beta ? XO_FLAG_BETA_0 : 0);
#define XO_AT(I, J) (8 == typesize ? ((const double*)a_dense)[(size_t)(I) * lda + (J)] : (double)((const float*)a_dense)[(size_t)(I) * lda + (J)])
  for (i = 0; i < M; ++i) for (j = 0; j < K; ++j) if (XO_AT(i, j) != 0.0) ++n; /* LIBXSMM_NEQ :88-94 */
  h->nnz = n;
  h->rowptr = (unsigned*)calloc((size_t)M + 1, sizeof(unsigned));
  h->colidx = (unsigned*)calloc(n ? n : 1, sizeof(unsigned));
  h->values = (double*)calloc(n ? n : 1, sizeof(double));
  n = 0;
  for (i = 0; i < M; ++i) { /* :102-113 */
    h->rowptr[i] = n;
    for (j = 0; j < K; ++j) if (XO_AT(i, j) != 0.0) { h->values[n] = XO_AT(i, j); h->colidx[n] = (unsigned)j; ++n; }
  }
  h->rowptr[M] = n;
  /* sparse attempt (:116-126): N_chunksize 8 (f64) / 16 (f32); succeeds only on AVX-512 with <= 31 unique values */
  h->sparse = 0;
  if (0 < n && 0 != have_avx512) {
    const int nu = xo_csr_reg_unique(h->values, n);
    if (0 < nu && nu <= 31) { h->sparse = 1; h->N_chunksize = (8 == typesize ? 8 : 16); }
  }
  if (0 == h->sparse) { /* dense fallback (:134-142): col-major SMM on the transposed problem, A copied tight */
    h->N_chunksize = 16;
    h->a_dense = malloc((size_t)M * K * typesize);
    for (i = 0; i < M; ++i) for (j = 0; j < K; ++j) {
      if (8 == typesize) ((double*)h->a_dense)[(size_t)i * K + j] = ((const double*)a_dense)[(size_t)i * lda + j];
      else ((float*)h->a_dense)[(size_t)i * K + j] = ((const float*)a_dense)[(size_t)i * lda + j];
    }
  }
#undef XO_AT
  return h;
}

void xo_fsspmdm_execute(const xo_fsspmdm* h, const void* B, void* C)
{
  int i;
  for (i = 0; i < h->N; i += h->N_chunksize) { /* :260-291 */
    if (h->sparse) {
      if (8 == h->typesize) {
        xo_dcsr_reg(h->flags, h->M, h->N_chunksize, h->K, h->ldb, h->ldc, h->rowptr, h->colidx, h->values,
          (const double*)B + i, (double*)C + i);
      }
      else { /* values narrowed to fp32 (generator_spgemm_csr_asparse_reg.c:210-212) */
        unsigned p; float* fv = (float*)malloc(sizeof(float) * (h->nnz ? h->nnz : 1));
        for (p = 0; p < h->nnz; ++p) fv[p] = (float)h->values[p];
        xo_scsr_reg(h->flags, h->M, h->N_chunksize, h->K, h->ldb, h->ldc, h->rowptr, h->colidx, fv,
          (const float*)B + i, (float*)C + i);
        free(fv);
      }
    }
    else { /* kernel(B+i, a_dense, C+i) with m=N_chunk, n=M, k=K, lda=ldb, ldb=K, ldc=ldc  (:137) -- JIT => FMA */
      xo_smm(XO_ARITH_FMA, h->typesize, h->flags, h->N_chunksize, h->M, h->K, h->ldb, h->K, h->ldc,
        (const char*)B + (size_t)i * h->typesize, h->a_dense, (char*)C + (size_t)i * h->typesize);
    }
  }
}

void xo_fsspmdm_destroy(xo_fsspmdm* h)
{
  if (NULL != h) { free(h->rowptr); free(h->colidx); free(h->values); free(h->a_dense); free(h); }
}

/* ------------------------------------------------------------------------------------------------
 * spmdm
 * ------------------------------------------------------------------------------------------------ */
void xo_spmdm_init(int M, int N, int K, int max_threads, int bn_isa, xo_spmdm_handle* h)
{ /* libxsmm_spmdm.c:540-608 */
  const double tol = 1.1;
  double avg_work, imb1, avg_blocks, imb2, imb;
  int max_work, max_blocks;
  h->m = M; h->n = N; h->k = K;
  h->bm = (M >= 4096 || M <= 1024) ? 512 : 256;
  h->bn = bn_isa;
  h->bk = 128;
  h->mb = (h->m + h->bm - 1) / h->bm;
  h->nb = (h->n + h->bn - 1) / h->bn;
  h->kb = (h->k + h->bk - 1) / h->bk;
  max_work = h->bm * h->bn;
  avg_work = (double)((size_t)h->m * h->n) / ((size_t)h->mb * h->nb);
  imb1 = max_work / avg_work;
  max_blocks = (h->mb * h->nb + max_threads - 1) / max_threads;
  avg_blocks = (double)h->mb * h->nb / max_threads;
  imb2 = max_blocks / avg_blocks;
  imb = imb1 * imb2;
  while (32 < h->bm && imb > tol) {
    h->bm--;
    h->mb = (h->m + h->bm - 1) / h->bm;
    max_blocks = (h->mb * h->nb + max_threads - 1) / max_threads;
    avg_blocks = (double)h->mb * h->nb / max_threads;
    imb2 = max_blocks / avg_blocks;
    max_work = h->bm * h->bn;
    avg_work = (double)((size_t)h->m * h->n) / ((size_t)h->mb * h->nb);
    imb1 = max_work / avg_work;
    imb = imb1 * imb2;
  }
}

xo_csr_slice* xo_spmdm_alloc_slices(const xo_spmdm_handle* h)
{ /* capacity per slice: libxsmm_spmdm.c:109-112 */
  const int ns = h->mb * h->kb;
  int i;
  xo_csr_slice* s = (xo_csr_slice*)calloc((size_t)ns, sizeof(*s));
  for (i = 0; i < ns; ++i) {
    s[i].rowidx = (uint16_t*)calloc((size_t)h->bm + 1, sizeof(uint16_t));
    s[i].colidx = (uint16_t*)calloc((size_t)h->bm * h->bk, sizeof(uint16_t));
    s[i].values = (float*)calloc((size_t)h->bm * h->bk, sizeof(float));
  }
  return s;
}

void xo_spmdm_free_slices(const xo_spmdm_handle* h, xo_csr_slice* s)
{
  const int ns = h->mb * h->kb; int i;
  if (NULL == s) return;
  for (i = 0; i < ns; ++i) { free(s[i].rowidx); free(s[i].colidx); free(s[i].values); }
  free(s);
}

void xo_spmdm_create_slice(const xo_spmdm_handle* h, char transa, const float* a, xo_csr_slice* slices, int block_id)
{ /* template/libxsmm_spmdm_createSparseSlice_fp32_thread.tpl.c:47-141 */
  const int kb = block_id / h->mb, mb = block_id % h->mb;
  const int ta = ('T' == transa || 't' == transa);
  const size_t off = ta ? ((size_t)mb * h->bm + (size_t)kb * h->m * h->bk) : ((size_t)kb * h->bk + (size_t)mb * h->k * h->bm);
  xo_csr_slice s = slices[kb * h->mb + mb];
  const int nrows = ((mb + 1) * h->bm > h->m) ? (h->m - mb * h->bm) : h->bm;
  const int ncols = ((kb + 1) * h->bk > h->k) ? (h->k - kb * h->bk) : h->bk;
  const float* in = a + off;
  uint16_t cnt = 0;
  int i, k;
  for (i = 0; i < nrows; ++i) {
    s.rowidx[i] = cnt;
    for (k = 0; k < ncols; ++k) {
      const float v = ta ? in[(size_t)k * h->m + i] : in[(size_t)i * h->k + k];
      if (!(0.f == v)) { s.colidx[cnt] = (uint16_t)k; s.values[cnt] = v; ++cnt; } /* LIBXSMM_FEQ(0,v)?0:1 */
    }
  }
  s.rowidx[nrows] = cnt;
}

void xo_spmdm_compute(int arith, const xo_spmdm_handle* h, char transa, char transb, const float* alpha,
                      const xo_csr_slice* slices, const float* b, char transc, const float* beta, float* c, int block_id)
{ /* template/libxsmm_spmdm_compute_fp32_thread.tpl.c:32-558, full-tile ordering (:306-371): per C element
   * acc = beta*C (beta==0 -> 0 without reading C :81-105; beta==1 -> C :107-161; else beta*C :163-212);
   * for kb: for p in row: acc = fma(val_p, B[kb*bk+col_p][n], acc); alpha ignored (include/libxsmm_spmdm.h:104). */
  const int mb = block_id / h->nb, nb = block_id % h->nb;
  const int tb = ('T' == transb || 't' == transb), tc = ('T' == transc || 't' == transc);
  const int m0 = mb * h->bm, n0 = nb * h->bn;
  const int m1 = (m0 + h->bm > h->m) ? h->m : (m0 + h->bm);
  const int n1 = (n0 + h->bn > h->n) ? h->n : (n0 + h->bn);
  int m, n, kb;
  (void)transa; (void)alpha;
  for (m = m0; m < m1; ++m) {
    for (n = n0; n < n1; ++n) {
      float* pc = tc ? &c[(size_t)n * h->m + m] : &c[(size_t)m * h->n + n];
      float acc = (0.f == *beta) ? 0.f : ((1.f == *beta) ? *pc : (*beta) * (*pc));
      for (kb = 0; kb < h->kb; ++kb) {
        const xo_csr_slice s = slices[kb * h->mb + mb];
        const int ml = m - m0;
        unsigned p;
        for (p = s.rowidx[ml]; p < s.rowidx[ml + 1]; ++p) {
          const int kk = kb * h->bk + s.colidx[p];
          const float bv = tb ? b[(size_t)n * h->k + kk] : b[(size_t)kk * h->n + n];
          if (XO_ARITH_FMA == arith) acc = fmaf(s.values[p], bv, acc);
          else { const float pr = s.values[p] * bv; acc = acc + pr; }
        }
      }
      *pc = acc;
    }
  }
}

void xo_spmdm_exec(int arith, int M, int N, int K, int bn_isa, char transa, char transb, char transc,
                   float beta, const float* a, const float* b, float* c)
{ /* call sequence of samples/spmdm/spmdm.c:74-112 */
  xo_spmdm_handle h; xo_csr_slice* s; int i; const float alpha = 1.f;
  xo_spmdm_init(M, N, K, 1, bn_isa, &h);
  s = xo_spmdm_alloc_slices(&h);
  for (i = 0; i < h.mb * h.kb; ++i) xo_spmdm_create_slice(&h, transa, a, s, i);
  for (i = 0; i < h.mb * h.nb; ++i) xo_spmdm_compute(arith, &h, transa, transb, &alpha, s, b, transc, &beta, c, i);
  xo_spmdm_free_slices(&h, s);
}

void xo_spmdm_exec_bf16(int arith, int M, int N, int K, int bn_isa, char transa, char transb, char transc,
                        unsigned short beta_bits, const unsigned short* a, const unsigned short* b, float* c)
{ /* bfloat16 twins: template/libxsmm_spmdm_createSparseSlice_bfloat16_thread.tpl.c:47-144 and
   * template/libxsmm_spmdm_compute_bfloat16_thread.tpl.c differ from the fp32 templates only in widening every input
   * element (bits << 16, libxsmm_spmdm_begin.h:69-75) before the compare / the copy into scratch_B. The scalar `*beta`
   * (a libxsmm_bfloat16*, i.e. unsigned short*) is used as a number as it stands (:91,113,164,187): no widening. */
  const size_t na = (size_t)M * K, nb = (size_t)K * N; size_t i;
  float* const wa = (float*)malloc(sizeof(float) * (na ? na : 1));
  float* const wb = (float*)malloc(sizeof(float) * (nb ? nb : 1));
  if (NULL != wa && NULL != wb) {
    for (i = 0; i < na; ++i) { union { unsigned u; float f; } v; v.u = (unsigned)a[i] << 16; wa[i] = v.f; }
    for (i = 0; i < nb; ++i) { union { unsigned u; float f; } v; v.u = (unsigned)b[i] << 16; wb[i] = v.f; }
    xo_spmdm_exec(arith, M, N, K, bn_isa, transa, transb, transc, (float)beta_bits, wa, wb, c);
  }
  free(wa); free(wb);
}

void xo_spmdm_exec_batch(int arith, int M, int N, int K, int bn_isa, char transa, char transb, char transc,
                         float beta, const float* a, const float* b, float* c, long long batch, int nthreads)
{
  long long i;
  (void)nthreads;
#if defined(_OPENMP)
# pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (i = 0; i < batch; ++i) {
    xo_spmdm_exec(arith, M, N, K, bn_isa, transa, transb, transc, beta,
      a + i * (long long)M * K, b + i * (long long)K * N, c + i * (long long)M * N);
  }
}

/* ------------------------------------------------------------------------------------------------
 * blocked_gemm
 * ------------------------------------------------------------------------------------------------ */
int xo_bgemm_init(xo_bgemm* h, int typesize, int m, int n, int k, int bm, int bn, int bk,
                  int b_m1, int b_n1, int b_k1, int b_k2, double alpha, double beta, int order)
{ /* libxsmm_blocked_gemm.c:55-68: block sizes clamp to the extents, divisibility checks */
  const int mm = (bm < m ? bm : m), kk = (bk < k ? bk : k), nn = (bn < n ? bn : n);
  memset(h, 0, sizeof(*h));
  if (!(0 < m && 0 < n && 0 < k && 0 < mm && 0 < nn && 0 < kk)) return -1;
  if (0 != (m % mm) || 0 != (n % nn) || 0 != (k % kk) || 0 != (m % b_m1) || 0 != (n % b_n1) || 0 != (k % b_k1) ||
      0 != ((k / b_k1 / b_k2) % kk) || 0 != ((n / b_n1) % nn) || 0 != ((m / b_m1) % mm)) return -1;
  if (1.0 != alpha || (1.0 != beta && 0.0 != beta)) return -1; /* descriptor init would fail (NO_BYPASS) */
  h->typesize = typesize; h->m = m; h->n = n; h->k = k; h->bm = mm; h->bn = nn; h->bk = kk;
  h->mb = m / mm; h->nb = n / nn; h->kb = k / kk;
  h->b_m1 = b_m1; h->b_n1 = b_n1; h->b_k1 = b_k1; h->b_k2 = b_k2; h->order = order;
  h->flags = (0.0 == beta ? XO_FLAG_BETA_0 : 0);
  return 0;
}

#define XO_COPY(T, DST, SRC) *(T*)(DST) = *(const T*)(SRC)
static void xo_copy_elem(int ts, void* d, const void* s) { if (8 == ts) XO_COPY(double, d, s); else XO_COPY(float, d, s); }

void xo_bgemm_copyin_a(const xo_bgemm* h, const void* src, int ld, void* dst)
{ /* template/libxsmm_blocked_gemm_copyin_a.tpl.c:32-45: dst[mb][kb][bk][bm] = src[(kb*bk+bk')*ld + mb*bm+bm'] */
  int mb, kb, bk, bm; const int ts = h->typesize;
  for (mb = 0; mb < h->mb; ++mb) for (kb = 0; kb < h->kb; ++kb) for (bk = 0; bk < h->bk; ++bk) for (bm = 0; bm < h->bm; ++bm) {
    const size_t d = (((size_t)mb * h->kb + kb) * h->bk + bk) * h->bm + bm;
    const size_t s = ((size_t)kb * h->bk + bk) * ld + (size_t)mb * h->bm + bm;
    xo_copy_elem(ts, (char*)dst + d * ts, (const char*)src + s * ts);
  }
}

void xo_bgemm_copyin_b(const xo_bgemm* h, const void* src, int ld, void* dst)
{ /* template/libxsmm_blocked_gemm_copyin_b.tpl.c: dst[nb][kb][bn][bk] = src[(nb*bn+bn')*ld + kb*bk+bk'] */
  int nb, kb, bn, bk; const int ts = h->typesize;
  for (nb = 0; nb < h->nb; ++nb) for (kb = 0; kb < h->kb; ++kb) for (bn = 0; bn < h->bn; ++bn) for (bk = 0; bk < h->bk; ++bk) {
    const size_t d = (((size_t)nb * h->kb + kb) * h->bn + bn) * h->bk + bk;
    const size_t s = ((size_t)nb * h->bn + bn) * ld + (size_t)kb * h->bk + bk;
    xo_copy_elem(ts, (char*)dst + d * ts, (const char*)src + s * ts);
  }
}

void xo_bgemm_copyin_c(const xo_bgemm* h, const void* src, int ld, void* dst)
{ /* template/libxsmm_blocked_gemm_copyin_c.tpl.c: dst[nb][mb][bn][bm] = src[(nb*bn+bn')*ld + mb*bm+bm'] */
  int nb, mb, bn, bm; const int ts = h->typesize;
  for (nb = 0; nb < h->nb; ++nb) for (mb = 0; mb < h->mb; ++mb) for (bn = 0; bn < h->bn; ++bn) for (bm = 0; bm < h->bm; ++bm) {
    const size_t d = (((size_t)nb * h->mb + mb) * h->bn + bn) * h->bm + bm;
    const size_t s = ((size_t)nb * h->bn + bn) * ld + (size_t)mb * h->bm + bm;
    xo_copy_elem(ts, (char*)dst + d * ts, (const char*)src + s * ts);
  }
}

void xo_bgemm_copyout_c(const xo_bgemm* h, const void* src, int ld, void* dst)
{ /* template/libxsmm_blocked_gemm_copyout_c.tpl.c:32-45 (inverse of copyin_c) */
  int nb, mb, bn, bm; const int ts = h->typesize;
  for (nb = 0; nb < h->nb; ++nb) for (mb = 0; mb < h->mb; ++mb) for (bn = 0; bn < h->bn; ++bn) for (bm = 0; bm < h->bm; ++bm) {
    const size_t s = (((size_t)nb * h->mb + mb) * h->bn + bn) * h->bm + bm;
    const size_t d = ((size_t)nb * h->bn + bn) * ld + (size_t)mb * h->bm + bm;
    xo_copy_elem(ts, (char*)dst + d * ts, (const char*)src + s * ts);
  }
}

void xo_bgemm_convert_b_to_a(const xo_bgemm* h, const void* src, void* dst)
{ /* template/libxsmm_blocked_gemm_convert_b_to_a.tpl.c:32-46: the two outer block indexes trade places */
  int mb, nb, bn, bm; const int ts = h->typesize;
  for (mb = 0; mb < h->mb; ++mb) for (nb = 0; nb < h->nb; ++nb) for (bn = 0; bn < h->bn; ++bn) for (bm = 0; bm < h->bm; ++bm) {
    const size_t d = (((size_t)mb * h->nb + nb) * h->bn + bn) * h->bm + bm;
    const size_t s = (((size_t)nb * h->mb + mb) * h->bn + bn) * h->bm + bm;
    xo_copy_elem(ts, (char*)dst + d * ts, (const char*)src + s * ts);
  }
}

void xo_bgemm_transpose_b(const xo_bgemm* h, const void* src, void* dst)
{ /* template/libxsmm_blocked_gemm_transpose_b.tpl.c:32-65: src is [kb][nb][bk][bn], dst is [.][kb][bn][bk] */
  int kb, nb, bk, bn; const int ts = h->typesize;
  const int square = (h->n == h->k && h->bn == h->bk);
  for (kb = 0; kb < h->kb; ++kb) for (nb = 0; nb < h->nb; ++nb) for (bk = 0; bk < h->bk; ++bk) for (bn = 0; bn < h->bn; ++bn) {
    const size_t s = (((size_t)kb * h->nb + nb) * h->bk + bk) * h->bn + bn;
    size_t d;
    if (square) d = (((size_t)nb * h->kb + kb) * h->bn + bn) * h->bk + bk;
    else { /* :51-62 */
      const long long job = ((long long)kb * h->bk + bk) * h->n + ((long long)nb * h->bn + bn);
      const long long ii = job / h->k, jj = job % h->k, jobt = jj * h->n + ii;
      const long long q = jobt / h->k, r = jobt % h->k;
      d = (((size_t)(q / h->bn) * h->kb + (size_t)(r / h->bk)) * h->bn + (size_t)(q % h->bn)) * h->bk + (size_t)(r % h->bk);
    }
    xo_copy_elem(ts, (char*)dst + d * ts, (const char*)src + s * ts);
  }
}

void xo_bgemm_order(int order, int w_i, int nw_i, int nw_j, int nw_k, int* i2, int* j2, int* k2)
{ /* internal_bgemm_order, libxsmm_blocked_gemm.c:469-506 */
  switch (order) {
    case 0: *j2 = w_i / (nw_i * nw_k); *i2 = (w_i - *j2 * (nw_i * nw_k)) / nw_k; *k2 = w_i % nw_k; break; /* JIK */
    case 1: *i2 = w_i / (nw_j * nw_k); *j2 = (w_i - *i2 * (nw_j * nw_k)) / nw_k; *k2 = w_i % nw_k; break; /* IJK */
    case 2: *j2 = w_i / (nw_k * nw_i); *k2 = (w_i - *j2 * (nw_k * nw_i)) / nw_i; *i2 = w_i % nw_i; break; /* JKI */
    case 3: *i2 = w_i / (nw_k * nw_j); *k2 = (w_i - *i2 * (nw_k * nw_j)) / nw_j; *j2 = w_i % nw_j; break; /* IKJ */
    case 4: *k2 = w_i / (nw_j * nw_i); *j2 = (w_i - *k2 * (nw_j * nw_i)) / nw_i; *i2 = w_i % nw_i; break; /* KJI */
    default: *k2 = w_i / (nw_i * nw_j); *i2 = (w_i - *k2 * (nw_i * nw_j)) / nw_j; *j2 = w_i % nw_j; break; /* KIJ */
  }
}

void xo_bgemm_st(int arith, const xo_bgemm* h, const void* a, const void* b, void* c)
{ /* template/libxsmm_blocked_gemm.tpl.c:32-165 with nthreads == 1 (ltid = 0): a thread-local block l_out
   * accumulates consecutive work items of the same (i2,j2) and is flushed with C_block += l_out (:93-111,139-160). */
  const int ts = h->typesize;
  const int mm = h->m / h->b_m1, nn = h->n / h->b_n1, kk = h->k / h->b_k1;
  const int nw_i = mm / h->bm, nw_j = nn / h->bn, nw_k = kk / h->bk, nw = nw_i * nw_j;
  const size_t blk = (size_t)h->bm * h->bn;
  void* l_out = calloc(blk, ts);
  int mb, nb, kb, m, n, k, nw_k2 = nw_k;
  size_t e;
#define XO_FLUSH(I2, J2) do { \
    char* cb = (char*)c + ((((size_t)(J2)) * h->mb + (I2)) * blk) * ts; \
    for (e = 0; e < blk; ++e) { \
      if (8 == ts) { ((double*)cb)[e] += ((double*)l_out)[e]; ((double*)l_out)[e] = 0; } \
      else { ((float*)cb)[e] += ((float*)l_out)[e]; ((float*)l_out)[e] = 0; } \
    } } while (0)
  for (mb = 0, m = 0; mb < h->b_m1; ++mb, m += nw_i) {
    for (nb = 0, n = 0; nb < h->b_n1; ++nb, n += nw_j) {
      for (kb = 0, k = 0; kb < h->b_k1; ++kb, k += nw_k2) {
        const int nw_k3 = nw_k / h->b_k2, nw2 = nw * nw_k3;
        int w_i, o_i2 = 0, o_j2 = 0;
        nw_k2 = nw_k3;
        for (w_i = 0; w_i < nw2; ++w_i) {
          int i2, j2, k2, ki, ki2;
          xo_bgemm_order(h->order, w_i, nw_i, nw_j, nw_k2, &i2, &j2, &k2);
          i2 += m; j2 += n; k2 += k;
          if (0 == w_i) { o_i2 = i2; o_j2 = j2; }
          else if (o_i2 != i2 || o_j2 != j2) { XO_FLUSH(o_i2, o_j2); o_i2 = i2; o_j2 = j2; }
          for (ki2 = 0, ki = h->b_k2 * k2; ki2 < h->b_k2; ++ki2, ++ki) {
            const char* ab = (const char*)a + ((((size_t)i2) * h->kb + ki) * h->bk * h->bm) * ts;
            const char* bb = (const char*)b + ((((size_t)j2) * h->kb + ki) * h->bn * h->bk) * ts;
            xo_smm(arith, ts, h->flags, h->bm, h->bn, h->bk, h->bm, h->bk, h->bm, ab, bb, l_out);
          }
          if (w_i == nw2 - 1) { o_i2 = i2; o_j2 = j2; XO_FLUSH(o_i2, o_j2); }
        }
      }
    }
  }
#undef XO_FLUSH
  free(l_out);
}

/* ------------------------------------------------------------------------------------------------
 * generators of the samples' inputs
 * ------------------------------------------------------------------------------------------------ */
#define XO_DEFINE_MATINIT(NAME, T)                                                                 \
void NAME(int seed, T* dst, int nrows, int ncols, int ld, double scale)                            \
{ /* include/libxsmm_frontend.h:414-431, seed != 0 */                                              \
  const double seed1 = scale * (double)seed + scale;                                               \
  int i, j;                                                                                        \
  for (i = 0; i < ncols; ++i) {                                                                    \
    for (j = 0; j < nrows; ++j) { const int kk = i * ld + j; dst[kk] = (T)(seed1 / (1.0 + kk)); }  \
    for (; j < ld; ++j) { const int kk = i * ld + j; dst[kk] = (T)seed; }                          \
  }                                                                                                \
}
XO_DEFINE_MATINIT(xo_matinit_f64, double)
XO_DEFINE_MATINIT(xo_matinit_f32, float)

/* POSIX drand48: X(n+1) = (a*X(n) + c) mod 2^48, a = 0x5DEECE66D, c = 0xB; srand48(s): X = (s << 16) | 0x330E */
static uint64_t xo_rng_state = 0x1234ABCD330EULL;
void xo_rng_seed(unsigned seed) { xo_rng_state = (((uint64_t)seed) << 16) | 0x330EULL; }
double xo_rng_f64(void)
{
  xo_rng_state = (0x5DEECE66DULL * xo_rng_state + 0xBULL) & 0xFFFFFFFFFFFFULL;
  return (double)xo_rng_state / 281474976710656.0; /* 2^48 */
}

void xo_matdiff(int typesize, int m, int n, const void* ref, const void* tst, int ldref, int ldtst,
                double* linf_abs, double* normf_rel)
{ /* src/template/libxsmm_matdiff.tpl.c:40-140; normf_rel = sqrt(sum d^2 / sum ref^2) (libxsmm_math.c:140-146) */
  double linf = 0, l2 = 0, nr = 0; int i, j;
  for (i = 0; i < n; ++i) for (j = 0; j < m; ++j) {
    const double r = (8 == typesize ? ((const double*)ref)[(size_t)i * ldref + j] : (double)((const float*)ref)[(size_t)i * ldref + j]);
    const double t = (8 == typesize ? ((const double*)tst)[(size_t)i * ldtst + j] : (double)((const float*)tst)[(size_t)i * ldtst + j]);
    const double d = fabs(r - t);
    if (!(t == t) || isinf(t)) { linf = INFINITY; l2 = INFINITY; continue; }
    if (linf < d) linf = d;
    l2 += d * d; nr += r * r;
  }
  if (NULL != linf_abs) *linf_abs = linf;
  if (NULL != normf_rel) *normf_rel = (0 < nr ? sqrt(l2 / nr) : sqrt(l2));
}

/* ------------------------------------------------------------------------------------------------
 * SOA kernels (EDGE/SeisSol "fused runs"): B and/or A and C are [row][col][v] with v = soa width innermost.
 * Every kernel is element-wise in v. Arithmetic: fused multiply-add, accumulator seeded with C (or 0 for beta == 0),
 * products in ascending k -- what the AVX2/AVX-512 generators emit (vfmadd231 on a register accumulator):
 *  - generator_spgemm_csr_asparse_soa.c:212-330: rows of A without non-zeros are not touched (not even for beta == 0);
 *  - generator_spgemm_csc_bsparse_soa.c:177-420, generator_spgemm_csr_bsparse_soa.c:160-330: every C column of the
 *    block is loaded/zeroed and stored; for k ascending the first entry (k, n) of the pattern contributes;
 *  - generator_gemm_rm_ac_soa.c / generator_gemm_rm_bc_soa.c: dense counterparts (gold loops of
 *    samples/edge/dense_rmacsoa.c:60-80, dense_rmbcsoa.c).
 * ------------------------------------------------------------------------------------------------ */
#define XO_DEFINE_SOA(SUFFIX, T, FMAF)                                                                  \
void xo_soa_csr_asparse_##SUFFIX(int flags, int m, int n, int k, int ldb, int ldc, int v,               \
  const unsigned* rowptr, const unsigned* colidx, const T* a_vals, const T* b, T* c)                     \
{                                                                                                       \
  int im, in, iv; unsigned p; (void)k;                                                                  \
  for (im = 0; im < m; ++im) {                                                                          \
    if (rowptr[im] == rowptr[im + 1]) continue;                                                         \
    for (in = 0; in < n; ++in) for (iv = 0; iv < v; ++iv) {                                             \
      T acc = (0 != (flags & XO_FLAG_BETA_0)) ? (T)0 : c[((size_t)im * ldc + in) * v + iv];             \
      for (p = rowptr[im]; p < rowptr[im + 1]; ++p) acc = FMAF(a_vals[p], b[((size_t)colidx[p] * ldb + in) * v + iv], acc); \
      c[((size_t)im * ldc + in) * v + iv] = acc;                                                        \
    }                                                                                                   \
  }                                                                                                     \
}                                                                                                       \
void xo_soa_bsparse_##SUFFIX(int flags, int csr, int m, int n, int k, int lda, int ldc, int v,          \
  const unsigned* ptr, const unsigned* idx, const T* a, const T* b_vals, T* c)                           \
{                                                                                                       \
  int im, in, iv, ik; unsigned p;                                                                       \
  for (im = 0; im < m; ++im) for (in = 0; in < n; ++in) for (iv = 0; iv < v; ++iv) {                    \
    T acc = (0 != (flags & XO_FLAG_BETA_0)) ? (T)0 : c[((size_t)im * ldc + in) * v + iv];               \
    for (ik = 0; ik < k; ++ik) {                                                                        \
      if (0 != csr) { /* row ik of B: first entry in column `in` */                                     \
        for (p = ptr[ik]; p < ptr[ik + 1]; ++p) if (idx[p] == (unsigned)in) { acc = FMAF(a[((size_t)im * lda + ik) * v + iv], b_vals[p], acc); break; } \
      }                                                                                                 \
      else { /* column `in` of B: first entry in row ik */                                              \
        for (p = ptr[in]; p < ptr[in + 1]; ++p) if (idx[p] == (unsigned)ik) { acc = FMAF(a[((size_t)im * lda + ik) * v + iv], b_vals[p], acc); break; } \
      }                                                                                                 \
    }                                                                                                   \
    c[((size_t)im * ldc + in) * v + iv] = acc;                                                          \
  }                                                                                                     \
}                                                                                                       \
void xo_soa_rm_ac_##SUFFIX(int flags, int m, int n, int k, int lda, int ldb, int ldc, int v, const T* a, const T* b, T* c) \
{                                                                                                       \
  int im, in, iv, ik;                                                                                   \
  for (im = 0; im < m; ++im) for (in = 0; in < n; ++in) for (iv = 0; iv < v; ++iv) {                    \
    T acc = (0 != (flags & XO_FLAG_BETA_0)) ? (T)0 : c[((size_t)im * ldc + in) * v + iv];               \
    for (ik = 0; ik < k; ++ik) acc = FMAF(a[((size_t)im * lda + ik) * v + iv], b[(size_t)ik * ldb + in], acc); \
    c[((size_t)im * ldc + in) * v + iv] = acc;                                                          \
  }                                                                                                     \
}                                                                                                       \
void xo_soa_rm_bc_##SUFFIX(int flags, int m, int n, int k, int lda, int ldb, int ldc, int v, const T* a, const T* b, T* c) \
{                                                                                                       \
  int im, in, iv, ik;                                                                                   \
  for (im = 0; im < m; ++im) for (in = 0; in < n; ++in) for (iv = 0; iv < v; ++iv) {                    \
    T acc = (0 != (flags & XO_FLAG_BETA_0)) ? (T)0 : c[((size_t)im * ldc + in) * v + iv];               \
    for (ik = 0; ik < k; ++ik) acc = FMAF(a[(size_t)im * lda + ik], b[((size_t)ik * ldb + in) * v + iv], acc); \
    c[((size_t)im * ldc + in) * v + iv] = acc;                                                          \
  }                                                                                                     \
}
XO_DEFINE_SOA(f64, double, fma)
XO_DEFINE_SOA(f32, float, fmaf)

/* ---- low-precision dense kernels: the gold loops of samples/xgemm/kernel.c (see xsmm_oracle.h) ---- */
static float xo_bf16_to_f32(unsigned short v) { union { unsigned int u; float f; } t; t.u = (unsigned int)v << 16; return t.f; }
static unsigned short xo_f32_to_bf16_trunc(float f) { union { unsigned int u; float f; } t; t.f = f; return (unsigned short)(t.u >> 16); }

int xo_gemm_lowp(int kind, int beta0, int m, int n, int k, int lda, int ldb, int ldc,
                 const unsigned short* a, const unsigned short* b, void* c, float scf)
{
  int i, j, s, k2;
  if (kind < 0 || kind > 3 || 0 != (k % 2) || lda < m || ldb < k || ldc < m) return 1;
  for (j = 0; j < n; ++j) {
    for (i = 0; i < m; ++i) {
      const size_t ci = (size_t)j * ldc + i;
      if (0 == kind) { /* kernel.c:915-927; the int sum wraps like the hardware's (unsigned arithmetic here: defined) */
        unsigned int acc = beta0 ? 0u : (unsigned int)((int*)c)[ci];
        for (s = 0; s < k / 2; ++s) for (k2 = 0; k2 < 2; ++k2) {
          acc += (unsigned int)((int)(short)a[(size_t)s * lda * 2 + (size_t)i * 2 + k2] * (int)(short)b[(size_t)j * ldb + (size_t)s * 2 + k2]);
        }
        ((int*)c)[ci] = (int)acc;
      }
      else if (1 == kind) { /* kernel.c:1007-1021: (float)iprod * scf, then the add -- three roundings, no contraction */
        volatile float acc = beta0 ? 0.f : ((float*)c)[ci];
        for (s = 0; s < k / 2; ++s) for (k2 = 0; k2 < 2; ++k2) {
          const int iprod = (int)(short)a[(size_t)s * lda * 2 + (size_t)i * 2 + k2] * (int)(short)b[(size_t)j * ldb + (size_t)s * 2 + k2];
          volatile float fprod = (float)iprod;
          volatile float scaled = fprod * scf;
          acc = acc + scaled;
        }
        ((float*)c)[ci] = acc;
      }
      else { /* kernel.c:1104-1123 and :1207-1229: the product of two bf16 values is exact in a float, one rounding per add */
        volatile float acc = beta0 ? 0.f : (2 == kind ? ((float*)c)[ci] : xo_bf16_to_f32(((unsigned short*)c)[ci]));
        for (s = 0; s < k / 2; ++s) for (k2 = 0; k2 < 2; ++k2) {
          volatile float prod = xo_bf16_to_f32(a[(size_t)s * lda * 2 + (size_t)i * 2 + k2]) * xo_bf16_to_f32(b[(size_t)j * ldb + (size_t)s * 2 + k2]);
          acc = acc + prod;
        }
        if (2 == kind) ((float*)c)[ci] = acc; else ((unsigned short*)c)[ci] = xo_f32_to_bf16_trunc(acc);
      }
    }
  }
  return 0;
}
