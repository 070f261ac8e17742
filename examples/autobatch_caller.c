/* Auto-batching of intercepted BLAS calls, the flow of the reference's samples/utilities/wrap/autobatch.c (:97-126):
 * between libxsmm_mmbatch_begin and libxsmm_mmbatch_end the wrapped dgemm_ calls (here called by their wrapper name, as
 * the sample does when it is not relinked) that match the announced shape are recorded and executed as ONE device batch at
 * the end; calls of another shape go through at once. Reference API only; every C block is checked against a plain loop.
 * Build: gcc -I include examples/autobatch_caller.c -L libxsmm-1_amd/lib -lxsmm -lm */
#include <libxsmm.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define GEMM LIBXSMM_FSYMBOL(LIBXSMM_CONCATENATE(__wrap_, LIBXSMM_TPREFIX(double, gemm)))

static double gold_diff(const double* a, const double* b, const double* c, const double* c0, int m, int n, int k, double beta)
{
  double diff = 0; int i, j, p;
  for (j = 0; j < n; ++j) for (i = 0; i < m; ++i) {
    double s = 0;
    for (p = 0; p < k; ++p) s += a[p * m + i] * b[j * k + p];
    diff = fmax(diff, fabs(c[j * m + i] - (s + beta * c0[j * m + i])));
  }
  return diff;
}

int main(void)
{
  const libxsmm_blasint m = 23, n = 17, k = 29, lda = 23, ldb = 29, ldc = 23, size = 500;
  const libxsmm_blasint m2 = 12, n2 = 12, k2 = 12;
  const double alpha = 1.0, beta = 1.0;
  const char transa = 'N', transb = 'N';
  const int flags = LIBXSMM_GEMM_FLAGS(transa, transb);
  const size_t asz = (size_t)m * k, bsz = (size_t)k * n, csz = (size_t)m * n;
  double *a, *b, *c, *c0, other[12 * 12], other0[12 * 12], diff = 0;
  int i;
  libxsmm_init();
  a = (double*)malloc(sizeof(double) * asz * size); b = (double*)malloc(sizeof(double) * bsz * size);
  c = (double*)malloc(sizeof(double) * csz * size); c0 = (double*)malloc(sizeof(double) * csz * size);
  if (0 == a || 0 == b || 0 == c || 0 == c0) return 100;
  for (i = 0; i < size; ++i) {
    LIBXSMM_MATINIT(double, 42 + i, a + i * asz, m, k, m, 1.0 / size);
    LIBXSMM_MATINIT(double, 24 + i, b + i * bsz, k, n, k, 1.0 / size);
    LIBXSMM_MATINIT(double, 22 + i, c + i * csz, m, n, m, 1.0 / size);
  }
  for (i = 0; i < (int)(csz * size); ++i) c0[i] = c[i];
  for (i = 0; i < 144; ++i) other0[i] = other[i] = 0.5 - i / 144.0;

  libxsmm_mmbatch_begin(LIBXSMM_GEMM_PRECISION_F64, &flags, &m, &n, &k, &lda, &ldb, &ldc, &alpha, &beta);
  for (i = 0; i < size; ++i) {
    GEMM(&transa, &transb, &m, &n, &k, &alpha, a + i * asz, &lda, b + i * bsz, &ldb, &beta, c + i * csz, &ldc);
    if (i == size / 2) { /* a call that does not match the recording: executed at once */
      GEMM(&transa, &transb, &m2, &n2, &k2, &alpha, a, &m2, b, &k2, &beta, other, &m2);
      diff = fmax(diff, gold_diff(a, b, other, other0, m2, n2, k2, beta));
    }
  }
  libxsmm_mmbatch_end(); /* the recorded products are executed here */
  for (i = 0; i < size; ++i) diff = fmax(diff, gold_diff(a + i * asz, b + i * bsz, c + i * csz, c0 + i * csz, m, n, k, beta));
  libxsmm_finalize();
  free(a); free(b); free(c); free(c0);
  if (!(diff <= 1e-12)) { fprintf(stderr, "autobatch_caller: diff %g\n", diff); return 1; }
  printf("autobatch_caller: %d recorded dgemm calls (one batch) and one direct call agree with the plain loops\n", (int)size);
  return 0;
}
