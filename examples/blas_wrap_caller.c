/* An application that knows nothing about LIBXSMM: it calls the Fortran BLAS symbol dgemm_/sgemm_ (column-major, all
 * arguments by reference). Relinked with -Wl,--wrap=dgemm_,--wrap=sgemm_ against libxsmm.so its calls are served by the GPU
 * engine (reference: documentation/libxsmm_mm.md "Call Wrapper", src/libxsmm_ext_gemm.c:314-587, sample
 * samples/utilities/wrap). Shapes inside and outside the SMM domain, general alpha/beta, transposes; checked against plain loops.
 * Build: gcc examples/blas_wrap_caller.c -Wl,--wrap=dgemm_,--wrap=sgemm_ -L libxsmm-1_amd/lib -lxsmm -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void dgemm_(const char*, const char*, const int*, const int*, const int*, const double*, const double*, const int*,
            const double*, const int*, const double*, double*, const int*);
void sgemm_(const char*, const char*, const int*, const int*, const int*, const float*, const float*, const int*,
            const float*, const int*, const float*, float*, const int*);

static double value(int seed, int i) { return ((seed * 7919 + i * 104729) % 2003) / 2003.0 - 0.5; }

static int check_d(char ta, char tb, int m, int n, int k, double alpha, double beta)
{
  const int lda = ('N' == ta ? m : k) + 3, ldb = ('N' == tb ? k : n) + 1, ldc = m + 2;
  const int acols = ('N' == ta ? k : m), bcols = ('N' == tb ? n : k);
  double *a = (double*)malloc(sizeof(double) * lda * acols), *b = (double*)malloc(sizeof(double) * ldb * bcols);
  double *c = (double*)malloc(sizeof(double) * ldc * n), *g = (double*)malloc(sizeof(double) * ldc * n);
  double diff = 0, scale = 0; int i, j, p;
  for (i = 0; i < lda * acols; ++i) a[i] = value(1, i);
  for (i = 0; i < ldb * bcols; ++i) b[i] = value(2, i);
  for (i = 0; i < ldc * n; ++i) g[i] = c[i] = value(3, i);
  for (j = 0; j < n; ++j) for (i = 0; i < m; ++i) {
    double s = 0;
    for (p = 0; p < k; ++p) s += ('N' == ta ? a[p * lda + i] : a[i * lda + p]) * ('N' == tb ? b[j * ldb + p] : b[p * ldb + j]);
    g[j * ldc + i] = alpha * s + beta * g[j * ldc + i];
  }
  dgemm_(&ta, &tb, &m, &n, &k, &alpha, a, &lda, b, &ldb, &beta, c, &ldc);
  for (i = 0; i < ldc * n; ++i) { diff = fmax(diff, fabs(c[i] - g[i])); scale = fmax(scale, fabs(g[i])); }
  free(a); free(b); free(c); free(g);
  if (!(diff <= 1e-12 * scale * k)) { fprintf(stderr, "dgemm %c%c %dx%dx%d alpha=%g beta=%g: diff %g\n", ta, tb, m, n, k, alpha, beta, diff); return 1; }
  return 0;
}

static int check_s(int m, int n, int k)
{
  const char nn = 'N'; const float alpha = 1.f, beta = 1.f;
  float *a = (float*)malloc(sizeof(float) * m * k), *b = (float*)malloc(sizeof(float) * k * n), *c = (float*)malloc(sizeof(float) * m * n);
  double* g = (double*)malloc(sizeof(double) * m * n);
  double diff = 0, scale = 0; int i, j, p;
  for (i = 0; i < m * k; ++i) a[i] = (float)value(4, i);
  for (i = 0; i < k * n; ++i) b[i] = (float)value(5, i);
  for (i = 0; i < m * n; ++i) g[i] = c[i] = (float)value(6, i);
  for (j = 0; j < n; ++j) for (p = 0; p < k; ++p) for (i = 0; i < m; ++i) g[j * m + i] += (double)a[p * m + i] * (double)b[j * k + p];
  sgemm_(&nn, &nn, &m, &n, &k, &alpha, a, &m, b, &k, &beta, c, &m);
  for (i = 0; i < m * n; ++i) { diff = fmax(diff, fabs((double)c[i] - g[i])); scale = fmax(scale, fabs(g[i])); }
  free(a); free(b); free(c); free(g);
  if (!(diff <= 1e-6 * scale * k)) { fprintf(stderr, "sgemm %dx%dx%d: diff %g\n", m, n, k, diff); return 1; }
  return 0;
}

int main(void)
{
  int result = 0;
  result |= check_d('N', 'N', 23, 23, 23, 1.0, 1.0);   /* SMM domain */
  result |= check_d('N', 'N', 32, 8, 64, 1.0, 0.0);
  result |= check_d('N', 'T', 13, 17, 5, 1.0, 1.0);
  result |= check_d('T', 'N', 20, 10, 30, 1.0, 1.0);   /* outside: transposed A */
  result |= check_d('N', 'N', 40, 50, 60, -0.5, 2.0);  /* outside: general alpha/beta */
  result |= check_d('T', 'T', 100, 70, 130, 1.5, 0.0);
  result |= check_s(32, 32, 32);
  result |= check_s(64, 48, 96);
  if (0 == result) printf("blas_wrap_caller: wrapped dgemm_/sgemm_ agree with the plain loops\n");
  return result;
}
