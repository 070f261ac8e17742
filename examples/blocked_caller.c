/* The flow of the reference's samples/blocked_gemm/blocked_gemm.c (:121-150,181) against the reference API only: create a
 * handle, copy A, B, C into the block-major layouts, run libxsmm_blocked_gemm_omp, copy C out, compare with a plain GEMM.
 * Build: gcc -I include examples/blocked_caller.c -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib -lm */
#include <libxsmm.h>
#include <math.h>
#include <stdio.h>

int main(void)
{
  const libxsmm_blasint m = 128, n = 96, k = 160, bm = 32, bn = 32, bk = 32, one = 1;
  const float alpha = 1.f, beta = 1.f;
  const libxsmm_blocked_gemm_order order = LIBXSMM_BLOCKED_GEMM_ORDER_JIK;
  float *a, *b, *c, *ab, *bb, *cb, *out;
  double maxerr = 0, scale = 0;
  libxsmm_blocked_gemm_handle* h;
  int i, j, p;

  libxsmm_init();
  a = (float*)libxsmm_malloc(sizeof(float) * m * k); b = (float*)libxsmm_malloc(sizeof(float) * k * n); c = (float*)libxsmm_malloc(sizeof(float) * m * n);
  ab = (float*)libxsmm_malloc(sizeof(float) * m * k); bb = (float*)libxsmm_malloc(sizeof(float) * k * n); cb = (float*)libxsmm_malloc(sizeof(float) * m * n);
  out = (float*)libxsmm_malloc(sizeof(float) * m * n);
  if (NULL == a || NULL == b || NULL == c || NULL == ab || NULL == bb || NULL == cb || NULL == out) return 100;
  LIBXSMM_MATINIT(float, 42, a, m, k, m, 1.0);
  LIBXSMM_MATINIT(float, 24, b, k, n, k, 1.0);
  LIBXSMM_MATINIT(float, 22, c, m, n, m, 1.0);
  h = libxsmm_blocked_gemm_handle_create(1, LIBXSMM_GEMM_PRECISION_F32, LIBXSMM_GEMM_PRECISION_F32, m, n, k, &bm, &bn, &bk,
    &one, &one, &one, &one, &alpha, &beta, NULL, NULL, &order);
  if (NULL == h) { fprintf(stderr, "handle_create failed\n"); return 1; }
  if (EXIT_SUCCESS != libxsmm_blocked_gemm_copyin_a(h, a, &m, ab) || EXIT_SUCCESS != libxsmm_blocked_gemm_copyin_b(h, b, &k, bb)
   || EXIT_SUCCESS != libxsmm_blocked_gemm_copyin_c(h, c, &m, cb)) return 2;
  libxsmm_blocked_gemm_omp(h, ab, bb, cb, 1);
  if (EXIT_SUCCESS != libxsmm_blocked_gemm_copyout_c(h, cb, &m, out)) return 3;
  libxsmm_blocked_gemm_handle_destroy(h);
  for (j = 0; j < n; ++j) for (i = 0; i < m; ++i) {
    double sum = c[j * m + i];
    for (p = 0; p < k; ++p) sum += (double)a[p * m + i] * (double)b[j * k + p];
    if (fabs(sum - out[j * m + i]) > maxerr) maxerr = fabs(sum - out[j * m + i]);
    if (fabs(sum) > scale) scale = fabs(sum);
  }
  libxsmm_free(a); libxsmm_free(b); libxsmm_free(c); libxsmm_free(ab); libxsmm_free(bb); libxsmm_free(cb); libxsmm_free(out);
  libxsmm_finalize();
  printf("blocked_caller: max error %g (scale %g)\n", maxerr, scale);
  return maxerr <= 2e-5 * scale ? 0 : 4;
}
