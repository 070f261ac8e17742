/* A caller written against the reference API only (no libxsmm_amd_* call): the flow of the reference's
 * samples/smm/specialized.cpp (streamed case, :143-237) and of a CP2K-style stack (samples/cp2k/cp2k.cpp:328-360):
 *   dispatch a kernel, (1) call it once per item through the bare function pointer, (2) hand the same batch to
 *   libxsmm_gemm_batch with index arrays, (3) accumulate a stack of products into a few C blocks; compare each result
 *   with a plain triple loop. Operands come from libxsmm_malloc (memory both the CPU and the GPU address), initialised
 *   with the reference's LIBXSMM_MATINIT.
 * Build: gcc -I include examples/smm_caller.c -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib -lm
 * Exit code 0 = all checks passed. */
#include <libxsmm.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

static void naive(int m, int n, int k, const double* a, const double* b, double* c)
{ /* column-major, C += A * B */
  int i, j, p;
  for (j = 0; j < n; ++j) for (p = 0; p < k; ++p) for (i = 0; i < m; ++i) c[j * m + i] += a[p * m + i] * b[j * k + p];
}

static double max_rel_diff(const double* x, const double* y, size_t count)
{
  double d = 0, s = 0; size_t i;
  for (i = 0; i < count; ++i) { const double e = fabs(x[i] - y[i]); if (e > d) d = e; if (fabs(y[i]) > s) s = fabs(y[i]); }
  return d / (s > 0 ? s : 1);
}

int main(void)
{
  const int m = 23, n = 23, k = 23, s = 200, nc = 8;
  const size_t asz = (size_t)m * k, bsz = (size_t)k * n, csz = (size_t)m * n;
  double *a, *b, *c, *gold, *cstack, *gstack;
  libxsmm_blasint *ia, *ib, *ic;
  libxsmm_dmmfunction kernel;
  const double alpha = 1, beta = 1;
  int i, result = 0;

  libxsmm_init();
  a = (double*)libxsmm_malloc(sizeof(double) * asz * s); b = (double*)libxsmm_malloc(sizeof(double) * bsz * s);
  c = (double*)libxsmm_malloc(sizeof(double) * csz * s); gold = (double*)malloc(sizeof(double) * csz * s);
  cstack = (double*)libxsmm_malloc(sizeof(double) * csz * nc); gstack = (double*)malloc(sizeof(double) * csz * nc);
  ia = (libxsmm_blasint*)malloc(sizeof(libxsmm_blasint) * s); ib = (libxsmm_blasint*)malloc(sizeof(libxsmm_blasint) * s);
  ic = (libxsmm_blasint*)malloc(sizeof(libxsmm_blasint) * s);
  if (NULL == a || NULL == b || NULL == c || NULL == gold || NULL == cstack || NULL == gstack || NULL == ia || NULL == ib || NULL == ic) return 100;
  for (i = 0; i < s; ++i) { /* seeds as in the sample: 42+i, 24+i, 22+i, scale 1/s */
    LIBXSMM_MATINIT(double, 42 + i, a + i * asz, m, k, m, 1.0 / s);
    LIBXSMM_MATINIT(double, 24 + i, b + i * bsz, k, n, k, 1.0 / s);
    LIBXSMM_MATINIT(double, 22 + i, c + i * csz, m, n, m, 1.0 / s);
  }
  memcpy(gold, c, sizeof(double) * csz * s);
  for (i = 0; i < s; ++i) naive(m, n, k, a + i * asz, b + i * bsz, gold + i * csz);

  /* (1) one kernel call per item */
  kernel = libxsmm_dmmdispatch(m, n, k, NULL, NULL, NULL, &alpha, &beta, NULL, NULL);
  if (NULL == kernel) { fprintf(stderr, "no kernel\n"); return 1; }
  for (i = 0; i < s; ++i) kernel(a + i * asz, b + i * bsz, c + i * csz);
  if (max_rel_diff(c, gold, csz * s) > 1e-12) { fprintf(stderr, "per-call kernel: mismatch\n"); result |= 2; }

  /* (2) the same batch through libxsmm_gemm_batch (index arrays, distinct C) */
  for (i = 0; i < s; ++i) {
    LIBXSMM_MATINIT(double, 22 + i, c + i * csz, m, n, m, 1.0 / s);
    ia[i] = (libxsmm_blasint)(i * asz); ib[i] = (libxsmm_blasint)(i * bsz); ic[i] = (libxsmm_blasint)(i * csz);
  }
  libxsmm_gemm_batch(LIBXSMM_GEMM_PRECISION_F64, LIBXSMM_GEMM_PRECISION_F64, "N", "N", m, n, k, &alpha, a, NULL, b, NULL, &beta, c, NULL,
    0, (libxsmm_blasint)sizeof(libxsmm_blasint), ia, ib, ic, s);
  if (max_rel_diff(c, gold, csz * s) > 1e-12) { fprintf(stderr, "gemm_batch: mismatch\n"); result |= 4; }

  /* (3) a stack: consecutive products accumulate into one of nc C blocks */
  for (i = 0; i < nc; ++i) LIBXSMM_MATINIT(double, 7 + i, cstack + i * csz, m, n, m, 1.0);
  memcpy(gstack, cstack, sizeof(double) * csz * nc);
  for (i = 0; i < s; ++i) { ic[i] = (libxsmm_blasint)((i * nc / s) * csz); naive(m, n, k, a + i * asz, b + i * bsz, gstack + (i * nc / s) * csz); }
  libxsmm_gemm_batch(LIBXSMM_GEMM_PRECISION_F64, LIBXSMM_GEMM_PRECISION_F64, "N", "N", m, n, k, &alpha, a, NULL, b, NULL, &beta, cstack, NULL,
    0, (libxsmm_blasint)sizeof(libxsmm_blasint), ia, ib, ic, s);
  if (max_rel_diff(cstack, gstack, csz * nc) > 1e-12) { fprintf(stderr, "stack: mismatch\n"); result |= 8; }

  libxsmm_free(a); libxsmm_free(b); libxsmm_free(c); libxsmm_free(cstack);
  free(gold); free(gstack); free(ia); free(ib); free(ic);
  libxsmm_finalize();
  if (0 == result) printf("smm_caller: per-call kernel, gemm_batch and stack accumulate agree with the plain loops\n");
  return result;
}
