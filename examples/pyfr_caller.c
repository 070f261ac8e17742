/* The panel loop of the reference's samples/pyfr/pyfr_driver_asp_reg.c (:275-309,351-380) against the reference API only:
 * a fixed sparse operator A (M x K, row-major, ~15 % non-zeros from a small palette) is applied to column panels of a wide
 * row-major B (K x N_total) into C (M x N_total): libxsmm_dfsspmdm_create once, libxsmm_dfsspmdm_execute per panel of N
 * columns, compared with the sample's gold loop.
 * Build: gcc -I include examples/pyfr_caller.c -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib -lm */
#include <libxsmm.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

int main(void)
{
  const int M = 35, K = 35, N = 96, npanels = 12, ntot = N * npanels;
  static const double palette[7] = { 0.25, -0.5, 0.75, 1.0, -1.25, 1.5, -2.0 };
  double *a, *b, *c, *gold, maxerr = 0;
  libxsmm_dfsspmdm* h;
  int i, j, p, beta_i, result = 0;

  libxsmm_init();
  a = (double*)libxsmm_aligned_malloc(sizeof(double) * M * K, 64); b = (double*)libxsmm_aligned_malloc(sizeof(double) * K * ntot, 64);
  c = (double*)libxsmm_aligned_malloc(sizeof(double) * M * ntot, 64); gold = (double*)malloc(sizeof(double) * M * ntot);
  if (NULL == a || NULL == b || NULL == c || NULL == gold) return 100;
  libxsmm_rng_set_seed(1);
  for (i = 0; i < M * K; ++i) { const double r = libxsmm_rng_f64(); a[i] = (r < 0.15 ? palette[(int)(r * 1000) % 7] : 0.0); }
  for (i = 0; i < M; ++i) a[i * K + (i % K)] = palette[i % 7]; /* no empty rows */
  for (i = 0; i < K * ntot; ++i) b[i] = libxsmm_rng_f64();
  for (beta_i = 1; beta_i >= 0; --beta_i) {
    const double beta = (double)beta_i;
    for (i = 0; i < M * ntot; ++i) c[i] = gold[i] = libxsmm_rng_f64();
    for (i = 0; i < M; ++i) for (j = 0; j < ntot; ++j) { /* gold (pyfr_driver_asp_reg.c:275-293) */
      double sum = beta * gold[i * ntot + j];
      for (p = 0; p < K; ++p) sum += a[i * K + p] * b[p * ntot + j];
      gold[i * ntot + j] = sum;
    }
    h = libxsmm_dfsspmdm_create(M, N, K, K, ntot, ntot, 1.0, beta, a);
    if (NULL == h) { fprintf(stderr, "fsspmdm_create failed\n"); return 1; }
    for (p = 0; p < npanels; ++p) libxsmm_dfsspmdm_execute(h, b + p * N, c + p * N);
    libxsmm_dfsspmdm_destroy(h);
    for (i = 0; i < M * ntot; ++i) if (fabs(c[i] - gold[i]) > maxerr) maxerr = fabs(c[i] - gold[i]);
    if (maxerr > 1e-11) { fprintf(stderr, "beta=%g: max error %g\n", beta, maxerr); result |= (2 << beta_i); }
  }
  libxsmm_free(a); libxsmm_free(b); libxsmm_free(c); free(gold);
  libxsmm_finalize();
  if (0 == result) printf("pyfr_caller: max error %g\n", maxerr);
  return result;
}
