/* The call sequence of the reference's samples/spmdm/spmdm.c (:74-112,205-243) against the reference API only:
 * libxsmm_spmdm_init, createSparseSlice over all blocks, compute over all blocks, destroy; inputs from libxsmm_rng_f64 with
 * seed 1 (A first, then B, then C; A sparsified with the sample's `r > 0.85` rule), result compared with a plain loop.
 * Build: gcc -I include examples/spmdm_caller.c -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib -lm */
#include <libxsmm.h>
#include <math.h>
#include <stdio.h>

int main(void)
{
  const int M = 300, N = 130, K = 260, nthreads = 1;
  const float alpha = 1.f, beta = 0.f;
  libxsmm_spmdm_handle handle;
  libxsmm_CSR_sparseslice* slices = NULL;
  float *a, *b, *c;
  double maxerr = 0;
  int i, j, p, blk, nblk;

  libxsmm_init();
  a = (float*)libxsmm_aligned_malloc(sizeof(float) * M * K, 64); b = (float*)libxsmm_aligned_malloc(sizeof(float) * K * N, 64);
  c = (float*)libxsmm_aligned_malloc(sizeof(float) * M * N, 64);
  if (NULL == a || NULL == b || NULL == c) return 100;
  libxsmm_rng_set_seed(1);
  for (i = 0; i < M * K; ++i) { const double r = libxsmm_rng_f64(); a[i] = (float)(r > 0.85 ? r : 0.0); }
  for (i = 0; i < K * N; ++i) b[i] = (float)libxsmm_rng_f64();
  for (i = 0; i < M * N; ++i) c[i] = (float)libxsmm_rng_f64();

  libxsmm_spmdm_init(M, N, K, nthreads, &handle, &slices);
  if (NULL == slices) { fprintf(stderr, "spmdm_init failed\n"); return 1; }
  nblk = libxsmm_spmdm_get_num_createSparseSlice_blocks(&handle);
  for (blk = 0; blk < nblk; ++blk) libxsmm_spmdm_createSparseSlice_fp32_thread(&handle, 'N', a, slices, blk, 0, nthreads);
  nblk = libxsmm_spmdm_get_num_compute_blocks(&handle);
  for (blk = 0; blk < nblk; ++blk) libxsmm_spmdm_compute_fp32_thread(&handle, 'N', 'N', &alpha, slices, b, 'N', &beta, c, blk, 0, nthreads);
  libxsmm_spmdm_destroy(&handle);

  for (i = 0; i < M; ++i) for (j = 0; j < N; ++j) { /* the sample's gold: naive loop, max absolute error */
    double sum = 0;
    for (p = 0; p < K; ++p) sum += (double)a[i * K + p] * (double)b[p * N + j];
    if (fabs(sum - (double)c[i * N + j]) > maxerr) maxerr = fabs(sum - (double)c[i * N + j]);
  }
  libxsmm_free(a); libxsmm_free(b); libxsmm_free(c);
  libxsmm_finalize();
  printf("spmdm_caller: max error %g\n", maxerr);
  return maxerr <= 1e-4 ? 0 : 2;
}
