/* Low-precision kernels called the way the reference's harness samples/xgemm/kernel.c does (:1055-1123 bf16 -> f32,
 * :870-935 i16 -> i32, :962-1030 i16 -> f32 with a scaling factor): descriptor via libxsmm_gemm_descriptor_dinit2, kernel
 * via libxsmm_xmmdispatch, call kernel(a, b, c, NULL, NULL, NULL[, &scf]); A is stored in pairs of k. Checked against the
 * harness' gold loops, bit for bit. Reference API only.
 * Build: gcc -I include examples/lowp_caller.c -L libxsmm-1_amd/lib -lxsmm -lm */
#include <libxsmm.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void)
{
  const libxsmm_blasint m = 32, n = 24, k = 48, lda = 32, ldb = 48, ldc = 32;
  const int kb = 2;
  libxsmm_descriptor_blob blob;
  const libxsmm_gemm_descriptor* desc;
  libxsmm_xmmfunction kernel;
  libxsmm_bfloat16 *a_bf, *b_bf; short *a_w, *b_w;
  float *c_f, *g_f, scf = 0.0625f; int *c_i, *g_i;
  int i, j, s, k2, result = 0;
  libxsmm_init();
  a_bf = (libxsmm_bfloat16*)libxsmm_aligned_malloc(sizeof(libxsmm_bfloat16) * lda * k, 64);
  b_bf = (libxsmm_bfloat16*)libxsmm_aligned_malloc(sizeof(libxsmm_bfloat16) * ldb * n, 64);
  a_w = (short*)libxsmm_aligned_malloc(sizeof(short) * lda * k, 64); b_w = (short*)libxsmm_aligned_malloc(sizeof(short) * ldb * n, 64);
  c_f = (float*)libxsmm_aligned_malloc(sizeof(float) * ldc * n, 64); g_f = (float*)malloc(sizeof(float) * ldc * n);
  c_i = (int*)libxsmm_aligned_malloc(sizeof(int) * ldc * n, 64); g_i = (int*)malloc(sizeof(int) * ldc * n);
  libxsmm_rng_set_seed(1);
  for (i = 0; i < lda * k; ++i) { libxsmm_bfloat16_hp t; t.f = (float)libxsmm_rng_f64(); a_bf[i] = t.i[1]; a_w[i] = (short)(libxsmm_rng_f64() * 200.0 - 100.0); }
  for (i = 0; i < ldb * n; ++i) { libxsmm_bfloat16_hp t; t.f = (float)libxsmm_rng_f64(); b_bf[i] = t.i[1]; b_w[i] = (short)(libxsmm_rng_f64() * 200.0 - 100.0); }

  /* bf16 -> f32 */
  for (i = 0; i < ldc * n; ++i) c_f[i] = g_f[i] = 0.f;
  desc = libxsmm_gemm_descriptor_dinit2(&blob, LIBXSMM_GEMM_PRECISION_BF16, LIBXSMM_GEMM_PRECISION_F32, m, n, k, lda, ldb, ldc, 1.0, 1.0, LIBXSMM_GEMM_FLAG_NONE, LIBXSMM_GEMM_PREFETCH_NONE);
  kernel = libxsmm_xmmdispatch(desc);
  if (NULL == kernel.bsmm) { fprintf(stderr, "no bf16 kernel\n"); return 10; }
  kernel.bsmm(a_bf, b_bf, c_f, NULL, NULL, NULL);
  for (j = 0; j < n; ++j) for (s = 0; s < k / kb; ++s) for (i = 0; i < m; ++i) for (k2 = 0; k2 < kb; ++k2) {
    libxsmm_bfloat16_hp ta, tb;
    ta.i[1] = a_bf[(s * (lda * kb)) + (i * kb) + k2]; ta.i[0] = 0;
    tb.i[1] = b_bf[(j * ldb) + (s * kb) + k2]; tb.i[0] = 0;
    { volatile float prod = ta.f * tb.f; volatile float sum = g_f[(j * ldc) + i] + prod; g_f[(j * ldc) + i] = sum; }
  }
  if (0 != memcmp(c_f, g_f, sizeof(float) * ldc * n)) { fprintf(stderr, "bf16 -> f32 differs from the gold loop\n"); result |= 1; }

  /* i16 -> i32 */
  for (i = 0; i < ldc * n; ++i) c_i[i] = g_i[i] = 0;
  desc = libxsmm_gemm_descriptor_dinit2(&blob, LIBXSMM_GEMM_PRECISION_I16, LIBXSMM_GEMM_PRECISION_I32, m, n, k, lda, ldb, ldc, 1.0, 1.0, LIBXSMM_GEMM_FLAG_NONE, LIBXSMM_GEMM_PREFETCH_NONE);
  kernel = libxsmm_xmmdispatch(desc);
  if (NULL == kernel.wimm) { fprintf(stderr, "no i16 kernel\n"); return 11; }
  kernel.wimm(a_w, b_w, c_i, NULL, NULL, NULL);
  for (j = 0; j < n; ++j) for (s = 0; s < k / kb; ++s) for (i = 0; i < m; ++i) for (k2 = 0; k2 < kb; ++k2) {
    g_i[(j * ldc) + i] += a_w[(s * (lda * kb)) + (i * kb) + k2] * b_w[(j * ldb) + (s * kb) + k2];
  }
  if (0 != memcmp(c_i, g_i, sizeof(int) * ldc * n)) { fprintf(stderr, "i16 -> i32 differs from the gold loop\n"); result |= 2; }

  /* i16 -> f32, scaled: the 7th argument points to the scaling factor */
  for (i = 0; i < ldc * n; ++i) c_f[i] = g_f[i] = 0.f;
  desc = libxsmm_gemm_descriptor_dinit2(&blob, LIBXSMM_GEMM_PRECISION_I16, LIBXSMM_GEMM_PRECISION_F32, m, n, k, lda, ldb, ldc, 1.0, 1.0, LIBXSMM_GEMM_FLAG_NONE, LIBXSMM_GEMM_PREFETCH_NONE);
  kernel = libxsmm_xmmdispatch(desc);
  if (NULL == kernel.wsmm) { fprintf(stderr, "no i16 -> f32 kernel\n"); return 12; }
  kernel.wsmm(a_w, b_w, c_f, NULL, NULL, NULL, &scf);
  for (j = 0; j < n; ++j) for (s = 0; s < k / kb; ++s) for (i = 0; i < m; ++i) for (k2 = 0; k2 < kb; ++k2) {
    const int iprod = (int)a_w[(s * (lda * kb)) + (i * kb) + k2] * (int)b_w[(j * ldb) + (s * kb) + k2];
    volatile float fprod = (float)iprod; volatile float scaled = fprod * scf; volatile float sum = g_f[(j * ldc) + i] + scaled;
    g_f[(j * ldc) + i] = sum;
  }
  if (0 != memcmp(c_f, g_f, sizeof(float) * ldc * n)) { fprintf(stderr, "i16 -> f32 differs from the gold loop\n"); result |= 4; }

  libxsmm_free(a_bf); libxsmm_free(b_bf); libxsmm_free(a_w); libxsmm_free(b_w); libxsmm_free(c_f); libxsmm_free(c_i); free(g_f); free(g_i);
  libxsmm_finalize();
  if (0 == result) printf("lowp_caller: bf16 -> f32, i16 -> i32 and i16 -> f32 kernels match the gold loops bit for bit\n");
  return result;
}
