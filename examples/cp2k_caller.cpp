/* C++ caller following the flow of the reference's samples/cp2k/cp2k.cpp (:214-330): streams of A and B tiles from
 * libxsmm_malloc, each chunk of u products accumulated into a private tile on the caller's stack (plain host memory, so
 * one call mixes device-visible and pageable operands), the tile then added into its C tile; once through the
 * dispatched functor, once through the libxsmm_gemm<T> overload, once through libxsmm_blas_gemm<T>; all three are
 * compared with a plain loop. Reference API only.
 * Build: g++ -I include examples/cp2k_caller.cpp -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib */
#include <libxsmm.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#define MAX_SIZE (32 * 32)

template<typename T> static void add(T* dst, const T* src, int m, int n, int ldc)
{
  for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) dst[j * ldc + i] += src[j * ldc + i];
}

template<typename T> static int run(int m, int n, int k, int s, int u, double tol)
{
  const int ldc = m, csize = (s + u - 1) / u;
  const size_t asz = (size_t)m * k, bsz = (size_t)k * n, csz = (size_t)ldc * n;
  T* const a = static_cast<T*>(libxsmm_malloc(sizeof(T) * asz * s));
  T* const b = static_cast<T*>(libxsmm_malloc(sizeof(T) * bsz * s));
  std::vector<T> c(csz * csize, (T)0), c2(csz * csize, (T)0), c3(csz * csize, (T)0);
  std::vector<double> gold(csz * csize, 0.0);
  if (0 == a || 0 == b || (size_t)MAX_SIZE < csz) return 100;
  for (int i = 0; i < s; ++i) {
    LIBXSMM_MATINIT(T, 42 + i, a + i * asz, m, k, m, 1.0 / s);
    LIBXSMM_MATINIT(T, 24 + i, b + i * bsz, k, n, k, 1.0 / s);
  }
  for (int i = 0; i < s; ++i) for (int j = 0; j < n; ++j) for (int p = 0; p < k; ++p) for (int r = 0; r < m; ++r) {
    gold[(size_t)(i / u) * csz + j * ldc + r] += (double)a[i * asz + p * m + r] * (double)b[i * bsz + j * k + p];
  }
  const libxsmm_mmfunction<T> xmm(LIBXSMM_GEMM_FLAG_NONE, m, n, k, (T)1, (T)1);
  if (!xmm) return 1;
  const T one = (T)1;
  const libxsmm_blasint bm = m, bn = n, bk = k;
  for (int i = 0; i < s; i += u) {
    T tmp[MAX_SIZE], tmp2[MAX_SIZE], tmp3[MAX_SIZE];
    std::memset(tmp, 0, sizeof(tmp)); std::memset(tmp2, 0, sizeof(tmp2)); std::memset(tmp3, 0, sizeof(tmp3));
    for (int j = i; j < i + u && j < s; ++j) {
      xmm(a + j * asz, b + j * bsz, tmp);
      libxsmm_gemm(0/*transa*/, 0/*transb*/, bm, bn, bk, &one, a + j * asz, 0/*lda*/, b + j * bsz, 0/*ldb*/, &one, tmp2, 0/*ldc*/);
      libxsmm_blas_gemm(0, 0, bm, bn, bk, &one, a + j * asz, 0, b + j * bsz, 0, &one, tmp3, 0);
    }
    add(&c[(size_t)(i / u) * csz], tmp, m, n, ldc);
    add(&c2[(size_t)(i / u) * csz], tmp2, m, n, ldc);
    add(&c3[(size_t)(i / u) * csz], tmp3, m, n, ldc);
  }
  double d = 0, scale = 0;
  for (size_t i = 0; i < csz * csize; ++i) {
    d = std::fmax(d, std::fabs((double)c[i] - gold[i]));
    d = std::fmax(d, std::fabs((double)c2[i] - gold[i]));
    d = std::fmax(d, std::fabs((double)c3[i] - gold[i]));
    scale = std::fmax(scale, std::fabs(gold[i]));
  }
  libxsmm_free(a); libxsmm_free(b);
  if (!(d <= tol * scale)) std::fprintf(stderr, "%dx%dx%d: diff %g vs scale %g\n", m, n, k, d, scale);
  return d <= tol * scale ? 0 : 2;
}

int main()
{
  libxsmm_init();
  int result = run<double>(23, 23, 23, 96, 8, 1e-12);
  result |= run<float>(13, 5, 7, 60, 7, 1e-5);
  libxsmm_finalize();
  if (0 == result) std::printf("cp2k_caller: functor, libxsmm_gemm and libxsmm_blas_gemm agree with the plain loops\n");
  return result;
}
