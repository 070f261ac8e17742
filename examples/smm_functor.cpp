/* C++ caller as in the reference's samples/smm/specialized.cpp (:131-190): libxsmm_mmfunction<T> constructed for a shape,
 * called per item (streamed operands from libxsmm_aligned_malloc), compared with a plain loop; plus the
 * LIBXSMM_MMCALL macro on the functor's kernel. Reference API only.
 * Build: g++ -I include examples/smm_functor.cpp -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib */
#include <libxsmm.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

template<typename T> static int run(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, int s, double tol)
{
  const size_t asz = (size_t)m * k, bsz = (size_t)k * n, csz = (size_t)m * n;
  T* const a = static_cast<T*>(libxsmm_aligned_malloc(sizeof(T) * asz * s, 64));
  T* const b = static_cast<T*>(libxsmm_aligned_malloc(sizeof(T) * bsz * s, 64));
  T* const c = static_cast<T*>(libxsmm_aligned_malloc(sizeof(T) * csz * s, 64));
  std::vector<double> gold(csz * s);
  if (0 == a || 0 == b || 0 == c) return 100;
  for (int i = 0; i < s; ++i) {
    LIBXSMM_MATINIT(T, 42 + i, a + i * asz, m, k, m, 1.0 / s);
    LIBXSMM_MATINIT(T, 24 + i, b + i * bsz, k, n, k, 1.0 / s);
    LIBXSMM_MATINIT(T, 22 + i, c + i * csz, m, n, m, 1.0 / s);
  }
  for (size_t i = 0; i < csz * s; ++i) gold[i] = c[i];
  for (int i = 0; i < s; ++i) for (int j = 0; j < n; ++j) for (int p = 0; p < k; ++p) for (int r = 0; r < m; ++r) {
    gold[i * csz + j * m + r] += (double)a[i * asz + p * m + r] * (double)b[i * bsz + j * k + p];
  }
  const libxsmm_mmfunction<T> xmm(LIBXSMM_GEMM_FLAG_NONE, m, n, k, (T)1, (T)1);
  if (!xmm) { std::fprintf(stderr, "no kernel for %dx%dx%d\n", (int)m, (int)n, (int)k); return 1; }
  for (int i = 0; i < s; ++i) {
    if (i & 1) xmm(a + i * asz, b + i * bsz, c + i * csz);
    else if (sizeof(T) == 8) LIBXSMM_MMCALL(xmm.kernel().dmm, (const double*)(a + i * asz), (const double*)(b + i * bsz), (double*)(c + i * csz), m, n, k);
    else LIBXSMM_MMCALL(xmm.kernel().smm, (const float*)(a + i * asz), (const float*)(b + i * bsz), (float*)(c + i * csz), m, n, k);
  }
  double d = 0, scale = 0;
  for (size_t i = 0; i < csz * s; ++i) { d = std::fmax(d, std::fabs((double)c[i] - gold[i])); scale = std::fmax(scale, std::fabs(gold[i])); }
  libxsmm_free(a); libxsmm_free(b); libxsmm_free(c);
  return d <= tol * scale ? 0 : 2;
}

int main()
{
  libxsmm_init();
  const libxsmm_mmfunction<double> unsupported(LIBXSMM_GEMM_FLAG_NONE, 23, 23, 23, 2.0, 1.0); // alpha != 1: no kernel, as in the reference
  int result = unsupported ? 50 : 0;
  result |= run<double>(23, 23, 23, 64, 1e-12);
  result |= run<float>(32, 32, 32, 64, 1e-5);
  libxsmm_finalize();
  if (0 == result) std::printf("smm_functor: libxsmm_mmfunction<double|float> agree with the plain loops\n");
  return result;
}
