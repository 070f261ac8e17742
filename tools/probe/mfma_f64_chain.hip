// Probe: is v_mfma_f64_16x16x4_f64 bit-for-bit the k-ordered fma chain D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,C))))?
// (the guide states this for the f32-input MFMA forms; for f64 it decides whether the dense f64 kernels may use the matrix
// cores without giving up bit-exact parity). Build: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f64_chain.hip -o mfma_f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, const double* C, double* D, int ksteps)
{ // A: 16 x K (row i, col k), B: K x 16, C/D: 16 x 16 row-major; K = 4 * ksteps
  const int l = threadIdx.x, i = l & 15, q = l >> 4, K = 4 * ksteps;
  d4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[(q + 4 * r) * 16 + i]; // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
  for (int s = 0; s < ksteps; ++s) {
    const double a = A[i * K + 4 * s + q];   // A[row = lane & 15][k = lane >> 4]
    const double b = B[(4 * s + q) * 16 + i]; // B[k = lane >> 4][col = lane & 15]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(q + 4 * r) * 16 + i] = acc[r];
}

int main()
{
  const int ksteps = 8, K = 4 * ksteps, trials = 200;
  std::vector<double> A(16 * K), B(K * 16), C(256), D(256), G(256);
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
  srand48(1);
  long long mismatches = 0, fused_only = 0;
  for (int t = 0; t < trials; ++t) {
    const double scale = std::ldexp(1.0, (t % 7) * 10 - 30); // vary magnitudes so that cancellation and rounding cases occur
    for (auto& v : A) v = (drand48() - 0.5) * scale;
    for (auto& v : B) v = drand48() - 0.5;
    for (auto& v : C) v = (drand48() - 0.5) * scale;
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, ksteps);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double acc = C[i * 16 + j], alt = C[i * 16 + j];
      for (int k = 0; k < K; ++k) { acc = std::fma(A[i * K + k], B[k * 16 + j], acc); alt = alt + A[i * K + k] * B[k * 16 + j]; }
      if (0 != std::memcmp(&acc, &D[i * 16 + j], 8)) ++mismatches;
      if (0 != std::memcmp(&acc, &alt, 8)) ++fused_only;
    }
  }
  std::printf("v_mfma_f64_16x16x4_f64 vs k-ordered fma chain: %lld mismatching elements of %d (the chain differs from mul+add in %lld)\n",
              mismatches, trials * 256, fused_only);
  return 0 == mismatches ? 0 : 1;
}
