// Cost of handing one lane's value to a whole wave (gfx950): v_readlane_b32 with an immediate / SGPR lane select, with the
// result consumed by a VALU instruction, against a broadcast ds_read_b64 and ds_bpermute_b32. Cycles per operation from
// s_memtime, 1..4 waves per SIMD. hipcc --offload-arch=gfx950 -O3 tools/probe/readlane_cost.hip -o tools/probe/readlane_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template<int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, long long* cycles, int iters)
{
  __shared__ float2 lds[256];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x & 255] = float2{ (float)lane, 1.f };
  __syncthreads();
  float x = (float)lane, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int xi = lane;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { // immediate lane select
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
        acc0 = __builtin_fmaf(s, x, acc0); acc1 = __builtin_fmaf(s, x, acc1); acc2 = __builtin_fmaf(s, x, acc2); acc3 = __builtin_fmaf(s, x, acc3);
      }
    }
    else if (MODE == 1) { // SGPR lane select
      const int base = __builtin_amdgcn_readfirstlane(it & 15);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), base + j));
        acc0 = __builtin_fmaf(s, x, acc0); acc1 = __builtin_fmaf(s, x, acc1); acc2 = __builtin_fmaf(s, x, acc2); acc3 = __builtin_fmaf(s, x, acc3);
      }
    }
    else if (MODE == 2) { // broadcast read from LDS
      const float2* p = lds + (it & 15);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 e = p[j];
        acc0 = __builtin_fmaf(e.x, x, acc0); acc1 = __builtin_fmaf(e.x, x, acc1); acc2 = __builtin_fmaf(e.y, x, acc2); acc3 = __builtin_fmaf(e.y, x, acc3);
      }
    }
    else if (MODE == 3) { // ds_bpermute
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float s = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * ((it + j) & 63), xi));
        acc0 = __builtin_fmaf(s, x, acc0); acc1 = __builtin_fmaf(s, x, acc1); acc2 = __builtin_fmaf(s, x, acc2); acc3 = __builtin_fmaf(s, x, acc3);
      }
    }
    else { // the four fma alone
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float s = 0.5f + j;
        acc0 = __builtin_fmaf(s, x, acc0); acc1 = __builtin_fmaf(s, x, acc1); acc2 = __builtin_fmaf(s, x, acc2); acc3 = __builtin_fmaf(s, x, acc3);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
  if (0 == lane) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main()
{
  const int iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 256 * 16 * sizeof(long long));
  const char* names[] = { "v_readlane imm + 4 fma", "v_readlane sgpr + 4 fma", "ds_read_b64 bcast + 4 fma", "ds_bpermute + 4 fma", "4 fma only" };
  for (int threads : { 256, 512, 1024 }) {
    for (int mode = 0; mode < 5; ++mode) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        default: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
      }
      hipDeviceSynchronize();
      const int nw = 256 * threads / 64;
      std::vector<long long> h(nw);
      hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
      double sum = 0; for (long long v : h) sum += (double)v;
      const double per_wave_op = sum / nw / (iters * 16.0);
      printf("%d waves/SIMD  %-28s %6.1f cycles per (op + 4 fma) per wave = %5.1f per SIMD\n", threads / 256, names[mode], per_wave_op, per_wave_op / (threads / 256));
    }
  }
  return 0;
}
