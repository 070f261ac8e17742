// tools/stream_probe.hip -- developer microbenchmark: what does MI355X HBM sustain for the SMM batch's traffic mix
// (three read streams + one write stream, each item 4 KiB contiguous per stream)? Not part of the product.
// build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (hipSuccess != e_) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// grid-stride, one float4 per thread per stream
template<bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_flat(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, long long n4)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 x, y, z;
    if (NTL) { x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); z = __builtin_nontemporal_load(c + i); }
    else { x = a[i]; y = b[i]; z = c[i]; }
    const f32x4 r = x + y + z;
    if (NTS) __builtin_nontemporal_store(r, c + i); else c[i] = r;
  }
}

// wave-per-item: every wave moves whole 4 KiB items (4 float4 per lane per stream), like the SMM kernels, prefetching the next item
template<bool NTL, bool NTS, int WRITE>
__global__ __launch_bounds__(256) void k_item(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, long long items)
{
  const int lane = threadIdx.x & 63;
  const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, W = ((long long)gridDim.x * blockDim.x) >> 6;
  f32x4 ra[4], rb[4], rc[4];
  if (w >= items) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long o = w * 256 + 64 * j + lane;
    ra[j] = NTL ? __builtin_nontemporal_load(a + o) : a[o]; rb[j] = NTL ? __builtin_nontemporal_load(b + o) : b[o]; rc[j] = NTL ? __builtin_nontemporal_load(c + o) : c[o];
  }
  for (long long it = w; it < items; it += W) {
    f32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = ra[j] + rb[j] + rc[j];
    const long long nx = it + W;
    if (nx < items) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long o = nx * 256 + 64 * j + lane;
        ra[j] = NTL ? __builtin_nontemporal_load(a + o) : a[o]; rb[j] = NTL ? __builtin_nontemporal_load(b + o) : b[o]; rc[j] = NTL ? __builtin_nontemporal_load(c + o) : c[o];
      }
    }
    if (WRITE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long o = it * 256 + 64 * j + lane;
        if (NTS) __builtin_nontemporal_store(r[j], c + o); else c[o] = r[j];
      }
    }
    else if (r[0][0] == 123.456f) c[it] = r[0]; // keep the loads alive
  }
}

// the same, CH consecutive items per wave and step (a wave's accesses are CH x 4 KiB contiguous per stream)
template<int CH>
__global__ __launch_bounds__(256) void k_item_ch(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, long long items)
{
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long groups = items / CH;
  f32x4 ra[CH][4], rb[CH][4], rc[CH][4];
  if (w >= groups) return;
#pragma unroll
  for (int q = 0; q < CH; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long o = (w * CH + q) * 256 + 64 * j + lane;
      ra[q][j] = __builtin_nontemporal_load(a + o); rb[q][j] = __builtin_nontemporal_load(b + o); rc[q][j] = __builtin_nontemporal_load(c + o);
    }
  for (long long it = w; it < groups; it += W) {
    f32x4 r[CH][4];
#pragma unroll
    for (int q = 0; q < CH; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) r[q][j] = ra[q][j] + rb[q][j] + rc[q][j];
    const long long nx = it + W;
    if (nx < groups) {
#pragma unroll
      for (int q = 0; q < CH; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long o = (nx * CH + q) * 256 + 64 * j + lane;
          ra[q][j] = __builtin_nontemporal_load(a + o); rb[q][j] = __builtin_nontemporal_load(b + o); rc[q][j] = __builtin_nontemporal_load(c + o);
        }
    }
#pragma unroll
    for (int q = 0; q < CH; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(r[q][j], c + (it * CH + q) * 256 + 64 * j + lane);
  }
}

// a wave per item, the batch cut into R regions that are walked side by side (R windows per stream instead of one): wave w works in
// region w % R
__global__ __launch_bounds__(256) void k_item_regions(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, long long items, int R)
{
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long per = items / R, region = w % R, first = w / R, step = W / R;
  if (first >= per) return;
  f32x4 ra[4], rb[4], rc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long o = (region * per + first) * 256 + 64 * j + lane;
    ra[j] = __builtin_nontemporal_load(a + o); rb[j] = __builtin_nontemporal_load(b + o); rc[j] = __builtin_nontemporal_load(c + o);
  }
  for (long long i = first; i < per; i += step) {
    f32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = ra[j] + rb[j] + rc[j];
    const long long nx = i + step;
    if (nx < per) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long o = (region * per + nx) * 256 + 64 * j + lane;
        ra[j] = __builtin_nontemporal_load(a + o); rb[j] = __builtin_nontemporal_load(b + o); rc[j] = __builtin_nontemporal_load(c + o);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(r[j], c + (region * per + i) * 256 + 64 * j + lane);
  }
}

__global__ void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ c, long long n4)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) c[i] = a[i];
}

template<typename F> float time_ms(F f, int reps)
{
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f(); f(); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv)
{
  const long long items = (argc > 1 ? atoll(argv[1]) : 1048576);
  const long long bytes = items * 4096, n4 = bytes / 16;
  f32x4 *a, *b, *c;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
  const double gb4 = 4.0 * bytes / 1e9, gb3 = 3.0 * bytes / 1e9, gb2 = 2.0 * bytes / 1e9;
  for (int bpc : {4, 8, 16}) {
    const unsigned grid = 256u * bpc;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, c, n4); }, 5);
    printf("copy        bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb2 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_flat<false, false>), dim3(grid), dim3(256), 0, 0, a, b, c, n4); }, 5);
    printf("flat        bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_flat<true, true>), dim3(grid), dim3(256), 0, 0, a, b, c, n4); }, 5);
    printf("flat nt     bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_flat<false, true>), dim3(grid), dim3(256), 0, 0, a, b, c, n4); }, 5);
    printf("flat nt-st  bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
  }
  for (int bpc : {2, 3, 4, 6, 8}) {
    const unsigned grid = 256u * bpc;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL((k_item<false, false, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item        bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_item<true, true, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item nt     bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_item<false, true, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item nt-st  bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_item<false, false, 0>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item rd-only bpc=%2d  %.3f ms  %.0f GB/s (3 reads)\n", bpc, t, gb3 / t * 1e3);
  }
  for (int bpc : {1, 2, 3, 4}) {
    const unsigned grid = 256u * bpc;
    float t;
    t = time_ms([&] { hipLaunchKernelGGL((k_item_ch<1>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item nt x1  bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_item_ch<2>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item nt x2  bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
    t = time_ms([&] { hipLaunchKernelGGL((k_item_ch<4>), dim3(grid), dim3(256), 0, 0, a, b, c, items); }, 5);
    printf("item nt x4  bpc=%2d  %.3f ms  %.0f GB/s\n", bpc, t, gb4 / t * 1e3);
  }
  for (int R : {1, 2, 4, 8, 16, 32, 64, 128, 256, 768}) {
    const unsigned grid = 256u * 3;
    const float t = time_ms([&] { hipLaunchKernelGGL(k_item_regions, dim3(grid), dim3(256), 0, 0, a, b, c, items, R); }, 5);
    printf("item nt, %3d regions  bpc= 3  %.3f ms  %.0f GB/s\n", R, t, gb4 / t * 1e3);
  }
  return 0;
}
