// smm_common.cuh -- device-side addressing shared by the dense SMM kernels (gfx950).
#ifndef XSMM_SMM_COMMON_CUH
#define XSMM_SMM_COMMON_CUH

#include <hip/hip_runtime.h>
#include "../xsmm_internal.hpp"

namespace xsmm {

// Trivially-copyable view of the batch addressing (libxsmm_mmbatch_kernel's three modes,
// reference src/libxsmm_gemm.c:1333-1364 index arrays, :1426-1461 pointer arrays) plus constant strides.
struct DevAddr {
  const char* a; const char* b; char* c;
  const char* ia; const char* ib; const char* ic;
  long long sa, sb, sc;
  int index_base, index_stride, mode;
  const int* flags; // SYNC_DEVICE*: {#equal neighbouring C, #out-of-order repeats of C} from c_order_kernel, else null
};

inline DevAddr make_addr(const SmmBatch& s)
{
  DevAddr d;
  d.a = (const char*)s.a; d.b = (const char*)s.b; d.c = (char*)s.c;
  d.ia = (const char*)s.ia; d.ib = (const char*)s.ib; d.ic = (const char*)s.ic;
  d.sa = s.sa; d.sb = s.sb; d.sc = s.sc;
  d.index_base = s.index_base; d.index_stride = s.index_stride; d.mode = s.mode;
  d.flags = (SYNC_DEVICE == s.sync ? s.devflags : nullptr);
  return d;
}

template<typename T>
__device__ __forceinline__ T* resolve(const char* base, const char* idx, long long stride, const DevAddr& ad, long long i)
{
  if (ADDR_STRIDED == ad.mode) {
    return (T*)base + i * stride;
  }
  else if (ADDR_INDEX == ad.mode) {
    if (nullptr == idx) return (T*)base;
    const int v = *(const int*)(idx + i * (long long)ad.index_stride);
    return (T*)base + ((long long)v - ad.index_base);
  }
  return *(T* const*)(base + i * stride); // ADDR_POINTER
}

// The same in two steps, for kernels that look the addresses of an item up one iteration before they request its operands:
// raw_of issues the load an index / pointer batch needs (nothing depends on it yet), cooked turns what arrived into the address.
// Done in one step (resolve), the index load is a full memory round trip in front of every operand request.
__device__ __forceinline__ long long raw_of(const char* base, const char* idx, long long stride, const DevAddr& ad, long long i)
{
  if (ADDR_STRIDED == ad.mode) return i * stride;
  if (ADDR_INDEX == ad.mode) return (nullptr == idx) ? (long long)ad.index_base : (long long)*(const int*)(idx + i * (long long)ad.index_stride);
  return *(const long long*)(base + i * stride); // ADDR_POINTER
}
template<typename T>
__device__ __forceinline__ T* cooked(const char* base, const DevAddr& ad, long long raw)
{
  if (ADDR_STRIDED == ad.mode) return (T*)base + raw;
  if (ADDR_INDEX == ad.mode) return (T*)base + (raw - ad.index_base);
  return (T*)raw;
}

template<typename T> __device__ __forceinline__ const T* addr_a(const DevAddr& ad, long long i) { return resolve<const T>(ad.a, ad.ia, ad.sa, ad, i); }
template<typename T> __device__ __forceinline__ const T* addr_b(const DevAddr& ad, long long i) { return resolve<const T>(ad.b, ad.ib, ad.sb, ad, i); }
template<typename T> __device__ __forceinline__ T* addr_c(const DevAddr& ad, long long i) { return resolve<T>(ad.c, ad.ic, ad.sc, ad, i); }

__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// wave-private LDS hand-off: LDS operations of one wave execute in order, so only the compiler has to be
// kept from moving reads above writes.
__device__ __forceinline__ void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace xsmm

#endif
