// smm_lowp.hip -- low-precision dense SMM kernels for gfx950: i16 -> i32, i16 -> f32 (scaled), bf16 -> f32, bf16 -> bf16.
//
// Replaces the kernels behind libxsmm_wimmdispatch / wsmmdispatch / bsmmdispatch / bmmdispatch (reference
// src/libxsmm_main.c:2198-2259; generator constraints src/generator_gemm.c:121-147: k even, no TRANS_B, bf16 output needs
// m % 16 == 0). Operand layout and arithmetic are those of the gold loops the reference's harness checks these kernels
// with (samples/xgemm/kernel.c:915-927, :1007-1021, :1104-1123, :1207-1229): A in pairs of k
// (a[(k/2)*lda*2 + m*2 + k%2]), B and C column-major, per C element the terms in ascending k; the int sum wraps, the
// float forms round the product (exact for bf16 x bf16) and the add separately, a bf16 result is the upper half of the
// float sum.
//
// One work-group per item, walking the batch with the stride of the grid. A and B of an item are parked in LDS as 32-bit
// k pairs (A: [k/2][m], lanes along m: conflict-free; B: [n][k/2], one pair broadcast to the lanes of a column); a thread
// owns the C elements e, e + 256, ... of the column-major tile. The path is HBM-bound at 2 bytes per input element like
// its fp32 sibling; it is parity-tested, not tuned.
#include "smm_common.cuh"

namespace xsmm {

namespace {

__device__ __forceinline__ float bf16_to_f32(unsigned v) { return __uint_as_float(v << 16); }

template<int KIND> struct LowpOut;
template<> struct LowpOut<1> { typedef int type; };            // i16 -> i32
template<> struct LowpOut<2> { typedef float type; };          // i16 -> f32
template<> struct LowpOut<3> { typedef float type; };          // bf16 -> f32
template<> struct LowpOut<4> { typedef unsigned short type; }; // bf16 -> bf16

template<int KIND, bool STAGED>
__global__ __launch_bounds__(256) void smm_lowp_kernel(DevAddr ad, int m, int n, int k, int lda, int ldb, int ldc, int beta0,
                                                       float scf, long long batch)
{
  typedef typename LowpOut<KIND>::type TC;
  extern __shared__ __align__(16) unsigned lowp_lds[];
  const int kh = k >> 1, t = threadIdx.x;
  unsigned* const As = lowp_lds;            // [kh][m]
  unsigned* const Bs = lowp_lds + kh * m;   // [n][kh]
  for (long long item = blockIdx.x; item < batch; item += gridDim.x) {
    const unsigned short* const a = addr_a<unsigned short>(ad, item);
    const unsigned short* const b = addr_b<unsigned short>(ad, item);
    TC* const c = addr_c<TC>(ad, item);
    if (STAGED) {
      __syncthreads(); // the previous item's readers are done
      for (int e = t; e < kh * m; e += 256) {
        const int s = e / m, i = e - s * m;
        const unsigned short* const p = a + ((size_t)s * lda + i) * 2;
        As[e] = (unsigned)p[0] | ((unsigned)p[1] << 16);
      }
      for (int e = t; e < n * kh; e += 256) {
        const int j = e / kh, s = e - j * kh;
        const unsigned short* const p = b + (size_t)j * ldb + 2 * s;
        Bs[e] = (unsigned)p[0] | ((unsigned)p[1] << 16);
      }
      __syncthreads();
    }
    for (int e = t; e < m * n; e += 256) {
      const int j = e / m, i = e - j * m;
      TC* const pc = c + (size_t)j * ldc + i;
      unsigned iacc = 0; float facc = 0.f;
      if (0 == beta0) {
        if (1 == KIND) iacc = (unsigned)*reinterpret_cast<const int*>(pc);
        else if (4 == KIND) facc = bf16_to_f32(*reinterpret_cast<const unsigned short*>(pc));
        else facc = *reinterpret_cast<const float*>(pc);
      }
      for (int s = 0; s < kh; ++s) {
        unsigned pa, pb;
        if (STAGED) { pa = As[s * m + i]; pb = Bs[j * kh + s]; }
        else {
          const unsigned short* const qa = a + ((size_t)s * lda + i) * 2;
          const unsigned short* const qb = b + (size_t)j * ldb + 2 * s;
          pa = (unsigned)qa[0] | ((unsigned)qa[1] << 16); pb = (unsigned)qb[0] | ((unsigned)qb[1] << 16);
        }
        if (KIND <= 2) {
          const int p0 = (int)(short)(pa & 0xFFFFu) * (int)(short)(pb & 0xFFFFu), p1 = (int)(short)(pa >> 16) * (int)(short)(pb >> 16);
          if (1 == KIND) { iacc += (unsigned)p0; iacc += (unsigned)p1; }
          else {
            facc = __fadd_rn(facc, __fmul_rn((float)p0, scf));
            facc = __fadd_rn(facc, __fmul_rn((float)p1, scf));
          }
        }
        else {
          facc = __fadd_rn(facc, __fmul_rn(bf16_to_f32(pa & 0xFFFFu), bf16_to_f32(pb & 0xFFFFu)));
          facc = __fadd_rn(facc, __fmul_rn(bf16_to_f32(pa >> 16), bf16_to_f32(pb >> 16)));
        }
      }
      if (1 == KIND) *reinterpret_cast<int*>(pc) = (int)iacc;
      else if (4 == KIND) *reinterpret_cast<unsigned short*>(pc) = (unsigned short)(__float_as_uint(facc) >> 16);
      else *reinterpret_cast<float*>(pc) = facc;
    }
  }
}

// Batch-reduce form of the bf16 kernels (libxsmm_bsmmdispatch_reducebatch / libxsmm_bmmdispatch_reducebatch,
// src/libxsmm_main.c:2290-2315): C (+)= sum_i A_i * B_i over `count` products given as arrays of pointers. A thread owns one
// element of C and walks the products in batch order, k ascending inside a product, every add rounded to fp32 -- the chain of
// `count` consecutive calls of the plain bf16 -> f32 kernel; a bf16 C is widened once at the start and truncated once at the end
// (the reference's kernel keeps the sums in fp32 registers across the batch).
template<int KIND>
__global__ __launch_bounds__(256) void smm_lowp_reduce_kernel(DevAddr ad, int m, int n, int k, int lda, int ldb, int ldc, int beta0, long long count)
{
  typedef typename LowpOut<KIND>::type TC;
  const int kh = k >> 1;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)m * n) return;
  const int j = (int)(e / m), i = (int)(e - (long long)j * m);
  TC* const pc = reinterpret_cast<TC*>(ad.c) + (size_t)j * ldc + i;
  float facc = 0.f;
  if (0 == beta0) facc = (4 == KIND) ? bf16_to_f32(*reinterpret_cast<const unsigned short*>(pc)) : *reinterpret_cast<const float*>(pc);
  for (long long item = 0; item < count; ++item) {
    const unsigned short* const a = addr_a<unsigned short>(ad, item);
    const unsigned short* const b = addr_b<unsigned short>(ad, item);
    for (int s = 0; s < kh; ++s) {
      const unsigned short* const qa = a + ((size_t)s * lda + i) * 2;
      const unsigned short* const qb = b + (size_t)j * ldb + 2 * s;
      facc = __fadd_rn(facc, __fmul_rn(bf16_to_f32(qa[0]), bf16_to_f32(qb[0])));
      facc = __fadd_rn(facc, __fmul_rn(bf16_to_f32(qa[1]), bf16_to_f32(qb[1])));
    }
  }
  if (4 == KIND) *reinterpret_cast<unsigned short*>(pc) = (unsigned short)(__float_as_uint(facc) >> 16);
  else *reinterpret_cast<float*>(pc) = facc;
}

template<int KIND>
int launch_kind(const SmmBatch& s, hipStream_t st)
{
  const size_t lds = ((size_t)(s.k / 2) * s.m + (size_t)s.n * (s.k / 2)) * sizeof(unsigned);
  long long blocks = s.batch;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  const DevAddr ad = make_addr(s);
  if (lds <= 48 * 1024) {
    hipLaunchKernelGGL((smm_lowp_kernel<KIND, true>), dim3((unsigned)blocks), dim3(256), lds, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
                       (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? 1 : 0, s.scf, s.batch);
  }
  else {
    hipLaunchKernelGGL((smm_lowp_kernel<KIND, false>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc,
                       (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? 1 : 0, s.scf, s.batch);
  }
  return (int)hipGetLastError();
}

} // namespace

int launch_smm_lowp_reduce(const SmmBatch& s, void* stream, const char** name)
{ // s.a / s.b: device arrays of pointers (stride s.sa / s.sb bytes), s.c: the one C, s.batch: number of products
  hipStream_t st = (hipStream_t)stream;
  if (0 != (s.k & 1) || s.m <= 0 || s.n <= 0 || s.k <= 0 || (3 != s.lowp && 4 != s.lowp) || ADDR_POINTER != s.mode) return (int)hipErrorInvalidValue;
  DevAddr ad = make_addr(s);
  ad.c = static_cast<char*>(const_cast<void*>(static_cast<const void*>(s.c)));
  const unsigned blocks = (unsigned)(((long long)s.m * s.n + 255) / 256);
  const int beta0 = (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? 1 : 0;
  if (3 == s.lowp) { *name = "smm_bf16f32_reduce_lowp"; hipLaunchKernelGGL((smm_lowp_reduce_kernel<3>), dim3(blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, beta0, s.batch); }
  else { *name = "smm_bf16_reduce_lowp"; hipLaunchKernelGGL((smm_lowp_reduce_kernel<4>), dim3(blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, beta0, s.batch); }
  return (int)hipGetLastError();
}

int launch_smm_lowp(const SmmBatch& s, void* stream, const char** name)
{
  hipStream_t st = (hipStream_t)stream;
  if (0 != (s.k & 1) || s.m <= 0 || s.n <= 0 || s.k <= 0) return (int)hipErrorInvalidValue;
  switch (s.lowp) {
    case 1: *name = "smm_i16i32_lowp"; return launch_kind<1>(s, st);
    case 2: *name = "smm_i16f32_lowp"; return launch_kind<2>(s, st);
    case 3: *name = "smm_bf16f32_lowp"; return launch_kind<3>(s, st);
    case 4: *name = "smm_bf16_lowp"; return launch_kind<4>(s, st);
    default: return (int)hipErrorInvalidValue;
  }
}

} // namespace xsmm
