// smm_special.hip -- tuned batched SMM kernels for the shapes the engine is measured on (gfx950).
//
// smm32_*: M=N=K=32, fp32, tight leading dimensions (BASELINE config 2, reference samples/smm/specialized.cpp).
// One wavefront owns one problem at a time and walks the batch with a stride of all resident waves, so
// neighbouring waves stream neighbouring 4 KiB matrices. Per problem: A and B arrive as four 16-byte loads
// per lane (1 KiB per wave instruction), are parked in the wave's private 8 KiB of LDS, C is read and written
// directly. The loads of problem i+1 are issued before the arithmetic of problem i (the GPU analogue of the
// reference's prefetch chaining, src/libxsmm_gemm.c:1348).
//   "fma"  variant: 4x4 register tile per lane, v_fma_f32, k ascending -- bit-identical to the reference's
//                   per-element fma chain.
//   "mfma" variant: v_mfma_f32_32x32x2_f32; the two k of one instruction are (s, 16+s), i.e. the chain per C
//                   element runs k = 0,16,1,17,...; same products, different association (tolerance parity).
#include "smm_common.cuh"

namespace xsmm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bool aligned16(const void* p) { return 0 == (reinterpret_cast<uintptr_t>(p) & 15); }

// 4 x float4 per lane covering a tight 32x32 fp32 matrix: chunk index c = 64*j + lane (16-byte chunks)
__device__ __forceinline__ void load_mat32(const float* p, int lane, f32x4 (&r)[4])
{
  if (aligned16(p)) {
    const f32x4* const v = reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = v[64 * j + lane];
  }
  else { // operands that are only element-aligned (arbitrary index arrays)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* const q = p + 4 * (64 * j + lane);
      r[j] = f32x4{ q[0], q[1], q[2], q[3] };
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// scalar-FMA variant. Lane (tx = lane & 7, ty = lane >> 3) owns C rows 4tx..4tx+3 of columns 4ty..4ty+3.
// LDS image (floats): A linear [k][32]; B as 16-byte chunks (n, q = k/4) at position n*8 + (q ^ ty(n)), ty(n) = n>>2,
// so the eight distinct B rows a wave touches per read fall into different bank groups.
// ---------------------------------------------------------------------------------------------------------------
template<bool BETA0>
__global__ __launch_bounds__(256)
void smm32_f32_fma_kernel(DevAddr ad, long long batch)
{
  __shared__ __align__(16) float lds[4][2048];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx = lane & 7, ty = lane >> 3;
  float* const As = lds[wave];
  float* const Bs = lds[wave] + 1024;
  const long long w = (long long)blockIdx.x * 4 + wave, W = (long long)gridDim.x * 4;
  if (w >= batch) return;

  f32x4 ra[4], rb[4], rc[4];
  {
    load_mat32(addr_a<float>(ad, w), lane, ra);
    load_mat32(addr_b<float>(ad, w), lane, rb);
    if (!BETA0) {
      const float* const pc = addr_c<float>(ad, w);
      const bool al = aligned16(pc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* const q = pc + (4 * ty + j) * 32 + 4 * tx;
        rc[j] = al ? *reinterpret_cast<const f32x4*>(q) : f32x4{ q[0], q[1], q[2], q[3] };
      }
    }
  }
  for (long long item = w; item < batch; item += W) {
    float* const pc = addr_c<float>(ad, item);
    // park A and B in LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 64 * j + lane;
      *reinterpret_cast<f32x4*>(As + 4 * c) = ra[j];
      const int n = c >> 3, q = c & 7;
      *reinterpret_cast<f32x4*>(Bs + 4 * (n * 8 + (q ^ (n >> 2)))) = rb[j];
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = BETA0 ? f32x4{ 0.f, 0.f, 0.f, 0.f } : rc[j];
    // issue the next problem's loads before computing this one
    const long long next = item + W;
    if (next < batch) {
      load_mat32(addr_a<float>(ad, next), lane, ra);
      load_mat32(addr_b<float>(ad, next), lane, rb);
      if (!BETA0) {
        const float* const pn = addr_c<float>(ad, next);
        const bool al = aligned16(pn);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* const q = pn + (4 * ty + j) * 32 + 4 * tx;
          rc[j] = al ? *reinterpret_cast<const f32x4*>(q) : f32x4{ q[0], q[1], q[2], q[3] };
        }
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) { // four k per step
      f32x4 av[4], bv[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) av[kk] = *reinterpret_cast<const f32x4*>(As + (4 * q + kk) * 32 + 4 * tx);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(Bs + 4 * ((4 * ty + j) * 8 + (q ^ ty)));
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = xfma(av[kk][i], bv[j][kk], acc[j][i]);
        }
      }
    }
    const bool al = aligned16(pc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* const q = pc + (4 * ty + j) * 32 + 4 * tx;
      if (al) *reinterpret_cast<f32x4*>(q) = acc[j];
      else { q[0] = acc[j][0]; q[1] = acc[j][1]; q[2] = acc[j][2]; q[3] = acc[j][3]; }
    }
    wave_lds_sync();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA variant: D[i=n][j=m] += sum_kk Bt[n][kk] * A[kk][m] with v_mfma_f32_32x32x2_f32.
//   a-operand (lane l: row i = l&31, kk = l>>5)  <- B[k][n], n = l&31, k = 16*(l>>5) + s
//   b-operand (lane l: col j = l&31, kk = l>>5)  <- A[m][k], m = l&31, k = 16*(l>>5) + s
//   D register r of lane l: n = (r&3) + 8*(r>>2) + 4*(l>>5), m = l&31  -> every C access is two full 128-byte rows.
// LDS image: A linear; B chunks (n, q) at n*8 + (q ^ ((n>>1)&7)) so that the 16 lanes of a ds_read_b128 group hit
// 16 different 16-byte slots.
// ---------------------------------------------------------------------------------------------------------------
template<bool BETA0>
__global__ __launch_bounds__(256)
void smm32_f32_mfma_kernel(DevAddr ad, long long batch)
{
  __shared__ __align__(16) float lds[4][2048];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 31, hi = lane >> 5;
  float* const As = lds[wave];
  float* const Bs = lds[wave] + 1024;
  const long long w = (long long)blockIdx.x * 4 + wave, W = (long long)gridDim.x * 4;
  if (w >= batch) return;

  f32x4 ra[4], rb[4];
  float rc[16];
  load_mat32(addr_a<float>(ad, w), lane, ra);
  load_mat32(addr_b<float>(ad, w), lane, rb);
  if (!BETA0) {
    const float* const pc = addr_c<float>(ad, w);
#pragma unroll
    for (int r = 0; r < 16; ++r) rc[r] = pc[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo];
  }
  for (long long item = w; item < batch; item += W) {
    float* const pc = addr_c<float>(ad, item);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 64 * j + lane;
      *reinterpret_cast<f32x4*>(As + 4 * c) = ra[j];
      const int n = c >> 3, q = c & 7;
      *reinterpret_cast<f32x4*>(Bs + 4 * (n * 8 + (q ^ ((n >> 1) & 7)))) = rb[j];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = BETA0 ? 0.f : rc[r];
    const long long next = item + W;
    if (next < batch) {
      load_mat32(addr_a<float>(ad, next), lane, ra);
      load_mat32(addr_b<float>(ad, next), lane, rb);
      if (!BETA0) {
        const float* const pn = addr_c<float>(ad, next);
#pragma unroll
        for (int r = 0; r < 16; ++r) rc[r] = pn[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo];
      }
    }
    wave_lds_sync();
    f32x4 bt[4]; // B[16*hi + 4t + e][n = lo]
#pragma unroll
    for (int t = 0; t < 4; ++t) bt[t] = *reinterpret_cast<const f32x4*>(Bs + 4 * (lo * 8 + ((4 * hi + t) ^ ((lo >> 1) & 7))));
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float av = As[(16 * hi + s) * 32 + lo]; // A[m = lo][k = 16*hi + s]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bt[s >> 2][s & 3], av, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) pc[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + lo] = acc[r];
    wave_lds_sync();
  }
}

bool is_smm32_f32(const SmmBatch& s)
{
  return 4 == s.typesize && 32 == s.m && 32 == s.n && 32 == s.k && 32 == s.lda && 32 == s.ldb && 32 == s.ldc
      && 0 == (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) && SYNC_NONE == s.sync && 0 == s.general;
}

} // namespace

// returns -1 when no specialised kernel applies (caller falls back to the generic family)
int launch_smm_special(const SmmBatch& s, void* stream, const char** name)
{
  hipStream_t st = (hipStream_t)stream;
  if (is_smm32_f32(s)) {
    const bool beta0 = (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0));
    long long blocks = (s.batch + 3) / 4;
    const long long resident = 256LL * 4; // 4 work-groups (16 waves) per CU
    if (blocks > resident) blocks = resident;
    const DevAddr ad = make_addr(s);
    if (0 != s.use_mfma) {
      *name = "smm_f32_32x32x32_mfma";
      if (beta0) hipLaunchKernelGGL((smm32_f32_mfma_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch);
      else hipLaunchKernelGGL((smm32_f32_mfma_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch);
    }
    else {
      *name = "smm_f32_32x32x32_fma";
      if (beta0) hipLaunchKernelGGL((smm32_f32_fma_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch);
      else hipLaunchKernelGGL((smm32_f32_fma_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch);
    }
    return (int)hipGetLastError();
  }
  return -1;
}

} // namespace xsmm
