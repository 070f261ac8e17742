// smm_special.hip -- tuned batched SMM kernels for the shapes the engine is measured on (gfx950).
//
// smm32_*: M=N=K=32, fp32, tight leading dimensions (BASELINE config 2, reference samples/smm/specialized.cpp).
// One wavefront owns one problem at a time and walks the batch with a stride of all resident waves, so
// neighbouring waves stream neighbouring 4 KiB matrices. Per problem: A and B arrive as four 16-byte loads
// per lane (1 KiB per wave instruction), are parked in the wave's private 8 KiB of LDS, C is read and written
// directly. The loads of problem i+1 are issued before the arithmetic of problem i (the GPU analogue of the
// reference's prefetch chaining, src/libxsmm_gemm.c:1348). Every byte is touched exactly once, so loads and
// stores carry the non-temporal hint.
//   "fma"  variant: 4x4 register tile per lane, v_fma_f32, k ascending -- bit-identical to the reference's
//                   per-element fma chain.
//   "mfma" variants: v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain; the two k of instruction s are (2s, 2s+1), so the
//                   chain per C element runs k = 0,1,2,...: bit-identical to the "fma" variant and the reference's chain.
#include "smm_common.cuh"
#include <mutex>

namespace xsmm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool aligned16(const void* p) { return 0 == (reinterpret_cast<uintptr_t>(p) & 15); }

// GLB: operand accesses in the global address space. The addresses come out of a run-time choice between three
// addressing modes (one of which loads the pointer), so left to the compiler every access is a FLAT instruction (counts on
// lgkmcnt as well as vmcnt). XSMM_SMM32_VARIANT=1 selects GLB for same-process A/B runs (tools/sweep_smm32.py).
template<bool GLB> struct Space { typedef float* ptr; typedef const float* cptr; typedef f32x4* vptr; typedef const f32x4* cvptr; };
template<> struct Space<true> {
  typedef __attribute__((address_space(1))) float* ptr; typedef const __attribute__((address_space(1))) float* cptr;
  typedef __attribute__((address_space(1))) f32x4* vptr; typedef const __attribute__((address_space(1))) f32x4* cvptr;
};
template<bool NT, bool GLB> __device__ __forceinline__ f32x4 ld4(const float* p, bool al)
{
  const typename Space<GLB>::cptr g = (typename Space<GLB>::cptr)p;
  if (al) return NT ? __builtin_nontemporal_load((typename Space<GLB>::cvptr)g) : *(typename Space<GLB>::cvptr)g;
  return f32x4{ g[0], g[1], g[2], g[3] }; // operands that are only element-aligned (arbitrary index arrays)
}
template<bool NT, bool GLB> __device__ __forceinline__ void st4(float* p, bool al, f32x4 v)
{
  const typename Space<GLB>::ptr g = (typename Space<GLB>::ptr)p;
  if (al) { if (NT) __builtin_nontemporal_store(v, (typename Space<GLB>::vptr)g); else *(typename Space<GLB>::vptr)g = v; }
  else { g[0] = v[0]; g[1] = v[1]; g[2] = v[2]; g[3] = v[3]; }
}
template<bool NT, bool GLB> __device__ __forceinline__ float ld1(const float* p) { const typename Space<GLB>::cptr g = (typename Space<GLB>::cptr)p; return NT ? __builtin_nontemporal_load(g) : *g; }
template<bool NT, bool GLB> __device__ __forceinline__ void st1(float* p, float v) { const typename Space<GLB>::ptr g = (typename Space<GLB>::ptr)p; if (NT) __builtin_nontemporal_store(v, g); else *g = v; }

// 4 x float4 per lane covering a tight 32x32 fp32 matrix: chunk index c = 64*j + lane (16-byte chunks)
template<bool NT, bool GLB> __device__ __forceinline__ void load_mat32(const float* p, int lane, f32x4 (&r)[4])
{
  const bool al = aligned16(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = ld4<NT, GLB>(p + 4 * (64 * j + lane), al);
}

// A linear, B as 16-byte chunks (n, q = k/4) at position n*8 + (q ^ key(n)); key spreads the rows a wave reads
// at once over different bank groups (KEYSHIFT = 2 for the 4x4-tile reads, 1 for the MFMA operand reads).
template<int KEYSHIFT> __device__ __forceinline__ void park_ab(float* As, float* Bs, int lane, const f32x4 (&ra)[4], const f32x4 (&rb)[4])
{
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 64 * j + lane;
    *reinterpret_cast<f32x4*>(As + 4 * c) = ra[j];
    const int n = c >> 3, q = c & 7;
    *reinterpret_cast<f32x4*>(Bs + 4 * (n * 8 + (q ^ ((n >> KEYSHIFT) & 7)))) = rb[j];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// scalar-FMA variant. Lane (tx = lane & 7, ty = lane >> 3) owns C rows 4tx..4tx+3 of columns 4ty..4ty+3.
// ---------------------------------------------------------------------------------------------------------------
template<bool BETA0, bool NT, bool GLB>
__global__ __launch_bounds__(256, 4)
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
  load_mat32<NT, GLB>(addr_a<float>(ad, w), lane, ra);
  load_mat32<NT, GLB>(addr_b<float>(ad, w), lane, rb);
  if (!BETA0) {
    const float* const pc = addr_c<float>(ad, w) + 4 * ty * 32 + 4 * tx;
    const bool al = aligned16(pc);
#pragma unroll
    for (int j = 0; j < 4; ++j) rc[j] = ld4<NT, GLB>(pc + j * 32, al);
  }
  for (long long item = w; item < batch; item += W) {
    float* const pc = addr_c<float>(ad, item) + 4 * ty * 32 + 4 * tx;
    park_ab<2>(As, Bs, lane, ra, rb);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = BETA0 ? f32x4{ 0.f, 0.f, 0.f, 0.f } : rc[j];
    const long long next = item + W; // issue the next problem's loads before computing this one
    if (next < batch) {
      load_mat32<NT, GLB>(addr_a<float>(ad, next), lane, ra);
      load_mat32<NT, GLB>(addr_b<float>(ad, next), lane, rb);
      if (!BETA0) {
        const float* const pn = addr_c<float>(ad, next) + 4 * ty * 32 + 4 * tx;
        const bool al = aligned16(pn);
#pragma unroll
        for (int j = 0; j < 4; ++j) rc[j] = ld4<NT, GLB>(pn + j * 32, al);
      }
    }
    wave_lds_sync();
#pragma unroll 2
    for (int q = 0; q < 8; ++q) { // four k per step (limited unrolling: the tile must stay within 128 VGPRs)
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
    for (int j = 0; j < 4; ++j) st4<NT, GLB>(pc + j * 32, al, acc[j]);
    wave_lds_sync();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA variant with v_mfma_f32_32x32x2_f32; per instruction (step s) lane l supplies k = 2s + (l>>5) for both operands:
//   A[m = l&31][k] from the linear LDS image, B[k][n = l&31] out of eight 16-byte reads of the swizzled image.
// D[i=n][j=m] (B as a-operand): register r of lane l is C[n = (r&3)+8(r>>2)+4(l>>5)][m = l&31], so every C access is a
// dword per lane, two full 128-byte rows per wave instruction. (The transposed operand order, where C moves as four
// 16-byte pieces per lane, was measured 15-40 % slower and has been removed.)
// ---------------------------------------------------------------------------------------------------------------
template<bool BETA0, bool NT, bool GLB, bool RUNS>
__global__ __launch_bounds__(256, 4)
void smm32_f32_mfma_kernel(DevAddr ad, long long batch, int runlen)
{ // RUNS: a unit is a run of `runlen` consecutive items with one C block (blocked GEMM), C stays in the accumulators
  __shared__ __align__(16) float lds[4][2048];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 31, hi = lane >> 5;
  float* const As = lds[wave];
  float* const Bs = lds[wave] + 1024;
  const long long w = (long long)blockIdx.x * 4 + wave, W = (long long)gridDim.x * 4;
  const long long nunits = RUNS ? batch / runlen : batch;
  if (w >= nunits) return;
  const int coff = 4 * hi * 32 + lo; // lane's first C element

  f32x4 ra[4], rb[4];
  float rc[16];
  auto load_c = [&](const float* pc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rc[r] = ld1<NT, GLB>(pc + ((r & 3) + 8 * (r >> 2)) * 32);
  };
  long long unit = w;
  int r0 = 0; // position inside the run
  {
    const long long first = RUNS ? unit * runlen : unit;
    load_mat32<NT, GLB>(addr_a<float>(ad, first), lane, ra);
    load_mat32<NT, GLB>(addr_b<float>(ad, first), lane, rb);
    if (!BETA0) load_c(addr_c<float>(ad, first) + coff);
  }
  f32x16 acc;
  for (;;) {
    const long long item = RUNS ? unit * runlen + r0 : unit;
    park_ab<1>(As, Bs, lane, ra, rb);
    if (!RUNS || 0 == r0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = BETA0 ? 0.f : rc[r];
    }
    int r1 = RUNS ? r0 + 1 : 0; long long unit1 = unit;
    if (!RUNS || r1 == runlen) { r1 = 0; unit1 += W; }
    const bool more = unit1 < nunits;
    if (more) {
      const long long next = RUNS ? unit1 * runlen + r1 : unit1;
      load_mat32<NT, GLB>(addr_a<float>(ad, next), lane, ra);
      load_mat32<NT, GLB>(addr_b<float>(ad, next), lane, rb);
      if (!BETA0 && (!RUNS || 0 == r1)) load_c(addr_c<float>(ad, next) + coff);
    }
    wave_lds_sync();
    // The instruction is a k-ordered fmaf chain (one rounding per product): with the two k of step s being 2s and 2s + 1 every
    // C element receives fma(A[m,k], B[k,n], acc) for k = 0, 1, ..., 31 -- the reference's chain, bit for bit.
    f32x4 bt[8]; // B[4t + e][n = lo]: the whole column (both halves of the wave read the same words: broadcast)
#pragma unroll
    for (int t = 0; t < 8; ++t) bt[t] = *reinterpret_cast<const f32x4*>(Bs + 4 * (lo * 8 + (t ^ ((lo >> 1) & 7))));
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float av = As[(2 * s + hi) * 32 + lo]; // A[m = lo][k = 2s + hi]
      const float bv = (0 != hi) ? bt[s >> 1][2 * (s & 1) + 1] : bt[s >> 1][2 * (s & 1)]; // B[k = 2s + hi][n = lo]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0);
    }
    if (!RUNS || r0 + 1 == runlen) {
      float* const pc = addr_c<float>(ad, item) + coff;
#pragma unroll
      for (int r = 0; r < 16; ++r) st1<NT, GLB>(pc + ((r & 3) + 8 * (r >> 2)) * 32, acc[r]);
    }
    wave_lds_sync();
    if (!more) break;
    unit = unit1; r0 = r1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// smm64: M=N=K=64, fp32, tight leading dimensions -- the largest member of the (M,N,K) <= 64 family, where the vector ALU
// (8 flop per byte at 64^3) rather than HBM limits the register-tiled kernels. One work-group of four waves owns one item
// at a time; wave (mq, nq) computes the 32x32 quadrant C[32mq.., 32nq..] with 32 v_mfma_f32_32x32x2_f32 whose k pairs are
// (2s, 2s + 1): per C element the chain fma(A[m,k], B[k,n], acc), k = 0..63 -- the reference's order, bit for bit.
// LDS images (16 KiB each, read conflict-free):
//   A: word k*64 + (m ^ 32(k&1))  -- a wave reads A[32mq + lo][2s + hi]: the two half-waves land in opposite bank halves;
//   B: 16-byte chunk (n, q = k/4) at n*16 + (q ^ (n&15)) -- a lane reads its own column, 16 lanes cover all bank groups.
// The operands of the work-group's next item are loaded into registers before the arithmetic of the current one.
// ---------------------------------------------------------------------------------------------------------------
template<bool BETA0>
__global__ __launch_bounds__(256, 4)
void smm64_f32_mfma_kernel(DevAddr ad, long long batch)
{
  __shared__ __align__(16) float As[4096];
  __shared__ __align__(16) float Bs[4096];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lo = lane & 31, hi = lane >> 5;
  const int mq = wave & 1, nq = wave >> 1;
  const int m = 32 * mq + lo, n = 32 * nq + lo;
  const int coff = (32 * nq + 4 * hi) * 64 + m; // lane's first C element
  long long item = blockIdx.x;
  if (item >= batch) return; // the whole work-group

  f32x4 ra[4], rb[4];
  float rc[16];
  auto load_ab = [&](long long i) {
    const float* const pa = addr_a<float>(ad, i);
    const float* const pb = addr_b<float>(ad, i);
    const bool ala = aligned16(pa), alb = aligned16(pb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = ld4<true, false>(pa + 4 * (256 * j + t), ala);
      rb[j] = ld4<true, false>(pb + 4 * (256 * j + t), alb);
    }
  };
  auto load_c = [&](const float* pc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rc[r] = ld1<true, false>(pc + ((r & 3) + 8 * (r >> 2)) * 64);
  };
  load_ab(item);
  if (!BETA0) load_c(addr_c<float>(ad, item) + coff);
  for (; item < batch; item += gridDim.x) {
    float* const pc = addr_c<float>(ad, item) + coff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 256 * j + t, row = c >> 4, q = c & 15; // row: k of A, n of B
      *reinterpret_cast<f32x4*>(As + row * 64 + ((4 * q) ^ ((row & 1) << 5))) = ra[j];
      *reinterpret_cast<f32x4*>(Bs + row * 64 + 4 * (q ^ (row & 15))) = rb[j];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = BETA0 ? 0.f : rc[r];
    const long long next = item + gridDim.x;
    if (next < batch) {
      load_ab(next);
      if (!BETA0) load_c(addr_c<float>(ad, next) + coff);
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 bt[8]; // B[32 half + 4u + e][n]
#pragma unroll
      for (int u = 0; u < 8; ++u) bt[u] = *reinterpret_cast<const f32x4*>(Bs + n * 64 + 4 * ((8 * half + u) ^ (n & 15)));
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float av = As[(32 * half + 2 * s + hi) * 64 + (m ^ (hi << 5))]; // A[m][k = 32 half + 2s + hi]
        const float bv = (0 != hi) ? bt[s >> 1][2 * (s & 1) + 1] : bt[s >> 1][2 * (s & 1)]; // B[k][n]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) st1<true, false>(pc + ((r & 3) + 8 * (r >> 2)) * 64, acc[r]);
    __syncthreads(); // all reads of the images are done before the next item is parked
  }
}

// ---------------------------------------------------------------------------------------------------------------
// smm_f32_mfma_wg: the same plan for any fp32 shape with 32 < max(M, N) <= 64 and K <= 64, any leading dimensions -- the
// class the register-tiled work-group kernels serve at about half of the HBM peak. Operands travel as dwords (lanes along
// a column: whole 128/256-byte rows of every column), both LDS images are k-major:
//   A: word k*64 + (m ^ 32(k&1));  B (transposed while parking): word k*64 + (n ^ bkey32(k)), see below.
// K is padded to an even count with A = -0, B = +0: the extra product is -0 and x + (-0) = x for every x, signed zeros
// included, so the chain stays the reference's. Rows m >= M and columns n >= N of the images hold the same padding; the
// C elements they would produce are neither loaded nor stored.
// ---------------------------------------------------------------------------------------------------------------
// Swizzle keys of the k-major B images: word k*64 + (n ^ key(k)). The transposing writes put 64 (fp32) resp. 64 (fp64)
// different k of one column into one instruction, the reads the two (fp32) resp. four (fp64) k of an MFMA step with 32 resp.
// 16 neighbouring n each. The LDS serves a dword instruction in passes of 32 lanes and a qword instruction in passes of 16
// (measured: keys that were only distinct over the full wave showed SQ_LDS_BANK_CONFLICT), so the low bits of the key
// follow k itself and one more bit separates the k of a step as well as the two halves of the wave.
__device__ __forceinline__ int bkey32(int k) { return (k & 31) | ((((k >> 5) ^ k) & 1) << 5); }
__device__ __forceinline__ int bkey64(int k) { return (k & 15) | ((((k >> 4) ^ k) & 1) << 4); }

// The operands of an item are the same for the whole work-group: with the base in scalar registers a load is "scalar base
// + one 32-bit lane offset", and the 48 loads of an item share three offset registers instead of holding 48 addresses.
template<typename T> __device__ __forceinline__ T* wave_uniform(T* p)
{
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

template<bool BETA0, bool TIGHT>
__global__ __launch_bounds__(256, 4)
void smm_f32_mfma_wg_kernel(DevAddr ad, int M, int N, int K, int lda, int ldb, int ldc, long long batch, int runlen)
{
  __shared__ __align__(16) float As[4096];
  __shared__ __align__(16) float Bs[4096];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lo = lane & 31, hi = lane >> 5;
  const int mq = wave & 1, nq = wave >> 1;
  const int m = 32 * mq + lo, n = 32 * nq + lo;
  const int ksteps = (K + 1) >> 1;
  const bool active = (32 * mq < M) && (32 * nq < N); // the wave's quadrant holds part of C
  // A unit is a run of `runlen` consecutive items with one C block (blocked GEMM: the k blocks of a C block); C stays in the
  // accumulators across the run. runlen = 1: independent items.
  const long long nunits = batch / runlen;
  long long unit = blockIdx.x;
  int r0 = 0; // position inside the run
  if (unit >= nunits) return; // the whole work-group

  // TIGHT (lda = M, ldb = K, M*K and K*N multiples of four): A and B of an item are contiguous arrays and travel as 16-byte
  // chunks (chunk c = 256j + t holds elements 4c..4c+3) -- a few wide loads instead of one dword load per column; the
  // elements are scattered into the images one by one. Rows k >= K of the images (odd K) are written once, up front.
  float ra[16], rb[16], rc[16];
  const int mk = M * K, kn = K * N;
  const float rcpm = 1.0f / (float)M, rcpk = 1.0f / (float)K;
  if (TIGHT && 0 != (K & 1)) { As[K * 64 + (t & 63)] = -0.f; Bs[K * 64 + (t & 63)] = 0.f; }
  const unsigned offa = (unsigned)(wave * lda + lane), offb = (unsigned)(wave * ldb + lane);
  const unsigned offc = (unsigned)((32 * nq + 4 * hi) * ldc + m);
  auto load_ab = [&](long long it) {
    const float* const pa = wave_uniform(addr_a<float>(ad, it));
    const float* const pb = wave_uniform(addr_b<float>(ad, it));
    if (TIGHT) {
      const bool ala = aligned16(pa), alb = aligned16(pb);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * (256 * j + t);
        if (e < mk) { const f32x4 v = ld4<true, true>(pa + e, ala); ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3]; }
        if (e < kn) { const f32x4 v = ld4<true, true>(pb + e, alb); rb[4 * j] = v[0]; rb[4 * j + 1] = v[1]; rb[4 * j + 2] = v[2]; rb[4 * j + 3] = v[3]; }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { // element (row = lane, column = 4j + wave) of the 64x64 frame
      const int col = 4 * j + wave;
      ra[j] = (lane < M && col < K) ? ld1<true, true>(pa + (size_t)(4 * j) * lda + offa) : -0.f;
      rb[j] = (lane < K && col < N) ? ld1<true, true>(pb + (size_t)(4 * j) * ldb + offb) : 0.f;
    }
  };
  auto load_c = [&](const float* pc0) {
    const float* const pc = wave_uniform(pc0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nr = (r & 3) + 8 * (r >> 2);
      rc[r] = (m < M && 32 * nq + 4 * hi + nr < N) ? ld1<true, true>(pc + (size_t)nr * ldc + offc) : 0.f;
    }
  };
  load_ab(unit * runlen);
  if (!BETA0 && active) load_c(addr_c<float>(ad, unit * runlen));
  f32x16 acc;
  for (;;) {
    const long long item = unit * runlen + r0;
    if (TIGHT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = 4 * (256 * j + t) + u;
          if (e < mk) { const int col = (int)(((float)e + 0.5f) * rcpm), row = e - col * M; As[col * 64 + (row ^ ((col & 1) << 5))] = ra[4 * j + u]; } // A[m = row][k = col]
          if (e < kn) { const int col = (int)(((float)e + 0.5f) * rcpk), row = e - col * K; Bs[row * 64 + (col ^ bkey32(row))] = rb[4 * j + u]; }       // B[k = row][n = col]
        }
      }
    }
    else
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int col = 4 * j + wave;
      As[col * 64 + (lane ^ ((col & 1) << 5))] = ra[j];                         // A[m = lane][k = col]
      Bs[lane * 64 + (col ^ bkey32(lane))] = rb[j];                               // B[k = lane][n = col]
    }
    if (0 == r0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = BETA0 ? 0.f : rc[r];
    }
    int r1 = r0 + 1; long long unit1 = unit;
    if (r1 == runlen) { r1 = 0; unit1 += gridDim.x; }
    const bool more = unit1 < nunits;
    if (more) {
      load_ab(unit1 * runlen + r1);
      if (!BETA0 && active && 0 == r1) load_c(addr_c<float>(ad, unit1 * runlen));
    }
    __syncthreads();
    if (active) {
      for (int s = 0; s < ksteps; ++s) {
        const int k = 2 * s + hi;
        const float av = As[k * 64 + (m ^ (hi << 5))];       // A[m][k]
        const float bv = Bs[k * 64 + (n ^ bkey32(k))];       // B[k][n]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc, 0, 0, 0);
      }
      if (r0 + 1 == runlen) {
        float* const pc = wave_uniform(addr_c<float>(ad, item));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nr = (r & 3) + 8 * (r >> 2);
          if (m < M && 32 * nq + 4 * hi + nr < N) st1<true, true>(pc + (size_t)nr * ldc + offc, acc[r]);
        }
      }
    }
    __syncthreads(); // all reads of the images are done before the next item is parked
    if (!more) break;
    unit = unit1; r0 = r1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// smm_f64_mfma_wg: the fp64 member: v_mfma_f64_16x16x4_f64 (probed to be the k-ordered fma chain bit for bit,
// tools/probe/mfma_f64_chain.hip), 2x2 tiles of 16x16 per wave, step s feeds k = 4s + q (q = lane >> 4); B is the first
// operand, so register r of lane (i, q) is C[n = q + 4r][m = i] of a tile: 128-byte rows. Images k-major, only the
// 4*ceil(K/4) rows in use are allocated (dynamic LDS: 1 KiB per k), so shorter K leave room for a third work-group per CU:
//   A: word k*64 + (m ^ 16(k&1));  B: word k*64 + (n ^ bkey64(k))
// K is padded to a multiple of four with A = -0, B = +0 (see above).
// ---------------------------------------------------------------------------------------------------------------
template<bool BETA0>
__global__ __launch_bounds__(256, 3)
void smm_f64_mfma_wg_kernel(DevAddr ad, int M, int N, int K, int lda, int ldb, int ldc, long long batch, int runlen)
{
  extern __shared__ __align__(16) double lds64[];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 15, q = lane >> 4;
  const int mq = wave & 1, nq = wave >> 1;
  const int ksteps = (K + 3) >> 2, kp = 4 * ksteps;
  double* const As = lds64;
  double* const Bs = lds64 + kp * 64;
  const int m0 = 32 * mq + i, n0 = 32 * nq + i;
  const bool tm1 = (32 * mq + 16 < M), tn1 = (32 * nq + 16 < N);
  const bool active = (32 * mq < M) && (32 * nq < N); // the wave's quadrant holds part of C
  // A unit is a run of `runlen` consecutive items with one C block (blocked GEMM: the k blocks of a C block); C stays in the
  // accumulators across the run. runlen = 1: independent items.
  const long long nunits = batch / runlen;
  long long unit = blockIdx.x;
  int r0 = 0; // position inside the run
  if (unit >= nunits) return; // the whole work-group

  double ra[16], rb[16], rc[16];
  const unsigned offa = (unsigned)(wave * lda + lane), offb = (unsigned)(wave * ldb + lane);
  const unsigned offc = (unsigned)((32 * nq + q) * ldc + m0);
  typedef const __attribute__((address_space(1))) double* gcptr;
  typedef __attribute__((address_space(1))) double* gptr;
  auto load_ab = [&](long long it) {
    const double* const pa = wave_uniform(addr_a<double>(ad, it));
    const double* const pb = wave_uniform(addr_b<double>(ad, it));
#pragma unroll
    for (int j = 0; j < 16; ++j) { // element (row = lane, column = 4j + wave) of the 64x64 frame
      const int col = 4 * j + wave;
      ra[j] = (lane < M && col < K) ? __builtin_nontemporal_load((gcptr)(pa + (size_t)(4 * j) * lda + offa)) : -0.0;
      rb[j] = (lane < K && col < N) ? __builtin_nontemporal_load((gcptr)(pb + (size_t)(4 * j) * ldb + offb)) : 0.0;
    }
  };
  // C element e = 8 tn + 4 tm + r: n = 32 nq + 16 tn + q + 4r, m = 32 mq + 16 tm + i
  auto load_c = [&](const double* pc0) {
    const double* const pc = wave_uniform(pc0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int nr = 16 * (e >> 3) + 4 * (e & 3), mr = 16 * ((e >> 2) & 1);
      rc[e] = (m0 + mr < M && 32 * nq + q + nr < N) ? __builtin_nontemporal_load((gcptr)(pc + (size_t)nr * ldc + mr + offc)) : 0.0;
    }
  };
  load_ab(unit * runlen);
  if (!BETA0 && active) load_c(addr_c<double>(ad, unit * runlen));
  f64x4 acc[2][2]; // [tn][tm]
  for (;;) {
    const long long item = unit * runlen + r0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int col = 4 * j + wave;
      if (col < kp) As[col * 64 + (lane ^ ((col & 1) << 4))] = ra[j];                                      // A[m = lane][k = col]
      if (lane < kp) Bs[lane * 64 + (col ^ bkey64(lane))] = rb[j];                                         // B[k = lane][n = col]
    }
    if (0 == r0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e >> 3][(e >> 2) & 1][e & 3] = BETA0 ? 0.0 : rc[e];
    }
    int r1 = r0 + 1; long long unit1 = unit;
    if (r1 == runlen) { r1 = 0; unit1 += gridDim.x; }
    const bool more = unit1 < nunits;
    if (more) {
      load_ab(unit1 * runlen + r1);
      if (!BETA0 && active && 0 == r1) load_c(addr_c<double>(ad, unit1 * runlen));
    }
    __syncthreads();
    if (active) {
      for (int s = 0; s < ksteps; ++s) {
        const int k = 4 * s + q, sa = (q & 1) << 4, sb = bkey64(k);
        const double a0 = As[k * 64 + (m0 ^ sa)], b0 = Bs[k * 64 + (n0 ^ sb)];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
        if (tm1) {
          const double a1 = As[k * 64 + ((m0 + 16) ^ sa)];
          acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[0][1], 0, 0, 0);
          if (tn1) {
            const double b1 = Bs[k * 64 + ((n0 + 16) ^ sb)];
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
          }
        }
        else if (tn1) {
          const double b1 = Bs[k * 64 + ((n0 + 16) ^ sb)];
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[1][0], 0, 0, 0);
        }
      }
      if (r0 + 1 == runlen) {
        double* const pc = wave_uniform(addr_c<double>(ad, item));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int nr = 16 * (e >> 3) + 4 * (e & 3), mr = 16 * ((e >> 2) & 1);
          if (m0 + mr < M && 32 * nq + q + nr < N) __builtin_nontemporal_store(acc[e >> 3][(e >> 2) & 1][e & 3], (gptr)(pc + (size_t)nr * ldc + mr + offc));
        }
      }
    }
    __syncthreads(); // all reads of the images are done before the next item is parked
    if (!more) break;
    unit = unit1; r0 = r1;
  }
}

// Two-stage form for K > 32: the images hold 32 k at a time (32 KiB per work-group, so three work-groups fit a CU where the
// one-stage form has room for two); the halves of A and B are parked and consumed one after the other, and the registers of a
// half are refilled with the next item's as soon as they are parked. B is loaded k-fastest (32 consecutive k of a column per
// half wave) so that a register belongs to one half.
template<bool BETA0>
__global__ __launch_bounds__(256, 3)
void smm_f64_mfma_wg2_kernel(DevAddr ad, int M, int N, int K, int lda, int ldb, int ldc, long long batch, int runlen)
{
  extern __shared__ __align__(16) double lds64[];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = lane & 15, q = lane >> 4;
  const int mq = wave & 1, nq = wave >> 1;
  const int ksteps = (K + 3) >> 2, kp = 4 * ksteps;
  double* const As = lds64;
  double* const Bs = lds64 + 32 * 64;
  const int m0 = 32 * mq + i, n0 = 32 * nq + i;
  const bool tm1 = (32 * mq + 16 < M), tn1 = (32 * nq + 16 < N);
  const bool active = (32 * mq < M) && (32 * nq < N); // the wave's quadrant holds part of C
  // A unit is a run of `runlen` consecutive items with one C block (blocked GEMM: the k blocks of a C block); C stays in the
  // accumulators across the run. runlen = 1: independent items.
  const long long nunits = batch / runlen;
  long long unit = blockIdx.x;
  int r0 = 0; // position inside the run
  if (unit >= nunits) return; // the whole work-group

  double ra[16], rb[16], rc[16];
  const int kk = t & 31, nb = t >> 5; // B: row within a half, first column
  const unsigned offa = (unsigned)(wave * lda + lane), offb = (unsigned)(nb * ldb + kk);
  const unsigned offc = (unsigned)((32 * nq + q) * ldc + m0);
  typedef const __attribute__((address_space(1))) double* gcptr;
  typedef __attribute__((address_space(1))) double* gptr;
  auto load_half = [&](long long it, int h) {
    const double* const pa = wave_uniform(addr_a<double>(ad, it));
    const double* const pb = wave_uniform(addr_b<double>(ad, it));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = 32 * h + 4 * j + wave; // A[m = lane][k = col]
      const double va = (lane < M && col < K) ? __builtin_nontemporal_load((gcptr)(pa + (size_t)(32 * h + 4 * j) * lda + offa)) : -0.0;
      const int n = nb + 8 * j;              // B[k = 32h + kk][n]
      const double vb = (32 * h + kk < K && n < N) ? __builtin_nontemporal_load((gcptr)(pb + (size_t)(8 * j) * ldb + 32 * h + offb)) : 0.0;
      if (0 == h) { ra[j] = va; rb[j] = vb; } else { ra[8 + j] = va; rb[8 + j] = vb; }
    }
  };
  auto park_half = [&](int h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = 4 * j + wave, k = 32 * h + kk; // local row of A, global k of B
      if (32 * h + col < kp) As[col * 64 + (lane ^ ((col & 1) << 4))] = (0 == h) ? ra[j] : ra[8 + j];
      if (k < kp) Bs[kk * 64 + ((nb + 8 * j) ^ bkey64(k))] = (0 == h) ? rb[j] : rb[8 + j];
    }
  };
  auto compute_half = [&](int h, f64x4 (&acc)[2][2]) {
    const int s1 = (ksteps < 8 * h + 8) ? ksteps : 8 * h + 8;
    for (int s = 8 * h; s < s1; ++s) {
      const int k = 4 * s + q, kl = k - 32 * h, sa = (q & 1) << 4, sb = bkey64(k);
      const double a0 = As[kl * 64 + (m0 ^ sa)], b0 = Bs[kl * 64 + (n0 ^ sb)];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
      if (tm1) {
        const double a1 = As[kl * 64 + ((m0 + 16) ^ sa)];
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[0][1], 0, 0, 0);
        if (tn1) {
          const double b1 = Bs[kl * 64 + ((n0 + 16) ^ sb)];
          acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
      }
      else if (tn1) {
        const double b1 = Bs[kl * 64 + ((n0 + 16) ^ sb)];
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[1][0], 0, 0, 0);
      }
    }
  };
  // C element e = 8 tn + 4 tm + r: n = 32 nq + 16 tn + q + 4r, m = 32 mq + 16 tm + i
  auto load_c = [&](const double* pc0) {
    const double* const pc = wave_uniform(pc0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int nr = 16 * (e >> 3) + 4 * (e & 3), mr = 16 * ((e >> 2) & 1);
      rc[e] = (m0 + mr < M && 32 * nq + q + nr < N) ? __builtin_nontemporal_load((gcptr)(pc + (size_t)nr * ldc + mr + offc)) : 0.0;
    }
  };
  load_half(unit * runlen, 0);
  load_half(unit * runlen, 1);
  if (!BETA0 && active) load_c(addr_c<double>(ad, unit * runlen));
  f64x4 acc[2][2]; // [tn][tm]
  for (;;) {
    const long long item = unit * runlen + r0;
    int r1 = r0 + 1; long long unit1 = unit;
    if (r1 == runlen) { r1 = 0; unit1 += gridDim.x; }
    const bool more = unit1 < nunits;
    const long long next = unit1 * runlen + r1;
    park_half(0);
    if (0 == r0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e >> 3][(e >> 2) & 1][e & 3] = BETA0 ? 0.0 : rc[e];
    }
    if (more) load_half(next, 0);
    __syncthreads();
    if (active) compute_half(0, acc);
    __syncthreads(); // the first half has been read
    park_half(1);
    if (more) {
      load_half(next, 1);
      if (!BETA0 && active && 0 == r1) load_c(addr_c<double>(ad, unit1 * runlen));
    }
    __syncthreads();
    if (active) {
      compute_half(1, acc);
      if (r0 + 1 == runlen) {
        double* const pc = wave_uniform(addr_c<double>(ad, item));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int nr = 16 * (e >> 3) + 4 * (e & 3), mr = 16 * ((e >> 2) & 1);
          if (m0 + mr < M && 32 * nq + q + nr < N) __builtin_nontemporal_store(acc[e >> 3][(e >> 2) & 1][e & 3], (gptr)(pc + (size_t)nr * ldc + mr + offc));
        }
      }
    }
    __syncthreads(); // all reads of the images are done before the next item is parked
    if (!more) break;
    unit = unit1; r0 = r1;
  }
}

// c[i] = a[i] + b[i] + c[i] in whole 4 KiB items per wave with the same prefetch structure as the SMM kernels: the
// traffic mix of a beta=1 SMM batch (3 reads : 1 write) without arithmetic or LDS -- the measured ceiling the SMM
// kernels are compared against (bench.py "stream_ceiling").
__global__ __launch_bounds__(256)
void stream_abc_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, long long items)
{
  const int lane = threadIdx.x & 63;
  const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, W = ((long long)gridDim.x * blockDim.x) >> 6;
  if (w >= items) return;
  f32x4 ra[4], rb[4], rc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long o = w * 256 + 64 * j + lane;
    ra[j] = __builtin_nontemporal_load(a + o); rb[j] = __builtin_nontemporal_load(b + o); rc[j] = __builtin_nontemporal_load(c + o);
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
        ra[j] = __builtin_nontemporal_load(a + o); rb[j] = __builtin_nontemporal_load(b + o); rc[j] = __builtin_nontemporal_load(c + o);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(r[j], c + it * 256 + 64 * j + lane);
  }
}

int env_int(const char* name, int fallback)
{
  const char* const v = getenv(name);
  return (nullptr != v && 0 != *v) ? atoi(v) : fallback;
}

bool is_smm32_f32(const SmmBatch& s)
{
  return 4 == s.typesize && 32 == s.m && 32 == s.n && 32 == s.k && 32 == s.lda && 32 == s.ldb && 32 == s.ldc
      && 0 == (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) && SYNC_NONE == s.sync && 0 == s.general;
}

bool is_smm64(const SmmBatch& s, int typesize)
{
  return typesize == s.typesize && 64 == s.m && 64 == s.n && 64 == s.k && 64 == s.lda && 64 == s.ldb && 64 == s.ldc
      && 0 == (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) && SYNC_NONE == s.sync && 0 == s.general && 0 != s.use_mfma;
}

// Work units of the matrix-core work-group kernels: independent items, or runs of a fixed length the caller vouches for
// (blocked GEMM: the k blocks of a C block follow each other). 0: not for these kernels.
long long mfma_wg_units(const SmmBatch& s)
{
  if (SYNC_NONE == s.sync) return s.batch;
  if (SYNC_RUNS == s.sync && 0 < s.uniform_run && 0 == s.batch % s.uniform_run) return s.batch / s.uniform_run;
  return 0;
}
int mfma_wg_runlen(const SmmBatch& s) { return SYNC_NONE == s.sync ? 1 : s.uniform_run; }

template<bool NT, bool GLB>
int launch_smm32(const SmmBatch& s, hipStream_t st, unsigned blocks, const char** name)
{
  const bool beta0 = (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0));
  const DevAddr ad = make_addr(s);
  if (0 == s.use_mfma) {
    *name = "smm_f32_32x32x32_fma";
    if (beta0) hipLaunchKernelGGL((smm32_f32_fma_kernel<true, NT, GLB>), dim3(blocks), dim3(256), 0, st, ad, s.batch);
    else hipLaunchKernelGGL((smm32_f32_fma_kernel<false, NT, GLB>), dim3(blocks), dim3(256), 0, st, ad, s.batch);
  }
  else {
    *name = "smm_f32_32x32x32_mfma";
    if (beta0) hipLaunchKernelGGL((smm32_f32_mfma_kernel<true, NT, GLB, false>), dim3(blocks), dim3(256), 0, st, ad, s.batch, 1);
    else hipLaunchKernelGGL((smm32_f32_mfma_kernel<false, NT, GLB, false>), dim3(blocks), dim3(256), 0, st, ad, s.batch, 1);
  }
  return (int)hipGetLastError();
}

} // namespace

int launch_smm_special(const SmmBatch& s, void* stream, const char** name)
{
  hipStream_t st = (hipStream_t)stream;
  if (is_smm32_f32(s)) {
    // tuning knobs (developer use; re-read on every launch so that one process can sweep them):
    // work-groups per CU of the persistent grid, address space of the operand accesses (1 = global), non-temporal hint
    // Defaults from same-process sweeps (profiles/r1_smm32_variant_sweep.txt): 3 work-groups per CU for both kernels; the
    // MFMA kernel is 4 % faster with generic pointers (FLAT accesses), the scalar-FMA kernel 10 % faster with global ones.
    const int bpc = env_int("XSMM_SMM32_BPC", 3), nt = env_int("XSMM_SMM32_NT", 1);
    const int variant = env_int("XSMM_SMM32_VARIANT", 0 != s.use_mfma ? 0 : 1);
    long long blocks = (s.batch + 3) / 4;
    const long long resident = 256LL * (bpc > 0 ? bpc : 3);
    if (blocks > resident) blocks = resident;
    if (1 == variant) return (0 != nt) ? launch_smm32<true, true>(s, st, (unsigned)blocks, name) : launch_smm32<false, true>(s, st, (unsigned)blocks, name);
    return (0 != nt) ? launch_smm32<true, false>(s, st, (unsigned)blocks, name) : launch_smm32<false, false>(s, st, (unsigned)blocks, name);
  }
  if (4 == s.typesize && 32 == s.m && 32 == s.n && 32 == s.k && 32 == s.lda && 32 == s.ldb && 32 == s.ldc && 0 == (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B)
    && SYNC_RUNS == s.sync && 0 < s.uniform_run && 0 == s.batch % s.uniform_run && 0 == s.general && 0 != s.use_mfma && 0 != env_int("XSMM_SMM32_RUNS", 1))
  { // blocked GEMM with 32^3 blocks: a wave per C block, the k blocks of the run through the matrix cores
    const long long units = s.batch / s.uniform_run;
    const int bpc = env_int("XSMM_SMM32_RUNS_BPC", 4);
    long long blocks = (units + 3) / 4;
    const long long resident = 256LL * (bpc > 0 ? bpc : 4);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) return -1;
    const DevAddr ad = make_addr(s);
    *name = "smm_f32_32x32x32_mfma_runs";
    if (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) hipLaunchKernelGGL((smm32_f32_mfma_kernel<true, false, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch, s.uniform_run);
    else hipLaunchKernelGGL((smm32_f32_mfma_kernel<false, false, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch, s.uniform_run);
    return (int)hipGetLastError();
  }
  if (is_smm64(s, 4) && 0 != env_int("XSMM_SMM64_MFMA", 1) && 0 != env_int("XSMM_SMM64_TIGHT", 1)) {
    const int bpc = env_int("XSMM_SMM64_BPC", 4);
    long long blocks = s.batch;
    const long long resident = 256LL * (bpc > 0 ? bpc : 3);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) return -1;
    const DevAddr ad = make_addr(s);
    *name = "smm_f32_64x64x64_mfma";
    if (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) hipLaunchKernelGGL((smm64_f32_mfma_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch);
    else hipLaunchKernelGGL((smm64_f32_mfma_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.batch);
    return (int)hipGetLastError();
  }
  if (4 == s.typesize && (32 < s.m || 32 < s.n) && s.m <= 64 && s.n <= 64 && 0 < s.k && s.k <= 64 && s.lda >= s.m && s.ldb >= s.k && s.ldc >= s.m
    && 0 == (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) && mfma_wg_units(s) > 0 && 0 == s.general && 0 != s.use_mfma && 0 != env_int("XSMM_SMM64_MFMA", 1))
  {
    const int bpc = env_int("XSMM_SMM64_BPC", 4);
    long long blocks = mfma_wg_units(s);
    const long long resident = 256LL * (bpc > 0 ? bpc : 4);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) return -1;
    const DevAddr ad = make_addr(s);
    const int runlen = mfma_wg_runlen(s);
    *name = (1 == runlen) ? "smm_f32_mfma_wg" : "smm_f32_mfma_wg_runs";
    const bool beta0 = 0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0);
    const bool tight = s.lda == s.m && s.ldb == s.k && 0 == ((s.m * s.k) & 3) && 0 == ((s.k * s.n) & 3) && 0 != env_int("XSMM_SMM64_WIDE", 1);
    if (tight) {
      if (beta0) hipLaunchKernelGGL((smm_f32_mfma_wg_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
      else hipLaunchKernelGGL((smm_f32_mfma_wg_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
    }
    else {
      if (beta0) hipLaunchKernelGGL((smm_f32_mfma_wg_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
      else hipLaunchKernelGGL((smm_f32_mfma_wg_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
    }
    return (int)hipGetLastError();
  }
  if (8 == s.typesize && (32 < s.m || 32 < s.n) && s.m <= 64 && s.n <= 64 && 0 < s.k && s.k <= 64 && s.lda >= s.m && s.ldb >= s.k && s.ldc >= s.m
    && 0 == (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) && mfma_wg_units(s) > 0 && 0 == s.general && 0 != s.use_mfma && 0 != env_int("XSMM_SMM64_MFMA", 1))
  {
    const bool two = s.k > 32 && 0 != env_int("XSMM_SMM64_TWOSTAGE", 1);
    const size_t lds = two ? (size_t)2 * 32 * 64 * sizeof(double) : (size_t)2 * (4 * ((s.k + 3) / 4)) * 64 * sizeof(double);
    int fit = (int)((160u * 1024u) / lds);
    if (fit > 3) fit = 3;
    const int bpc = env_int("XSMM_SMM64_BPC", fit);
    long long blocks = mfma_wg_units(s);
    const long long resident = 256LL * (bpc > 0 ? bpc : fit);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) return -1;
    const DevAddr ad = make_addr(s);
    const int runlen = mfma_wg_runlen(s);
    *name = (1 == runlen) ? "smm_f64_mfma_wg" : "smm_f64_mfma_wg_runs";
    static std::once_flag once; // more than 64 KiB of dynamic LDS has to be asked for
    std::call_once(once, []() {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smm_f64_mfma_wg_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smm_f64_mfma_wg_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    });
    if (two) {
      if (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) hipLaunchKernelGGL((smm_f64_mfma_wg2_kernel<true>), dim3((unsigned)blocks), dim3(256), lds, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
      else hipLaunchKernelGGL((smm_f64_mfma_wg2_kernel<false>), dim3((unsigned)blocks), dim3(256), lds, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
      return (int)hipGetLastError();
    }
    if (0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) hipLaunchKernelGGL((smm_f64_mfma_wg_kernel<true>), dim3((unsigned)blocks), dim3(256), lds, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
    else hipLaunchKernelGGL((smm_f64_mfma_wg_kernel<false>), dim3((unsigned)blocks), dim3(256), lds, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen);
    return (int)hipGetLastError();
  }
  return -1;
}

int launch_stream_abc(const void* a, const void* b, void* c, long long bytes, void* stream)
{
  const long long items = bytes / 4096;
  const int bpc = env_int("XSMM_STREAM_BPC", 3);
  hipLaunchKernelGGL(stream_abc_kernel, dim3(256u * (unsigned)(bpc > 0 ? bpc : 3)), dim3(256), 0, (hipStream_t)stream,
    (const f32x4*)a, (const f32x4*)b, (f32x4*)c, items);
  return (int)hipGetLastError();
}

} // namespace xsmm
