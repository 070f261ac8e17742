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
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; // (uniform by construction: addresses and indexes live in scalar registers)
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
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; // (uniform by construction: addresses and indexes live in scalar registers)
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
  // what follows (unit, r0) in this wave's walk
  auto step = [&](long long u, int r, long long& u1, int& r1) { r1 = RUNS ? r + 1 : 0; u1 = u; if (!RUNS || r1 == runlen) { r1 = 0; u1 += W; } };
  float* pc_cur; float* pc_nxt = nullptr; // C of the current / the next item
  {
    const long long first = RUNS ? unit * runlen : unit;
    load_mat32<NT, GLB>(addr_a<float>(ad, first), lane, ra);
    load_mat32<NT, GLB>(addr_b<float>(ad, first), lane, rb);
    pc_cur = addr_c<float>(ad, first);
    if (!BETA0) load_c(pc_cur + coff);
  }
  // Addresses are looked up one item further ahead than the operands: an index (or pointer) batch needs a load per operand
  // before the operand can be requested, and looked up on the spot that load is a memory round trip in front of every item's
  // requests (1 M items through libxsmm_gemm_batch with index arrays: 63 % of the HBM peak against 72 % for the strided form).
  long long raw_a = 0, raw_b = 0, raw_c = 0;
  if (!RUNS && unit + W < nunits) {
    raw_a = raw_of(ad.a, ad.ia, ad.sa, ad, unit + W); raw_b = raw_of(ad.b, ad.ib, ad.sb, ad, unit + W); raw_c = raw_of(ad.c, ad.ic, ad.sc, ad, unit + W);
  }
  f32x16 acc;
  // The result of an item leaves at the top of the next iteration, *before* that iteration issues its loads: the counter a
  // wave waits on (vmcnt) retires loads and stores in the order of issue, so with the stores issued right after the
  // arithmetic -- younger than the prefetched operands -- the wait for the operands at the top of the loop was also a wait for
  // the stores of the previous item to reach memory (measured on the probe tools/probe/mfma_wave.hip: 7.3 -> 2.2 us per item
  // and wave). Deferred, a store has the whole next item to complete.
  f32x16 pend; float* pend_pc = nullptr;
  auto store_pend = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) st1<NT, GLB>(pend_pc + ((r & 3) + 8 * (r >> 2)) * 32, pend[r]);
  };
  for (;;) {
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the operands of this item (nothing younger is in flight)
    if (nullptr != pend_pc) { store_pend(); pend_pc = nullptr; }
    park_ab<1>(As, Bs, lane, ra, rb);
    if (!RUNS || 0 == r0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = BETA0 ? 0.f : rc[r];
    }
    int r1; long long unit1; step(unit, r0, unit1, r1);
    const bool more = unit1 < nunits;
    if (more) {
      if (RUNS) { // (blocked GEMM work lists: looked up on the spot -- the two-step form was measured 6 % slower there, 4096^3 in 32^3 blocks)
        const long long next = unit1 * runlen + r1;
        load_mat32<NT, GLB>(addr_a<float>(ad, next), lane, ra);
        load_mat32<NT, GLB>(addr_b<float>(ad, next), lane, rb);
        if (0 == r1) { pc_nxt = addr_c<float>(ad, next); if (!BETA0) load_c(pc_nxt + coff); } else pc_nxt = pc_cur;
      }
      else { // the next item's operands, at the addresses looked up during the previous item
        load_mat32<NT, GLB>(cooked<const float>(ad.a, ad, raw_a), lane, ra);
        load_mat32<NT, GLB>(cooked<const float>(ad.b, ad, raw_b), lane, rb);
        pc_nxt = cooked<float>(ad.c, ad, raw_c);
        if (!BETA0) load_c(pc_nxt + coff);
        const long long next2 = unit1 + W; // ... and the addresses of the item after it
        if (next2 < nunits) { raw_a = raw_of(ad.a, ad.ia, ad.sa, ad, next2); raw_b = raw_of(ad.b, ad.ib, ad.sb, ad, next2); raw_c = raw_of(ad.c, ad.ic, ad.sc, ad, next2); }
      }
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
    if (!RUNS || r0 + 1 == runlen) { pend = acc; pend_pc = pc_cur + coff; }
    wave_lds_sync();
    if (!more) break;
    unit = unit1; r0 = r1; pc_cur = pc_nxt;
  }
  if (nullptr != pend_pc) store_pend();
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
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lo = lane & 31, hi = lane >> 5;
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

#include "smm_mfma_wg.inc"

// c[i] = a[i] + b[i] + c[i] in whole 4 KiB items per wave with the same prefetch structure as the SMM kernels: the
// traffic mix of a beta=1 SMM batch (3 reads : 1 write) without arithmetic or LDS -- the measured ceiling the SMM
// kernels are compared against (bench.py "stream_ceiling").
__global__ __launch_bounds__(256)
void stream_abc_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, long long items)
{
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = ((long long)gridDim.x * blockDim.x) >> 6; // (blockDim.x is a multiple of 64)
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
    // (index / pointer batches: four work-groups per CU -- the address lookups one item ahead add latency a fourth work-group covers:
    // 1 M items through libxsmm_gemm_batch with index arrays 3.15 -> 2.96 ms, profiles/r3_smm32_sweep.txt; strided batches lose with four)
    const int bpc = env_int("XSMM_SMM32_BPC", (ADDR_STRIDED == s.mode || 0 == s.use_mfma) ? 3 : 4), nt = env_int("XSMM_SMM32_NT", 1);
    const int variant = env_int("XSMM_SMM32_VARIANT", 0 != s.use_mfma ? 0 : 1);
    long long blocks = (s.batch + 3) / 4;
    const int grid_env = env_int("XSMM_SMM32_GRID", 0); // developer knob: the persistent grid in work-groups (0: 256 x bpc)
    const long long resident = (0 < grid_env) ? grid_env : 256LL * (bpc > 0 ? bpc : 3);
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
    // C as a contiguous array through LDS when its columns are not whole 128-byte lines (see kernels/smm_mfma_wg.inc)
    const bool tightc = s.ldc == s.m && 0 == ((s.m * s.n) & 3) && 0 != (s.m & 31) && 0 != env_int("XSMM_SMM64_TIGHTC", 1);
    const int bpc = env_int("XSMM_SMM64_BPC", tightc ? 3 : 4); // (48 KiB of LDS with the C image)
    long long blocks = mfma_wg_units(s);
    const long long resident = 256LL * (bpc > 0 ? bpc : 4);
    if (blocks > resident) blocks = resident;
    if (blocks < 1) return -1;
    const DevAddr ad = make_addr(s);
    const int runlen = mfma_wg_runlen(s);
    *name = (1 == runlen) ? "smm_f32_mfma_wg" : "smm_f32_mfma_wg_runs";
    const bool beta0 = 0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0);
    const bool tight = s.lda == s.m && s.ldb == s.k && 0 == ((s.m * s.k) & 3) && 0 == ((s.k * s.n) & 3) && 0 != env_int("XSMM_SMM64_WIDE", 1);
#define XSMM_MW(B0, TI, TC) hipLaunchKernelGGL((smm_f32_mfma_wg_kernel<B0, TI, TC>), dim3((unsigned)blocks), dim3(256), 0, st, ad, s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.batch, runlen)
    if (tightc) { if (tight) { if (beta0) XSMM_MW(true, true, true); else XSMM_MW(false, true, true); } else { if (beta0) XSMM_MW(true, false, true); else XSMM_MW(false, false, true); } }
    else { if (tight) { if (beta0) XSMM_MW(true, true, false); else XSMM_MW(false, true, false); } else { if (beta0) XSMM_MW(true, false, false); else XSMM_MW(false, false, false); } }
#undef XSMM_MW
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
