// xsmm_jit_smm.cpp -- dense SMM kernels specialised per descriptor with hiprtc (the gfx950 analogue of
// libxsmm_build's JIT, reference src/libxsmm_main.c:1246-1683: one kernel per (precision, M, N, K, flags)).
//
// The pre-compiled kernels in kernels/smm_generic.hip serve every descriptor; for large batches of small, tightly
// packed matrices (lda == m, ldb == k, ldc == m) a kernel with M, N, K baked in is generated at first use:
//   * one wavefront per item, walking the batch with a stride of all resident waves;
//   * A, B, C are fetched as flat, fully coalesced, non-temporal loads (the widest access the alignment of the item
//     size allows) one item ahead of the arithmetic, parked in wave-private LDS, and C leaves the same way;
//   * 8x8 lanes, TM x TN register tile, v_fma in ascending k: the reference's per-element chain, bit for bit.
#include "xsmm_internal.hpp"

#include <hip/hip_runtime_api.h>

#include <cstring>
#include <mutex>
#include <string>
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace xsmm {

extern const char* const SMM_MFMA_WG_SOURCE; // kernels/smm_mfma_wg.inc as text (Makefile)

namespace {

const char* const SMM_JIT_PRELUDE = R"XSMM(
#if XFLAT
#define XGLOBAL
#else
#define XGLOBAL __attribute__((address_space(1)))
#endif
// ---- batch addressing (same structure and meaning as kernels/smm_common.cuh) ----
struct DevAddr {
  const char* a; const char* b; char* c;
  const char* ia; const char* ib; const char* ic;
  long long sa, sb, sc;
  int index_base, index_stride, mode;
  const int* flags; // device-side verdict on how C blocks repeat: [0] equal neighbours, [1] out-of-order repeats (or null)
};
template<typename P> __device__ __forceinline__ P* resolve(const char* base, const char* idx, long long stride, const DevAddr& ad, long long i)
{
  if (0 == ad.mode) return (P*)base + i * stride;
  if (1 == ad.mode) { if (nullptr == idx) return (P*)base; const int v = *(const XGLOBAL int*)(idx + i * (long long)ad.index_stride); return (P*)base + ((long long)v - ad.index_base); }
  return *(P* const XGLOBAL*)(base + i * stride);
}
__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
typedef short xs16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int xfma(int a, int b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(xs16x2, a), __builtin_bit_cast(xs16x2, b), c, false); } // two i16 x i16 products + i32, wrapping
__device__ __forceinline__ void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

)XSMM";

// the part that depends on the shape (macros XM, XN, XK, ...); XGROUPED: instantiated once per shape inside a namespace
const char* const SMM_JIT_SHAPE = R"XSMM(
// (XLOWP 1, i16 -> i32: the k pairs stay packed -- one v_dot2_i32_i16 per pair -- so the kernel's k runs over pairs)
constexpr int M = XM, N = XN, K = (1 == XLOWP) ? (XK / 2) : XK;
// XPACK items that follow each other in memory (tight strided batches) are handled by a wave at a time: their operands are
// one contiguous piece (wider, fully used loads even when a single item is a few hundred bytes or not 16-byte sized), the
// 64 lanes split into XPACK groups with one item each. The host passes strides and the count in units of XPACK items.
constexpr int G = XPACK;
// Leading dimensions as in memory. LDS always holds the tight images; only the transfers know about the gaps: an operand is
// fetched as the one span of memory it occupies (gaps included), elements in the gaps are dropped on the way into LDS and
// never written on the way out.
constexpr int LDA = XLDA, LDB = XLDB, LDC = XLDC;
constexpr bool TIGHT_A = (LDA == M), TIGHT_B = (LDB == (XTRANSB ? N : K)), TIGHT_C = (LDC == M);
constexpr int AE1 = LDA * (K - 1) + M, BE1 = XTRANSB ? (LDB * (K - 1) + N) : (LDB * (N - 1) + K), CE1 = LDC * (N - 1) + M; // span of one item's operand
constexpr int CT1 = M * N;                                          // tight C image of one item
constexpr int AE = G * AE1, BE = G * BE1, CE = G * CE1;            // ... of what a wave handles at a time (G > 1: tight only)
constexpr int TS = (int)sizeof(T);
#if (2 == XRUNS)
// work-group form: 4 waves share one product; a wave owns NQ = ceil(N/4) columns of C, its 64 lanes are 16 (along m) x 4
constexpr int UT = 256;                                            // threads that stream one item's operands
constexpr int TGM = 16, TGN = 4;
constexpr int NQ = (N + 3) / 4;
constexpr int TM = (M + TGM - 1) / TGM, TN = (NQ + TGN - 1) / TGN;
constexpr int NPAD = 3 * NQ + TGN * TN;                            // highest column index a lane may touch, plus one
#else
// wave form: one wave per item, 8 x 8 lanes (XPACK items: 64 / XPACK lanes each)
constexpr int UT = 64;
constexpr int TGM = (G >= 4) ? ((G >= 16) ? 2 : 4) : 8, TGN = 64 / (G * TGM);
constexpr int TM = (M + TGM - 1) / TGM, TN = (N + TGN - 1) / TGN;
constexpr int NPAD = TGN * TN;
#endif
// widest access (in elements) that every item of a strided batch is aligned for
constexpr int vw(int elems) { return (0 == (elems * TS) % 16) ? 16 / TS : ((0 == (elems * TS) % 8) ? 8 / TS : 1); }
// XSCALAR: index/pointer batches guarantee element alignment only
constexpr int VA = XSCALAR ? 1 : vw(AE), VB = XSCALAR ? 1 : vw(BE), VC = (XSCALAR || !TIGHT_C) ? 1 : vw(CE);
constexpr int NLA = (AE + UT * VA - 1) / (UT * VA), NLB = (BE + UT * VB - 1) / (UT * VB), NLC = (CE + UT * VC - 1) / (UT * VC);
// LDS strides: A as [k][M] (lanes with equal ty read the same words, lanes with different tx adjacent ones);
// B as [n][KP] (TRANS_B: [k][N]) with KP chosen so that the column groups of one instruction fall into different banks
// (TN * TS / 4 a multiple of 16 -- fp64, eight columns per lane -- has no such stride: an odd one then)
constexpr int pick_kp() { if (0 == (TN * (TS / 4)) % 16) return K | 1; int kp = K; while (0 == (TN * kp * (TS / 4)) % 16) ++kp; return kp; }
constexpr int KP = pick_kp();
constexpr int AS1 = ((K * M + TGM * TM + 3) / 4) * 4;                 // per item
constexpr int BS1 = XTRANSB ? (((K * N + NPAD + 3) / 4) * 4) : ((NPAD * KP + 3) / 4) * 4;
constexpr int AS_SIZE = G * AS1, BS_SIZE = G * BS1;
constexpr int CS_SIZE = ((G * CT1 + 3) / 4) * 4;                     // the items' C blocks stay contiguous (flat copy in and out)
// (wave run form: C is in LDS only while a run is opened or closed, when no operand image is alive -- its image lies over theirs;
// a third less LDS per wave is a third more chains resident per CU)
constexpr bool C_OVER_AB = (1 == XRUNS);
constexpr int WAVE_LDS = C_OVER_AB ? ((AS_SIZE + BS_SIZE > CS_SIZE) ? (AS_SIZE + BS_SIZE) : CS_SIZE) : (AS_SIZE + BS_SIZE + CS_SIZE); // elements (wave form)
constexpr int WG_BUF = AS_SIZE + BS_SIZE;                            // elements per operand buffer (work-group form)
constexpr int WG_NBUF = (2 * WG_BUF * TS <= 65536) ? 2 : 1;          // double-buffered when 64 KiB allow

template<int V> struct Vec { typedef T type __attribute__((ext_vector_type(V))); };
template<> struct Vec<1> { typedef T type; };

// Operands are in global memory, but their addresses come out of a run-time choice between three addressing modes (one of
// them loads the pointer): left generic, every access becomes a FLAT instruction, which also counts on lgkmcnt -- each wait
// for an LDS operation would then wait for the prefetched operands as well. Hence the explicit address space.
template<int V, int NL, int E> __device__ __forceinline__ void load_flat(const T* p, int lane, T (&r)[NL][V])
{
  const XGLOBAL T* const g = (const XGLOBAL T*)p;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    // lanes past the end fetch the last piece again (never used): no divergent control flow around the loads, so the
    // compiler's count of what is in flight at a wait stays exact instead of falling back to "everything"
    const int e0 = (UT * j + lane) * V, e = (e0 < E - V) ? e0 : (E - V);
    if constexpr (1 == V) r[j][0] = __builtin_nontemporal_load(g + e);
    else {
      const typename Vec<V>::type v = __builtin_nontemporal_load(reinterpret_cast<const XGLOBAL typename Vec<V>::type*>(g + e));
#pragma unroll
      for (int q = 0; q < V; ++q) r[j][q] = v[q];
    }
  }
}

// registers -> LDS (A as stored, B with the padded row stride); `lane` is the thread's index among the UT streaming threads
__device__ __forceinline__ void park_ab(T* As, T* Bs, int lane, const T (&ra)[NLA][VA], const T (&rb)[NLB][VB])
{
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
#pragma unroll
    for (int q = 0; q < VA; ++q) {
      const int e = (UT * j + lane) * VA + q;
      if (e < AE) {
        if (TIGHT_A) As[(1 == G) ? e : ((e / AE1) * AS1 + (e % AE1))] = ra[j][q];
        else { const int k = e / LDA, m = e - k * LDA; if (m < M) As[k * M + m] = ra[j][q]; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
#pragma unroll
    for (int q = 0; q < VB; ++q) {
      const int e = (UT * j + lane) * VB + q;
      if (e < BE) {
        const int g = (1 == G) ? 0 : (e / BE1), r = (1 == G) ? e : (e % BE1);
        if (XTRANSB) { if (TIGHT_B) Bs[g * BS1 + r] = rb[j][q]; else { const int k = r / LDB, n = r - k * LDB; if (n < N) Bs[k * N + n] = rb[j][q]; } }
        else { const int n = r / LDB, k = r - n * LDB; if (k < K) Bs[g * BS1 + n * KP + k] = rb[j][q]; }
      }
    }
  }
}
__device__ __forceinline__ void park_c(T* Cs, int lane, const T (&rc)[NLC][VC])
{
#pragma unroll
  for (int j = 0; j < NLC; ++j) {
#pragma unroll
    for (int q = 0; q < VC; ++q) {
      const int e = (64 * j + lane) * VC + q;
      if (e < CE) { if (TIGHT_C) Cs[e] = rc[j][q]; else { const int n = e / LDC, m = e - n * LDC; if (m < M) Cs[n * M + m] = rc[j][q]; } }
    }
  }
}
#if XLOWP
// 16-bit inputs (XLOWP 1: i16, T = int; 3: bf16, T = float), stored as the reference's low-precision kernels expect them: A in
// pairs of k (a[(k/2)*M*2 + m*2 + k%2]), B column-major -- both are sequences of 32-bit k pairs. They are fetched as such and
// widened on the way into LDS, where the images are the ones of the fp32 / int kernels; everything after that is shared.
constexpr int PA1 = (M * XK) / 2, PB1 = (XK * N) / 2;              // k pairs per operand and item (XK is even)
constexpr int PA = G * PA1, PB = G * PB1;                          // ... of the XPACK items a wave handles at a time (back to back in memory)
constexpr int VPA = (0 == (PA * 4) % 16 && !XSCALAR) ? 4 : 1, VPB = (0 == (PB * 4) % 16 && !XSCALAR) ? 4 : 1;
constexpr int NPA = (PA + 64 * VPA - 1) / (64 * VPA), NPB = (PB + 64 * VPB - 1) / (64 * VPB);
template<int V> struct PVec { typedef unsigned type __attribute__((ext_vector_type(V))); };
template<> struct PVec<1> { typedef unsigned type; };
template<int V, int NL, int E> __device__ __forceinline__ void load_pairs(const unsigned* p, int lane, unsigned (&r)[NL][V])
{
  const XGLOBAL unsigned* const g = (const XGLOBAL unsigned*)p;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (64 * j + lane) * V;
    if (e < E) {
      if constexpr (1 == V) r[j][0] = __builtin_nontemporal_load(g + e);
      else {
        const typename PVec<V>::type v = __builtin_nontemporal_load(reinterpret_cast<const XGLOBAL typename PVec<V>::type*>(g + e));
#pragma unroll
        for (int q = 0; q < V; ++q) r[j][q] = v[q];
      }
    }
  }
}
__device__ __forceinline__ T widen_lo(unsigned p) { return (1 == XLOWP) ? (T)(int)(short)(p & 0xFFFFu) : (T)__uint_as_float(p << 16); }
__device__ __forceinline__ T widen_hi(unsigned p) { return (1 == XLOWP) ? (T)(int)(short)(p >> 16) : (T)__uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ void park_pairs(T* As, T* Bs, int lane, const unsigned (&ra)[NPA][VPA], const unsigned (&rb)[NPB][VPB])
{
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
#pragma unroll
    for (int q = 0; q < VPA; ++q) {
      const int e = (64 * j + lane) * VPA + q;
      if (e < PA) {
        const int it = e / PA1, e1 = e - it * PA1, sp = e1 / M, m = e1 - sp * M;
        T* const Ai = As + it * AS1;
        if (1 == XLOWP) Ai[sp * M + m] = (T)ra[j][q]; // packed pair
        else { Ai[(2 * sp) * M + m] = widen_lo(ra[j][q]); Ai[(2 * sp + 1) * M + m] = widen_hi(ra[j][q]); }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NPB; ++j) {
#pragma unroll
    for (int q = 0; q < VPB; ++q) {
      const int e = (64 * j + lane) * VPB + q;
      if (e < PB) {
        const int it = e / PB1, e1 = e - it * PB1, n = e1 / (XK / 2), sp = e1 - n * (XK / 2);
        T* const Bi = Bs + it * BS1;
        if (1 == XLOWP) Bi[n * KP + sp] = (T)rb[j][q];
        else { Bi[n * KP + 2 * sp] = widen_lo(rb[j][q]); Bi[n * KP + 2 * sp + 1] = widen_hi(rb[j][q]); }
      }
    }
  }
}
#if (2 == XLOWP)
// bf16 -> bf16: C travels as 32-bit pairs of bf16 as well (M is a multiple of 16: a pair never straddles a column); the sums
// are float, a result is the upper half of the float (truncation, as the reference's harness does)
constexpr int PC = G * ((M * N) / 2); // (the items' C blocks are contiguous in memory and in the wave's C buffer alike)
constexpr int VPC = (0 == (PC * 4) % 16 && !XSCALAR) ? 4 : 1;
constexpr int NPC = (PC + 64 * VPC - 1) / (64 * VPC);
template<int V> __device__ __forceinline__ void store_pvec(const unsigned* v, XGLOBAL unsigned* dst)
{
  typename PVec<V>::type w;
#pragma unroll
  for (int q = 0; q < V; ++q) w[q] = v[q];
  __builtin_nontemporal_store(w, reinterpret_cast<XGLOBAL typename PVec<V>::type*>(dst));
}
template<> __device__ __forceinline__ void store_pvec<1>(const unsigned* v, XGLOBAL unsigned* dst) { __builtin_nontemporal_store(v[0], dst); }
__device__ __forceinline__ void park_c_pairs(T* Cs, int lane, const unsigned (&rc)[NPC][VPC])
{
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
#pragma unroll
    for (int q = 0; q < VPC; ++q) { const int e = (64 * j + lane) * VPC + q; if (e < PC) { Cs[2 * e] = widen_lo(rc[j][q]); Cs[2 * e + 1] = widen_hi(rc[j][q]); } }
  }
}
__device__ __forceinline__ void store_c_pairs(T* Cs, unsigned* pc, int lane, int tx, int ty, const T (&acc)[TM][TN], T* Ci)
{ // (Ci: this lane's item inside the wave's C buffer Cs)
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) Ci[n * M + m] = acc[i][j]; }
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const int e = (64 * j + lane) * VPC;
    if (e < PC) {
      unsigned v[VPC];
#pragma unroll
      for (int q = 0; q < VPC; ++q) v[q] = (__float_as_uint(Cs[2 * (e + q)]) >> 16) | (__float_as_uint(Cs[2 * (e + q) + 1]) & 0xFFFF0000u);
      store_pvec<VPC>(v, (XGLOBAL unsigned*)pc + e);
    }
  }
  wave_lds_sync();
}
#endif
#endif
// acc(i,j) = fma(A(m,k), B(k,n), acc(i,j)) for k ascending: the reference's per-element chain
__device__ __forceinline__ void multiply(const T* As, const T* Bs, int tx, int ncol0, T (&acc)[TM][TN])
{
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    T av[TM], bv[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[i] = As[k * M + tx * TM + i];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = XTRANSB ? Bs[k * N + ncol0 + j] : Bs[(ncol0 + j) * KP + k];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = xfma(av[i], bv[j], acc[i][j]);
    }
  }
}
#if XRUNS
// The run forms keep few waves per CU busy (a run is a sequential chain), so nothing hides the LDS round trip between an
// operand read and its fma. Software pipeline over groups of KU k-steps: the reads of group g+1 are issued before the fmas
// of group g (two register sets, fully unrolled: all indices are compile-time constants).
constexpr int KU = (2 == XRUNS) ? 4 : 2;
constexpr int NG = (K + KU - 1) / KU;
__device__ __forceinline__ void read_group(const T* As, const T* Bs, int tx, int ncol0, int g, T (&av)[KU][TM], T (&bv)[KU][TN])
{
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int k = g * KU + u;
    if (k < K) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[u][i] = As[k * M + tx * TM + i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[u][j] = XTRANSB ? Bs[k * N + ncol0 + j] : Bs[(ncol0 + j) * KP + k];
    }
  }
}
__device__ __forceinline__ void fma_group(int g, const T (&av)[KU][TM], const T (&bv)[KU][TN], T (&acc)[TM][TN])
{
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    if (g * KU + u < K) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = xfma(av[u][i], bv[u][j], acc[i][j]);
      }
    }
  }
}
__device__ __forceinline__ void multiply_pipelined(const T* As, const T* Bs, int tx, int ncol0, T (&acc)[TM][TN])
{
  T a0[KU][TM], b0[KU][TN], a1[KU][TM], b1[KU][TN];
  read_group(As, Bs, tx, ncol0, 0, a0, b0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (0 == (g & 1)) { if (g + 1 < NG) read_group(As, Bs, tx, ncol0, g + 1, a1, b1); fma_group(g, a0, b0, acc); }
    else { if (g + 1 < NG) read_group(As, Bs, tx, ncol0, g + 1, a0, b0); fma_group(g, a1, b1, acc); }
  }
}
#endif

__device__ __forceinline__ void acc_from_c(const T* Cs, int tx, int ty, T (&acc)[TM][TN], bool zero)
{
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int m = tx * TM + i, n = ty * TN + j;
      acc[i][j] = (!zero && m < M && n < N) ? Cs[n * M + m] : (T)0;
    }
  }
}
// C leaves through LDS so that the stores are flat and coalesced
// (Ci: this lane's item inside the wave's C buffer Cs; the two are the same unless XPACK > 1)
__device__ __forceinline__ void store_c(T* Cs, T* pc, int lane, int tx, int ty, const T (&acc)[TM][TN], T* Ci = nullptr)
{
  if (nullptr == Ci) Ci = Cs;
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) Ci[n * M + m] = acc[i][j]; }
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < NLC; ++j) {
    const int e = (64 * j + lane) * VC;
    if (e < CE) {
      if constexpr (!TIGHT_C) { const int n = e / LDC, m = e - n * LDC; if (m < M) __builtin_nontemporal_store(Cs[n * M + m], (XGLOBAL T*)pc + e); } // (VC == 1)
      else if constexpr (1 == VC) __builtin_nontemporal_store(Cs[e], (XGLOBAL T*)pc + e);
      else __builtin_nontemporal_store(*reinterpret_cast<const typename Vec<VC>::type*>(Cs + e), reinterpret_cast<XGLOBAL typename Vec<VC>::type*>((XGLOBAL T*)pc + e));
    }
  }
  wave_lds_sync();
}

// The streaming form defers the stores of an item to the top of the next iteration, in front of that iteration's loads:
// vmcnt retires loads and stores in the order of issue, so stores issued right behind the arithmetic -- younger than the
// prefetched operands -- made the wait for the operands a wait for the previous item's C to reach memory as well (measured
// on tools/probe/mfma_wave.hip: 7.3 -> 2.2 us per item and wave).
__device__ __forceinline__ void c_to_lds(T* Ci, int tx, int ty, const T (&acc)[TM][TN])
{
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) Ci[n * M + m] = acc[i][j]; }
  }
  wave_lds_sync();
}
__device__ __forceinline__ void lds_to_mem(const T* Cs, T* pc, int lane)
{
#pragma unroll
  for (int j = 0; j < NLC; ++j) {
    if constexpr (!TIGHT_C) { const int e = 64 * j + lane; if (e < CE) { const int n = e / LDC, m = e - n * LDC; if (m < M) __builtin_nontemporal_store(Cs[n * M + m], (XGLOBAL T*)pc + e); } } // (VC == 1)
    else { // (lanes past the end repeat the last piece: same data to the same place)
      const int e0 = (64 * j + lane) * VC, e = (e0 < CE - VC) ? e0 : (CE - VC);
      if constexpr (1 == VC) __builtin_nontemporal_store(Cs[e], (XGLOBAL T*)pc + e);
      else __builtin_nontemporal_store(*reinterpret_cast<const typename Vec<VC>::type*>(Cs + e), reinterpret_cast<XGLOBAL typename Vec<VC>::type*>((XGLOBAL T*)pc + e));
    }
  }
}

#if XRUNS
__device__ __forceinline__ void xatomic_add(double* p, double v) { (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v); }
__device__ __forceinline__ void xatomic_add(float* p, float v) { (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v); }
// a segment's sum joins C: through LDS so that the atomics of a wave cover consecutive addresses
__device__ __forceinline__ void atomic_c(T* Cs, T* pc, int lane, int tx, int ty, const T (&acc)[TM][TN])
{
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) Cs[n * M + m] = acc[i][j]; }
  }
  wave_lds_sync();
  for (int e = lane; e < M * N; e += 64) xatomic_add(pc + (TIGHT_C ? e : ((e / M) * LDC + (e % M))), Cs[e]);
  wave_lds_sync();
}
#endif

)XSMM";

// walking a batch run by run (shared by the register-tiled run forms below and the matrix-core run form): needs T, DevAddr /
// resolve, XRUNS, XSPLIT, XDEPTH
const char* const SMM_JIT_CHAIN = R"XSMM(
// bit l: item first + l starts a run, i.e. its C differs from its predecessor's (item 0 always does)
__device__ __forceinline__ unsigned long long head_mask(const DevAddr& ad, long long first, int lane, long long batch)
{
  const long long j = first + lane;
  bool head = false;
  if (j < batch) head = (0 == j) || (resolve<T>(ad.c, ad.ic, ad.sc, ad, j - 1) != resolve<T>(ad.c, ad.ic, ad.sc, ad, j));
  return __ballot(head);
}

// Items [first, end) a wave / work-group walks when it is dealt the chunk [chunk, chunk + 64): from the chunk's first run
// head through the chunk's other runs to the end of the run that is still open at the chunk's end (the next head at or
// beyond chunk + 64). Returns false if the chunk holds no head (an earlier chunk's walk covers it).
__device__ __forceinline__ bool chain_of_chunk(const DevAddr& ad, long long chunk, int lane, long long batch,
                                               unsigned long long& heads, long long& first, long long& end)
{
  heads = head_mask(ad, chunk, lane, batch);
  if (0 == heads) return false;
  first = chunk + (__ffsll((long long)heads) - 1);
  end = chunk + 64;
  while (end < batch) {
    const unsigned long long mk = head_mask(ad, end, lane, batch);
    if (0 != mk) { end += (__ffsll((long long)mk) - 1); break; }
    end += 64;
  }
  if (end > batch) end = batch;
  return true;
}
__device__ __forceinline__ bool is_head(unsigned long long heads, long long chunk, long long i)
{ // beyond the chunk the walk only continues through non-heads
  return (i < chunk + 64) && (0 != ((heads >> (int)(i - chunk)) & 1ULL));
}

#if XRUNS
// When a batch is not walked run by run. (1) C blocks repeat out of order: no walk in batch order can keep them apart.
// (2) XSPLIT -- relaxed order, the caller's reference path is itself multi-threaded with a lock per C
// (libxsmm_gemm_batch_omp, mmbatch with several tasks) -- and the batch is a handful of long runs, i.e. of sequential
// chains that leave the chip idle. Such a batch is cut into segments of `len` items; a wave sums the products of a run
// inside its segment from zero and adds the sum to C with floating-point atomics (one C-sized atomic update per segment
// instead of a C read and write per run). Returns 0 when the batch is walked run by run, in batch order.
__device__ __forceinline__ int segment_len(const int* flags, long long batch)
{
  if (nullptr == flags) return 0;
  const long long peers = (0 < flags[3]) ? flags[3] : 1; // batches that run beside this one in the same (grouped) launch, about as large
  if (0 == flags[1]) {
    const long long runs = batch - flags[0];
    // (the runs of all batches of the launch fill the chip together: 27 CP2K groups of 172 runs each are 4644 chains, and walked in
    // batch order -- no C-sized atomic update per segment -- they are the faster form: 0.81 against 1.02 ms, profiles/r3_cp2k_stacks.txt)
    if (!XSPLIT || runs * peers >= 2048 || 16 * runs > batch) return 0;
  }
  const long long len = (batch * peers + 4095) / 4096;
  return (int)(len < 8 ? 8 : (len > 64 ? 64 : len));
}
#endif

constexpr int D = XDEPTH; // products whose operands are in flight (register stages)

// Work-group barrier that orders LDS traffic only. __syncthreads() carries a work-group-scope fence, which on gfx9 drains
// vmcnt -- i.e. every global load in flight, the whole register ring of prefetched products -- before each s_barrier and
// turns the D-deep prefetch into depth one (measured: 2.2 us per 32^3 product instead of well under one).
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Operand addresses of 64 consecutive items, one item per lane. Index and pointer batches need a load per item and operand
// before the operand itself can be requested; done item by item that load is a full memory round trip on the critical
// path of a run. Here 64 of them travel together and single addresses are picked with v_readlane.
struct AddrWindow { long long base; unsigned long long a, b; };
__device__ __forceinline__ void window_fill(AddrWindow& w, const DevAddr& ad, long long base, int lane, long long end)
{
  const long long j = base + lane;
  w.base = base;
  w.a = (j < end) ? (unsigned long long)resolve<const T>(ad.a, ad.ia, ad.sa, ad, j) : 0ULL;
  w.b = (j < end) ? (unsigned long long)resolve<const T>(ad.b, ad.ib, ad.sb, ad, j) : 0ULL;
  // The addresses may be load results (pointer batches). Settle that here, once per 64 items: otherwise the compiler must
  // assume a pending load at every later v_readlane and drains vmcnt -- the prefetched operands -- before each item.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(w.a), "+v"(w.b));
}
__device__ __forceinline__ const T* window_pick(unsigned long long v, int src)
{
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, src), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src);
  return (const T*)(((unsigned long long)hi << 32) | lo);
}
// addresses of item i (i never decreases between calls; the window slides forward in steps that keep i inside)
#define WINDOW_AB(W, I, PA, PB) \
  if ((I) - (W).base >= 64) window_fill((W), ad, (I), lane, end); \
  const int w_src_ = __builtin_amdgcn_readfirstlane((int)((I) - (W).base)); \
  const T* const PA = window_pick((W).a, w_src_); const T* const PB = window_pick((W).b, w_src_)

)XSMM";

const char* const SMM_JIT_SHAPE_KERNELS = R"XSMM(
#if (2 == XRUNS)
// Work-group form for long runs (CP2K stacks, batch-reduce): the four waves of a work-group share every product of a run.
// All 256 threads stream A and B of the products D ahead into register stages while the current one is multiplied out of
// LDS (two operand buffers when 64 KiB allow: one barrier per product); wave v owns columns [v*NQ, v*NQ + NQ) of C and
// keeps them in registers for the whole run, each element still receiving its products in batch order, k ascending --
// the sequential reference's chain. A long run is a latency chain (HBM round trip per product): the depth of the register
// ring, not the thread count, is what shortens it.
#if XGROUPED
__device__ XENTRY_ATTR void xsmm_entry(const DevAddr& ad, long long batch, unsigned xbid, unsigned xgrid, T* lds)
{
#else
extern "C" __global__ __launch_bounds__(256) void xsmm_smm_op(DevAddr ad, long long batch)
{
  __shared__ __attribute__((aligned(16))) T lds[WG_NBUF * WG_BUF];
  const unsigned xbid = blockIdx.x, xgrid = gridDim.x;
#endif
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int tx = lane & (TGM - 1), ty = lane >> 4;
  const int ncol0 = wave * NQ + ty * TN;
  if (nullptr != ad.flags) {
    if (8LL * ad.flags[0] < 7LL * batch) return;                     // runs shorter than 8 on average: the wave form owns it
    if (0 != segment_len(ad.flags, batch)) return;                   // not walked run by run: the wave form cuts it into segments
  }
  T ra[D][NLA][VA], rb[D][NLB][VB];
  int buf = 0;
  // every work-group takes a contiguous range of chunks (dealt round-robin, runs of a uniform length of 128, 256, ... items
  // would put all run heads into the chunks of every 2nd, 4th, ... work-group)
  const long long nchunks = (batch + 63) / 64, cpw = (nchunks + xgrid - 1) / xgrid;
  const long long c_end = ((long long)(xbid + 1) * cpw < nchunks) ? (long long)(xbid + 1) * cpw : nchunks;
  for (long long ci = (long long)xbid * cpw; ci < c_end; ++ci) {
    const long long chunk = ci * 64;
    unsigned long long heads; long long first, end;
    if (!chain_of_chunk(ad, chunk, lane, batch, heads, first, end)) continue; // identical in the four waves
    AddrWindow win; window_fill(win, ad, first, lane, end);
#pragma unroll
    for (int s = 0; s < D; ++s) {
      if (first + s < end) {
        WINDOW_AB(win, first + s, pa, pb);
        load_flat<VA, NLA, AE>(pa, t, ra[s]);
        load_flat<VB, NLB, BE>(pb, t, rb[s]);
      }
    }
    T acc[TM][TN];
    T* pc = nullptr;
    for (long long i0 = first; i0 < end; i0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const long long i = i0 + s;
        if (i < end) {
          if (is_head(heads, chunk, i)) { // item i opens a run: C goes straight between HBM and registers
            if (nullptr != pc) {
#pragma unroll
              for (int ii = 0; ii < TM; ++ii) {
#pragma unroll
                for (int j = 0; j < TN; ++j) { const int m = tx * TM + ii, n = ncol0 + j; if (m < M && ty * TN + j < NQ && n < N) ((XGLOBAL T*)pc)[n * LDC + m] = acc[ii][j]; }
              }
            }
            pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, i);
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) {
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const int m = tx * TM + ii, n = ncol0 + j;
                acc[ii][j] = (!XBETA0 && m < M && ty * TN + j < NQ && n < N) ? ((const XGLOBAL T*)pc)[n * LDC + m] : (T)0;
              }
            }
          }
          T* const As = lds + buf * WG_BUF;
          T* const Bs = As + AS_SIZE;
          if (1 == WG_NBUF) lds_barrier(); // single buffer: the previous product's readers must be done
          park_ab(As, Bs, t, ra[s], rb[s]);
          if (i + D < end) {
            WINDOW_AB(win, i + D, pa, pb);
            load_flat<VA, NLA, AE>(pa, t, ra[s]);
            load_flat<VB, NLB, BE>(pb, t, rb[s]);
          }
          lds_barrier();
          multiply_pipelined(As, Bs, tx, ncol0, acc);
          buf ^= (WG_NBUF - 1);
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < TM; ++ii) {
#pragma unroll
      for (int j = 0; j < TN; ++j) { const int m = tx * TM + ii, n = ncol0 + j; if (m < M && ty * TN + j < NQ && n < N) ((XGLOBAL T*)pc)[n * LDC + m] = acc[ii][j]; }
    }
  }
}
#else
#if XGROUPED
__device__ XENTRY_ATTR void xsmm_entry(const DevAddr& ad, long long batch, unsigned xbid, unsigned xgrid, T* lds)
{
  if ((int)(threadIdx.x >> 6) >= XWAVES) return; // (the grouped kernel's work-groups have four waves)
#else
extern "C" __global__ __launch_bounds__(64 * XWAVES) void xsmm_smm_op(DevAddr ad, long long batch)
{
  __shared__ __attribute__((aligned(16))) T lds[XWAVES * WAVE_LDS];
  const unsigned xbid = blockIdx.x, xgrid = gridDim.x;
#endif
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int LPI = 64 / G;                      // lanes per item
  const int grp = lane / LPI, tx = (lane % LPI) % TGM, ty = (lane % LPI) / TGM;
  T* const As = lds + wave * WAVE_LDS;
  T* const Bs = As + AS_SIZE;
  T* const Cs = C_OVER_AB ? As : (Bs + BS_SIZE);
  const long long w = (long long)xbid * XWAVES + wave, W = (long long)xgrid * XWAVES;
#if XRUNS
  // Consecutive items that share one C form a run (CP2K stacks, batch-reduce): the wave that owns the run's first item
  // keeps C in registers and adds the products in batch order -- what the reference's sequential loop does. Chunks of 64
  // items are dealt round-robin to the waves; run heads are found 64 items at a time (one item per lane, __ballot). A
  // wave walks its chain (see chain_of_chunk) item by item with the operands of the next D items in flight (and the C of
  // the next item, if it starts a run). A batch of distinct C blocks is the special case "all heads". Batches that
  // cannot (or need not) be walked in batch order go segment by segment, see segment_len.
  const int seg = segment_len(ad.flags, batch);
  if (XHASWG && nullptr != ad.flags && 0 == seg && 8LL * ad.flags[0] >= 7LL * batch) return; // runs of 8 and more on average: the work-group form owns this batch
  const long long step = (0 != seg ? seg : 64);
  T ra[D][NLA][VA], rb[D][NLB][VB], rc[NLC][VC];
  // every wave takes a contiguous range of chunks (see the work-group form)
  const long long nchunks = (batch + step - 1) / step, cpw = (nchunks + W - 1) / W;
  const long long c_end = ((w + 1) * cpw < nchunks) ? (w + 1) * cpw : nchunks;
  for (long long ci = w * cpw; ci < c_end; ++ci) {
    const long long chunk = ci * step;
    unsigned long long heads; long long first, end;
    if (0 != seg) { // a segment is walked from its first item, whoever opened the run it starts in
      heads = head_mask(ad, chunk, lane, batch) | 1ULL;
      first = chunk; end = (chunk + seg < batch ? chunk + seg : batch);
    }
    else if (!chain_of_chunk(ad, chunk, lane, batch, heads, first, end)) continue;
    AddrWindow win; window_fill(win, ad, first, lane, end);
#pragma unroll
    for (int s = 0; s < D; ++s) {
      if (first + s < end) {
        WINDOW_AB(win, first + s, pa, pb);
        load_flat<VA, NLA, AE>(pa, lane, ra[s]);
        load_flat<VB, NLB, BE>(pb, lane, rb[s]);
      }
    }
    if (!XBETA0 && 0 == seg) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, first), lane, rc);
    T acc[TM][TN];
    T* pc = nullptr;
    for (long long i0 = first; i0 < end; i0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const long long i = i0 + s;
        if (i < end) {
          if (is_head(heads, chunk, i)) { // item i opens a run: close the previous one, take over its C
            if (0 != seg) {
              if (nullptr != pc) atomic_c(Cs, pc, lane, tx, ty, acc);
              pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, i);
              acc_from_c(Cs, tx, ty, acc, true);
            }
            else {
              if (nullptr != pc) store_c(Cs, pc, lane, tx, ty, acc);
              pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, i);
              if (!XBETA0) { park_c(Cs, lane, rc); wave_lds_sync(); }
              acc_from_c(Cs, tx, ty, acc, XBETA0);
              if (!XBETA0) wave_lds_sync(); // (the operand images are parked over C's next)
            }
          }
          park_ab(As, Bs, lane, ra[s], rb[s]);
          if (i + D < end) {
            WINDOW_AB(win, i + D, pa, pb);
            load_flat<VA, NLA, AE>(pa, lane, ra[s]);
            load_flat<VB, NLB, BE>(pb, lane, rb[s]);
          }
          if (!XBETA0 && 0 == seg && i + 1 < end && is_head(heads, chunk, i + 1)) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, i + 1), lane, rc);
          wave_lds_sync();
          multiply_pipelined(As, Bs, tx, ty * TN, acc);
          wave_lds_sync();
        }
      }
    }
    if (0 != seg) atomic_c(Cs, pc, lane, tx, ty, acc);
    else store_c(Cs, pc, lane, tx, ty, acc);
  }
#else
  if (w >= batch) return;
#if XLOWP
  unsigned ra[NPA][VPA], rb[NPB][VPB];
#if (2 == XLOWP)
  unsigned rc[NPC][VPC];
#else
  T rc[NLC][VC];
#endif
  // (the batch addresses 16-bit operands in elements of 16 bits -- strides, index arrays -- whichever the addressing mode)
  load_pairs<VPA, NPA, PA>((const unsigned*)resolve<const unsigned short>(ad.a, ad.ia, ad.sa, ad, w), lane, ra);
  load_pairs<VPB, NPB, PB>((const unsigned*)resolve<const unsigned short>(ad.b, ad.ib, ad.sb, ad, w), lane, rb);
#else
  T ra[NLA][VA], rb[NLB][VB], rc[NLC][VC];
  load_flat<VA, NLA, AE>(resolve<const T>(ad.a, ad.ia, ad.sa, ad, w), lane, ra);
  load_flat<VB, NLB, BE>(resolve<const T>(ad.b, ad.ib, ad.sb, ad, w), lane, rb);
#endif
#if (2 == XLOWP)
  if (!XBETA0) load_pairs<VPC, NPC, PC>((const unsigned*)resolve<const unsigned short>(ad.c, ad.ic, ad.sc, ad, w), lane, rc);
#else
  if (!XBETA0) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, w), lane, rc);
#endif
#if (2 != XLOWP)
  T* pend = nullptr; // C block whose result waits in the LDS image
#endif
  for (long long item = w; item < batch; item += W) {
#if (2 == XLOWP)
    unsigned* const pc = (unsigned*)resolve<unsigned short>(ad.c, ad.ic, ad.sc, ad, item);
#else
    T* const pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, item);
    if (XDEFER) {
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this item's operands; the only other instructions in flight are older stores
      if (nullptr != pend) { lds_to_mem(Cs, pend, lane); wave_lds_sync(); }
    }
#endif
#if XLOWP
    park_pairs(As, Bs, lane, ra, rb);
#else
    park_ab(As, Bs, lane, ra, rb);
#endif
#if (2 == XLOWP)
    if (!XBETA0) park_c_pairs(Cs, lane, rc);
#else
    if (!XBETA0) park_c(Cs, lane, rc);
#endif
    // ---- next item's loads go out before this item's arithmetic
    const long long next = item + W;
    if (next < batch) {
#if XLOWP
      load_pairs<VPA, NPA, PA>((const unsigned*)resolve<const unsigned short>(ad.a, ad.ia, ad.sa, ad, next), lane, ra);
      load_pairs<VPB, NPB, PB>((const unsigned*)resolve<const unsigned short>(ad.b, ad.ib, ad.sb, ad, next), lane, rb);
#else
      load_flat<VA, NLA, AE>(resolve<const T>(ad.a, ad.ia, ad.sa, ad, next), lane, ra);
      load_flat<VB, NLB, BE>(resolve<const T>(ad.b, ad.ib, ad.sb, ad, next), lane, rb);
#endif
#if (2 == XLOWP)
      if (!XBETA0) load_pairs<VPC, NPC, PC>((const unsigned*)resolve<const unsigned short>(ad.c, ad.ic, ad.sc, ad, next), lane, rc);
#else
      if (!XBETA0) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, next), lane, rc);
#endif
    }
    wave_lds_sync();
    T acc[TM][TN];
    acc_from_c(Cs + grp * CT1, tx, ty, acc, XBETA0);
    multiply(As + grp * AS1, Bs + grp * BS1, tx, ty * TN, acc);
#if (2 == XLOWP)
    store_c_pairs(Cs, pc, lane, tx, ty, acc, Cs + grp * CT1);
#else
    if (XDEFER) { c_to_lds(Cs + grp * CT1, tx, ty, acc); pend = pc; }
    else store_c(Cs, pc, lane, tx, ty, acc, Cs + grp * CT1);
#endif
  }
#if (2 != XLOWP)
  if (XDEFER && nullptr != pend) lds_to_mem(Cs, pend, lane);
#endif
#endif
}
#endif
)XSMM";

// ---- 32 < max(M, N) <= 64 on the matrix cores with ONE WAVE PER ITEM (tight operands, M and K multiples of four) --------
// The work-group-per-item kernels (kernels/smm_mfma_wg.inc) synchronise four waves twice per item and keep few items in
// flight per CU; below 64^3 they are latency-bound (40^3: 46 % of the HBM peak). Here a wave owns an item: C, A and B
// arrive as whole 16-byte chunks of the contiguous arrays (lanes past the end of an array repeat its last chunk: no
// divergent control flow around memory instructions, so the compiler's wait counts stay exact), C is redistributed through
// LDS into the tile layout of the accumulators, A and B are parked as LDS images that the operand fetches of
// v_mfma_{f32,f64}_16x16x4 read conflict-free (A: [k][MS], rows of 16 lanes at a stride that spreads the four k of a fetch
// over the banks; B: [n][KSD] with KSD/VEC odd), the result goes back through LDS and leaves as whole lines. Both
// instructions are k-ordered fma chains (tools/probe/mfma_f64_chain.hip; kernels/sparse.hip uses the fp32 one): C equals
// the reference's per-element chain bit for bit. The stores of item i are issued at the top of iteration i + 1, before
// the loads of item i + 2: vmcnt retires in the order of issue, so whenever the wave waits for its operands nothing younger
// is in flight and it never waits for a store to reach memory (tools/probe/mfma_wave.hip: 7.3 -> 2.2 us per item and wave).
const char* const SMM_JIT_MFMA_WAVE_BODY = R"XSMM(
constexpr int M = XM, N = XN, K = XK;
constexpr int TS = (int)sizeof(T);
// XVEC: elements per memory access -- a 16-byte chunk where the shape and the operands' alignment allow it (M a multiple of the
// chunk, K of four, 16-byte aligned items), else 1: any M, N, K and element-aligned operands (index and pointer batches)
constexpr int VEC = XVEC;
template<int W> struct VecOf { typedef T type __attribute__((ext_vector_type(W))); };
template<> struct VecOf<1> { typedef T type; };
typedef VecOf<VEC>::type V;
typedef T ACC __attribute__((ext_vector_type(4)));
typedef float ACC32 __attribute__((ext_vector_type(4)));
typedef double ACC64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ACC32 xmfma(float a, float b, ACC32 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ACC64 xmfma(double a, double b, ACC64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
constexpr bool F64 = (8 == TS);
// K is padded to a multiple of four with A = -0 and B = +0: the product -0 is the identity of the addition for every sum,
// signs of zeros included, so the chain stays the reference's bit for bit (as in kernels/smm_mfma_wg.inc)
constexpr int MI = (M + 15) / 16, NI = (N + 15) / 16, KS = (K + 3) / 4, KP4 = 4 * KS;
constexpr int MS = (M <= 16) ? 16 : (M <= 48 ? 48 : 64);
constexpr bool ASWZ = (64 == MS);
constexpr int kstride() { int s = ((KP4 + VEC - 1) / VEC); if (0 == (s & 1)) ++s; return s * VEC; }
constexpr int KSD = kstride();
// TRANS_B (memory holds B^T: element (k, n) at k * ldb + n): the image is k-major like A's, [k][NS], fetched the same way
constexpr int NS = (N <= 16) ? 16 : (N <= 48 ? 48 : 64);
constexpr bool BSWZ = (64 == NS);
// image of C: column stride such that the four column groups of a tile access fall into different banks
constexpr int cstride() { int s = M; for (;; s += VEC) { if (F64 ? (16 == s % 32) : (4 == s % 16 || 12 == s % 16)) break; } return s; }
constexpr int CSD = cstride();
constexpr int C_ELEMS = N * CSD;
constexpr int A_ELEMS = (C_ELEMS > KP4 * MS) ? C_ELEMS : KP4 * MS, B_ELEMS = XTRANSB ? KP4 * NS : N * KSD;
#if XLOWP
// bf16 inputs (XLOWP 3: fp32 result, 2: bf16 result) as the reference's low-precision kernels store them: A in pairs of k
// (a[(k/2)*M*2 + m*2 + k%2]), B column-major -- both sequences of 32-bit k pairs. A 16-byte chunk is four pairs (A: four
// rows m of one pair of k; B: eight consecutive k of one column); the halves are widened on the way into the same fp32 LDS
// images (a bf16 product is exact in fp32, so fma(a, b, acc) is the gold loop's product-then-add, samples/xgemm/kernel.c).
typedef unsigned UV __attribute__((ext_vector_type(4)));
constexpr int LDA = M, LDB = K, LDC = M;
constexpr bool TIGHT = true;
constexpr int NCA = M * K / 8, NCB = K * N / 8;                    // chunks per operand
constexpr int NCC = (2 == XLOWP) ? (M * N / 8) : (M * N / 4);
static_assert(0 == M % 4 && 0 == K % 8 && (2 != XLOWP || 0 == M % 8), "shape");
__device__ __forceinline__ V widen_lo(UV p) { return V{ __uint_as_float(p[0] << 16), __uint_as_float(p[1] << 16), __uint_as_float(p[2] << 16), __uint_as_float(p[3] << 16) }; }
__device__ __forceinline__ V widen_hi(UV p) { return V{ __uint_as_float(p[0] & 0xFFFF0000u), __uint_as_float(p[1] & 0xFFFF0000u), __uint_as_float(p[2] & 0xFFFF0000u), __uint_as_float(p[3] & 0xFFFF0000u) }; }
// eight consecutive bf16 (four pairs) as two vectors of four floats in memory order
__device__ __forceinline__ void widen8(UV p, V& v0, V& v1)
{
  v0 = V{ __uint_as_float(p[0] << 16), __uint_as_float(p[0] & 0xFFFF0000u), __uint_as_float(p[1] << 16), __uint_as_float(p[1] & 0xFFFF0000u) };
  v1 = V{ __uint_as_float(p[2] << 16), __uint_as_float(p[2] & 0xFFFF0000u), __uint_as_float(p[3] << 16), __uint_as_float(p[3] & 0xFFFF0000u) };
}
#else
// Leading dimensions as in memory (gaps only in the element-wise build): an operand is fetched as the one span of memory it
// occupies, elements in the gaps are dropped on the way into the images and never written on the way out.
constexpr int LDA = XLDA, LDB = XLDB, LDC = XLDC;
constexpr bool TIGHT = (LDA == M && LDB == (XTRANSB ? N : K) && LDC == M);
constexpr int NCA = (LDA * (K - 1) + M) / VEC, NCB = (XTRANSB ? (LDB * (K - 1) + N) : (LDB * (N - 1) + K)) / VEC, NCC = (LDC * (N - 1) + M) / VEC;
typedef V UV;
static_assert(1 == VEC || (TIGHT && 0 == M % VEC && 0 == K % 4 && (!XTRANSB || 0 == N % VEC)), "shape");
#endif
constexpr int CA = (NCA + 63) / 64, CB = (NCB + 63) / 64, CC = (NCC + 63) / 64;
__device__ __forceinline__ int clampi(int v, int hi) { return v < hi ? v : hi; }
// row of C a lane's accumulator register r belongs to (the fp32 and fp64 instructions differ)
#define XNROW(r) (F64 ? (lq + 4 * (r)) : (4 * lq + (r)))
// what the batch's strides and index arrays count: elements of the operands' types (16-bit inputs, and a bf16 result, in 16-bit elements)
#if XLOWP
typedef unsigned short XOPI;
#if (2 == XLOWP)
typedef unsigned short XOPC;
#else
typedef T XOPC;
#endif
#else
typedef T XOPI; typedef T XOPC;
#endif

#if (2 != XNSPLIT)
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XWPE))) void xsmm_smm_op(DevAddr ad, long long batch, int runlen)
{
  extern __shared__ __align__(16) unsigned char smem[];
  T* const As = reinterpret_cast<T*>(smem);
  T* const Bs = As + A_ELEMS;
  T* const Cs = As;                      // the image of C shares the place of A's (never alive together)
  T* const dummy = Bs + B_ELEMS;         // a word per lane for the writes of lanes outside C
  const int lane = threadIdx.x, l16 = lane & 15, lq = lane >> 4;
  (void)runlen;
  UV ra[CA], rb[CB], rc[CC]; // (strides of the batch are in 32-bit words: the host halves those of 16-bit operands)
  auto load_ab = [&](long long item) {
    const XGLOBAL UV* const pa = (const XGLOBAL UV*)resolve<const XOPI>(ad.a, ad.ia, ad.sa, ad, item);
    const XGLOBAL UV* const pb = (const XGLOBAL UV*)resolve<const XOPI>(ad.b, ad.ib, ad.sb, ad, item);
#pragma unroll
    for (int j = 0; j < CA; ++j) ra[j] = __builtin_nontemporal_load(pa + clampi(64 * j + lane, NCA - 1));
#pragma unroll
    for (int j = 0; j < CB; ++j) rb[j] = __builtin_nontemporal_load(pb + clampi(64 * j + lane, NCB - 1));
  };
  auto load_c = [&](long long item) {
    const XGLOBAL UV* const pc = (const XGLOBAL UV*)resolve<const XOPC>(ad.c, ad.ic, ad.sc, ad, item);
#pragma unroll
    for (int j = 0; j < CC; ++j) rc[j] = __builtin_nontemporal_load(pc + clampi(64 * j + lane, NCC - 1));
  };
  auto store_c = [&](long long item) { // the image of C -> memory, whole lines
    XGLOBAL UV* const pc = (XGLOBAL UV*)resolve<XOPC>(ad.c, ad.ic, ad.sc, ad, item);
#pragma unroll
    for (int j = 0; j < CC; ++j) {
      const int ch = clampi(64 * j + lane, NCC - 1);
#if (2 == XLOWP)
      const int e = ch * 8, n = e / M, m = e % M; // a bf16 result is the upper half of the float sum (truncation, as the harness does)
      const V v0 = *reinterpret_cast<const V*>(Cs + n * CSD + m), v1 = *reinterpret_cast<const V*>(Cs + n * CSD + m + 4);
      const UV w = UV{ (__float_as_uint(v0[0]) >> 16) | (__float_as_uint(v0[1]) & 0xFFFF0000u), (__float_as_uint(v0[2]) >> 16) | (__float_as_uint(v0[3]) & 0xFFFF0000u),
                       (__float_as_uint(v1[0]) >> 16) | (__float_as_uint(v1[1]) & 0xFFFF0000u), (__float_as_uint(v1[2]) >> 16) | (__float_as_uint(v1[3]) & 0xFFFF0000u) };
      __builtin_nontemporal_store(w, pc + ch);
#else
      const int e = ch * VEC, n = e / LDC, m = e % LDC;
      if (TIGHT || m < M) __builtin_nontemporal_store(*reinterpret_cast<const UV*>(Cs + n * CSD + (TIGHT ? m : clampi(m, M - 1))), pc + ch);
#endif
    }
  };
  long long item = blockIdx.x, prev = -1;
  if (item >= batch) return;
  load_ab(item);
  if (!XBETA0) load_c(item);
  if (KP4 > K) { // the padding of B's image: written once (its place is not shared)
    if (XTRANSB) { for (int e = lane; e < (KP4 - K) * NS; e += 64) Bs[K * NS + e] = T(0); }
    else { for (int e = lane; e < N * (KP4 - K); e += 64) Bs[(e / (KP4 - K)) * KSD + K + e % (KP4 - K)] = T(0); }
  }
  for (;;) {
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this item's operands (the only other instructions in flight are older stores)
    if (0 <= prev) { store_c(prev); wave_lds_sync(); }
    // (the places in the images a lane parks its pieces at do not change from item to item: left to the compiler they are all
    // computed once and kept in registers -- a hundred of them in the element-wise form, which then spills. Recomputed per item
    // from a lane index the compiler cannot see through.)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    ACC acc[NI][MI];
    if (!XBETA0) {
#pragma unroll
      for (int j = 0; j < CC; ++j) {
        const int ch = clampi(64 * j + ln, NCC - 1);
#if (2 == XLOWP)
        const int e = ch * 8, n = e / M, m = e % M;
        V v0, v1; widen8(rc[j], v0, v1);
        *reinterpret_cast<V*>(Cs + n * CSD + m) = v0; *reinterpret_cast<V*>(Cs + n * CSD + m + 4) = v1;
#else
        const int e = ch * VEC, n = e / LDC, m = e % LDC;
        *reinterpret_cast<UV*>((TIGHT || m < M) ? Cs + n * CSD + m : dummy + lane) = rc[j];
#endif
      }
      wave_lds_sync();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = clampi(16 * ni + XNROW(r), N - 1), m = clampi(16 * mi + l16, M - 1);
            acc[ni][mi][r] = Cs[n * CSD + m];
          }
      wave_lds_sync();
    }
    else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = ACC{ 0, 0, 0, 0 };
    }
    if (KP4 > K) { // the padding rows of A's image (the image of C has been lying over them)
      for (int e = lane; e < (KP4 - K) * MS; e += 64) As[K * MS + e] = -T(0);
    }
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      const int ch = clampi(64 * j + ln, NCA - 1);
#if XLOWP
      const int w = ch * 4, sp = w / M, m = w % M, k0 = 2 * sp, k1 = 2 * sp + 1; // four rows of the k pair (2 sp, 2 sp + 1)
      *reinterpret_cast<V*>(As + k0 * MS + (ASWZ ? (m ^ ((k0 & 3) << 4)) : m)) = widen_lo(ra[j]);
      *reinterpret_cast<V*>(As + k1 * MS + (ASWZ ? (m ^ ((k1 & 3) << 4)) : m)) = widen_hi(ra[j]);
#else
      const int e = ch * VEC, k = e / LDA, m = e % LDA;
      *reinterpret_cast<V*>((TIGHT || m < M) ? As + k * MS + (ASWZ ? (m ^ ((k & 3) << 4)) : m) : dummy + lane) = ra[j];
#endif
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int ch = clampi(64 * j + ln, NCB - 1);
#if XLOWP
      const int e = ch * 8, n = e / K, k = e % K; // eight consecutive k of column n
      V v0, v1; widen8(rb[j], v0, v1);
      *reinterpret_cast<V*>(Bs + n * KSD + k) = v0; *reinterpret_cast<V*>(Bs + n * KSD + k + 4) = v1;
#else
#if XTRANSB
      const int e = ch * VEC, k = e / LDB, n = e % LDB;
      *reinterpret_cast<V*>((TIGHT || n < N) ? Bs + k * NS + (BSWZ ? (n ^ ((k & 3) << 4)) : n) : dummy + lane) = rb[j];
#else
      const int e = ch * VEC, n = e / LDB, k = e % LDB;
      *reinterpret_cast<V*>((TIGHT || k < K) ? Bs + n * KSD + k : dummy + lane) = rb[j];
#endif
#endif
    }
    const long long next = item + gridDim.x;
    if (next < batch) { load_ab(next); if (!XBETA0) load_c(next); }
    wave_lds_sync();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      T af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) { const int m = 16 * mi + l16; af[mi] = As[(4 * ks + lq) * MS + (ASWZ ? (m ^ (lq << 4)) : m)]; }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = XTRANSB ? (16 * ni + l16) : clampi(16 * ni + l16, N - 1);
        bf[ni] = XTRANSB ? Bs[(4 * ks + lq) * NS + (BSWZ ? (n ^ (lq << 4)) : n)] : Bs[n * KSD + 4 * ks + lq];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = xmfma(bf[ni], af[mi], acc[ni][mi]);
    }
    wave_lds_sync(); // (the images are dead now)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 16 * ni + XNROW(r), m = 16 * mi + l16;
          const bool inside = (16 * ni + 15 < N || n < N) && (16 * mi + 15 < M || m < M);
          T* const dst = inside ? Cs + n * CSD + m : dummy + lane;
          *dst = acc[ni][mi][r];
        }
    wave_lds_sync();
    prev = item;
    if (next >= batch) break;
    item = next;
  }
  store_c(prev);
}
#else
// XNSPLIT 2: the columns of C are worked on in two halves against one image of A (items whose images would not leave room
// for four waves per CU: fp64 56^3 needs 55 KB). Per half: its C and B columns arrive, C goes through the place of B's half
// image into the accumulators, B is parked there, after the arithmetic the result waits there for the deferred stores. The
// operands of the next half -- or the next item's A and first half -- are in flight meanwhile. A's image is tight (stride M:
// two-way bank conflicts on its fetches, irrelevant next to 64-cycle fp64 matrix instructions).
constexpr int AMS = M;
constexpr int NH = N / 2, NIH = (NH + 15) / 16;
constexpr int CSH = M;
constexpr int BH_ELEMS = (NH * KSD > NH * CSH) ? NH * KSD : NH * CSH;
constexpr int CBH = (K * NH / VEC + 63) / 64, CCH = (M * NH / VEC + 63) / 64;
static_assert(0 == N % 2 && 0 == (K * NH) % VEC && 0 == (M * NH) % VEC && 0 == XLOWP, "shape");
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XWPE))) void xsmm_smm_op(DevAddr ad, long long batch, int runlen)
{
  extern __shared__ __align__(16) unsigned char smem[];
  T* const As = reinterpret_cast<T*>(smem);
  T* const Bs = As + K * AMS;
  T* const Cs = Bs;                      // the half image of C shares the place of B's half (never alive together)
  T* const dummy = Bs + BH_ELEMS;
  const int lane = threadIdx.x, l16 = lane & 15, lq = lane >> 4;
  (void)runlen;
  V ra[CA], rb[CBH], rc[CCH];
  auto load_a = [&](long long item) {
    const XGLOBAL V* const pa = (const XGLOBAL V*)resolve<const T>(ad.a, ad.ia, ad.sa, ad, item);
#pragma unroll
    for (int j = 0; j < CA; ++j) ra[j] = __builtin_nontemporal_load(pa + clampi(64 * j + lane, M * K / VEC - 1));
  };
  auto load_bc = [&](long long item, int h) {
    const XGLOBAL V* const pb = (const XGLOBAL V*)(resolve<const T>(ad.b, ad.ib, ad.sb, ad, item) + h * (K * NH));
#pragma unroll
    for (int j = 0; j < CBH; ++j) rb[j] = __builtin_nontemporal_load(pb + clampi(64 * j + lane, K * NH / VEC - 1));
    if (!XBETA0) {
      const XGLOBAL V* const pc = (const XGLOBAL V*)(resolve<const T>(ad.c, ad.ic, ad.sc, ad, item) + h * (M * NH));
#pragma unroll
      for (int j = 0; j < CCH; ++j) rc[j] = __builtin_nontemporal_load(pc + clampi(64 * j + lane, M * NH / VEC - 1));
    }
  };
  auto store_c = [&](long long item, int h) {
    XGLOBAL V* const pc = (XGLOBAL V*)(resolve<T>(ad.c, ad.ic, ad.sc, ad, item) + h * (M * NH));
#pragma unroll
    for (int j = 0; j < CCH; ++j) {
      const int ch = clampi(64 * j + lane, M * NH / VEC - 1), e = ch * VEC, n = e / M, m = e % M;
      __builtin_nontemporal_store(*reinterpret_cast<const V*>(Cs + n * CSH + m), pc + ch);
    }
  };
  long long item = blockIdx.x, prev = -1; int prevh = 0;
  if (item >= batch) return;
  load_a(item); load_bc(item, 0);
  for (;;) {
    const long long next = item + gridDim.x;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this half's operands (the only other instructions in flight are older stores)
      if (0 <= prev) { store_c(prev, prevh); wave_lds_sync(); }
      ACC acc[NIH][MI];
      if (!XBETA0) {
#pragma unroll
        for (int j = 0; j < CCH; ++j) {
          const int ch = clampi(64 * j + lane, M * NH / VEC - 1), e = ch * VEC, n = e / M, m = e % M;
          *reinterpret_cast<V*>(Cs + n * CSH + m) = rc[j];
        }
        wave_lds_sync();
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = clampi(16 * ni + XNROW(r), NH - 1), m = clampi(16 * mi + l16, M - 1);
              acc[ni][mi][r] = Cs[n * CSH + m];
            }
        wave_lds_sync();
      }
      else {
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = ACC{ 0, 0, 0, 0 };
      }
      if (0 == h) {
#pragma unroll
        for (int j = 0; j < CA; ++j) {
          const int ch = clampi(64 * j + lane, M * K / VEC - 1), e = ch * VEC, k = e / M, m = e % M;
          *reinterpret_cast<V*>(As + k * AMS + m) = ra[j];
        }
      }
#pragma unroll
      for (int j = 0; j < CBH; ++j) {
        const int ch = clampi(64 * j + lane, K * NH / VEC - 1), e = ch * VEC, n = e / K, k = e % K;
        *reinterpret_cast<V*>(Bs + n * KSD + k) = rb[j];
      }
      if (0 == h) load_bc(item, 1);
      else if (next < batch) { load_a(next); load_bc(next, 0); }
      wave_lds_sync();
#pragma unroll 2
      for (int ks = 0; ks < KS; ++ks) { // (limited unrolling: with the whole k loop unrolled the operand fetches are hoisted and the registers spill)
        T af[MI], bf[NIH];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { const int m = clampi(16 * mi + l16, M - 1); af[mi] = As[(4 * ks + lq) * AMS + m]; }
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni) { const int n = clampi(16 * ni + l16, NH - 1); bf[ni] = Bs[n * KSD + 4 * ks + lq]; }
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = xmfma(bf[ni], af[mi], acc[ni][mi]);
      }
      wave_lds_sync();
#pragma unroll
      for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 16 * ni + XNROW(r), m = 16 * mi + l16;
            const bool inside = (16 * ni + 15 < NH || n < NH) && (16 * mi + 15 < M || m < M);
            T* const dst = inside ? Cs + n * CSH + m : dummy + lane;
            *dst = acc[ni][mi][r];
          }
      wave_lds_sync();
      prev = item; prevh = h;
    }
    if (next >= batch) break;
    item = next;
  }
  store_c(prev, prevh);
}
#endif
)XSMM";

const char* const SMM_JIT_MFMA_RUNS_CONST = R"XSMM(
// ---- run form on the matrix cores: M, N <= 32, K <= 64, fp32 / fp64, any addressing mode, any leading dimensions -------------
// The register-tiled run form above spends a product's time on LDS round trips (every k step fetches TM + TN operands per lane
// for TM x TN fma: the LDS pipe is half busy, the waves wait on it) and holds 250-300 registers for a 32^3 fp64 product -- two
// waves per SIMD, spills in the grouped kernel. Here a wave still owns a run (C in the accumulators across its products, the
// products added in batch order), but:
//   * A never touches LDS: v_mfma_{f32,f64}_16x16x4 wants, in lane (l16, lq), the element A(m = 16 mi + l16, k = 4 ks + lq) --
//     sixteen lanes along a column of A, i.e. contiguous in memory for any lda. The fragments of a product are loaded straight
//     from global memory into the registers the instructions read, one k step at a time, and the registers of k step ks are
//     refilled with the NEXT product's right behind the instructions that consumed them;
//   * B (lanes along n would stride through memory) arrives as a flat, coalesced array and is parked in LDS as [n][KSD], KSD
//     chosen so that a fragment fetch is conflict-free;
//   * C goes straight between memory and the accumulators when a run opens and closes (a lane's elements of one register are
//     sixteen consecutive rows of a column).
// Both instructions are k-ordered fma chains with one rounding per product (tools/probe/mfma_f64_chain.hip), K is padded to a
// multiple of four with A = -0 / B = +0 (the product -0 is the identity of the addition for every sum, signs of zeros
// included): C equals the reference's sequential chain bit for bit, as in the register-tiled form.
constexpr int M = XM, N = XN, K = XK;
constexpr int LDA = XLDA, LDB = XLDB, LDC = XLDC;
constexpr int TS = (int)sizeof(T);
constexpr bool F64 = (8 == TS);
typedef T ACC __attribute__((ext_vector_type(4)));
typedef float ACC32 __attribute__((ext_vector_type(4)));
typedef double ACC64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ACC32 xmfma(float a, float b, ACC32 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ACC64 xmfma(double a, double b, ACC64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// K goes through the registers and B's image in NCH chunks of KC (a multiple of four) -- one chunk up to K = 64; beyond that only the
// streaming form (the run form keeps a whole product's fragments of A in flight)
// (beyond 64 in chunks of at most 32: two chunk variants of fragments and address arithmetic alive at once cost registers)
constexpr int NCH = (XSTREAM && K > 64) ? (K + 31) / 32 : 1;
constexpr int KC = 4 * (((K + NCH - 1) / NCH + 3) / 4);
constexpr int MI = (M + 15) / 16, NI = (N + 15) / 16, KS = KC / 4, KP4 = KC;
// row stride of B's image: fp64 fragments are fetched sixteen lanes (one k, sixteen n) at a time -- an odd stride spreads them
// over all bank pairs; fp32 fragments thirty-two lanes (two k) at a time -- a stride of 2 mod 4 keeps the two k apart as well
constexpr int KSD = F64 ? (KP4 + 1) : (KP4 + 2);
constexpr int BE = LDB * (N - 1) + K;                  // span of B in memory
constexpr int NLB = (1 == NCH) ? (BE + 63) / 64 : (N * KC + 63) / 64; // one chunk: the flat span; several: N x KC elements per chunk
constexpr int WAVE_LDS = ((N * KSD + 3) / 4) * 4;      // elements
#define XNROW(r) (F64 ? (lq + 4 * (r)) : (4 * lq + (r)))
__device__ __forceinline__ int clampi(int v, int hi) { return v < hi ? v : hi; }
__device__ __forceinline__ void xatomic_add(double* p, double v) { (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v); }
__device__ __forceinline__ void xatomic_add(float* p, float v) { (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v); }
)XSMM";

const char* const SMM_JIT_MFMA_RUNS_KERNEL = R"XSMM(
// fragment of A for k step ks, tile mi (rows beyond M repeat row M - 1: they only reach rows of C that are never stored)
__device__ __forceinline__ T load_a_frag(const T* pa, int ks, int moff, int lq, int k0 = 0)
{ // (k0: first k of the chunk, a constant after unrolling)
  const int k = k0 + 4 * ks + lq;
  const XGLOBAL T* const g = (const XGLOBAL T*)pa;
  if (k0 + 4 * ks + 3 < K) return __builtin_nontemporal_load(g + k * LDA + moff);
  const T v = __builtin_nontemporal_load(g + clampi(k, K - 1) * LDA + moff); // (no divergent control flow around the load)
  return (k < K) ? v : -T(0);
}
__device__ __forceinline__ void load_b_flat(const T* pb, int lane, T (&rb)[NLB])
{
  const XGLOBAL T* const g = (const XGLOBAL T*)pb;
#pragma unroll
  for (int j = 0; j < NLB; ++j) rb[j] = __builtin_nontemporal_load(g + clampi(64 * j + lane, BE - 1));
}
__device__ __forceinline__ void park_b(T* Bs, int lane, const T (&rb)[NLB])
{
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int e = 64 * j + lane, n = e / LDB, k = e - n * LDB;
    if (e < BE && k < K) Bs[n * KSD + k] = rb[j];
  }
}
// chunk c of B (several chunks: K > 64): the KC x N window, lanes along k -- runs of KC elements of a column; k beyond K: +0
__device__ __forceinline__ void load_b_chunk(const T* pb, int c, int lane, T (&rb)[NLB])
{
  const XGLOBAL T* const g = (const XGLOBAL T*)pb;
  asm volatile("" : "+v"(lane)); // (the places are recomputed per chunk: hoisted out of the loop over the items they cost a register each)
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int e = 64 * j + lane, n = e / KC, kk = e - n * KC, k = c * KC + kk;
    const T v = __builtin_nontemporal_load(g + clampi(n, N - 1) * LDB + clampi(k, K - 1)); // (no control flow around the load)
    rb[j] = (k < K) ? v : T(0);
  }
}
__device__ __forceinline__ void park_b_chunk(T* Bs, int lane, const T (&rb)[NLB])
{
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int e = 64 * j + lane, n = e / KC, kk = e - n * KC;
    if (e < N * KC) Bs[n * KSD + kk] = rb[j];
  }
}
__device__ __forceinline__ void acc_load(const T* pc, int l16, int lq, ACC (&acc)[NI][MI])
{
  const XGLOBAL T* const g = (const XGLOBAL T*)pc;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ni + XNROW(r), m = 16 * mi + l16;
        acc[ni][mi][r] = g[clampi(n, N - 1) * LDC + clampi(m, M - 1)];
      }
}
__device__ __forceinline__ void acc_store(T* pc, int l16, int lq, const ACC (&acc)[NI][MI], bool atomic)
{
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ni + XNROW(r), m = 16 * mi + l16;
        if (n < N && m < M) {
          if (atomic) xatomic_add(pc + n * LDC + m, acc[ni][mi][r]);
          else ((XGLOBAL T*)pc)[n * LDC + m] = acc[ni][mi][r];
        }
      }
}

#if XHANDWAIT
// The compiler's own wait counts are useless for the software pipeline of the run form: in front of the first matrix instruction
// of a product it waits for ALL of the product's A fragments (vmcnt(#B loads just issued)) -- the ones requested at the very end of
// the previous product included: a memory round trip per product, 1.8 us for a 32^3 fp64 product whose matrix instructions take 0.9.
// The loads whose results live across iterations are therefore issued as inline assembly -- invisible to the compiler's
// scoreboard -- and waited for by hand: vmcnt retires in the order of issue, and the number of loads issued after the ones a step
// needs is a constant of the pipeline (below). Loads the compiler does see (C at a run head, address windows) are drained inside
// their branch, so that nothing pending escapes into the loop.
__device__ __forceinline__ T xload_nt(const XGLOBAL T* p, int off)
{ // off: bytes, a constant below 4096 after unrolling (the instruction's immediate: one address register pair serves a 4 KiB window)
  T v;
  if constexpr (F64) asm volatile("global_load_dwordx2 %0, %1, off offset:%2 nt" : "=v"(v) : "v"(p), "n"(off));
  else asm volatile("global_load_dword %0, %1, off offset:%2 nt" : "=v"(v) : "v"(p), "n"(off));
  return v;
}
// B's flat span, 64 elements per load; the last load of a span that does not end on 64 elements repeats the span's last element
__device__ __forceinline__ void xload_b(const T* pb, int lane, T (&rb)[NLB])
{
  const XGLOBAL T* const b0 = (const XGLOBAL T*)pb + lane;
#pragma unroll
  for (int jj = 0; jj < NLB; ++jj) {
    const int byte = jj * 64 * TS;
    if (64 * jj + 63 < BE) rb[jj] = xload_nt(b0 + (byte / 4096) * (4096 / TS), byte % 4096);
    else rb[jj] = xload_nt((const XGLOBAL T*)pb + clampi(64 * jj + lane, BE - 1), 0);
  }
}
// A's fragments of k step ks (rows beyond M repeat row M - 1, k beyond K repeats k = K - 1: replaced by -0 when used)
__device__ __forceinline__ void xload_a(const T* pa, int ks, const int (&moff)[MI], int lq, T (&af)[MI])
{
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int byte = 4 * ks * LDA * TS;
    if (4 * ks + 3 < K) af[mi] = xload_nt((const XGLOBAL T*)pa + lq * LDA + moff[mi] + (byte / 4096) * (4096 / TS), byte % 4096);
    else af[mi] = xload_nt((const XGLOBAL T*)pa + clampi(4 * ks + lq, K - 1) * LDA + moff[mi], 0);
  }
}
constexpr int vmc(int n) { return n < 63 ? n : 63; } // (the counter has six bits; fewer is stricter, never wrong)
template<int CNT> __device__ __forceinline__ void wait_on(T& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template<int CNT> __device__ __forceinline__ void wait_on(T& a, T& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT)); }
#endif

#if XSTREAM
// Every item owns its C (strided batches; index / pointer batches under the caller's promise): a wave per item, walking the batch with
// a stride of all resident waves. The next item's operands -- B's flat array, C in the layout of the accumulators, A's fragments
// behind the instructions that free their registers -- are in flight during this item's arithmetic; the stores of an item are
// issued behind the loads of the next (older loads: the wait for them is not a wait for the stores).
extern "C" __global__ __launch_bounds__(64 * XWAVES) void xsmm_smm_op(DevAddr ad, long long batch)
{
  __shared__ __attribute__((aligned(16))) T lds[XWAVES * WAVE_LDS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, l16 = lane & 15, lq = lane >> 4;
  T* const Bs = lds + wave * WAVE_LDS;
  const long long w = (long long)blockIdx.x * XWAVES + wave, W = (long long)gridDim.x * XWAVES;
  if (w >= batch) return;
  if (1 == NCH && KP4 > K) { // the padding of B's image (+0): written once, never parked over (several chunks: the loads deliver it)
    for (int e = lane; e < N * (KP4 - K); e += 64) Bs[(e / (KP4 - K)) * KSD + K + e % (KP4 - K)] = T(0);
  }
  int moff[MI], boff[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) moff[mi] = clampi(16 * mi + l16, M - 1);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) boff[ni] = clampi(16 * ni + l16, N - 1) * KSD + lq;
  T af[KS][MI], rb[NLB];
  ACC cn[NI][MI];
  T* pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, w);
  const T* pa_cur = resolve<const T>(ad.a, ad.ia, ad.sa, ad, w);
  const T* pb_cur = resolve<const T>(ad.b, ad.ib, ad.sb, ad, w);
  if (1 == NCH) load_b_flat(pb_cur, lane, rb); else load_b_chunk(pb_cur, 0, lane, rb);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[ks][mi] = load_a_frag(pa_cur, ks, moff[mi], lq);
  if (!XBETA0) acc_load(pc, l16, lq, cn);
  for (long long item = w; item < batch; item += W) {
    T* const pc_cur = pc;
    ACC acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = XBETA0 ? ACC{ 0, 0, 0, 0 } : cn[ni][mi];
    // (no control flow around the loads: the last item of a wave fetches its own operands once more, never used)
    const long long next = (item + W < batch) ? (item + W) : item;
    const T* const pa_next = resolve<const T>(ad.a, ad.ia, ad.sa, ad, next);
    const T* const pb_next = resolve<const T>(ad.b, ad.ib, ad.sb, ad, next);
#pragma unroll
    for (int c = 0; c < NCH; ++c) { // (unrolled: which chunk comes next, and where K ends, are constants)
      if (1 == NCH) park_b(Bs, lane, rb); else park_b_chunk(Bs, lane, rb);
      const bool last = (c + 1 == NCH);
      const T* const pan = last ? pa_next : pa_cur;
      const int cnext = last ? 0 : c + 1;
      if (1 == NCH) load_b_flat(pb_next, lane, rb); else load_b_chunk(last ? pb_next : pb_cur, cnext, lane, rb);
      if (last) {
        pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, next);
        if (!XBETA0) acc_load(pc, l16, lq, cn);
      }
      wave_lds_sync();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        T bf[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[ni] = Bs[boff[ni] + 4 * ks];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = xmfma(bf[ni], af[ks][mi], acc[ni][mi]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[ks][mi] = load_a_frag(pan, ks, moff[mi], lq, cnext * KC);
      }
      wave_lds_sync(); // (B's image is parked over next)
    }
    pa_cur = pa_next; pb_cur = pb_next;
    acc_store(pc_cur, l16, lq, acc, false);
  }
}
#else
#if XGROUPED
__device__ XENTRY_ATTR void xsmm_entry(const DevAddr& ad, long long batch, unsigned xbid, unsigned xgrid, T* lds)
{
#if !XTILEWG
  if ((int)(threadIdx.x >> 6) >= XWAVES) return; // (the grouped kernel's work-groups may have more waves than this body uses)
#endif
#else
extern "C" __global__ __launch_bounds__(64 * XWAVES) void xsmm_smm_op(DevAddr ad, long long batch)
{
  __shared__ __attribute__((aligned(16))) T lds[XWAVES * WAVE_LDS];
  const unsigned xbid = blockIdx.x, xgrid = gridDim.x;
#endif
  // (XTILEWG: the waves of a work-group are the tiles of C of ONE walk -- each wave is the only wave of its body, lds is its own)
  const int wave = XTILEWG ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, l16 = lane & 15, lq = lane >> 4;
  T* const Bs = lds + wave * WAVE_LDS;
  const long long w = (long long)xbid * XWAVES + wave, W = (long long)xgrid * XWAVES;
  // (walking the batch: chunks of 64 items, run heads, segments -- see the register-tiled wave form)
  const int seg = segment_len(ad.flags, batch);
  const long long step = (0 != seg ? seg : 64);
  const long long nchunks = (batch + step - 1) / step, cpw = (nchunks + W - 1) / W;
  const long long c_end = ((w + 1) * cpw < nchunks) ? (w + 1) * cpw : nchunks;
  if (w * cpw >= c_end) return;
  if (KP4 > K) { // the padding of B's image (+0): written once, never parked over
    for (int e = lane; e < N * (KP4 - K); e += 64) Bs[(e / (KP4 - K)) * KSD + K + e % (KP4 - K)] = T(0);
  }
  int moff[MI], boff[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) moff[mi] = clampi(16 * mi + l16, M - 1);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) boff[ni] = clampi(16 * ni + l16, N - 1) * KSD + lq; // (columns beyond N repeat column N - 1: never stored)
  for (long long ci = w * cpw; ci < c_end; ++ci) {
    const long long chunk = ci * step;
    unsigned long long heads; long long first, end;
    if (0 != seg) { // a segment is walked from its first item, whoever opened the run it starts in
      heads = head_mask(ad, chunk, lane, batch) | 1ULL;
      first = chunk; end = (chunk + seg < batch ? chunk + seg : batch);
    }
    else if (!chain_of_chunk(ad, chunk, lane, batch, heads, first, end)) continue;
    AddrWindow win; window_fill(win, ad, first, lane, end);
    // D products in flight (register sets; the walk is unrolled over them so that every index is a constant): a light product's
    // arithmetic is over in a tenth of a microsecond -- one product ahead, a run of 13^3 products is a chain of memory round trips
    T af[D][KS][MI], rb[D][NLB];
#if XHANDWAIT
    constexpr int LSET = NLB + KS * MI;                                   // loads per product
    constexpr int CNT_B = vmc(KS * MI + (D - 1) * LSET);                  // issued after a product's B when it is parked
    constexpr int CNT_A = vmc((KS - 1) * MI + NLB + (D - 1) * LSET);      // issued after the fragments of k step ks when they are used (any ks)
#endif
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const long long j = (first + s < end) ? (first + s) : (end - 1);
      WINDOW_AB(win, j, pa0, pb0);
#if XHANDWAIT
      xload_b(pb0, lane, rb[s]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xload_a(pa0, ks, moff, lq, af[s][ks]);
#else
      load_b_flat(pb0, lane, rb[s]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[s][ks][mi] = load_a_frag(pa0, ks, moff[mi], lq);
#endif
    }
    ACC acc[NI][MI];
    T* pc = nullptr;
    for (long long i0 = first; i0 < end; i0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const long long i = i0 + s;
        if (i >= end) break;
        if (is_head(heads, chunk, i)) { // item i opens a run: close the previous one, take over its C
          if (nullptr != pc) acc_store(pc, l16, lq, acc, 0 != seg);
          pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, i);
          if (XBETA0 || 0 != seg) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = ACC{ 0, 0, 0, 0 };
          }
          else {
            acc_load(pc, l16, lq, acc);
#if XHANDWAIT
            // (C has arrived before the branch ends: a load left pending here would make the compiler wait for everything in
            // front of the first matrix instruction of EVERY product, head or not)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(acc[ni][mi]));
#endif
          }
        }
#if XHANDWAIT
#pragma unroll
        for (int jj = 0; jj + 1 < NLB; jj += 2) wait_on<CNT_B>(rb[s][jj], rb[s][jj + 1]);
        if (NLB & 1) wait_on<CNT_B>(rb[s][NLB - 1]);
#endif
        park_b(Bs, lane, rb[s]);
        // The operands of the product D ahead take this set's place: B right away, A behind the instructions that free its
        // registers. No control flow around the loads -- the last D products of a walk fetch the last one's operands again (never
        // used): with conditional loads the compiler loses count of what is in flight and waits for everything, the operands just
        // requested included, in front of the first matrix instruction (one memory round trip per product: 1.4 instead of 0.8 us
        // for 13^3).
        const long long inext = (i + D < end) ? (i + D) : (end - 1);
        WINDOW_AB(win, inext, pa1, pb1);
#if XHANDWAIT
        xload_b(pb1, lane, rb[s]);
#else
        load_b_flat(pb1, lane, rb[s]);
#endif
        wave_lds_sync();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          T bf[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bf[ni] = Bs[boff[ni] + 4 * ks];
#if XHANDWAIT
          if (1 == MI) wait_on<CNT_A>(af[s][ks][0]); else wait_on<CNT_A>(af[s][ks][0], af[s][ks][MI - 1]);
          T av[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) av[mi] = (4 * ks + 3 < K || 4 * ks + lq < K) ? af[s][ks][mi] : -T(0); // (K padded to four: A = -0)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = xmfma(bf[ni], av[mi], acc[ni][mi]);
          xload_a(pa1, ks, moff, lq, af[s][ks]);
#else
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = xmfma(bf[ni], af[s][ks][mi], acc[ni][mi]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) af[s][ks][mi] = load_a_frag(pa1, ks, moff[mi], lq);
#endif
        }
        wave_lds_sync(); // (B's image is parked over next)
      }
    }
    acc_store(pc, l16, lq, acc, 0 != seg);
  }
}
#endif
)XSMM";

// LDS bytes of a wave of that kernel (mirrors the constexpr arithmetic of the source); 0: the shape is not served
static size_t smm_mfma_wave_lds(int typesize, int m, int n, int k, int vec = 0, bool transb = false)
{ // vec: elements per memory access (0: a 16-byte chunk)
  if (0 == vec) vec = 16 / typesize;
  if (m > 64 || n > 64 || k > 64 || m < 1 || n < 1 || k < 1) return 0;
  if (1 != vec && (0 != m % vec || 0 != k % 4)) return 0;
  const int kp4 = 4 * ((k + 3) / 4);
  const int ms = (m <= 16) ? 16 : (m <= 48 ? 48 : 64);
  int ksd = (kp4 + vec - 1) / vec; if (0 == (ksd & 1)) ++ksd; ksd *= vec;
  int csd = m; for (;; csd += vec) { if (8 == typesize ? (16 == csd % 32) : (4 == csd % 16 || 12 == csd % 16)) break; }
  const int a_elems = (n * csd > kp4 * ms) ? n * csd : kp4 * ms;
  const int ns = (n <= 16) ? 16 : (n <= 48 ? 48 : 64);
  if (transb && 1 != vec && 0 != n % vec) return 0;
  return (size_t)(a_elems + (transb ? kp4 * ns : n * ksd) + 64) * typesize;
}
// ... of the form that works on the columns of C in two halves (0: not served)
static size_t smm_mfma_wave2_lds(int typesize, int m, int n, int k)
{
  const int vec = 16 / typesize;
  if (0 != m % vec || 0 != k % 4 || m > 64 || n > 64 || k > 64 || k < 4 || 0 != (n & 1)) return 0;
  const int nh = n / 2;
  if (0 != (k * nh) % vec || 0 != (m * nh) % vec) return 0;
  int ksd = (k + vec - 1) / vec; if (0 == (ksd & 1)) ++ksd; ksd *= vec;
  const int bh = (nh * ksd > nh * m) ? nh * ksd : nh * m;
  return (size_t)(k * m + bh + 64) * typesize;
}
// waves per SIMD the kernel is compiled for: two where LDS leaves room for eight waves per CU, else one (the register file
// then holds an item's operands, the next item's and the accumulators without spilling)
static int smm_mfma_wave_wpe(size_t lds, int typesize = 0, int m = 0, int n = 0, int k = 0, int vec = 0, int lda = 0, int ldb = 0, int ldc = 0)
{
  if (lda < m) lda = m;
  if (ldb < k) ldb = k;
  if (ldc < m) ldc = m;
  static const int env = []() { const char* e = getenv("XSMM_SMMJIT_WAVE_WPE"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }(); // developer knob
  if (0 < env) return env;
  if (8 * lds > 160u * 1024u) return 1;
  if (1 == vec) { // element-wise form: a register per element in flight plus what parking them costs (45^3 fp32 spills at two waves per SIMD)
    const int w = typesize / 4, regs = w * ((lda * (k - 1) + m + 63) / 64 + (ldb * (n - 1) + k + 63) / 64 + (ldc * (n - 1) + m + 63) / 64) + w * 4 * ((m + 15) / 16) * ((n + 15) / 16);
    if (regs > 125) return 1;
  }
  return 2;
}

// ---- the run form on the matrix cores (SMM_JIT_MFMA_RUNS_*): LDS bytes of a wave (mirrors the source); 0: shape not served
static size_t smm_mfma_runs_lds(int typesize, int m, int n, int k, int ldb, bool stream = false)
{ // stream: the streaming form (every item owns its C), which takes K beyond 64 in chunks
  if (m < 1 || n < 1 || k < 1 || m > 32 || n > 32 || k > (stream ? 256 : 64) || (4 != typesize && 8 != typesize)) return 0;
  if (ldb < k) ldb = k;
  const int nch = (stream && k > 64) ? (k + 31) / 32 : 1;
  if (1 == nch && (long long)ldb * (n - 1) + k > 64 * 40) return 0; // B's span travels through registers, an element per lane and load
  const int kc = 4 * (((k + nch - 1) / nch + 3) / 4), ksd = (8 == typesize) ? kc + 1 : kc + 2;
  return (size_t)(((n * ksd + 3) / 4) * 4) * typesize;
}
// products in flight per wave (register sets): as many as fit ~80 registers, four at most
static int smm_mfma_runs_depth(int typesize, int m, int n, int k, int ldb, bool deep = false)
{ // deep: the tiles of a batch of few runs (smm_tile_split) -- a wave or two per SIMD, the registers are there
  static const int env = []() { const char* e = getenv("XSMM_SMMJIT_MFMA_RUNS_DEPTH"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }(); // developer knob
  if (0 < env) return env > 4 ? 4 : env;
  if (ldb < k) ldb = k;
  const int regs = (typesize / 4) * (((k + 3) / 4) * ((m + 15) / 16) + (ldb * (n - 1) + k + 63) / 64);
  const int d = (deep ? 128 : 80) / (regs > 0 ? regs : 1);
  return d < 1 ? 1 : (d > 4 ? 4 : d);
}
// hand-counted waits in the software pipeline of the run form (XSMM_SMMJIT_HANDWAIT=0: the compiler's: developer knob)
static int smm_mfma_handwait()
{
  static const int env = []() { const char* e = getenv("XSMM_SMMJIT_HANDWAIT"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
  return 0 != env ? 1 : 0;
}
// XSMM_SMMJIT_TILESPLIT=2 (developer knob, re-read on every call): the tiles of a run as the waves of ONE work-group instead of work-groups
// of one wave wherever the dispatcher puts them. Measured (profiles/r3_tile_split.txt): the tiles then share a CU's memory pipeline -- 10-20 %
// faster for batches of 16 000-30 000 items in short runs (the halves of A and B two tiles share are requested from one CU), but a long run
// is walked SLOWER than by a single wave (runs of 256: 0.41 ms against 0.18 ms for the tiles as groups and 0.31 ms for a wave per run; one
// CP2K stack 0.19 against 0.14 ms): not the default.
static int smm_tile_wg()
{
  const char* const e = getenv("XSMM_SMMJIT_TILESPLIT");
  return (nullptr != e && 2 == atoi(e)) ? 1 : 0;
}
static int smm_mfma_runs_waves(size_t lds) { return (0 == lds) ? 0 : ((4 * lds <= 65536) ? 4 : ((2 * lds <= 65536) ? 2 : 1)); }
// may batch s (shared C: runs) take that form?
static bool smm_mfma_runs_ok(const SmmBatch& s)
{
  static const int on = []() { const char* e = getenv("XSMM_SMMJIT_MFMA_RUNS"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }(); // developer knob
  if (0 == on || 0 == s.use_mfma || 0 != s.lowp || 0 != s.general || 0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) return false;
  if (s.lda < s.m || s.ldb < s.k || s.ldc < s.m) return false;
  return 0 != smm_mfma_runs_lds(s.typesize, s.m, s.n, s.k, s.ldb);
}
// ... the streaming form (items that own their C; K up to 256)
static bool smm_mfma_stream_ok(const SmmBatch& s)
{
  static const int on = []() { const char* e = getenv("XSMM_SMMJIT_MFMA_RUNS"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
  if (0 == on || 0 == s.use_mfma || 0 != s.lowp || 0 != s.general || 0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) || SYNC_NONE != s.sync) return false;
  if (s.lda < s.m || s.ldb < s.k || s.ldc < s.m) return false;
  return 0 != smm_mfma_runs_lds(s.typesize, s.m, s.n, s.k, s.ldb, true);
}

// ---- shapes with 32 < M or N <= 64: one work-group (256 threads, 16 x 16) per item, K in chunks of KC through LDS ------
const char* const SMM_JIT_BIG_BODY = R"XSMM(
#define XGLOBAL __attribute__((address_space(1)))
struct DevAddr {
  const char* a; const char* b; char* c;
  const char* ia; const char* ib; const char* ic;
  long long sa, sb, sc;
  int index_base, index_stride, mode;
  const int* flags;
};
template<typename P> __device__ __forceinline__ P* resolve(const char* base, const char* idx, long long stride, const DevAddr& ad, long long i)
{
  if (0 == ad.mode) return (P*)base + i * stride;
  if (1 == ad.mode) { if (nullptr == idx) return (P*)base; const int v = *(const XGLOBAL int*)(idx + i * (long long)ad.index_stride); return (P*)base + ((long long)v - ad.index_base); }
  return *(P* const XGLOBAL*)(base + i * stride);
}
__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// barrier that orders LDS traffic only (a work-group-scope fence would drain the prefetched global loads as well)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int M = XM, N = XN, K = XK, KC = XKC;
constexpr int TM = (M + 15) / 16, TN = (N + 15) / 16;             // register tile of a thread (tx = t & 15 along m, ty = t >> 4 along n)
constexpr int NCH = (K + KC - 1) / KC;                             // k-chunks per item
constexpr int AEL = KC * M, BEL = KC * N;                          // elements of one chunk
constexpr int NLA = (AEL + 255) / 256, NLB = (BEL + 255) / 256;    // per thread
constexpr int MP = 16 * TM, NP = 16 * TN;
// LDS images of a chunk: A as [kk][MP], B as [kk][NPP]: a thread's TM resp. TN operands of one k are contiguous (one or two
// 16-byte reads each). B arrives k-contiguous unless TRANS_B, so its LDS writes stride by a row; the 16-byte row padding
// keeps that at a 4-way bank conflict on the (16x rarer) writes while the reads stay aligned.
constexpr int NPP = NP + 16 / (int)sizeof(T);
constexpr int AS = KC * MP, BS = KC * NPP;
constexpr int BUF = ((AS + BS + 3) / 4) * 4;

// chunk ch of one item: A rows k0..k0+KC (contiguous KC*M elements), B rows k0..k0+KC of every column (TRANS_B: contiguous)
__device__ __forceinline__ void load_chunk(const T* pa, const T* pb, int ch, int t, T (&ra)[NLA], T (&rb)[NLB])
{
  const int k0 = ch * KC, kc = (K - k0 < KC) ? (K - k0) : KC;
  const XGLOBAL T* const ga = (const XGLOBAL T*)pa + (long long)k0 * M;
#pragma unroll
  for (int j = 0; j < NLA; ++j) { const int e = 256 * j + t; if (e < kc * M) ra[j] = __builtin_nontemporal_load(ga + e); }
  if (XTRANSB) {
    const XGLOBAL T* const gb = (const XGLOBAL T*)pb + (long long)k0 * N;
#pragma unroll
    for (int j = 0; j < NLB; ++j) { const int e = 256 * j + t; if (e < kc * N) rb[j] = __builtin_nontemporal_load(gb + e); }
  }
  else {
    const XGLOBAL T* const gb = (const XGLOBAL T*)pb + k0;
#pragma unroll
    for (int j = 0; j < NLB; ++j) { const int e = 256 * j + t, n = e / KC, kk = e - n * KC; if (e < BEL && kk < kc) rb[j] = __builtin_nontemporal_load(gb + (long long)n * K + kk); }
  }
}
__device__ __forceinline__ void park_chunk(T* As, T* Bs, int ch, int t, const T (&ra)[NLA], const T (&rb)[NLB])
{
  const int k0 = ch * KC, kc = (K - k0 < KC) ? (K - k0) : KC;
#pragma unroll
  for (int j = 0; j < NLA; ++j) { const int e = 256 * j + t; if (e < kc * M) { const int kk = e / M, m = e - kk * M; As[kk * MP + m] = ra[j]; } }
  if (XTRANSB) {
#pragma unroll
    for (int j = 0; j < NLB; ++j) { const int e = 256 * j + t; if (e < kc * N) { const int kk = e / N, n = e - kk * N; Bs[kk * NPP + n] = rb[j]; } }
  }
  else {
#pragma unroll
    for (int j = 0; j < NLB; ++j) { const int e = 256 * j + t, n = e / KC, kk = e - n * KC; if (e < BEL && kk < kc) Bs[kk * NPP + n] = rb[j]; }
  }
}

// runlen: the batch is a sequence of runs of `runlen` consecutive items that share one C (blocked GEMM: the k blocks of a C
// block; 1: every item owns its C). A work-group keeps C in registers across a run: the chain of the sequential reference.
extern "C" __global__ __launch_bounds__(256) void xsmm_smm_op(DevAddr ad, long long batch, long long runlen)
{
  __shared__ __attribute__((aligned(16))) T lds[2 * BUF];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const long long nruns = batch / runlen, G = gridDim.x;
  long long run = blockIdx.x;
  if (run >= nruns) return;
  long long item = run * runlen, p = 0;
  T ra[NLA], rb[NLB], rc[TM][TN], acc[TM][TN];
  // software pipeline over (item, chunk): the registers hold the chunk after the one that is being multiplied
  load_chunk(resolve<const T>(ad.a, ad.ia, ad.sa, ad, item), resolve<const T>(ad.b, ad.ib, ad.sb, ad, item), 0, t, ra, rb);
  if (!XBETA0) {
    const XGLOBAL T* const gc = (const XGLOBAL T*)resolve<T>(ad.c, ad.ic, ad.sc, ad, item);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) rc[i][j] = gc[n * M + m]; }
    }
  }
  int buf = 0;
  for (;;) {
    if (0 == p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = XBETA0 ? (T)0 : rc[i][j];
      }
    }
    const bool same_run = (p + 1 < runlen), has_next = same_run || (run + G < nruns);
    const long long next = same_run ? (item + 1) : ((run + G) * runlen);
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      T* const As = lds + buf * BUF;
      T* const Bs = As + AS;
      park_chunk(As, Bs, ch, t, ra, rb);
      if (ch + 1 < NCH) load_chunk(resolve<const T>(ad.a, ad.ia, ad.sa, ad, item), resolve<const T>(ad.b, ad.ib, ad.sb, ad, item), ch + 1, t, ra, rb);
      else if (has_next) { // first chunk of the next item, and its C if it opens a run
        load_chunk(resolve<const T>(ad.a, ad.ia, ad.sa, ad, next), resolve<const T>(ad.b, ad.ib, ad.sb, ad, next), 0, t, ra, rb);
        if (!XBETA0 && !same_run) {
          const XGLOBAL T* const gc = (const XGLOBAL T*)resolve<T>(ad.c, ad.ic, ad.sc, ad, next);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) rc[i][j] = gc[n * M + m]; }
          }
        }
      }
      lds_barrier();
      const int k0 = ch * KC, kc = (K - k0 < KC) ? (K - k0) : KC;
#pragma unroll 4
      for (int kk = 0; kk < kc; ++kk) {
        T av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = As[kk * MP + tx * TM + i];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = Bs[kk * NPP + ty * TN + j];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = xfma(av[i], bv[j], acc[i][j]);
        }
      }
      buf ^= 1;
    }
    if (!same_run) { // the run is complete
      XGLOBAL T* const gc = (XGLOBAL T*)resolve<T>(ad.c, ad.ic, ad.sc, ad, item);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) __builtin_nontemporal_store(acc[i][j], gc + n * M + m); }
      }
    }
    if (!has_next) break;
    if (same_run) ++p; else { p = 0; run += G; }
    item = next;
  }
}
)XSMM";

struct SmmKey {
  int typesize, m, n, k, flags, variant, lda, ldb, ldc;
  bool operator==(const SmmKey& o) const { return typesize == o.typesize && m == o.m && n == o.n && k == o.k && flags == o.flags && variant == o.variant
                                               && lda == o.lda && ldb == o.ldb && ldc == o.ldc; }
};
struct SmmKeyHash { size_t operator()(const SmmKey& k) const { return (size_t)(((((k.m * 131 + k.n) * 131 + k.k) * 8 + k.flags * 2 + (k.typesize == 8)) * 64 + k.variant) * 31 + k.lda * 7 + k.ldb * 3 + k.ldc); } };

// A specialised kernel is absent (never asked for), being compiled on the helper thread, ready, or failed (not retried).
struct JitSlot { int state; JitKernel* kernel; }; // state 0: compiling, 1: ready, 2: failed
std::mutex g_smm_lock;
std::unordered_map<SmmKey, JitSlot, SmmKeyHash> g_smm_cache;

// The kernel for `key`: from the table, else from the code-object cache on disk (milliseconds), else from the compiler -- on
// the helper thread (nullptr now: the caller falls back to the next best kernel, a later call finds this one ready) unless
// LIBXSMM_AMD_JIT_ASYNC=0 or `wait`.
template<typename KEY, typename MAP, typename GEN>
JitKernel* jit_resolve(MAP& table, const KEY& key, const char* fname, bool wait, GEN gen_source)
{
  {
    std::lock_guard<std::mutex> guard(g_smm_lock);
    auto it = table.find(key);
    if (it != table.end()) return (1 == it->second.state) ? it->second.kernel : nullptr;
    table.emplace(key, JitSlot{ 0, nullptr }); // this caller is in charge of it
  }
  const std::string src = gen_source();
  auto publish = [&table, key](JitKernel* k) {
    std::lock_guard<std::mutex> guard(g_smm_lock);
    JitSlot& slot = table[key];
    slot.kernel = k; slot.state = (nullptr != k ? 1 : 2);
  };
  JitKernel* k = jit_from_cache(src, fname);
  if (nullptr != k) { publish(k); return k; }
  auto compile = [src, fname, publish]() {
    std::string log;
    JitKernel* const kk = jit_compile(src, fname, &log);
    if (nullptr == kk && 0 != verbosity()) fprintf(stderr, "LIBXSMM WARNING: SMM JIT failed (%s); using the pre-compiled kernel\n", log.c_str());
    publish(kk);
    return kk;
  };
  if (wait || !jit_async_enabled()) return compile();
  jit_async([compile]() { (void)compile(); });
  return nullptr;
}

} // namespace

static int smm_jit_waves(int typesize, int m, int n, int k, int flags, int pack = 1, bool runs = false);

// k-chunk of the work-group-per-item form: the largest of 32/16/8 whose two LDS buffers fit 64 KiB (0: none does)
static size_t smm_jit_big_buf(int typesize, int m, int n, int kc, int flags)
{
  const size_t mp = 16 * (size_t)((m + 15) / 16), np = 16 * (size_t)((n + 15) / 16);
  (void)flags;
  const size_t as = (size_t)kc * mp, bs = (size_t)kc * (np + 16 / (size_t)typesize);
  return ((as + bs + 3) / 4) * 4 * (size_t)typesize;
}
static int smm_jit_big_kc(int typesize, int m, int n, int k, int flags)
{
  for (int kc = 32; kc >= 8; kc /= 2) {
    if (kc / 2 >= k && kc > 8) continue; // a shorter chunk already covers K
    if (2 * smm_jit_big_buf(typesize, m, n, kc, flags) <= 65536) return kc;
  }
  return 0;
}

// Products whose operands a run kernel keeps in flight (register stages). Measured on CP2K stacks (MI355X): once the
// per-item index loads are off the critical path (address windows) a run is bound by its on-chip work per product, not by
// the HBM round trip -- depths 2 and 4 were no faster (work-group form) or slower (wave form: register pressure).
// The ring stays in the source as a developer knob (XSMM_SMMJIT_DEPTH).
static int smm_jit_depth(int typesize, int m, int n, int k, int variant)
{
  (void)typesize; (void)m; (void)n; (void)k;
  if (0 == (variant & (SMM_JIT_RUNS | SMM_JIT_WGRUNS))) return 1;
  static const int env = []() { const char* e = getenv("XSMM_SMMJIT_DEPTH"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
  static const int env_bytes = []() { const char* e = getenv("XSMM_SMMJIT_DEPTH_BYTES"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }(); // (developer knob: only products up to this many operand bytes)
  if (0 < env_bytes && (long long)typesize * ((long long)m * k + (long long)k * n) > env_bytes) return 1;
  return (0 < env && env <= 8) ? env : 1;
}

// items per wave pass, encoded in the variant as log2 << 8
static int smm_jit_pack_of(int variant) { return 1 << ((variant >> 8) & 7); }
static int smm_jit_pack_bits(int pack) { int l = 0; while ((1 << l) < pack) ++l; return l << 8; }

std::string gen_smm_source(int typesize, int m, int n, int k, int flags, int variant, int lda, int ldb, int ldc)
{
  if (lda <= 0) lda = m;
  if (ldb <= 0) ldb = (flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? n : k;
  if (ldc <= 0) ldc = m;
  std::string s = "// generated by libxsmm-amd (dense SMM kernel, shape baked in)\n";
  const int lowp = (variant >> 11) & 3; // 16-bit inputs: 1 = i16 -> i32, 2 = bf16 -> bf16, 3 = bf16 -> f32 (0: none)
  s += std::string("typedef ") + (8 == typesize ? "double" : (1 == lowp ? "int" : "float")) + " T;\n";
  s += "#define XLOWP " + std::to_string(lowp) + "\n";
  s += "#define XM " + std::to_string(m) + "\n#define XN " + std::to_string(n) + "\n#define XK " + std::to_string(k) + "\n";
  s += std::string("#define XBETA0 ") + ((flags & LIBXSMM_GEMM_FLAG_BETA_0) ? "1" : "0") + "\n";
  s += std::string("#define XTRANSB ") + ((flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? "1" : "0") + "\n";
  if (0 != (variant & SMM_JIT_MFMA_RUNS)) { // run form on the matrix cores (a wave per run)
    s += "#define XLDA " + std::to_string(lda) + "\n#define XLDB " + std::to_string(ldb) + "\n#define XLDC " + std::to_string(ldc) + "\n";
    s += "#define XWAVES " + std::to_string(smm_mfma_runs_waves(smm_mfma_runs_lds(typesize, m, n, k, ldb, 0 == (variant & SMM_JIT_RUNS)))) + "\n";
    s += "#define XFLAT 0\n#define XRUNS 1\n#define XHASWG 0\n#define XGROUPED 0\n";
    s += std::string("#define XSTREAM ") + ((variant & SMM_JIT_RUNS) ? "0" : "1") + "\n"; // (without the run bit: every item owns its C)
    s += "#define XHANDWAIT " + std::to_string(smm_mfma_handwait()) + "\n";
    s += "#define XDEPTH " + std::to_string(smm_mfma_runs_depth(typesize, m, n, k, ldb)) + "\n";
    s += std::string("#define XSPLIT ") + ((variant & SMM_JIT_SPLIT) ? "1" : "0") + "\n";
    s += "#define XTILEWG 0\n";
    s += SMM_JIT_PRELUDE; s += SMM_JIT_MFMA_RUNS_CONST; s += SMM_JIT_CHAIN; s += SMM_JIT_MFMA_RUNS_KERNEL;
    return s;
  }
  if (0 != (variant & SMM_JIT_MFMA_WAVE2)) { // ... the columns of C in two halves
    s += "#define XFLAT 0\n#define XWPE 1\n#define XNSPLIT 2\n#define XVEC " + std::to_string(16 / typesize) + "\n";
    s += "#define XLDA " + std::to_string(m) + "\n#define XLDB " + std::to_string(k) + "\n#define XLDC " + std::to_string(m) + "\n";
    s += SMM_JIT_PRELUDE;
    s += SMM_JIT_MFMA_WAVE_BODY;
    return s;
  }
  if (0 != (variant & SMM_JIT_MFMA_WAVE)) { // matrix-core kernel, one wave per item
    const int wvec = (0 != (variant & SMM_JIT_SCALAR)) ? 1 : 16 / typesize;
    s += "#define XNSPLIT 1\n#define XVEC " + std::to_string(wvec) + "\n";
    const bool wtb = (0 != (flags & LIBXSMM_GEMM_FLAG_TRANS_B));
    s += "#define XLDA " + std::to_string(1 == wvec ? lda : m) + "\n#define XLDB " + std::to_string(1 == wvec ? ldb : (wtb ? n : k)) + "\n#define XLDC " + std::to_string(1 == wvec ? ldc : m) + "\n";
    s += "#define XFLAT 0\n#define XWPE " + std::to_string(smm_mfma_wave_wpe(smm_mfma_wave_lds(typesize, m, n, k, wvec, wtb), typesize, m, n, k, wvec, lda, wtb ? 0 : ldb, ldc)) + "\n";
    s += SMM_JIT_PRELUDE;
    s += SMM_JIT_MFMA_WAVE_BODY;
    return s;
  }
  if (0 != (variant & SMM_JIT_MFMA)) { // matrix-core work-group kernel, shape and leading dimensions baked in
    s += "#define XFLAT 0\n#define XMW_JIT 1\n";
    s += "#define XLDA " + std::to_string(lda) + "\n#define XLDB " + std::to_string(ldb) + "\n#define XLDC " + std::to_string(ldc) + "\n";
    s += std::string("#define XTIGHT ") + ((variant & SMM_JIT_MFMA_TIGHT) ? "1" : "0") + "\n";
    s += std::string("#define XTIGHTC ") + ((variant & SMM_JIT_MFMA_TIGHTC) ? "1" : "0") + "\n";
    s += "#define XMW_FORM " + std::to_string(8 == typesize ? (k > 32 ? 2 : 1) : 0) + "\n";
    s += SMM_JIT_PRELUDE;
    s += SMM_MFMA_WG_SOURCE;
    return s;
  }
  if (0 != (variant & SMM_JIT_BIG)) { // work-group per item, K chunked
    s += "#define XKC " + std::to_string(smm_jit_big_kc(typesize, m, n, k, flags)) + "\n";
    s += SMM_JIT_BIG_BODY;
    return s;
  }
  const int pack = smm_jit_pack_of(variant);
  s += "#define XLDA " + std::to_string(lda) + "\n#define XLDB " + std::to_string(ldb) + "\n#define XLDC " + std::to_string(ldc) + "\n"; // leading dimensions in memory
  s += "#define XPACK " + std::to_string(pack) + "\n";   // items per wave pass (streaming form of tight strided batches)
  const bool wave_runs = (0 != (variant & SMM_JIT_RUNS) && 0 == (variant & SMM_JIT_WGRUNS)); // (the wave run form keeps C's image over the operands')
  s += "#define XWAVES " + std::to_string(smm_jit_waves(typesize, m, n, k, flags, pack, wave_runs)) + "\n";
  s += std::string("#define XSCALAR ") + ((variant & SMM_JIT_SCALAR) ? "1" : "0") + "\n"; // element-wide loads/stores only
  // runs of equal C accumulate in registers: 1 = a wave per run, 2 = a work-group per run (long runs)
  s += std::string("#define XRUNS ") + ((variant & SMM_JIT_WGRUNS) ? "2" : ((variant & SMM_JIT_RUNS) ? "1" : "0")) + "\n";
  { // address space of the operand accesses: global for the run forms; the streaming (one wave per item) form measured
    // faster with generic pointers, i.e. FLAT instructions (f64 13^3: 60.7 vs 55.3 %, the fp32 32^3 kernels 72.5 vs 69 %)
    static const int flat_env = []() { const char* e = getenv("XSMM_SMMJIT_FLAT"); return (nullptr != e && 0 != *e) ? atoi(e) : -1; }();
    static const int defer_env0 = []() { const char* e = getenv("XSMM_SMMJIT_DEFER"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
    // (with the stores deferred the global form is the faster one: tools/sweep_defer.sh, profiles/r2_sweep_defer.txt)
    const int flat = (0 <= flat_env) ? flat_env : ((0 != (variant & (SMM_JIT_RUNS | SMM_JIT_WGRUNS)) || 0 != defer_env0) ? 0 : 1);
    s += std::string("#define XFLAT ") + (flat ? "1" : "0") + "\n";
  }
  s += "#define XDEPTH " + std::to_string(smm_jit_depth(typesize, m, n, k, variant)) + "\n";  // register stages of the run forms
  s += std::string("#define XSPLIT ") + ((variant & SMM_JIT_SPLIT) ? "1" : "0") + "\n";  // relaxed order: few long runs are cut into segments (atomics)
  s += std::string("#define XHASWG ") + ((variant & SMM_JIT_HASWG) ? "1" : "0") + "\n";   // wave form: leave long runs to the work-group form
  s += "#define XGROUPED 0\n";
  { // deferred stores of the streaming form (developer knob)
    static const int defer_env = []() { const char* e = getenv("XSMM_SMMJIT_DEFER"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
    s += std::string("#define XDEFER ") + (defer_env ? "1" : "0") + "\n";
  }
  s += SMM_JIT_PRELUDE;
  s += SMM_JIT_SHAPE; s += SMM_JIT_CHAIN; s += SMM_JIT_SHAPE_KERNELS;
  return s;
}

// LDS bytes one wave of the generated kernel needs (mirrors the constexpr arithmetic of the source)
static size_t smm_jit_wave_lds(int typesize, int m, int n, int k, int flags, int pack = 1, bool runs = false)
{ // runs: the wave run form (XRUNS 1), where C's image lies over the operand images
  const int tgm = (pack >= 4) ? ((pack >= 16) ? 2 : 4) : 8, tgn = 64 / (pack * tgm);
  const int tm = (m + tgm - 1) / tgm, tn = (n + tgn - 1) / tgn;
  int kp = k; if (0 == (tn * (typesize / 4)) % 16) kp = k | 1; else while (0 == (tn * kp * (typesize / 4)) % 16) ++kp;
  const size_t as = ((size_t)(k * m + tgm * tm + 3) / 4) * 4;
  const size_t bs = (flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (((size_t)(k * n + tgn * tn + 3) / 4) * 4) : ((((size_t)tgn * tn) * kp + 3) / 4) * 4;
  const size_t cs = ((size_t)(pack * m * n + 3) / 4) * 4;
  if (runs) return ((pack * (as + bs) > cs) ? pack * (as + bs) : cs) * typesize;
  return (pack * (as + bs) + cs) * typesize;
}

// wavefronts per work-group: as many (4, 2, 1) as fit 64 KiB of static LDS; 0 if even one wave does not fit
static int smm_jit_waves(int typesize, int m, int n, int k, int flags, int pack, bool runs)
{
  const size_t w = smm_jit_wave_lds(typesize, m, n, k, flags, pack, runs);
  return (4 * w <= 65536) ? 4 : ((2 * w <= 65536) ? 2 : ((w <= 65536) ? 1 : 0));
}

bool smm_jit_eligible(const SmmBatch& s)
{
  const char* const env_jit = getenv("LIBXSMM_AMD_JIT"); // re-read on every call: tests and tools toggle it
  const bool enabled = (nullptr == env_jit || 0 != atoi(env_jit));
  if (!enabled || 0 != s.general || SYNC_ATOMIC == s.sync) return false;
  if (SYNC_NONE != s.sync && 0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) return false; // (never chosen: beta == 0 needs no care)
  if (SYNC_DEVICE == s.sync && 0 == s.c_atomics) return false;  // C in host memory the GPU maps: the generic kernel adds by compare-and-swap (cas_add, kernels/smm_generic.hip)
  const bool tight = (s.lda == s.m && s.ldc == s.m && (0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (s.ldb == s.n) : (s.ldb == s.k)));
  if (!tight) { // leading dimensions with gaps: the wave forms fetch an operand's whole span -- as long as the gaps stay moderate
    if (s.m > 32 || s.n > 32 || (s.k > 64 && !smm_mfma_stream_ok(s))) return false;
    const long long span = (long long)s.lda * (s.k - 1) + s.m + (long long)s.ldb * ((0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? s.k : s.n) - 1)
                         + (0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? s.n : s.k) + (long long)s.ldc * (s.n - 1) + s.m;
    const long long used = (long long)s.m * s.k + (long long)s.k * s.n + (long long)s.m * s.n;
    if (2 * used < span) return false;
    if (span * s.typesize > 40960 && !(s.k > 64 && smm_mfma_stream_ok(s))) return false; // (a long K in chunks: no whole operand is ever on chip)
  }
  if (s.m <= 32 && s.n <= 32 && s.k > 64 && smm_mfma_stream_ok(s)) { /* the matrix-core streaming form takes K in chunks of up to 64 */ }
  else if (s.m > 32 || s.n > 32 || s.k > 64) { // work-group-per-item form: 16x16 threads x (<=4x4) tile, K chunked (also small M, N with a long K)
    if (s.m > 64 || s.n > 64 || s.k > 1024) return false;
    if (SYNC_NONE != s.sync && !(0 < s.uniform_run && 0 == s.batch % s.uniform_run)) return false; // shared C only as runs of a known, uniform length
    if (0 == smm_jit_big_kc(s.typesize, s.m, s.n, s.k, s.flags)) return false;
  }
  else {
    if (s.k > 64) return false;                                                // 8x8 lanes x (<=4x4) tile, whole K in LDS
    if (0 == smm_jit_waves(s.typesize, s.m, s.n, s.k, s.flags)) return false; // static LDS limit per work-group
  }
  // (the compiler works on a helper thread and its output is kept on disk, so a small batch is enough of a reason: the
  // specialised kernel is the faster one at every size measured -- fp64 23^3: 7.9 vs 12.2 us at 32 items, 8.4 vs 13.5 us at 1024,
  // 50 vs 77 us at 16 384; tools/bench_small_batches.py, profiles/r3_small_batches.txt)
  const char* const env_min = getenv("LIBXSMM_AMD_JIT_MINBATCH");
  const long long min_batch = (nullptr != env_min && 0 != *env_min) ? atoll(env_min) : 16LL;
  if (s.batch < min_batch && 0 == s.jit_always) return false;
  return true;
}

// LDS bytes of one operand buffer of the work-group form (mirrors the constexpr arithmetic of the source)
static size_t smm_jit_wg_buf(int typesize, int m, int n, int k, int flags)
{
  const int nq = (n + 3) / 4, tm = (m + 15) / 16, tn = (nq + 3) / 4, npad = 3 * nq + 4 * tn;
  int kp = k; if (0 == (tn * (typesize / 4)) % 16) kp = k | 1; else while (0 == (tn * kp * (typesize / 4)) % 16) ++kp;
  const size_t as = ((size_t)(k * m + 16 * tm + 3) / 4) * 4;
  const size_t bs = (flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (((size_t)(k * n + npad + 3) / 4) * 4) : (((size_t)npad * kp + 3) / 4) * 4;
  return (as + bs) * typesize;
}

// Which flavour of the generated kernel a batch needs: strided batches of tightly packed items whose bases are 16-byte
// aligned use the widest loads the item size allows; index/pointer batches (and anything else) are only known to be
// element-aligned.
static int smm_jit_width_variant(const SmmBatch& s)
{
  bool wide = false;
  if (ADDR_STRIDED == s.mode) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(s.a) | reinterpret_cast<uintptr_t>(s.b) | reinterpret_cast<uintptr_t>(s.c);
    wide = (0 == (bits & 15))
        && (s.sa == (long long)s.m * s.k || 0 == s.sa) && (s.sb == (long long)s.k * s.n || 0 == s.sb)
        && (s.sc == (long long)s.m * s.n || 0 == s.sc);
  }
  return wide ? 0 : SMM_JIT_SCALAR;
}

static JitKernel* smm_jit_get(const SmmKey& key, bool wait = false)
{
  return jit_resolve(g_smm_cache, key, "xsmm_smm_op", wait, [&key]() {
    return gen_smm_source(key.typesize, key.m, key.n, key.k, key.flags, key.variant, key.lda, key.ldb, key.ldc); });
}

// one launch of one flavour; -1 when the kernel is not available
static int smm_jit_launch_variant(const SmmBatch& s, int variant, void* stream)
{
  const SmmKey key = { s.typesize, s.m, s.n, s.k, s.flags & (LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_TRANS_B), variant, s.lda, s.ldb, s.ldc };
  JitKernel* const k = smm_jit_get(key);
  if (nullptr == k) return -1;
  struct { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; } ad;
  ad.a = (const char*)s.a; ad.b = (const char*)s.b; ad.c = (char*)s.c; ad.ia = (const char*)s.ia; ad.ib = (const char*)s.ib; ad.ic = (const char*)s.ic;
  ad.sa = s.sa; ad.sb = s.sb; ad.sc = s.sc; ad.index_base = s.index_base; ad.index_stride = s.index_stride; ad.mode = s.mode;
  ad.flags = (SYNC_DEVICE == s.sync ? s.devflags : nullptr);
  long long batch = s.batch;
  static const int bpc_env = []() { const char* e = getenv("XSMM_SMMJIT_BPC"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
  if (0 != (variant & SMM_JIT_BIG)) { // one work-group per item, walking the batch with a stride of the grid
    const size_t lds = 2 * smm_jit_big_buf(s.typesize, s.m, s.n, smm_jit_big_kc(s.typesize, s.m, s.n, s.k, s.flags), s.flags);
    long long per_cu = (long long)((160 * 1024) / (lds ? lds : 1));
    if (per_cu > 4) per_cu = 4;
    // a persistent grid must not exceed what is resident: a work-group that starts after the others have finished their
    // share is pure tail (measured: f64 64^3 with 150 VGPRs fits 3 per CU; a grid of 4 per CU loses 12 %)
    const int occ = jit_blocks_per_cu(k, 256);
    if (0 < occ && occ < per_cu) per_cu = occ;
    if (per_cu < 1) per_cu = 1;
    if (0 < bpc_env) per_cu = bpc_env;
    long long runlen = (0 < s.uniform_run ? s.uniform_run : 1);
    long long blocks = batch / runlen;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    if (blocks < 1) blocks = 1;
    void* args[] = { &ad, &batch, &runlen };
    return jit_launch_args(k, (unsigned)blocks, 256u, args, stream);
  }
  if (0 != (variant & SMM_JIT_WGRUNS)) { // work-groups of 256 threads, dealt chunks of 64 items
    const size_t buf = smm_jit_wg_buf(s.typesize, s.m, s.n, s.k, s.flags);
    const size_t lds = (2 * buf <= 65536 ? 2 : 1) * buf;
    long long per_cu = (long long)((160 * 1024) / (lds ? lds : 1));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    if (0 < bpc_env) per_cu = bpc_env;
    long long blocks = (batch + 63) / 64;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    if (blocks < 1) blocks = 1;
    return jit_launch_raw(k, (unsigned)blocks, 256u, &ad, sizeof(ad), &batch, stream);
  }
  if (0 != (variant & SMM_JIT_MFMA_RUNS)) { // a wave per run on the matrix cores, dealt chunks of 64 items (segments of 8 and more if the verdict cuts the batch up)
    const bool stream_form = (0 == (variant & SMM_JIT_RUNS)); // every item owns its C: a wave per item, stride of the resident waves
    const size_t wlds = smm_mfma_runs_lds(s.typesize, s.m, s.n, s.k, s.ldb, stream_form);
    const int waves = smm_mfma_runs_waves(wlds);
    if (0 == waves) return -1;
    const long long units = stream_form ? batch : ((nullptr != ad.flags) ? (batch + 7) / 8 : (batch + 63) / 64);
    long long blocks = (units + waves - 1) / waves;
    long long per_cu = (long long)((160 * 1024) / (wlds * (size_t)waves)); if (per_cu * waves > 16) per_cu = 16 / waves; if (per_cu < 1) per_cu = 1;
    if (stream_form) { const int occ = jit_blocks_per_cu(k, 64 * waves); if (0 < occ && occ < per_cu) per_cu = occ; } // (a persistent grid must not exceed what is resident)
    if (0 < bpc_env) per_cu = bpc_env;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    if (blocks < 1) blocks = 1;
    return jit_launch_raw(k, (unsigned)blocks, 64u * (unsigned)waves, &ad, sizeof(ad), &batch, stream);
  }
  const int pack = smm_jit_pack_of(variant);
  if (1 < pack) { // the launch covers batch / pack groups of `pack` consecutive items (the caller handles the remainder)
    ad.sa *= pack; ad.sb *= pack; ad.sc *= pack; batch /= pack;
    if (0 == batch) return 0;
  }
  const bool wave_runs = (0 != (variant & SMM_JIT_RUNS));
  const int waves = smm_jit_waves(s.typesize, s.m, s.n, s.k, s.flags, pack, wave_runs);
  const size_t lds = (size_t)waves * smm_jit_wave_lds(s.typesize, s.m, s.n, s.k, s.flags, pack, wave_runs);
  long long per_cu = (long long)((160 * 1024) / (lds ? lds : 1));
  if (per_cu * waves > 16) per_cu = 16 / waves; // the streaming rate peaks around 12-16 waves per CU
  if (per_cu < 1) per_cu = 1;
  if (0 < bpc_env) per_cu = bpc_env;
  // run form: a wave scans chunks of 64 items for run heads, so the grid is sized by chunks
  // (the verdict is on the device: segments of 8 items and more if the batch is cut up -- waves without a chunk leave at once)
  const long long units = (0 != (variant & SMM_JIT_RUNS)) ? ((nullptr != ad.flags ? (batch + 7) / 8 : (batch + 63) / 64)) : batch;
  long long blocks = (units + waves - 1) / waves;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  if (blocks < 1) blocks = 1;
  return jit_launch_raw(k, (unsigned)blocks, 64u * (unsigned)waves, &ad, sizeof(ad), &batch, stream);
}

// Items a wave handles at a time. Only for batches laid out back to back (strided, tight, the "wide" flavour): a small or
// oddly sized item (5^3 doubles are 1000 bytes, 13^3 are 1352) leaves most of a wave's lanes and of every load instruction
// idle. Developer knob: XSMM_SMMJIT_PACK.
static int smm_jit_pack(const SmmBatch& s, int width)
{
  static const int env = []() { const char* e = getenv("XSMM_SMMJIT_PACK"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
  if (0 != width || ADDR_STRIDED != s.mode || 0 == s.sa || 0 == s.sb || 0 == s.sc) return 1;
  int pack = 1;
  if (0 < env) pack = env;
  else { // measured on MI355X (tools/bench_dense.py, XSMM_SMMJIT_PACK sweep): items of 6 KB and more gain nothing; below, as
    // many as make up 16 KB -- f64 5^3: 22 -> 69 % of the HBM peak, 8^3: 43 -> 72 %, 13^3: 61 -> 67 %; f32 5^3: 13 -> 58 %, 8^3: 25 -> 72 %
    const size_t item = ((size_t)s.m * s.k + (size_t)s.k * s.n + (size_t)s.m * s.n) * s.typesize;
    if (item < 6000) while (pack < 16 && 2 * pack * item <= 16384) pack *= 2;
    // items up to ~13 KB that do not end on a 128-byte line: two at a time, so that a line is not fetched by two waves (with
    // deferred stores, profiles/r2_sweep_pack.txt: f32 23^3 62.1 -> 63.4 %, f32 13x23x32 61.2 -> 63.6 %, f64 13x23x32 65.1 -> 66.5 %)
    else if (item < 14000 && 0 != item % 128) pack = 2;
  }
  if (pack > 16) pack = 16;
  while (0 != (pack & (pack - 1))) --pack; // power of two
  // operands of `pack` items in flight per lane (registers) and in LDS
  while (1 < pack && ((size_t)pack * ((size_t)s.m * s.k + (size_t)s.k * s.n + (size_t)s.m * s.n) * s.typesize > 28672
                   || 0 == smm_jit_waves(s.typesize, s.m, s.n, s.k, s.flags, pack))) pack /= 2;
  return pack;
}

// ---- one launch for several batches (CP2K-style stacks: a batch per shape) -----------------------------------------------
// A batch of few long runs leaves the chip nearly empty (one sequential chain per C block: 172 chains per shape in the
// per-GPU share of BASELINE config 5), and batches issued one after the other on a stream run one after the other. Here
// the run kernels of all batches of a call become ONE kernel: the shape-dependent part of the generated source is
// instantiated once per (shape, form) in a namespace of its own, a dispatcher maps every work-group to its batch by its
// index (a small table in device memory: addressing, batch size, first work-group, which body) and calls that body -- all
// chains of all shapes are resident at the same time. Same code per shape as the single-batch kernels, same chains, same
// bits; one verdict slot per batch (fused check kernel).
namespace {
struct GroupedBody { int m, n, k, flags, variant, lda, ldb, ldc; };
inline bool operator==(const GroupedBody& a, const GroupedBody& b) { return 0 == memcmp(&a, &b, sizeof(a)); }

constexpr int GROUPED_BYVAL = 32; // table entries a grouped launch carries as a kernel argument
std::string gen_smm_grouped_source(int typesize, const std::vector<GroupedBody>& bodies, int threads)
{
  std::string s = "// generated by libxsmm-amd (dense SMM run kernels of several shapes behind one dispatcher)\n";
  s += std::string("typedef ") + (8 == typesize ? "double" : "float") + " T;\n#define XLOWP 0\n#define XFLAT 0\n#define XGROUPED 1\n";
  // threads > 64 (XSMM_SMMJIT_TILESPLIT=2): the entries are the 16 x 16 tiles of C of ONE batch (smm_tile_split) and wave t of every
  // work-group walks entry t -- the waves of a work-group walk the same runs
  const bool tilewg = (threads > 64);
  s += std::string("#define XTILEWG ") + (tilewg ? "1" : "0") + "\n";
  bool all_mfma = true;
  for (const GroupedBody& b : bodies) all_mfma = all_mfma && 0 != (b.variant & SMM_JIT_MFMA_RUNS);
  // The register-tiled bodies are called (inlined, the dispatcher carries the registers of all of them at once: measured slower);
  // the matrix-core bodies are lean enough to be inlined into the switch -- no call, no callee-saved registers through scratch.
  static const int inline_env = []() { const char* e = getenv("XSMM_SMMJIT_GROUPED_INLINE"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }(); // developer knob
  s += std::string("#define XENTRY_ATTR ") + ((all_mfma && 0 != inline_env) ? "__forceinline__" : "__attribute__((noinline))") + "\n";
  s += SMM_JIT_PRELUDE;
  for (size_t i = 0; i < bodies.size(); ++i) {
    const GroupedBody& b = bodies[i];
    s += "namespace xg" + std::to_string(i) + " {\n";
    s += "#define XM " + std::to_string(b.m) + "\n#define XN " + std::to_string(b.n) + "\n#define XK " + std::to_string(b.k) + "\n";
    s += std::string("#define XBETA0 ") + ((b.flags & LIBXSMM_GEMM_FLAG_BETA_0) ? "1" : "0") + "\n";
    s += std::string("#define XTRANSB ") + ((b.flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? "1" : "0") + "\n";
    s += "#define XLDA " + std::to_string(b.lda) + "\n#define XLDB " + std::to_string(b.ldb) + "\n#define XLDC " + std::to_string(b.ldc) + "\n";
    s += "#define XPACK 1\n";
    s += "#define XWAVES 1\n"; // (wave bodies: a work-group is one wave, see below)
    s += std::string("#define XSCALAR ") + ((b.variant & SMM_JIT_SCALAR) ? "1" : "0") + "\n";
    s += std::string("#define XRUNS ") + ((b.variant & SMM_JIT_WGRUNS) ? "2" : "1") + "\n";
    s += "#define XDEPTH " + std::to_string(0 != (b.variant & SMM_JIT_MFMA_RUNS) ? smm_mfma_runs_depth(typesize, b.m, b.n, b.k, b.ldb, 0 != (b.variant & SMM_JIT_DEEP)) : smm_jit_depth(typesize, b.m, b.n, b.k, b.variant)) + "\n";
    s += std::string("#define XSPLIT ") + ((b.variant & SMM_JIT_SPLIT) ? "1" : "0") + "\n";
    s += std::string("#define XHASWG ") + ((b.variant & SMM_JIT_HASWG) ? "1" : "0") + "\n";
    if (0 != (b.variant & SMM_JIT_MFMA_RUNS)) { s += "#define XSTREAM 0\n#define XHANDWAIT " + std::to_string(smm_mfma_handwait()) + "\n"; s += SMM_JIT_MFMA_RUNS_CONST; s += SMM_JIT_CHAIN; s += SMM_JIT_MFMA_RUNS_KERNEL; s += "#undef XNROW\n#undef XSTREAM\n#undef XHANDWAIT\n"; }
    else { s += SMM_JIT_SHAPE; s += SMM_JIT_CHAIN; s += SMM_JIT_SHAPE_KERNELS; }
    s += "#undef XM\n#undef XN\n#undef XK\n#undef XBETA0\n#undef XTRANSB\n#undef XLDA\n#undef XLDB\n#undef XLDC\n#undef XPACK\n#undef XWAVES\n"
         "#undef XSCALAR\n#undef XRUNS\n#undef XDEPTH\n#undef XSPLIT\n#undef XHASWG\n#undef WINDOW_AB\n}\n";
  }
  s += "struct GroupEntry { DevAddr ad; long long batch; unsigned block_begin, nblocks; int body, pad; };\n";
  // Work-groups of ONE wave: a wave whose chunk holds no run head leaves at once and its slot goes to the next work-group --
  // with four waves per work-group the slot of a whole work-group stayed taken until its longest chain was done (measured
  // on the 27 CP2K shapes: 1.37 ms with four waves per work-group).
  // (two waves per SIMD: the bodies are called, not inlined; without the bound the kernel is given the registers of the
  // hungriest body plus its own -- 308 for the 27 CP2K shapes in fp64 -- and one wave per SIMD)
  static const int grouped_wpe_env = []() { const char* e = getenv("XSMM_SMMJIT_GROUPED_WPE"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }(); // developer knob: waves per SIMD the dispatcher is compiled for
  // (two waves per SIMD for either kind of body. The matrix-core bodies of 32 x 32 fp64 keep ~155 registers alive -- A fragments,
  // B's flat image, accumulators -- and would just fit three, but inlined next to 26 others they spill at that bound: 27 CP2K
  // shapes, 524 288 products, batch order: 0.80-0.81 ms per call with two waves per SIMD, 0.82 ms with three (1.00 ms when the
  // light bodies keep four products in flight as well), 1.46 ms with four; profiles/r3_cp2k_stacks.txt)
  const int grouped_wpe = (0 < grouped_wpe_env) ? grouped_wpe_env : 2;
  // The table travels as a kernel argument (up to GROUPED_BYVAL entries: 3.8 KiB of the 4 KiB a launch may carry; the callers fuse at
  // most that many batches per launch -- the ordering check carries its table the same way) -- no staging copy on the stream in front of
  // the launch (a blit kernel of ~5 us per call: 27 CP2K calls of ~90 us each paid it 27 times), nothing that a stream capture must not see.
  s += "struct GroupTab { GroupEntry e[" + std::to_string(GROUPED_BYVAL) + "]; };\n";
  s += "extern \"C\" __global__ __launch_bounds__(" + std::to_string(threads) + ", " + std::to_string(grouped_wpe) + ") void xsmm_smm_grouped(const GroupTab tabv, int nentries)\n{\n";
  s += "  extern __shared__ __attribute__((aligned(16))) unsigned char xsmm_dyn_lds[];\n";

  if (!tilewg) {
    s += "  int e = 0;\n  while (e + 1 < nentries && blockIdx.x >= tabv.e[e + 1].block_begin) ++e;\n";
    s += "  const GroupEntry g = tabv.e[e];\n  const unsigned bid = blockIdx.x - g.block_begin;\n  T* const lds = reinterpret_cast<T*>(xsmm_dyn_lds);\n";
  }
  else { // (pad: bytes of LDS per wave)
    s += "  const int e = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);\n  if (e >= nentries) return;\n";
    s += "  const GroupEntry g = tabv.e[e];\n  const unsigned bid = blockIdx.x;\n  T* const lds = reinterpret_cast<T*>(xsmm_dyn_lds + (size_t)e * g.pad);\n";
  }
  s += "  switch (g.body) {\n";
  for (size_t i = 0; i < bodies.size(); ++i) s += "    case " + std::to_string(i) + ": xg" + std::to_string(i) + "::xsmm_entry(g.ad, g.batch, bid, g.nblocks, lds); break;\n";
  s += "    default: break;\n  }\n}\n";
  return s;
}

struct GroupedKey {
  int typesize, threads; std::vector<GroupedBody> bodies;
  bool operator==(const GroupedKey& o) const { return typesize == o.typesize && threads == o.threads && bodies == o.bodies; }
};
struct GroupedKeyHash {
  size_t operator()(const GroupedKey& k) const {
    size_t h = (size_t)k.typesize * 1024 + (size_t)k.threads;
    for (const GroupedBody& b : k.bodies) h = h * 1000003u + (size_t)(((b.m * 131 + b.n) * 131 + b.k) * 64 + b.variant + b.flags * 7 + b.lda + b.ldb * 3 + b.ldc * 5);
    return h;
  }
};
std::unordered_map<GroupedKey, JitSlot, GroupedKeyHash> g_grouped_cache; // (guarded by g_smm_lock)

// work-groups and LDS bytes one batch needs under a run-form body (the sizing of smm_jit_launch_variant)
void grouped_geometry(const SmmBatch& s, int variant, long long* blocks, size_t* lds)
{
  if (0 != (variant & SMM_JIT_MFMA_RUNS)) *lds = smm_mfma_runs_lds(s.typesize, s.m, s.n, s.k, s.ldb);
  else *lds = smm_jit_wave_lds(s.typesize, s.m, s.n, s.k, s.flags, 1, true); // one wave per work-group (run form)
  // a wave per chunk of 64 items (a wave whose chunk holds no run head leaves at once; if the verdict on the device cuts the
  // batch into shorter segments, a wave takes several). Sized for segments of 8 -- eight times as many work-groups, most of
  // them without work -- the 27 CP2K shapes took 0.95 ms instead of 0.6: the dispatcher, not the chains, set the pace.
  long long b = (s.batch + 63) / 64;
  if (b > 256 * 16) b = 256 * 16;
  *blocks = (b < 1 ? 1 : b);
}
} // namespace

// Can batch s be part of a grouped launch? (the run forms of the wave / work-group kernels: M, N <= 32, K <= 64)
bool smm_jit_grouped_eligible(const SmmBatch& s)
{
  if (!smm_jit_eligible(s)) return false;
  if (s.m > 32 || s.n > 32 || s.k > 64 || 0 != s.lowp) return false;
  if (SYNC_DEVICE != s.sync) return false;
  return 0 != smm_jit_waves(s.typesize, s.m, s.n, s.k, s.flags, 1);
}

// All batches (same precision, each with its verdict slot in devflags) in one launch. -1: not available (the caller launches
// them one by one).
namespace {
struct GroupedEntry { int group, body; long long blocks; };
struct GroupedPlan { GroupedKey key; std::vector<GroupedEntry> entries; size_t lds_max = 0; };
// which bodies (one per shape) a set of batches needs and how many work-groups each batch gets. Only the wave form: the
// work-group form (four waves share a product; better for a batch of large shapes on its own) as a second kernel beside
// the first was measured slower -- 27 CP2K shapes, fp64: 1.36 ms with it (its kernel took 0.94 ms next to the wave kernel,
// 0.32 ms alone), 1.18 ms with all chains on the wave form.
bool grouped_plan(const SmmBatch* groups, int ngroups, bool check_eligible, GroupedPlan& plan, bool tiles = false)
{ // tiles: the groups are the tiles of one batch of few runs (smm_tile_split)
  plan.key.typesize = groups[0].typesize; plan.key.threads = (tiles && 0 != smm_tile_wg()) ? 64 * ngroups : 64; plan.key.bodies.clear(); plan.entries.clear(); plan.lds_max = 0;
  for (int g = 0; g < ngroups; ++g) {
    const SmmBatch& s = groups[g];
    if (s.typesize != groups[0].typesize || (check_eligible && !smm_jit_grouped_eligible(s))) return false;
    const int variant = smm_mfma_runs_ok(s) ? (SMM_JIT_SCALAR | (0 != s.relaxed ? SMM_JIT_SPLIT : 0) | SMM_JIT_RUNS | SMM_JIT_MFMA_RUNS | (tiles ? SMM_JIT_DEEP : 0))
                                            : (smm_jit_width_variant(s) | (0 != s.relaxed ? SMM_JIT_SPLIT : 0) | SMM_JIT_RUNS);
    const GroupedBody body = { s.m, s.n, s.k, s.flags & (LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_TRANS_B), variant, s.lda, s.ldb, s.ldc };
    size_t bi = 0;
    while (bi < plan.key.bodies.size() && !(plan.key.bodies[bi] == body)) ++bi;
    if (bi == plan.key.bodies.size()) plan.key.bodies.push_back(body);
    GroupedEntry e; e.group = g; e.body = (int)bi; size_t lds = 0;
    grouped_geometry(s, variant, &e.blocks, &lds);
    if (lds > plan.lds_max) plan.lds_max = lds;
    plan.entries.push_back(e);
  }
  { // bodies in a canonical order: the same set of shapes gives the same kernel (and cache file) whatever the order of the groups
    std::vector<size_t> order(plan.key.bodies.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return 0 > memcmp(&plan.key.bodies[x], &plan.key.bodies[y], sizeof(GroupedBody)); });
    std::vector<GroupedBody> sorted(order.size()); std::vector<int> where(order.size());
    for (size_t i = 0; i < order.size(); ++i) { sorted[i] = plan.key.bodies[order[i]]; where[order[i]] = (int)i; }
    plan.key.bodies.swap(sorted);
    for (GroupedEntry& e : plan.entries) e.body = where[(size_t)e.body];
  }
  // the work-groups of a launch start in the order of their indexes: the batches with the longest chains go first, the short
  // ones fill the tail
  std::stable_sort(plan.entries.begin(), plan.entries.end(), [&](const GroupedEntry& x, const GroupedEntry& y) {
    const SmmBatch& a = groups[x.group]; const SmmBatch& b = groups[y.group];
    return (long long)a.m * a.k + (long long)a.k * a.n + (long long)a.m * a.n > (long long)b.m * b.k + (long long)b.k * b.n + (long long)b.m * b.n;
  });
  return true;
}

JitKernel* grouped_kernel(const GroupedKey& key)
{
  return jit_resolve(g_grouped_cache, key, "xsmm_smm_grouped", false, [&key]() { return gen_smm_grouped_source(key.typesize, key.bodies, key.threads); });
}
}

// the text a grouped launch of these batches compiles (tests: valid gfx950 code without a device)
std::string gen_smm_grouped_source_for(const SmmBatch* groups, int ngroups, bool tiles)
{
  GroupedPlan plan;
  if (ngroups < 1 || !grouped_plan(groups, ngroups, false, plan, tiles)) return std::string();
  return gen_smm_grouped_source(plan.key.typesize, plan.key.bodies, plan.key.threads);
}

static int launch_smm_jit_grouped_checked(const SmmBatch* groups, int ngroups, bool check_eligible, void* stream, const char** name);
int launch_smm_jit_grouped(const SmmBatch* groups, int ngroups, void* stream, const char** name)
{
  return launch_smm_jit_grouped_checked(groups, ngroups, true, stream, name);
}

static int launch_smm_jit_grouped_checked(const SmmBatch* groups, int ngroups, bool check_eligible, void* stream, const char** name)
{
  if (ngroups < 1) return -1;
  GroupedPlan plan;
  if (!grouped_plan(groups, ngroups, check_eligible, plan, !check_eligible) || plan.lds_max > 65536) return -1;
  JitKernel* const kern = grouped_kernel(plan.key);
  if (nullptr == kern) return -1;
  struct DevAddrH { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; };
  struct GroupEntryH { DevAddrH ad; long long batch; unsigned block_begin, nblocks; int body, pad; };
  *name = (8 == groups[0].typesize) ? "smm_f64_jit_shape_runs_grouped" : "smm_f32_jit_shape_runs_grouped";
  std::vector<GroupEntryH> tab(plan.entries.size());
  unsigned total = 0;
  for (size_t i = 0; i < plan.entries.size(); ++i) {
    const SmmBatch& s = groups[plan.entries[i].group];
    GroupEntryH& t = tab[i]; memset(&t, 0, sizeof(t));
    t.ad.a = (const char*)s.a; t.ad.b = (const char*)s.b; t.ad.c = (char*)s.c; t.ad.ia = (const char*)s.ia; t.ad.ib = (const char*)s.ib; t.ad.ic = (const char*)s.ic;
    t.ad.sa = s.sa; t.ad.sb = s.sb; t.ad.sc = s.sc; t.ad.index_base = s.index_base; t.ad.index_stride = s.index_stride; t.ad.mode = s.mode;
    t.ad.flags = s.devflags;
    t.batch = s.batch; t.block_begin = total; t.nblocks = (unsigned)plan.entries[i].blocks; t.body = plan.entries[i].body;
    total += t.nblocks;
  }
  size_t lds_bytes = plan.lds_max;
  if (plan.key.threads > 64) { // a work-group per chunk of 64 items, a wave per tile
    const size_t per_wave = (plan.lds_max + 15) / 16 * 16;
    total = tab[0].nblocks;
    for (GroupEntryH& t : tab) { t.block_begin = 0; t.nblocks = total; t.pad = (int)per_wave; }
    lds_bytes = per_wave * tab.size();
  }
  struct GroupTabH { GroupEntryH e[GROUPED_BYVAL]; };
  static_assert(sizeof(GroupTabH) + 16 <= 4096, "kernel arguments of a launch");
  int nentries = (int)tab.size();
  if (tab.size() > (size_t)GROUPED_BYVAL) return -1; // (never: the callers fuse at most 32 batches per launch)
  GroupTabH byval; // (a kernel argument: see gen_smm_grouped_source)
  memcpy(&byval, tab.data(), tab.size() * sizeof(GroupEntryH));
  void* args[] = { (void*)&byval, &nentries };
  return jit_launch_dyn(kern, total, (unsigned)plan.key.threads, (unsigned)lds_bytes, args, stream);
}

// Matrix-core work-group kernels (32 < max(M, N) <= 64, K <= 64; independent items, or runs of a uniform length) with the
// descriptor baked in. -1: not applicable / not ready (the pre-compiled kernel of the same plan serves: launch_smm_special).
int launch_smm_jit_mfma(const SmmBatch& s, void* stream, const char** name)
{
  const char* const env_jit = getenv("LIBXSMM_AMD_JIT");
  if (nullptr != env_jit && 0 == atoi(env_jit)) return -1;
  static const int on = []() { const char* e = getenv("XSMM_SMMJIT_MFMA"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }(); // developer knob
  if (0 == on || 0 == s.use_mfma || 0 != s.general || 0 != s.lowp) return -1;
  static const int wave_min = []() { const char* e = getenv("XSMM_SMMJIT_WAVE_MIN"); return (nullptr != e && 0 != *e) ? atoi(e) : 32; }(); // developer knob: the wave form below 33
  const bool transb = (0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B)); // (B^T in memory: served by the one-wave-per-item form only)
  if (!((wave_min < s.m || wave_min < s.n) && s.m <= 64 && s.n <= 64 && 0 < s.k && s.k <= 64 && s.lda >= s.m && s.ldb >= (transb ? s.n : s.k) && s.ldc >= s.m)) return -1;
  if (4 == s.typesize && 64 == s.m && 64 == s.n && 64 == s.k && 64 == s.lda && 64 == s.ldb && 64 == s.ldc && SYNC_NONE == s.sync) return -1; // the hand-tuned tight 64^3 kernel
  long long units = 0; int runlen = 1;
  if (SYNC_NONE == s.sync) units = s.batch;
  else if (SYNC_RUNS == s.sync && 0 < s.uniform_run && 0 == s.batch % s.uniform_run) { units = s.batch / s.uniform_run; runlen = (int)s.uniform_run; }
  if (units < 1) return -1;
  const char* const env_min = getenv("LIBXSMM_AMD_JIT_MINBATCH");
  if (s.batch < ((nullptr != env_min && 0 != *env_min) ? atoll(env_min) : 16LL) && 0 == s.jit_always) return -1;
  const bool f64 = (8 == s.typesize);
  { // one wave per item: independent items of a strided batch, tight and 16-byte aligned operands, at least four waves per CU
    static const int wave_on = []() { const char* e = getenv("XSMM_SMMJIT_MFMA_WAVE"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }(); // developer knob
    // 16-byte chunks for strided batches of suitably shaped and aligned items, else element by element (any shape, index and
    // pointer batches: the kernel resolves the addresses like every other batch kernel)
    const uintptr_t bits = reinterpret_cast<uintptr_t>(s.a) | reinterpret_cast<uintptr_t>(s.b) | reinterpret_cast<uintptr_t>(s.c)
                         | (uintptr_t)(s.sa * s.typesize) | (uintptr_t)(s.sb * s.typesize) | (uintptr_t)(s.sc * s.typesize);
    const int chunk = 16 / s.typesize;
    const int bdim = transb ? s.n : s.k, bcnt = transb ? s.k : s.n; // B in memory: bcnt columns of bdim elements at a distance of ldb
    const bool tight_ld = (s.lda == s.m && s.ldb == bdim && s.ldc == s.m);
    const bool wide = (tight_ld && ADDR_STRIDED == s.mode && 0 == (bits & 15) && 0 == s.m % chunk && 0 == s.k % 4 && (!transb || 0 == s.n % chunk));
    const size_t wlds = smm_mfma_wave_lds(s.typesize, s.m, s.n, s.k, wide ? chunk : 1, transb);
    // gaps in the leading dimensions: the element-wise build fetches the spans and drops the gaps (up to half as much again)
    // -- where the spans are short: with more than ~80 elements per lane in flight the work-group form is the faster one
    // (tools/bench_gaps.py: 43x9x27 ld 48/32/48 55.9 vs 51.4 %, 40x64x17 ld 40/17/44 48.6 vs 40.5 %, but 48^3 ld 56 28.8 vs 56.8 %)
    const long long span_loads = ((long long)s.lda * (s.k - 1) + s.m + 63) / 64 + ((long long)s.ldb * (bcnt - 1) + bdim + 63) / 64 + ((long long)s.ldc * (s.n - 1) + s.m + 63) / 64;
    const bool gaps_ok = tight_ld || (2 * s.lda <= 3 * s.m && 2 * s.ldb <= 3 * bdim && 2 * s.ldc <= 3 * s.m && span_loads <= 80);
    if (0 != wave_on && 0 != wlds && 4 * wlds <= 160u * 1024u && SYNC_NONE == s.sync && gaps_ok)
    {
      const SmmKey wkey = { s.typesize, s.m, s.n, s.k, s.flags & (LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_TRANS_B), SMM_JIT_MFMA_WAVE | (wide ? 0 : SMM_JIT_SCALAR), s.lda, s.ldb, s.ldc };
      JitKernel* const wk = smm_jit_get(wkey);
      if (nullptr != wk) {
        struct { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; } wad;
        wad.a = (const char*)s.a; wad.b = (const char*)s.b; wad.c = (char*)s.c; wad.ia = (const char*)s.ia; wad.ib = (const char*)s.ib; wad.ic = (const char*)s.ic;
        wad.sa = s.sa; wad.sb = s.sb; wad.sc = s.sc; wad.index_base = s.index_base; wad.index_stride = s.index_stride; wad.mode = s.mode; wad.flags = nullptr;
        long long wbatch = s.batch; int one = 1;
        int per_cu = (int)((160u * 1024u) / wlds);
        const int by_regs = 4 * smm_mfma_wave_wpe(wlds, s.typesize, s.m, s.n, s.k, wide ? chunk : 1, s.lda, s.ldb, s.ldc);
        if (per_cu > by_regs) per_cu = by_regs;
        { // The memory system is at its best with ~100 KB of operands in flight per CU (the fp32 32^3 kernel: twelve waves of 12 KB); a wave
          // here has a whole item of 19-55 KB in flight. Waves per CU, % of the HBM peak (tools/bench_dense.py, XSMM_SMMJIT_WAVE_PERCU sweep):
          // f32 40^3 3: 71.8, 4: 73.8, 8: 67.9, 12: 67.5 | f64 40^3 3: 72.2, 4: 69.3, 8: 69.0 | f32 48^3 3: 72.9, 4: 70.3, 8: 68.3 |
          // f64 48^3 3: 68.8, 8: 69.1, 12: 69.7 | f32 56^3 3: 69.8, 4: 68.6, 8: 67.6; five to seven waves (uneven over the four SIMDs) lose 3-15 points.
          const size_t item_bytes = (size_t)s.typesize * ((size_t)s.lda * s.k + (size_t)s.ldb * bcnt + (size_t)s.ldc * s.n);
          if (wide && item_bytes >= 16384) { const int fit = (int)(100u * 1024u / item_bytes); per_cu = (fit >= 4) ? 4 : 3; if (per_cu > by_regs) per_cu = by_regs; }
          const char* const e = getenv("XSMM_SMMJIT_WAVE_PERCU"); if (nullptr != e && 0 < atoi(e)) per_cu = atoi(e); // developer knob
        }
        long long wblocks = 256LL * per_cu;
        if (wblocks > s.batch) wblocks = s.batch;
        void* wargs[] = { &wad, &wbatch, &one };
        *name = f64 ? "smm_f64_mfma_wave_jit" : "smm_f32_mfma_wave_jit";
        return jit_launch_dyn(wk, (unsigned)wblocks, 64u, (unsigned)wlds, wargs, stream);
      }
    }
  }
  if (transb) return -1; // (the forms below read B as stored column by column)
  { // fp64 items too large for that: the two-halves form where it leaves room for four waves per CU (56^3: 66 % against 54-58 % on the
    // work-group form; 64 x 64 x K stays on the work-group form, which is the faster one there -- tools/probe/mfma_wave.hip)
    static const int wave_on = []() { const char* e = getenv("XSMM_SMMJIT_MFMA_WAVE"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
    const size_t wlds = f64 ? smm_mfma_wave2_lds(s.typesize, s.m, s.n, s.k) : 0;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(s.a) | reinterpret_cast<uintptr_t>(s.b) | reinterpret_cast<uintptr_t>(s.c)
                         | (uintptr_t)(s.sa * s.typesize) | (uintptr_t)(s.sb * s.typesize) | (uintptr_t)(s.sc * s.typesize);
    if (0 != wave_on && 0 != wlds && 4 * wlds <= 160u * 1024u && !(64 == s.m && 64 == s.n) && SYNC_NONE == s.sync && ADDR_STRIDED == s.mode && 0 == (bits & 15)
      && s.lda == s.m && s.ldb == s.k && s.ldc == s.m)
    {
      const SmmKey wkey = { s.typesize, s.m, s.n, s.k, s.flags & LIBXSMM_GEMM_FLAG_BETA_0, SMM_JIT_MFMA_WAVE2, s.lda, s.ldb, s.ldc };
      JitKernel* const wk = smm_jit_get(wkey);
      if (nullptr != wk) {
        struct { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; } wad;
        wad.a = (const char*)s.a; wad.b = (const char*)s.b; wad.c = (char*)s.c; wad.ia = wad.ib = wad.ic = nullptr;
        wad.sa = s.sa; wad.sb = s.sb; wad.sc = s.sc; wad.index_base = 0; wad.index_stride = 0; wad.mode = 0; wad.flags = nullptr;
        long long wbatch = s.batch; int one = 1;
        long long wblocks = 256LL * 4;
        if (wblocks > s.batch) wblocks = s.batch;
        void* wargs[] = { &wad, &wbatch, &one };
        *name = "smm_f64_mfma_wave2_jit";
        return jit_launch_dyn(wk, (unsigned)wblocks, 64u, (unsigned)wlds, wargs, stream);
      }
    }
  }
  if (!(32 < s.m || 32 < s.n)) return -1; // (the work-group forms below are for shapes beyond 32)
  const bool tight = !f64 && s.lda == s.m && s.ldb == s.k && 0 == ((s.m * s.k) & 3) && 0 == ((s.k * s.n) & 3);
  static const int tightc_on = []() { const char* e = getenv("XSMM_SMM64_TIGHTC"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
  const bool tightc = !f64 && s.ldc == s.m && 0 == ((s.m * s.n) & 3) && 0 != (s.m & 31) && 0 != tightc_on;
  const int variant = SMM_JIT_MFMA | (tight ? SMM_JIT_MFMA_TIGHT : 0) | (tightc ? SMM_JIT_MFMA_TIGHTC : 0);
  const SmmKey key = { s.typesize, s.m, s.n, s.k, s.flags & LIBXSMM_GEMM_FLAG_BETA_0, variant, s.lda, s.ldb, s.ldc };
  JitKernel* const k = smm_jit_get(key);
  if (nullptr == k) return -1;
  struct { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; } ad;
  ad.a = (const char*)s.a; ad.b = (const char*)s.b; ad.c = (char*)s.c; ad.ia = (const char*)s.ia; ad.ib = (const char*)s.ib; ad.ic = (const char*)s.ic;
  ad.sa = s.sa; ad.sb = s.sb; ad.sc = s.sc; ad.index_base = s.index_base; ad.index_stride = s.index_stride; ad.mode = s.mode; ad.flags = nullptr;
  long long batch = s.batch;
  size_t lds = 0; int fit = (tightc ? 3 : 4); // (48 KiB of LDS with the C image: three work-groups per CU)
  if (f64) { // (the sizes of the pre-compiled launch, kernels/smm_special.hip)
    lds = (s.k > 32) ? (size_t)2 * 32 * 64 * sizeof(double) : (size_t)2 * (4 * ((s.k + 3) / 4)) * 64 * sizeof(double);
    fit = (int)((160u * 1024u) / lds); if (fit > 3) fit = 3; if (fit < 1) fit = 1;
  }
  static const int bpc_env = []() { const char* e = getenv("XSMM_SMM64_BPC"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
  long long blocks = units;
  const long long resident = 256LL * (0 < bpc_env ? bpc_env : fit);
  if (blocks > resident) blocks = resident;
  void* args[] = { &ad, &batch, &runlen };
  *name = f64 ? (1 == runlen ? "smm_f64_mfma_wg_jit" : "smm_f64_mfma_wg_runs_jit") : (1 == runlen ? "smm_f32_mfma_wg_jit" : "smm_f32_mfma_wg_runs_jit");
  return jit_launch_dyn(k, (unsigned)blocks, 256u, (unsigned)lds, args, stream);
}

// 16-bit inputs (args.lowp 1: i16 -> i32, 3: bf16 -> f32): strided batches of tightly packed items with independent C go through
// the streaming form of the specialised kernel (the inputs are widened on their way into LDS). -1: not applicable.
int launch_smm_jit_lowp(const SmmBatch& s, void* stream, const char** name)
{
  const char* const env_jit = getenv("LIBXSMM_AMD_JIT");
  if (nullptr != env_jit && 0 == atoi(env_jit)) return -1;
  if ((1 != s.lowp && 3 != s.lowp && 4 != s.lowp) || 0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) return -1;
  if (4 == s.lowp && 0 != (s.m & 1)) return -1;
  const bool strided = (ADDR_STRIDED == s.mode); // (index and pointer batches: the streaming form with element-wide -- one k pair -- accesses, below)
  { // bf16 inputs beyond 32: the one-wave-per-item matrix-core kernel (fp32 instruction on the widened operands: the gold
    // loop's product-then-add bit for bit)
    static const int wave_on = []() { const char* e = getenv("XSMM_SMMJIT_MFMA_WAVE"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
    static const int wave_min = []() { const char* e = getenv("XSMM_SMMJIT_LOWP_WAVE_MIN"); return (nullptr != e && 0 != *e) ? atoi(e) : 31; }(); // developer knob (32^3: 73.8 vs 67.9 % for the fp32 result, 67.9 vs 49.5 % for bf16; 16^3 is better off on the streaming form)
    const size_t wlds = smm_mfma_wave_lds(4, s.m, s.n, s.k);
    const uintptr_t bits = strided ? (reinterpret_cast<uintptr_t>(s.a) | reinterpret_cast<uintptr_t>(s.b) | reinterpret_cast<uintptr_t>(s.c)
                         | (uintptr_t)(s.sa * 2) | (uintptr_t)(s.sb * 2) | (uintptr_t)(s.sc * (4 == s.lowp ? 2 : 4))) : 0;
    const char* const env_min = getenv("LIBXSMM_AMD_JIT_MINBATCH");
    // (index and pointer batches as well: their items are whole numbers of 16-byte chunks long -- M % 4 == 0, K % 8 == 0 -- so callers
    // that lay items out back to back keep the chunks aligned; an item that does not start on 16 bytes costs speed, not correctness)
    if (0 != wave_on && 0 != s.use_mfma && (3 == s.lowp || 4 == s.lowp) && (wave_min < s.m || wave_min < s.n) && 0 != wlds && 4 * wlds <= 160u * 1024u
      && 0 == (s.k & 7) && (3 == s.lowp || 0 == (s.m & 7)) && 0 == (bits & 15) && s.lda == s.m && s.ldb == s.k && s.ldc == s.m
      && s.batch >= ((nullptr != env_min && 0 != *env_min) ? atoll(env_min) : 16LL))
    {
      const SmmKey wkey = { 4, s.m, s.n, s.k, s.flags & LIBXSMM_GEMM_FLAG_BETA_0, SMM_JIT_MFMA_WAVE | ((4 == s.lowp ? 2 : 3) << 11), s.lda, s.ldb, s.ldc };
      JitKernel* const wk = smm_jit_get(wkey);
      if (nullptr != wk) {
        struct { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; } wad;
        wad.a = (const char*)s.a; wad.b = (const char*)s.b; wad.c = (char*)s.c; wad.ia = (const char*)s.ia; wad.ib = (const char*)s.ib; wad.ic = (const char*)s.ic;
        wad.sa = s.sa; wad.sb = s.sb; wad.sc = s.sc; // (in elements of the operands' types, as the kernel counts them)
        wad.index_base = s.index_base; wad.index_stride = s.index_stride; wad.mode = s.mode; wad.flags = nullptr;
        long long wbatch = s.batch; int one = 1;
        int per_cu = (int)((160u * 1024u) / wlds);
        const int by_regs = 4 * smm_mfma_wave_wpe(wlds);
        if (per_cu > by_regs) per_cu = by_regs;
        { // Eight waves per CU (tools/bench_dense.py lowp, XSMM_SMMJIT_WAVE_PERCU sweep, % of the HBM peak at 8 / as many as LDS and
          // registers allow: bf16 -> f32 48^3 70.5 / 67.4, 64^3 74.4 / 67.7; bf16 -> bf16 32^3 73.5 / 67.6, 48^3 73.5 / 62.5, 64^3 60.3 / 54.7;
          // three or four -- the optimum of the fp32 / fp64 wave kernels -- lose 5-30 points here: the widening and rounding work per item
          // wants more waves to overlap with)
          // (eight per CU also where the estimate above says fewer fit: whatever is not resident at once starts as the others finish)
          per_cu = 8;
          const char* const e = getenv("XSMM_SMMJIT_WAVE_PERCU"); if (nullptr != e && 0 < atoi(e)) per_cu = atoi(e); // developer knob
        }
        long long wblocks = 256LL * per_cu;
        if (wblocks > s.batch) wblocks = s.batch;
        void* wargs[] = { &wad, &wbatch, &one };
        *name = (4 == s.lowp) ? "smm_bf16_mfma_wave_jit_lowp" : "smm_bf16f32_mfma_wave_jit_lowp";
        return jit_launch_dyn(wk, (unsigned)wblocks, 64u, (unsigned)wlds, wargs, stream);
      }
    }
  }
  // (i16 -> i32 has no matrix-core form: beyond 32 the same streaming kernel with a larger tile per lane -- 48^3 18 -> 57 %,
  // 64^3 16 -> 56 % of the HBM peak; XSMM_SMMJIT_LOWP_BIG=0: the pre-compiled kernel)
  static const int big_i16 = []() { const char* e = getenv("XSMM_SMMJIT_LOWP_BIG"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
  const int lim = (0 != big_i16) ? 64 : 32; // (bf16 shapes the matrix-core form above does not take -- M or K not a multiple of 4 / 8 -- come here as well)
  if (s.m > lim || s.n > lim || s.k > 64 || 0 != (s.k & 1) || s.lda != s.m || s.ldb != s.k || s.ldc != s.m) return -1;
  // strided batches of items laid out back to back take 16-byte accesses; every other batch (other strides, index arrays -- in
  // elements of 16 bits, as the reference's libxsmm_mmbatch_kernel counts them --, arrays of pointers) one k pair per access
  const bool back_to_back = strided && s.sa == (long long)s.m * s.k && s.sb == (long long)s.k * s.n && s.sc == (long long)s.m * s.n;
  const char* const env_min = getenv("LIBXSMM_AMD_JIT_MINBATCH");
  if (s.batch < ((nullptr != env_min && 0 != *env_min) ? atoll(env_min) : 16LL)) return -1;
  if (0 == smm_jit_waves(4, s.m, s.n, s.k, s.flags)) return -1;
  SmmBatch j = s;
  j.typesize = 4; j.lowp = 0; j.sync = SYNC_NONE; // (the kernel itself addresses A and B -- and a bf16 C -- in elements of 16 bits)
  const uintptr_t bits = reinterpret_cast<uintptr_t>(s.a) | reinterpret_cast<uintptr_t>(s.b) | reinterpret_cast<uintptr_t>(s.c);
  const int lowp_bits = ((4 == s.lowp ? 2 : s.lowp) << 11); // XLOWP: 1 i16 -> i32, 2 bf16 -> bf16, 3 bf16 -> f32
  const bool wide = (back_to_back && 0 == (bits & 15));
  const int variant = (wide ? 0 : SMM_JIT_SCALAR) | lowp_bits;
  *name = (1 == s.lowp) ? "smm_i16i32_jit_shape_lowp" : (4 == s.lowp ? "smm_bf16_jit_shape_lowp" : "smm_bf16f32_jit_shape_lowp");
  if (wide) { // small items laid out back to back: several per wave and pass, as the fp32 / fp64 streaming form takes them (a 16^3 item is
    // 1.5-2 KB: one per wave leaves most lanes of every access idle and the kernel bound by its instruction count per item)
    const char* const pack_e = getenv("XSMM_SMMJIT_LOWP_PACK"); // developer knob (re-read on every call: the tests sweep it)
    const int pack_env = (nullptr != pack_e && 0 != *pack_e) ? atoi(pack_e) : 0;
    const size_t item = 2 * ((size_t)s.m * s.k + (size_t)s.k * s.n) + (size_t)(4 == s.lowp ? 2 : 4) * s.m * s.n;
    int pack = 1;
    if (0 < pack_env) pack = pack_env;
    else if (item < 6000) while (pack < 8 && 2 * pack * item <= 16384) pack *= 2;
    while (0 != (pack & (pack - 1))) --pack;
    while (1 < pack && ((size_t)pack * 4 * ((size_t)s.m * s.k + (size_t)s.k * s.n + (size_t)s.m * s.n) > 28672 || 0 == smm_jit_waves(4, s.m, s.n, s.k, s.flags, pack))) pack /= 2;
    if (1 < pack && s.batch >= pack) {
      const int e = smm_jit_launch_variant(j, variant | smm_jit_pack_bits(pack), stream);
      if (0 == e) {
        const long long done = (s.batch / pack) * pack;
        if (done == s.batch) return 0;
        SmmBatch rest = j; // (strides in elements of the operands: 16 bits, C of the result's width)
        rest.a = (const char*)s.a + done * s.sa * 2; rest.b = (const char*)s.b + done * s.sb * 2;
        rest.c = (char*)s.c + done * s.sc * (4 == s.lowp ? 2 : 4); rest.batch = s.batch - done;
        return smm_jit_launch_variant(rest, variant, stream);
      }
      if (0 < e) return e; // (< 0: the packed flavour is not available -- one item per wave)
    }
  }
  return smm_jit_launch_variant(j, variant, stream);
}

// A batch of few runs whose C has several 16 x 16 tiles (one call per CP2K stack: 19 418 products of 32^3 are 172 runs -- 172 waves on
// a chip that holds thousands, each a chain of ~110 dependent products): every tile of C becomes a batch of its own -- the sub-matrices
// A(16 mi.., :), B(:, 16 ni..), C(16 mi.., 16 ni..) under the same leading dimensions, index arrays and ordering verdict -- and the
// tiles run as the groups of one grouped launch: a wave per run AND tile, a quarter of the matrix instructions and half of the
// operand loads per product and wave. Every element of C still receives its own chain in batch order: the same bits.
// Returns the number of tile batches written to out[<= 4] (0: not split).
static int smm_tile_split(const SmmBatch& s, SmmBatch* out)
{
  const char* const on_env = getenv("XSMM_SMMJIT_TILESPLIT"); // developer knob (re-read on every call: the tests toggle it)
  const int on = (nullptr != on_env && 0 != *on_env) ? atoi(on_env) : 1;
  static const int max_waves = []() { const char* e = getenv("XSMM_SMMJIT_TILESPLIT_WAVES"); return (nullptr != e && 0 != *e) ? atoi(e) : 1280; }(); // developer knob
  const int mi = (s.m + 15) / 16, ni = (s.n + 15) / 16;
  if (0 == on || mi * ni < 2 || mi * ni > 4 || SYNC_DEVICE != s.sync || (ADDR_STRIDED != s.mode && ADDR_INDEX != s.mode)) return 0;
  // Every tile's wave fetches its rows of A and its columns of B: twice the requests of a wave per product. Measured (tools/bench_tile_split.py,
  // profiles/r3_tile_split.txt; fp64 32^3, ms split / not split): 2 000 items 0.08-0.16 / 0.12-0.31, 8 000 items 0.08-0.16 / 0.12-0.31, 16 000 items
  // equal for runs up to 32 and 0.17 / 0.31 for runs of 256, 30 000 items 0.20-0.28 / 0.13-0.27 for runs up to 32 (0.19 / 0.31 for runs of
  // 256), 60 000 items 0.37-0.61 / 0.18-0.37: split while the batch cannot fill the chip anyway (the run lengths are known on the device only).
  if (((s.batch + 63) / 64) * mi * ni > max_waves) return 0;
  int n = 0;
  for (int j = 0; j < ni; ++j) {
    for (int i = 0; i < mi; ++i) {
      SmmBatch t = s;
      t.m = (s.m - 16 * i < 16) ? (s.m - 16 * i) : 16; t.n = (s.n - 16 * j < 16) ? (s.n - 16 * j) : 16;
      t.a = static_cast<const char*>(s.a) + (size_t)16 * i * s.typesize;
      t.b = static_cast<const char*>(s.b) + (size_t)16 * j * s.ldb * s.typesize;
      t.c = static_cast<char*>(s.c) + ((size_t)16 * j * s.ldc + (size_t)16 * i) * s.typesize;
      // (the batch as a whole has passed smm_jit_eligible; its rule about leading dimensions with gaps -- an operand's whole span is
      // fetched -- does not apply to a tile: the matrix-core form requests A and C element by element through their leading dimensions,
      // and B's span of a tile is the tile's columns)
      if (!smm_mfma_runs_ok(t)) return 0;
      out[n++] = t;
    }
  }
  return n;
}

int launch_smm_jit(const SmmBatch& s, void* stream, const char** name)
{ // returns -1 when no specialised kernel is available
  const int width = smm_jit_width_variant(s);
  const bool f64 = (8 == s.typesize);
  if (s.m <= 32 && s.n <= 32 && s.k > 64 && smm_mfma_stream_ok(s)) { // long K, small M and N, every item its own C: K in chunks on the matrix cores
    const int e = smm_jit_launch_variant(s, SMM_JIT_SCALAR | SMM_JIT_MFMA_RUNS, stream);
    if (0 <= e) { *name = f64 ? "smm_f64_mfma_stream_jit" : "smm_f32_mfma_stream_jit"; return e; }
  }
  if (s.m > 32 || s.n > 32 || s.k > 64) { // (eligibility made sure of SYNC_NONE, or of runs of a uniform length)
    if (0 == smm_jit_big_kc(s.typesize, s.m, s.n, s.k, s.flags) || (SYNC_NONE != s.sync && !(0 < s.uniform_run && 0 == s.batch % s.uniform_run))) return -1;
    *name = f64 ? "smm_f64_jit_shape_wg" : "smm_f32_jit_shape_wg";
    return smm_jit_launch_variant(s, SMM_JIT_BIG, stream);
  }
  if (SYNC_NONE == s.sync) { // every item owns its C
    { // leading dimensions with gaps: the matrix-core run form addresses A's fragments and C through their leading dimensions (no
      // gap element is requested) -- every item is a run of its own there
      const char* const gaps_env = getenv("XSMM_SMMJIT_GAPS_MFMA"); // developer knob (re-read on every call: tests and tools toggle it): 0 off, 1 gaps only, 2 tight items too
      // (tools/bench_generic.py, fraction of the HBM peak in algorithmic bytes, register-tiled streaming form -> this one: fp64 23^3 ld 24
      // 54.1 -> 59.2 %, fp32 32^3 ld 40 42.3 -> 50.5 %, fp64 13^3 ld 16 44.1 -> 51.6 %; tight fp64 23^3 64.5 -> 61.5 %: tight items stay)
      const int gaps_mfma = (nullptr != gaps_env && 0 != *gaps_env) ? atoi(gaps_env) : 1;
      const bool tight_ld = (s.lda == s.m && s.ldc == s.m && s.ldb == s.k);
      if (0 != gaps_mfma && (!tight_ld || 2 == gaps_mfma) && smm_mfma_runs_ok(s)) {
        const int e = smm_jit_launch_variant(s, SMM_JIT_SCALAR | SMM_JIT_MFMA_RUNS, stream);
        if (0 <= e) { *name = f64 ? "smm_f64_mfma_stream_jit" : "smm_f32_mfma_stream_jit"; return e; }
      }
    }
    *name = f64 ? "smm_f64_jit_shape" : "smm_f32_jit_shape";
    const int pack = smm_jit_pack(s, width);
    if (1 < pack) { // groups of `pack` consecutive items per wave, then the few items that are left
      const int e = smm_jit_launch_variant(s, width | smm_jit_pack_bits(pack), stream);
      if (0 == e) {
        const long long done = (s.batch / pack) * pack;
        if (done == s.batch) return 0;
        SmmBatch rest = s;
        rest.a = (const char*)s.a + done * s.sa * s.typesize; rest.b = (const char*)s.b + done * s.sb * s.typesize;
        rest.c = (char*)s.c + done * s.sc * s.typesize; rest.batch = s.batch - done;
        return smm_jit_launch_variant(rest, SMM_JIT_SCALAR, stream);
      }
      if (0 < e) return e; // (< 0: the packed flavour did not compile -- one item per wave)
    }
    return smm_jit_launch_variant(s, width, stream);
  }
  if (smm_mfma_runs_ok(s)) { // shared C on the matrix cores: a wave per run (batch order; segments + atomics if the verdict or a relaxed order say so)
    { SmmBatch tiles[4];
      const int nt = smm_tile_split(s, tiles); // (at most four entries: the table is a kernel argument -- nothing staged, fine inside a stream capture)
      if (1 < nt) {
        const int e = launch_smm_jit_grouped_checked(tiles, nt, false, stream, name);
        if (0 <= e) { *name = f64 ? "smm_f64_mfma_runs_tiles_jit" : "smm_f32_mfma_runs_tiles_jit"; return e; }
      }
    }
    const int e = smm_jit_launch_variant(s, SMM_JIT_SCALAR | (0 != s.relaxed ? SMM_JIT_SPLIT : 0) | SMM_JIT_RUNS | SMM_JIT_MFMA_RUNS, stream);
    if (0 <= e) { *name = f64 ? "smm_f64_mfma_runs_jit" : "smm_f32_mfma_runs_jit"; return e; }
  }
  static const int wg_env = []() { const char* e = getenv("XSMM_SMMJIT_WG"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }(); // developer knob
  // the work-group form pays off once a product's operands are large (measured on CP2K stacks: 32^3 f64 yes, 23^3 no)
  const bool tight = (s.lda == s.m && s.ldc == s.m && (0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (s.ldb == s.n) : (s.ldb == s.k)));
  const bool wg_fits = (tight && 0 != wg_env && smm_jit_wg_buf(s.typesize, s.m, s.n, s.k, s.flags) <= 65536
                     && (2 == wg_env || (size_t)s.typesize * ((size_t)s.m * s.k + (size_t)s.k * s.n) >= 12288));
  if (SYNC_RUNS == s.sync) { // the host knows that C repeats in runs (one C for the whole batch, batch-reduce): long runs
    if (wg_fits) {
      const int e = smm_jit_launch_variant(s, width | SMM_JIT_WGRUNS, stream);
      if (0 <= e) { *name = f64 ? "smm_f64_jit_shape_wgruns" : "smm_f32_jit_shape_wgruns"; return e; }
    }
    *name = f64 ? "smm_f64_jit_shape_runs" : "smm_f32_jit_shape_runs";
    return smm_jit_launch_variant(s, width | SMM_JIT_RUNS, stream);
  }
  // SYNC_DEVICE: both run forms are launched; each reads the device-side verdict (average run length) and one of them works
  *name = f64 ? "smm_f64_jit_shape_runs" : "smm_f32_jit_shape_runs";
  const int split = (0 != s.relaxed ? SMM_JIT_SPLIT : 0); // the caller's reference path is unordered as well
  // (both kernels are resolved before anything is launched: a wave form that leaves long runs to a companion that then
  // fails to compile would have to be followed by a second wave-form launch -- which would add the short runs twice)
  auto available = [&](int variant) {
    const SmmKey key = { s.typesize, s.m, s.n, s.k, s.flags & (LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_TRANS_B), variant, s.lda, s.ldb, s.ldc };
    return nullptr != smm_jit_get(key);
  };
  const bool pair = wg_fits && available(width | split | SMM_JIT_WGRUNS) && available(width | split | SMM_JIT_RUNS | SMM_JIT_HASWG);
  if (!pair) return smm_jit_launch_variant(s, width | split | SMM_JIT_RUNS, stream); // the wave form alone takes long runs as well
  int e = smm_jit_launch_variant(s, width | split | SMM_JIT_RUNS | SMM_JIT_HASWG, stream);
  if (0 == e) e = smm_jit_launch_variant(s, width | split | SMM_JIT_WGRUNS, stream);
  return e;
}

// Code objects ahead of time (no device needed): for every shape the flavours a batch call may ask for -- strided batches
// (wide accesses, several items per wave where that is chosen), index / pointer batches (element-wide accesses: streaming,
// run forms and their relaxed-order twins) -- and, if `grouped`, the fused kernel of all shapes for index batches. Returns
// the number of code objects that could not be built.
int smm_jit_prebuild(const SmmBatch* shapes, int nshapes, int grouped, int* built)
{
  int failed = 0, done = 0;
  static const int wg_env = []() { const char* e = getenv("XSMM_SMMJIT_WG"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
  auto build = [&](const std::string& src) { std::string log; if (src.empty()) return; if (0 == jit_build_offline(src, &log)) ++done; else { ++failed; if (0 != verbosity()) fprintf(stderr, "LIBXSMM-AMD: prebuild: %s\n", log.c_str()); } };
  for (int i = 0; i < nshapes; ++i) {
    SmmBatch s = shapes[i];
    s.batch = 1 << 20; s.jit_always = 1; s.sync = SYNC_NONE;
    if (!smm_jit_eligible(s)) continue;
    const int flags = s.flags & (LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_TRANS_B);
    auto one = [&](int variant) { build(gen_smm_source(s.typesize, s.m, s.n, s.k, flags, variant, s.lda, s.ldb, s.ldc)); };
    if (s.m > 32 || s.n > 32 || s.k > 64) {
      one(SMM_JIT_BIG);
      if (s.m <= 64 && s.n <= 64 && s.k <= 64 && 0 == (flags & LIBXSMM_GEMM_FLAG_TRANS_B)) { // the matrix-core forms launch_smm_jit_mfma would pick
        const size_t wlds = smm_mfma_wave_lds(s.typesize, s.m, s.n, s.k);
        const bool f64 = (8 == s.typesize);
        const size_t w2lds = f64 ? smm_mfma_wave2_lds(s.typesize, s.m, s.n, s.k) : 0;
        if (0 != wlds && 4 * wlds <= 160u * 1024u && s.lda == s.m && s.ldb == s.k && s.ldc == s.m) { one(SMM_JIT_MFMA_WAVE); one(SMM_JIT_MFMA_WAVE | SMM_JIT_SCALAR); }
        else if (0 != w2lds && 4 * w2lds <= 160u * 1024u && !(64 == s.m && 64 == s.n) && s.lda == s.m && s.ldb == s.k && s.ldc == s.m) one(SMM_JIT_MFMA_WAVE2);
        else if (!(!f64 && 64 == s.m && 64 == s.n && 64 == s.k && 64 == s.lda && 64 == s.ldb && 64 == s.ldc)) {
          const bool tight = !f64 && s.lda == s.m && s.ldb == s.k && 0 == ((s.m * s.k) & 3) && 0 == ((s.k * s.n) & 3);
          const bool tightc = !f64 && s.ldc == s.m && 0 == ((s.m * s.n) & 3) && 0 != (s.m & 31);
          one(SMM_JIT_MFMA | (tight ? SMM_JIT_MFMA_TIGHT : 0) | (tightc ? SMM_JIT_MFMA_TIGHTC : 0));
        }
      }
      continue;
    }
    // strided batches: the wide flavour with the pack the launcher would choose, and the element-wide flavour for the remainder
    SmmBatch t = s; t.mode = ADDR_STRIDED; t.sa = (long long)s.m * s.k; t.sb = (long long)s.k * s.n; t.sc = (long long)s.m * s.n;
    const int pack = smm_jit_pack(t, 0);
    one(0); if (1 < pack) one(smm_jit_pack_bits(pack));
    one(SMM_JIT_SCALAR);
    if (0 == (flags & LIBXSMM_GEMM_FLAG_BETA_0)) { // shared C only matters with beta == 1
      const bool tight = (s.lda == s.m && s.ldc == s.m && (0 != (flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (s.ldb == s.n) : (s.ldb == s.k)));
      const bool wg_fits = (tight && 0 != wg_env && smm_jit_wg_buf(s.typesize, s.m, s.n, s.k, flags) <= 65536
                         && (2 == wg_env || (size_t)s.typesize * ((size_t)s.m * s.k + (size_t)s.k * s.n) >= 12288));
      for (int split = 0; split <= SMM_JIT_SPLIT; split += SMM_JIT_SPLIT) {
        { SmmBatch r = s; r.use_mfma = 1; if (smm_mfma_runs_ok(r)) one(SMM_JIT_SCALAR | split | SMM_JIT_RUNS | SMM_JIT_MFMA_RUNS); }
        one(SMM_JIT_SCALAR | split | SMM_JIT_RUNS);
        if (wg_fits) { one(SMM_JIT_SCALAR | split | SMM_JIT_RUNS | SMM_JIT_HASWG); one(SMM_JIT_SCALAR | split | SMM_JIT_WGRUNS); }
      }
    }
  }
  if (0 != grouped && 1 < nshapes) {
    for (int relaxed = 0; relaxed < 2; ++relaxed) {
      std::vector<SmmBatch> g;
      for (int i = 0; i < nshapes; ++i) {
        SmmBatch s = shapes[i]; s.mode = ADDR_INDEX; s.batch = 1 << 20; s.sync = SYNC_DEVICE; s.relaxed = relaxed; s.jit_always = 1; s.c_atomics = 1;
        s.use_mfma = 1; // (the default policy; a process that switches the matrix cores off compiles its own)
        if (smm_jit_grouped_eligible(s) && s.typesize == shapes[0].typesize) g.push_back(s);
      }
      if (1 < g.size()) build(gen_smm_grouped_source_for(g.data(), (int)g.size()));
    }
    // one call per shape with few runs: the tiles of C as groups (smm_tile_split)
    for (int i = 0; i < nshapes; ++i) {
      SmmBatch s = shapes[i]; s.mode = ADDR_INDEX; s.batch = 64; s.sync = SYNC_DEVICE; s.relaxed = 0; s.jit_always = 1; s.c_atomics = 1; s.use_mfma = 1;
      SmmBatch tiles[4];
      const int nt = (0 == (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? smm_tile_split(s, tiles) : 0;
      if (1 < nt) build(gen_smm_grouped_source_for(tiles, nt, true));
    }
  }
  if (nullptr != built) *built = done;
  return failed;
}

} // namespace xsmm
