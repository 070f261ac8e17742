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
#include <unordered_map>

namespace xsmm {

namespace {

const char* const SMM_JIT_BODY = R"XSMM(
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
  if (1 == ad.mode) { if (nullptr == idx) return (P*)base; const int v = *(const int*)(idx + i * (long long)ad.index_stride); return (P*)base + ((long long)v - ad.index_base); }
  return *(P* const*)(base + i * stride);
}
__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int M = XM, N = XN, K = XK;
constexpr int TGM = 8, TGN = 8;
constexpr int TM = (M + TGM - 1) / TGM, TN = (N + TGN - 1) / TGN;
constexpr int AE = M * K, BE = K * N, CE = M * N;                  // elements per operand (tight leading dimensions)
constexpr int TS = (int)sizeof(T);
// widest access (in elements) that every item of a strided batch is aligned for
constexpr int vw(int elems) { return (0 == (elems * TS) % 16) ? 16 / TS : ((0 == (elems * TS) % 8) ? 8 / TS : 1); }
// XSCALAR: index/pointer batches guarantee element alignment only
constexpr int VA = XSCALAR ? 1 : vw(AE), VB = XSCALAR ? 1 : vw(BE), VC = XSCALAR ? 1 : vw(CE);
constexpr int NLA = (AE + 64 * VA - 1) / (64 * VA), NLB = (BE + 64 * VB - 1) / (64 * VB), NLC = (CE + 64 * VC - 1) / (64 * VC);
// LDS strides: A as [k][M] (lanes with equal ty read the same words, lanes with different tx adjacent ones);
// B as [n][KP] (TRANS_B: [k][NP]) with KP chosen so that the eight column groups fall into different banks
constexpr int pick_kp() { int kp = K; while (0 == (TN * kp * (TS / 4)) % 16) ++kp; return kp; }
constexpr int KP = pick_kp();
constexpr int AS_SIZE = ((K * M + TGM * TM + 3) / 4) * 4;
constexpr int BS_SIZE = XTRANSB ? (((K * N + TGN * TN + 3) / 4) * 4) : (((TGN * TN) * KP + 3) / 4) * 4;
constexpr int CS_SIZE = ((CE + 3) / 4) * 4;
constexpr int WAVE_LDS = AS_SIZE + BS_SIZE + CS_SIZE;                // elements

template<int V> struct Vec { typedef T type __attribute__((ext_vector_type(V))); };
template<> struct Vec<1> { typedef T type; };

template<int V, int NL, int E> __device__ __forceinline__ void load_flat(const T* p, int lane, T (&r)[NL][V])
{
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int e = (64 * j + lane) * V;
    if (e < E) {
      if constexpr (1 == V) r[j][0] = __builtin_nontemporal_load(p + e);
      else {
        const typename Vec<V>::type v = __builtin_nontemporal_load(reinterpret_cast<const typename Vec<V>::type*>(p + e));
#pragma unroll
        for (int q = 0; q < V; ++q) r[j][q] = v[q];
      }
    }
  }
}

// registers -> wave-private LDS (A as stored, B with the padded row stride)
__device__ __forceinline__ void park_ab(T* As, T* Bs, int lane, const T (&ra)[NLA][VA], const T (&rb)[NLB][VB])
{
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
#pragma unroll
    for (int q = 0; q < VA; ++q) { const int e = (64 * j + lane) * VA + q; if (e < AE) As[e] = ra[j][q]; }
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
#pragma unroll
    for (int q = 0; q < VB; ++q) {
      const int e = (64 * j + lane) * VB + q;
      if (e < BE) { if (XTRANSB) Bs[e] = rb[j][q]; else Bs[(e / K) * KP + (e % K)] = rb[j][q]; }
    }
  }
}
__device__ __forceinline__ void park_c(T* Cs, int lane, const T (&rc)[NLC][VC])
{
#pragma unroll
  for (int j = 0; j < NLC; ++j) {
#pragma unroll
    for (int q = 0; q < VC; ++q) { const int e = (64 * j + lane) * VC + q; if (e < CE) Cs[e] = rc[j][q]; }
  }
}
// acc(i,j) = fma(A(m,k), B(k,n), acc(i,j)) for k ascending: the reference's per-element chain
__device__ __forceinline__ void multiply(const T* As, const T* Bs, int tx, int ty, T (&acc)[TM][TN])
{
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    T av[TM], bv[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[i] = As[k * M + tx * TM + i];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = XTRANSB ? Bs[k * N + ty * TN + j] : Bs[(ty * TN + j) * KP + k];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = xfma(av[i], bv[j], acc[i][j]);
    }
  }
}
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
__device__ __forceinline__ void store_c(T* Cs, T* pc, int lane, int tx, int ty, const T (&acc)[TM][TN])
{
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int m = tx * TM + i, n = ty * TN + j; if (m < M && n < N) Cs[n * M + m] = acc[i][j]; }
  }
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < NLC; ++j) {
    const int e = (64 * j + lane) * VC;
    if (e < CE) {
      if constexpr (1 == VC) __builtin_nontemporal_store(Cs[e], pc + e);
      else __builtin_nontemporal_store(*reinterpret_cast<const typename Vec<VC>::type*>(Cs + e), reinterpret_cast<typename Vec<VC>::type*>(pc + e));
    }
  }
  wave_lds_sync();
}

// bit l: item first + l starts a run, i.e. its C differs from its predecessor's (item 0 always does)
__device__ __forceinline__ unsigned long long head_mask(const DevAddr& ad, long long first, int lane, long long batch)
{
  const long long j = first + lane;
  bool head = false;
  if (j < batch) head = (0 == j) || (resolve<T>(ad.c, ad.ic, ad.sc, ad, j - 1) != resolve<T>(ad.c, ad.ic, ad.sc, ad, j));
  return __ballot(head);
}

extern "C" __global__ __launch_bounds__(64 * XWAVES) void xsmm_smm_op(DevAddr ad, long long batch)
{
  __shared__ __attribute__((aligned(16))) T lds[XWAVES * WAVE_LDS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx = lane & 7, ty = lane >> 3;
  T* const As = lds + wave * WAVE_LDS;
  T* const Bs = As + AS_SIZE;
  T* const Cs = Bs + BS_SIZE;
  const long long w = (long long)blockIdx.x * XWAVES + wave, W = (long long)gridDim.x * XWAVES;
  T ra[NLA][VA], rb[NLB][VB], rc[NLC][VC];
#if XRUNS
  // Consecutive items that share one C form a run (CP2K stacks, batch-reduce): the wave that owns the run's first item
  // keeps C in registers and adds the products in batch order -- what the reference's sequential loop does. Chunks of 64
  // items are dealt round-robin to the waves; run heads are found 64 items at a time (one item per lane, __ballot). A
  // wave starts at the first head of its chunk and walks on item by item -- through the other runs of the chunk and, past
  // the chunk's end, to the end of the run that is still open -- with the next item's operands (and its C, if it starts a
  // run) in flight during the current item's arithmetic. A batch of distinct C blocks is the special case "all heads".
  if (nullptr != ad.flags && 0 != ad.flags[1]) return; // C blocks repeat out of order: the atomic kernel owns this batch
  for (long long chunk = w * 64; chunk < batch; chunk += W * 64) {
    unsigned long long heads = head_mask(ad, chunk, lane, batch);
    if (0 == heads) continue; // covered by a run that started in an earlier chunk
    long long base = chunk;   // `heads` describes the items [base, base + 64)
    long long i = chunk + (__ffsll((long long)heads) - 1);
    load_flat<VA, NLA, AE>(resolve<const T>(ad.a, ad.ia, ad.sa, ad, i), lane, ra);
    load_flat<VB, NLB, BE>(resolve<const T>(ad.b, ad.ib, ad.sb, ad, i), lane, rb);
    if (!XBETA0) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, i), lane, rc);
    T acc[TM][TN];
    T* pc = nullptr;
    for (;;) {
      if (0 != ((heads >> (int)(i - base)) & 1ULL)) { // item i opens a run: close the previous one, take over its C
        if (nullptr != pc) store_c(Cs, pc, lane, tx, ty, acc);
        pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, i);
        if (!XBETA0) { park_c(Cs, lane, rc); wave_lds_sync(); }
        acc_from_c(Cs, tx, ty, acc, XBETA0);
      }
      park_ab(As, Bs, lane, ra, rb);
      const long long nx = i + 1;
      bool more = (nx < batch), nx_head = false;
      if (more) {
        if (nx - base >= 64) { base += 64; heads = head_mask(ad, base, lane, batch); }
        nx_head = (0 != ((heads >> (int)(nx - base)) & 1ULL));
        more = (nx < chunk + 64) || !nx_head; // beyond the own chunk only the tail of the open run is taken
      }
      if (more) {
        load_flat<VA, NLA, AE>(resolve<const T>(ad.a, ad.ia, ad.sa, ad, nx), lane, ra);
        load_flat<VB, NLB, BE>(resolve<const T>(ad.b, ad.ib, ad.sb, ad, nx), lane, rb);
        if (nx_head && !XBETA0) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, nx), lane, rc);
      }
      wave_lds_sync();
      multiply(As, Bs, tx, ty, acc);
      wave_lds_sync();
      if (!more) break;
      i = nx;
    }
    store_c(Cs, pc, lane, tx, ty, acc);
  }
#else
  if (w >= batch) return;
  load_flat<VA, NLA, AE>(resolve<const T>(ad.a, ad.ia, ad.sa, ad, w), lane, ra);
  load_flat<VB, NLB, BE>(resolve<const T>(ad.b, ad.ib, ad.sb, ad, w), lane, rb);
  if (!XBETA0) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, w), lane, rc);
  for (long long item = w; item < batch; item += W) {
    T* const pc = resolve<T>(ad.c, ad.ic, ad.sc, ad, item);
    park_ab(As, Bs, lane, ra, rb);
    if (!XBETA0) park_c(Cs, lane, rc);
    // ---- next item's loads go out before this item's arithmetic
    const long long next = item + W;
    if (next < batch) {
      load_flat<VA, NLA, AE>(resolve<const T>(ad.a, ad.ia, ad.sa, ad, next), lane, ra);
      load_flat<VB, NLB, BE>(resolve<const T>(ad.b, ad.ib, ad.sb, ad, next), lane, rb);
      if (!XBETA0) load_flat<VC, NLC, CE>(resolve<const T>(ad.c, ad.ic, ad.sc, ad, next), lane, rc);
    }
    wave_lds_sync();
    T acc[TM][TN];
    acc_from_c(Cs, tx, ty, acc, XBETA0);
    multiply(As, Bs, tx, ty, acc);
    store_c(Cs, pc, lane, tx, ty, acc);
  }
#endif
}
)XSMM";

struct SmmKey {
  int typesize, m, n, k, flags, variant;
  bool operator==(const SmmKey& o) const { return typesize == o.typesize && m == o.m && n == o.n && k == o.k && flags == o.flags && variant == o.variant; }
};
struct SmmKeyHash { size_t operator()(const SmmKey& k) const { return (size_t)((((k.m * 131 + k.n) * 131 + k.k) * 8 + k.flags * 2 + (k.typesize == 8)) * 4 + k.variant); } };

std::mutex g_smm_lock;
std::unordered_map<SmmKey, JitKernel*, SmmKeyHash> g_smm_cache; // nullptr value: compilation failed, do not retry

} // namespace

static int smm_jit_waves(int typesize, int m, int n, int k, int flags);

std::string gen_smm_source(int typesize, int m, int n, int k, int flags, int variant)
{
  std::string s = "// generated by libxsmm-amd (dense SMM kernel, shape baked in)\n";
  s += std::string("typedef ") + (8 == typesize ? "double" : "float") + " T;\n";
  s += "#define XM " + std::to_string(m) + "\n#define XN " + std::to_string(n) + "\n#define XK " + std::to_string(k) + "\n";
  s += std::string("#define XBETA0 ") + ((flags & LIBXSMM_GEMM_FLAG_BETA_0) ? "1" : "0") + "\n";
  s += std::string("#define XTRANSB ") + ((flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? "1" : "0") + "\n";
  s += "#define XWAVES " + std::to_string(smm_jit_waves(typesize, m, n, k, flags)) + "\n";
  s += std::string("#define XSCALAR ") + ((variant & SMM_JIT_SCALAR) ? "1" : "0") + "\n"; // element-wide loads/stores only
  s += std::string("#define XRUNS ") + ((variant & SMM_JIT_RUNS) ? "1" : "0") + "\n";     // runs of equal C accumulate in registers
  s += SMM_JIT_BODY;
  return s;
}

// LDS bytes one wave of the generated kernel needs (mirrors the constexpr arithmetic of the source)
static size_t smm_jit_wave_lds(int typesize, int m, int n, int k, int flags)
{
  const int tm = (m + 7) / 8, tn = (n + 7) / 8;
  int kp = k; while (0 == (tn * kp * (typesize / 4)) % 16) ++kp;
  const size_t as = ((size_t)(k * m + 8 * tm + 3) / 4) * 4;
  const size_t bs = (flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (((size_t)(k * n + 8 * tn + 3) / 4) * 4) : ((((size_t)8 * tn) * kp + 3) / 4) * 4;
  const size_t cs = ((size_t)(m * n + 3) / 4) * 4;
  return (as + bs + cs) * typesize;
}

// wavefronts per work-group: as many (4, 2, 1) as fit 64 KiB of static LDS; 0 if even one wave does not fit
static int smm_jit_waves(int typesize, int m, int n, int k, int flags)
{
  const size_t w = smm_jit_wave_lds(typesize, m, n, k, flags);
  return (4 * w <= 65536) ? 4 : ((2 * w <= 65536) ? 2 : ((w <= 65536) ? 1 : 0));
}

bool smm_jit_eligible(const SmmBatch& s)
{
  const char* const env_jit = getenv("LIBXSMM_AMD_JIT"); // re-read on every call: tests and tools toggle it
  const bool enabled = (nullptr == env_jit || 0 != atoi(env_jit));
  if (!enabled || 0 != s.general || SYNC_ATOMIC == s.sync) return false;
  if (SYNC_NONE != s.sync && 0 != (s.flags & LIBXSMM_GEMM_FLAG_BETA_0)) return false; // (never chosen: beta == 0 needs no care)
  if (s.lda != s.m || s.ldc != s.m) return false;
  if (0 != (s.flags & LIBXSMM_GEMM_FLAG_TRANS_B) ? (s.ldb != s.n) : (s.ldb != s.k)) return false;
  if (s.m > 32 || s.n > 32 || s.k > 64) return false;                       // 8x8 lanes x (<=4x4) tile
  if (0 == smm_jit_waves(s.typesize, s.m, s.n, s.k, s.flags)) return false;                // static LDS limit per work-group
  const char* const env_min = getenv("LIBXSMM_AMD_JIT_MINBATCH");
  const long long min_batch = (nullptr != env_min && 0 != *env_min) ? atoll(env_min) : 16384LL;
  if (s.batch < min_batch) return false;                                      // compile time must be worth it
  return true;
}

// Which flavour of the generated kernel a batch needs: strided batches of tightly packed items whose bases are 16-byte
// aligned use the widest loads the item size allows; index/pointer batches (and anything else) are only known to be
// element-aligned. Runs of equal C (SYNC_RUNS) take the accumulate-in-registers form.
static int smm_jit_variant(const SmmBatch& s)
{
  int v = (SYNC_RUNS == s.sync || SYNC_DEVICE == s.sync) ? SMM_JIT_RUNS : 0;
  bool wide = false;
  if (ADDR_STRIDED == s.mode) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(s.a) | reinterpret_cast<uintptr_t>(s.b) | reinterpret_cast<uintptr_t>(s.c);
    wide = (0 == (bits & 15))
        && (s.sa == (long long)s.m * s.k || 0 == s.sa) && (s.sb == (long long)s.k * s.n || 0 == s.sb)
        && (s.sc == (long long)s.m * s.n || 0 == s.sc);
  }
  if (!wide) v |= SMM_JIT_SCALAR;
  return v;
}

int launch_smm_jit(const SmmBatch& s, void* stream, const char** name)
{ // returns -1 when no specialised kernel is available
  const SmmKey key = { s.typesize, s.m, s.n, s.k, s.flags & (LIBXSMM_GEMM_FLAG_BETA_0 | LIBXSMM_GEMM_FLAG_TRANS_B), smm_jit_variant(s) };
  JitKernel* k = nullptr;
  {
    std::lock_guard<std::mutex> guard(g_smm_lock);
    auto it = g_smm_cache.find(key);
    if (it != g_smm_cache.end()) k = it->second;
    else {
      std::string log;
      k = jit_compile(gen_smm_source(key.typesize, key.m, key.n, key.k, key.flags, key.variant), "xsmm_smm_op", &log);
      if (nullptr == k && 0 != verbosity()) fprintf(stderr, "LIBXSMM WARNING: SMM JIT failed (%s); using the pre-compiled kernel\n", log.c_str());
      g_smm_cache.emplace(key, k);
    }
  }
  if (nullptr == k) return -1;
  struct { const char* a; const char* b; char* c; const char* ia; const char* ib; const char* ic; long long sa, sb, sc; int index_base, index_stride, mode; const int* flags; } ad;
  ad.a = (const char*)s.a; ad.b = (const char*)s.b; ad.c = (char*)s.c; ad.ia = (const char*)s.ia; ad.ib = (const char*)s.ib; ad.ic = (const char*)s.ic;
  ad.sa = s.sa; ad.sb = s.sb; ad.sc = s.sc; ad.index_base = s.index_base; ad.index_stride = s.index_stride; ad.mode = s.mode;
  ad.flags = (SYNC_DEVICE == s.sync ? s.devflags : nullptr);
  long long batch = s.batch;
  const int waves = smm_jit_waves(s.typesize, s.m, s.n, s.k, s.flags);
  const size_t lds = (size_t)waves * smm_jit_wave_lds(s.typesize, s.m, s.n, s.k, s.flags);
  long long per_cu = (long long)((160 * 1024) / (lds ? lds : 1));
  if (per_cu * waves > 16) per_cu = 16 / waves; // the streaming rate peaks around 12-16 waves per CU
  if (per_cu < 1) per_cu = 1;
  static const int bpc_env = []() { const char* e = getenv("XSMM_SMMJIT_BPC"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
  if (0 < bpc_env) per_cu = bpc_env;
  // run form: a wave scans chunks of 64 items for run heads, so the grid is sized by chunks
  const long long units = (0 != (key.variant & SMM_JIT_RUNS)) ? ((batch + 63) / 64) : batch;
  long long blocks = (units + waves - 1) / waves;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  if (blocks < 1) blocks = 1;
  *name = (0 != (key.variant & SMM_JIT_RUNS)) ? ((8 == s.typesize) ? "smm_f64_jit_shape_runs" : "smm_f32_jit_shape_runs")
                                                : ((8 == s.typesize) ? "smm_f64_jit_shape" : "smm_f32_jit_shape");
  return jit_launch_raw(k, (unsigned)blocks, 64u * (unsigned)waves, &ad, sizeof(ad), &batch, stream);
}

} // namespace xsmm
