// smm_generic.hip -- batched small dense GEMM for gfx950, any (M,N,K), any leading dimensions, f32/f64.
//
// Replaces the reference's dense JIT back end (src/generator_gemm*.c) behind libxsmm_mmbatch_kernel
// (src/libxsmm_gemm.c:1315-1608). Contract per C element (reference src/generator_gemm_noarch.c:59-84 and the
// AVX2 microkernel): C[n*ldc+m] = (beta ? C : 0) + sum over k ascending of A[k*lda+m]*B[n*ldb+k]
// (TRANS_B: B[k*ldb+n]), one fused multiply-add per step -- this kernel keeps exactly that chain, so its
// result is bit-identical to a k-ordered fma() loop (the reference's AVX2/AVX-512 path).
//
// Mapping: a "unit" is a group of G threads (G=64: one wavefront, G=256: one work-group) that owns one
// problem at a time: A (K x M) and B (N x K) are staged through LDS with coalesced loads along the
// contiguous axis, every thread keeps a TM x TN register tile of C, C is read/written directly in HBM.
// Units stride over the batch, so consecutive units stream consecutive problems.
#include "smm_common.cuh"
#include <cstring>

namespace xsmm {
namespace {

// *p += v by compare-and-swap at system scope (integer atomics travel over PCIe; hardware floating-point adds do not)
__device__ __forceinline__ void cas_add(float* p, float v)
{
  unsigned* const u = reinterpret_cast<unsigned*>(p);
  unsigned seen = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  while (!__hip_atomic_compare_exchange_strong(u, &seen, __float_as_uint(__uint_as_float(seen) + v), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {}
}
__device__ __forceinline__ void cas_add(double* p, double v)
{
  unsigned long long* const u = reinterpret_cast<unsigned long long*>(p);
  unsigned long long seen = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  while (!__hip_atomic_compare_exchange_strong(u, &seen, (unsigned long long)__double_as_longlong(__longlong_as_double((long long)seen) + v), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {}
}

template<int G> __device__ __forceinline__ void unit_sync()
{
  if constexpr (G == 64) wave_lds_sync(); else __syncthreads();
}

// T: element type; TM,TN: register tile; TGM x TGN = G threads per problem; GENERAL: alpha/beta/TRANS_A form
template<typename T, int TM, int TN, int TGM, int TGN, bool GENERAL>
__global__ __launch_bounds__(256)
void smm_generic_kernel(DevAddr ad, int M, int N, int K, int lda, int ldb, int ldc, int flags, int sync_arg,
                        long long batch_arg, int KC, int kshift, T alpha, T beta, int tiles_m, int tiles_n, int hw_atomics,
                        const unsigned long long* batch_ptr)
{
  // (deferred per-call kernels: the number of items was not known when this launch was queued -- the gate kernel in front of
  // it on the stream has left it in *batch_ptr)
  if (nullptr != batch_ptr) batch_arg = (long long)(*batch_ptr);
  // tiles_m * tiles_n > 1 (independent C only): a unit is one MP x NP tile of one item -- a single large product
  // (libxsmm_?gemm, a relinked BLAS caller) spreads over the chip instead of running on one work-group
  const int tiles = tiles_m * tiles_n;
  const long long batch = batch_arg * tiles;
  int sync = sync_arg;
  if (SYNC_DEVICE == sync_arg) { // how C blocks repeat was established on the device (c_order_kernel, same stream)
    sync = (0 != ad.flags[1]) ? SYNC_ATOMIC : ((0 != ad.flags[0]) ? SYNC_RUNS : SYNC_NONE);
  }
  constexpr int G = TGM * TGN;
  constexpr int PPB = 256 / G;
  constexpr int MP = TGM * TM;
  constexpr int NP = TGN * TN;
  extern __shared__ __align__(16) unsigned char smem_raw[];

  const int slot = threadIdx.x / G;
  const int t = threadIdx.x % G;
  const int tx = t % TGM, ty = t / TGM;
  const int KP = KC | 1;                              // odd row stride: conflict-free column reads of B
  const size_t slot_elems = (size_t)KC * MP + (size_t)NP * KP;
  T* const As = reinterpret_cast<T*>(smem_raw) + (size_t)slot * slot_elems;   // As[kc][MP]
  T* const Bs = As + (size_t)KC * MP;                                          // Bs[NP][KP]
  const bool transa = GENERAL && (0 != (flags & LIBXSMM_GEMM_FLAG_TRANS_A));
  const bool transb = (0 != (flags & LIBXSMM_GEMM_FLAG_TRANS_B));
  const bool beta0 = (0 != (flags & LIBXSMM_GEMM_FLAG_BETA_0));
  const int KL = 1 << kshift;                         // lanes along the contiguous axis while staging B

  const long long unit = (long long)blockIdx.x * PPB + slot;
  const long long nunits = (long long)gridDim.x * PPB;

  // Run mode: only the head of a run works. Dealt round-robin, the heads of runs of a uniform length that divides the number
  // of units (blocked GEMM: 32 k blocks per C block, 4096 units) would all land on the same few units -- every unit takes a
  // contiguous slice of the batch instead (a run that starts in the slice is walked to its end, wherever that is).
  const bool sliced = (SYNC_RUNS == sync);
  const long long per = sliced ? ((batch + nunits - 1) / nunits) : 1;
  const long long lo = sliced ? unit * per : unit, hi = sliced ? ((lo + per < batch) ? lo + per : batch) : batch, step = sliced ? 1 : nunits;
  for (long long work = lo; work < hi; work += step) {
    const long long item = (1 < tiles) ? work / tiles : work;
    const int tile = (1 < tiles) ? (int)(work - item * tiles) : 0;
    const int m_first = (1 < tiles) ? (tile % tiles_m) * MP : 0, m_last = (1 < tiles) ? m_first + MP : M;
    const int n_first = (1 < tiles) ? (tile / tiles_m) * NP : 0, n_last = (1 < tiles) ? n_first + NP : N;
    T* const pc = addr_c<T>(ad, item);
    long long count = 1;
    if (SYNC_RUNS == sync) { // only the head of a run of equal C works; it walks the run in batch order
      if (0 < item && addr_c<T>(ad, item - 1) == pc) continue;
      while (item + count < batch && addr_c<T>(ad, item + count) == pc) ++count;
    }
    for (int m0 = m_first; m0 < M && m0 < m_last; m0 += MP) {
      for (int n0 = n_first; n0 < N && n0 < n_last; n0 += NP) {
        T acc[TM][TN];
        T cv[GENERAL ? TM : 1][GENERAL ? TN : 1]; // general form: the value of C as the items of a run update it one after the other
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int m = m0 + tx * TM + i, n = n0 + ty * TN + j;
            acc[i][j] = (!GENERAL && !beta0 && SYNC_ATOMIC != sync && m < M && n < N) ? pc[(size_t)n * ldc + m] : T(0);
            if constexpr (GENERAL) cv[i][j] = (T(0) != beta && m < M && n < N) ? pc[(size_t)n * ldc + m] : T(0);
          }
        }
        for (long long r = 0; r < count; ++r) {
          const T* const pa = addr_a<T>(ad, item + r);
          const T* const pb = addr_b<T>(ad, item + r);
          if constexpr (GENERAL) { // C = alpha * A_i * B_i + beta * C item by item, in batch order (the reference's sequential loop, src/libxsmm_gemm.c:1778-1806)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = T(0);
            }
          }
          for (int k0 = 0; k0 < K; k0 += KC) {
            const int kc = (K - k0 < KC) ? (K - k0) : KC;
            unit_sync<G>(); // previous tile fully consumed
            // stage A[k0..k0+kc) x [m0..m0+MP): lanes run along m (contiguous in memory unless TRANS_A)
            for (int idx = t; idx < kc * MP; idx += G) {
              const int kk = idx / MP, mm = idx % MP;
              const int m = m0 + mm;
              T v = T(0);
              if (m < M) v = transa ? pa[(size_t)m * lda + (k0 + kk)] : pa[(size_t)(k0 + kk) * lda + m];
              As[idx] = v;
            }
            // stage B: Bs[nn][kk] = B[k0+kk][n0+nn]
            if (!transb) { // B[n*ldb+k]: lanes along k
              const int kk = t & (KL - 1);
              for (int nn = t >> kshift; nn < NP; nn += (G >> kshift)) {
                const int n = n0 + nn;
                for (int kb = kk; kb < kc; kb += KL) {
                  Bs[nn * KP + kb] = (n < N) ? pb[(size_t)n * ldb + (k0 + kb)] : T(0);
                }
              }
            }
            else { // B[k*ldb+n]: lanes along n
              for (int idx = t; idx < kc * NP; idx += G) {
                const int kk = idx / NP, nn = idx % NP;
                const int n = n0 + nn;
                Bs[nn * KP + kk] = (n < N) ? pb[(size_t)(k0 + kk) * ldb + n] : T(0);
              }
            }
            unit_sync<G>();
            for (int kk = 0; kk < kc; ++kk) {
              T av[TM], bv[TN];
#pragma unroll
              for (int i = 0; i < TM; ++i) av[i] = As[kk * MP + tx * TM + i];
#pragma unroll
              for (int j = 0; j < TN; ++j) bv[j] = Bs[(ty * TN + j) * KP + kk];
#pragma unroll
              for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = xfma(av[i], bv[j], acc[i][j]);
              }
            }
          }
          if constexpr (GENERAL) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
              for (int j = 0; j < TN; ++j) cv[i][j] = (T(0) == beta) ? (alpha * acc[i][j]) : (alpha * acc[i][j] + beta * cv[i][j]);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int m = m0 + tx * TM + i, n = n0 + ty * TN + j;
            if (m < M && n < N) {
              T* const dst = pc + (size_t)n * ldc + m;
              if constexpr (GENERAL) {
                *dst = cv[i][j];
              }
              else if (SYNC_ATOMIC == sync) {
                if (beta0) *dst = acc[i][j];
                else if (0 != hw_atomics) atomicAdd(dst, acc[i][j]);
                else cas_add(dst, acc[i][j]); // C in host memory the GPU maps: floating-point atomics do not reach it
              }
              else {
                *dst = acc[i][j];
              }
            }
          }
        }
      }
    }
  }
}

// Counts adjacent C operands that are equal (out[0]) or decreasing (out[1]). No launch in front of it has to clear
// anything: every block leaves its counts in its own pair out[4 + 2*block ..], the block that arrives last (a ticket
// counter, out[2], which atomicInc wraps back to zero) adds the pairs up and writes the totals.
// (bid, nblocks: this block's index among the blocks that inspect this batch)
template<typename T>
__device__ __forceinline__ void c_order_body(const DevAddr& ad, long long batch, int* out, unsigned bid, unsigned nblocks, int peers)
{ // peers: batches that run side by side with this one in one launch (out[3]: the run kernels decide with it whether this batch's few
  // long runs have to be cut into segments to fill the chip)
  __shared__ int red[2][4];
  __shared__ bool last;
  int eq = 0, dec = 0;
  for (long long i = (long long)bid * blockDim.x + threadIdx.x + 1; i < batch; i += (long long)nblocks * blockDim.x) {
    const T* const c0 = addr_c<T>(ad, i - 1);
    const T* const c1 = addr_c<T>(ad, i);
    eq += (c1 == c0) ? 1 : 0;
    dec += (c1 < c0) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) { eq += __shfl_xor(eq, o); dec += __shfl_xor(dec, o); }
  if (0 == (threadIdx.x & 63)) { red[0][threadIdx.x >> 6] = eq; red[1][threadIdx.x >> 6] = dec; }
  __syncthreads();
  if (0 == threadIdx.x) {
    __hip_atomic_store(out + 4 + 2 * bid, red[0][0] + red[0][1] + red[0][2] + red[0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + 5 + 2 * bid, red[1][0] + red[1][1] + red[1][2] + red[1][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = (nblocks - 1 == atomicInc(reinterpret_cast<unsigned*>(out + 2), nblocks - 1));
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  eq = dec = 0;
  for (unsigned b = threadIdx.x; b < nblocks; b += blockDim.x) {
    eq += __hip_atomic_load(out + 4 + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dec += __hip_atomic_load(out + 5 + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int o = 32; o > 0; o >>= 1) { eq += __shfl_xor(eq, o); dec += __shfl_xor(dec, o); }
  __syncthreads();
  if (0 == (threadIdx.x & 63)) { red[0][threadIdx.x >> 6] = eq; red[1][threadIdx.x >> 6] = dec; }
  __syncthreads();
  if (0 == threadIdx.x) {
    out[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    out[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out[3] = peers;
  }
}

template<typename T>
__global__ __launch_bounds__(256) void c_order_kernel(DevAddr ad, long long batch, int* out)
{
  c_order_body<T>(ad, batch, out, blockIdx.x, gridDim.x, 1);
}

// the same for up to ORDER_GROUPS batches in one launch (one verdict slot each): CP2K-style calls with a batch per shape
constexpr int ORDER_GROUPS = 32, ORDER_GROUP_BLOCKS = 16;
struct OrderGroup { DevAddr ad; long long batch; int* out; };
struct OrderGroups { OrderGroup g[ORDER_GROUPS]; int peers; };
static_assert(sizeof(OrderGroups) <= 4096, "passed by value");
template<typename T>
__global__ __launch_bounds__(256) void c_order_groups_kernel(OrderGroups tab)
{
  const unsigned g = blockIdx.x / ORDER_GROUP_BLOCKS;
  c_order_body<T>(tab.g[g].ad, tab.g[g].batch, tab.g[g].out, blockIdx.x % ORDER_GROUP_BLOCKS, ORDER_GROUP_BLOCKS, tab.peers);
}

template<typename T, int TM, int TGM, bool GENERAL>
int launch_generic_t(const SmmBatch& s, hipStream_t stream)
{
  constexpr int G = TGM * TGM, PPB = 256 / G, MP = TGM * TM, NP = TGM * TM;
  // K chunk: keep a unit's LDS tile <= 16 KiB (G=64) / 48 KiB (G=256) so several work-groups fit a CU
  const size_t budget = (G == 64 ? 16384 : 49152) / sizeof(T);
  int KC = s.k;
  while (KC > 1 && ((size_t)KC * MP + (size_t)NP * (KC | 1)) > budget) KC = (KC + 1) / 2;
  if (KC < 1) KC = 1;
  const size_t slot_elems = (size_t)KC * MP + (size_t)NP * (KC | 1);
  const size_t smem = slot_elems * sizeof(T) * PPB;
  int kshift = 0;
  while ((1 << kshift) < KC && (1 << kshift) < G) ++kshift; // lanes along k when staging B
  // independent C operands and more than one tile per item: tiles are units of their own
  int tiles_m = 1, tiles_n = 1;
  if (SYNC_NONE == s.sync && (s.m > MP || s.n > NP)) { tiles_m = (s.m + MP - 1) / MP; tiles_n = (s.n + NP - 1) / NP; }
  const long long units = s.batch * tiles_m * tiles_n;
  long long blocks = (units + PPB - 1) / PPB;
  const long long maxblocks = 256LL * 16; // 256 CUs, enough resident groups; units stride over the batch
  if (blocks > maxblocks) blocks = maxblocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((smm_generic_kernel<T, TM, TM, TGM, TGM, GENERAL>), dim3((unsigned)blocks), dim3(256), smem, stream,
    make_addr(s), s.m, s.n, s.k, s.lda, s.ldb, s.ldc, s.flags, s.sync, s.batch, KC, kshift, (T)s.alpha, (T)s.beta, tiles_m, tiles_n,
    (SYNC_DEVICE == s.sync || SYNC_ATOMIC == s.sync) ? s.c_atomics : 1, s.batch_ptr);
  return (int)hipGetLastError();
}

template<typename T, bool GENERAL>
int launch_generic(const SmmBatch& s, hipStream_t stream, const char** name)
{
  const int mx = (s.m > s.n ? s.m : s.n);
  static const char* const names_f32[] = {
    "smm_f32_generic_w8", "smm_f32_generic_w16", "smm_f32_generic_w24", "smm_f32_generic_w32",
    "smm_f32_generic_g48", "smm_f32_generic_g64" };
  static const char* const names_f64[] = {
    "smm_f64_generic_w8", "smm_f64_generic_w16", "smm_f64_generic_w24", "smm_f64_generic_w32",
    "smm_f64_generic_g48", "smm_f64_generic_g64" };
  const char* const* names = (sizeof(T) == 4 ? names_f32 : names_f64);
  if (mx <= 8) { *name = names[0]; return launch_generic_t<T, 1, 8, GENERAL>(s, stream); }
  if (mx <= 16) { *name = names[1]; return launch_generic_t<T, 2, 8, GENERAL>(s, stream); }
  if (mx <= 24) { *name = names[2]; return launch_generic_t<T, 3, 8, GENERAL>(s, stream); }
  if (mx <= 32) { *name = names[3]; return launch_generic_t<T, 4, 8, GENERAL>(s, stream); }
  if (mx <= 48) { *name = names[4]; return launch_generic_t<T, 3, 16, GENERAL>(s, stream); }
  *name = names[5]; return launch_generic_t<T, 4, 16, GENERAL>(s, stream);
}

} // namespace

int launch_smm_generic(const SmmBatch& s, void* stream, const char** name)
{
  hipStream_t st = (hipStream_t)stream;
  if (0 != s.general) {
    return (8 == s.typesize) ? launch_generic<double, true>(s, st, name) : launch_generic<float, true>(s, st, name);
  }
  return (8 == s.typesize) ? launch_generic<double, false>(s, st, name) : launch_generic<float, false>(s, st, name);
}

int launch_c_order_check(const SmmBatch& s, int* d_out, void* stream)
{
  hipStream_t st = (hipStream_t)stream;
  long long blocks = (s.batch + 255) / 256;
  if (blocks > FLAG_SLOT_BLOCKS) blocks = FLAG_SLOT_BLOCKS;
  if (blocks < 1) blocks = 1;
  if (8 == s.typesize) hipLaunchKernelGGL((c_order_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, make_addr(s), s.batch, d_out);
  else hipLaunchKernelGGL((c_order_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, make_addr(s), s.batch, d_out);
  return (int)hipGetLastError();
}

// Gate of a burst of deferred per-call kernels: one lane waits until the host has sealed the burst (bit 63 of *word, which
// lives in host memory the GPU reads directly) and leaves the number of recorded calls for the batch kernel that is queued
// right behind it. The wait is bounded: the host's helper thread seals a burst a few dozen microseconds after the last call;
// should that not happen (the process was stopped), the gate seals the burst itself after `limit_ticks` of the 100 MHz wall
// clock -- atomically, like the helper would: the caller's next call finds the seal and starts a new burst, nothing is lost.
__global__ __launch_bounds__(64) void defer_gate_kernel(unsigned long long* word, unsigned long long* count_out, unsigned long long limit_ticks)
{
  if (0 != threadIdx.x) return;
  const unsigned long long t0 = wall_clock64();
  unsigned long long w;
  for (;;) {
    w = __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (0 != (w >> 63)) break;
    if (wall_clock64() - t0 > limit_ticks) {
      w = __hip_atomic_fetch_or(word, 1ULL << 63, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM) | (1ULL << 62);
      break;
    }
    __builtin_amdgcn_s_sleep(32);
  }
  count_out[0] = w & 0xFFFFFFFFULL;
  count_out[1] = (w >> 62) & 1ULL;
}

int launch_defer_gate(unsigned long long* word, unsigned long long* count_out, void* stream)
{
  hipLaunchKernelGGL(defer_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, count_out, 200000000ULL /* 2 s */);
  return (int)hipGetLastError();
}

int launch_c_order_check_groups(const SmmBatch* groups, int ngroups, void* stream)
{
  if (ngroups < 1 || ngroups > ORDER_GROUPS) return -1;
  OrderGroups tab; memset(&tab, 0, sizeof(tab));
  for (int g = 0; g < ngroups; ++g) {
    if (groups[g].typesize != groups[0].typesize || nullptr == groups[g].devflags) return -1;
    SmmBatch s = groups[g]; s.sync = SYNC_NONE; // (make_addr: no flags pointer needed here)
    tab.g[g].ad = make_addr(s); tab.g[g].batch = s.batch; tab.g[g].out = const_cast<int*>(groups[g].devflags);
  }
  tab.peers = ngroups;
  hipStream_t st = (hipStream_t)stream;
  if (8 == groups[0].typesize) hipLaunchKernelGGL((c_order_groups_kernel<double>), dim3((unsigned)(ngroups * ORDER_GROUP_BLOCKS)), dim3(256), 0, st, tab);
  else hipLaunchKernelGGL((c_order_groups_kernel<float>), dim3((unsigned)(ngroups * ORDER_GROUP_BLOCKS)), dim3(256), 0, st, tab);
  return (int)hipGetLastError();
}

} // namespace xsmm
