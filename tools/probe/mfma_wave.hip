// Probe: wave-per-item matrix-core kernel for 32 < max(M,N) <= 64 built from 16x16x4 tiles (fp32 and fp64), shape baked in.
// hipcc -O3 --offload-arch=gfx950 -DXM=40 -DXN=40 -DXK=40 -DXF64=0 tools/probe/mfma_wave.hip -o /tmp/mfma_wave && /tmp/mfma_wave [batch] [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef XKUNROLL
#define XKUNROLL 2
#endif
#ifndef XF64
#define XF64 0
#endif
#ifndef XBETA0
#define XBETA0 0
#endif
#ifndef XCLDS
#define XCLDS 0
#endif
#if XF64
typedef double T; typedef double V __attribute__((ext_vector_type(2))); typedef double ACC __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#else
typedef float T; typedef float V __attribute__((ext_vector_type(4))); typedef float ACC __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif
constexpr int M = XM, N = XN, K = XK;
constexpr int VEC = 16 / sizeof(T);
constexpr int MI = (M + 15) / 16, NI = (N + 15) / 16, KS = K / 4;
constexpr int MS = (M <= 16) ? 16 : (M <= 48 ? 48 : 64);
constexpr bool ASWZ = (64 == MS);
constexpr int kstride() { int s = ((K + VEC - 1) / VEC); if (0 == (s & 1)) ++s; return s * VEC; }
constexpr int KSD = kstride();
// C image (XCLDS): column stride such that the four column groups of a wave access fall into different banks
constexpr int cstride() { int s = M; for (;; s += VEC) { if (XF64 ? (16 == s % 32) : (4 == s % 16 || 12 == s % 16)) break; } return s; }
constexpr int CSD = cstride();
constexpr int C_ELEMS = N * CSD;
constexpr int A_ELEMS = (C_ELEMS > K * MS) ? C_ELEMS : K * MS, B_ELEMS = N * KSD;
__device__ __forceinline__ int clampi(int v, int hi) { return v < hi ? v : hi; }
#define NROW(r) (XF64 ? (lq + 4 * (r)) : (4 * lq + (r)))
constexpr int CA = (M * K / VEC + 63) / 64, CB = (K * N / VEC + 63) / 64, CC = (M * N / VEC + 63) / 64;
constexpr size_t LDS_BYTES = (size_t)(A_ELEMS + B_ELEMS + 64) * sizeof(T);
static_assert(0 == M % VEC && 0 == K % 4, "shape");

#ifndef XWPE
#define XWPE 2
#endif
// One wave per item. Per item: C, A and B arrive as whole 16-byte chunks of the contiguous arrays (lanes past the end of
// an array repeat its last chunk: no divergent control flow around memory instructions, so the compiler's wait counts stay
// exact), C is redistributed through LDS into the tile layout of the accumulators, A and B are parked as LDS images that
// the operand fetches of v_mfma_*_16x16x4 read conflict-free, the result goes back through LDS and leaves as whole lines.
// The stores of item i are issued at the top of iteration i + 1, before the loads of item i + 2: whenever the wave waits
// for its loads nothing younger is in flight, so it never waits for a store to complete.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XWPE))) void kern(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ c, long long batch)
{
  extern __shared__ __align__(16) unsigned char smem[];
  T* const As = reinterpret_cast<T*>(smem);
  T* const Bs = As + A_ELEMS;
  T* const Cs = As;                      // the image of C shares the place of A's (never alive together)
  T* const dummy = Bs + B_ELEMS;         // a word per lane for the writes of lanes outside C
  const int lane = threadIdx.x, l16 = lane & 15, lq = lane >> 4;
  V ra[CA], rb[CB], rc[CC];
  auto load_ab = [&](long long item) {
    const V* const pa = reinterpret_cast<const V*>(a + item * (M * K));
    const V* const pb = reinterpret_cast<const V*>(b + item * (K * N));
#pragma unroll
    for (int j = 0; j < CA; ++j) ra[j] = __builtin_nontemporal_load(pa + clampi(64 * j + lane, M * K / VEC - 1));
#pragma unroll
    for (int j = 0; j < CB; ++j) rb[j] = __builtin_nontemporal_load(pb + clampi(64 * j + lane, K * N / VEC - 1));
  };
  auto load_c = [&](long long item) {
    const V* const pc = reinterpret_cast<const V*>(c + item * (M * N));
#pragma unroll
    for (int j = 0; j < CC; ++j) rc[j] = __builtin_nontemporal_load(pc + clampi(64 * j + lane, M * N / VEC - 1));
  };
  auto lds_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  auto store_c = [&](long long item) { // the image of C -> memory, whole lines
    V* const pc = reinterpret_cast<V*>(c + item * (M * N));
#pragma unroll
    for (int j = 0; j < CC; ++j) {
      const int ch = clampi(64 * j + lane, M * N / VEC - 1), e = ch * VEC, n = e / M, m = e % M;
      __builtin_nontemporal_store(*reinterpret_cast<const V*>(Cs + n * CSD + m), pc + ch);
    }
  };
  long long item = blockIdx.x, prev = -1;
  if (item >= batch) return;
  load_ab(item);
  if (!XBETA0) load_c(item);
  for (;;) {
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this item's operands (the only younger instructions are older stores)
    if (0 <= prev) { store_c(prev); lds_sync(); }
    ACC acc[NI][MI];
    if (!XBETA0) {
#pragma unroll
      for (int j = 0; j < CC; ++j) {
        const int ch = clampi(64 * j + lane, M * N / VEC - 1), e = ch * VEC, n = e / M, m = e % M;
        *reinterpret_cast<V*>(Cs + n * CSD + m) = rc[j];
      }
      lds_sync();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = clampi(16 * ni + NROW(r), N - 1), m = clampi(16 * mi + l16, M - 1);
            acc[ni][mi][r] = Cs[n * CSD + m];
          }
      lds_sync();
    }
    else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = ACC{ 0, 0, 0, 0 };
    }
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      const int ch = clampi(64 * j + lane, M * K / VEC - 1), e = ch * VEC, k = e / M, m = e % M;
      *reinterpret_cast<V*>(As + k * MS + (ASWZ ? (m ^ ((k & 3) << 4)) : m)) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int ch = clampi(64 * j + lane, K * N / VEC - 1), e = ch * VEC, n = e / K, k = e % K;
      *reinterpret_cast<V*>(Bs + n * KSD + k) = rb[j];
    }
    const long long next = item + gridDim.x;
    if (next < batch) { load_ab(next); if (!XBETA0) load_c(next); }
    lds_sync();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      T af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) { const int m = 16 * mi + l16; af[mi] = As[(4 * ks + lq) * MS + (ASWZ ? (m ^ (lq << 4)) : m)]; }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) { const int n = clampi(16 * ni + l16, N - 1); bf[ni] = Bs[n * KSD + 4 * ks + lq]; }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MFMA(bf[ni], af[mi], acc[ni][mi]);
    }
    lds_sync(); // (the images are dead now)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 16 * ni + NROW(r), m = 16 * mi + l16;
          const bool inside = (16 * ni + 15 < N || n < N) && (16 * mi + 15 < M || m < M);
          T* const dst = inside ? Cs + n * CSD + m : dummy + lane;
          *dst = acc[ni][mi][r];
        }
    lds_sync();
    prev = item;
    if (next >= batch) break;
    item = next;
  }
  store_c(prev);
}

#if defined(XNSPLIT) && (2 == XNSPLIT)
// Variant for items whose images do not leave room for four waves per CU (fp64 56^3: 55 KB): the columns of C are worked
// on in two halves against one image of A. Per half: its C and B columns arrive, C goes through the B region into the
// accumulators, B is parked there, after the arithmetic the result waits there for the deferred stores. The operands of the
// next half (or the next item's A and first half) are in flight meanwhile.
#ifndef XAMS
#define XAMS MS
#endif
constexpr int AMS = XAMS;                                  // row stride of A's image
constexpr int NH = N / 2, NIH = (NH + 15) / 16;
constexpr int CSH = M;                                     // column stride of the half C image
constexpr int BH_ELEMS = (NH * KSD > NH * CSH) ? NH * KSD : NH * CSH;
constexpr int CBH = (K * NH / VEC + 63) / 64, CCH = (M * NH / VEC + 63) / 64;
constexpr size_t LDS2_BYTES = (size_t)(K * AMS + BH_ELEMS + 64) * sizeof(T);
static_assert(0 == N % 2, "shape");
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XWPE))) void kern2(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ c, long long batch)
{
  extern __shared__ __align__(16) unsigned char smem[];
  T* const As = reinterpret_cast<T*>(smem);
  T* const Bs = As + K * AMS;
  T* const Cs = Bs;                      // the half image of C shares the place of B's half (never alive together)
  T* const dummy = Bs + BH_ELEMS;
  const int lane = threadIdx.x, l16 = lane & 15, lq = lane >> 4;
  V ra[CA], rb[CBH], rc[CCH];
  auto load_a = [&](long long item) {
    const V* const pa = reinterpret_cast<const V*>(a + item * (M * K));
#pragma unroll
    for (int j = 0; j < CA; ++j) ra[j] = __builtin_nontemporal_load(pa + clampi(64 * j + lane, M * K / VEC - 1));
  };
  auto load_bc = [&](long long item, int h) {
    const V* const pb = reinterpret_cast<const V*>(b + item * (K * N) + h * (K * NH));
#pragma unroll
    for (int j = 0; j < CBH; ++j) rb[j] = __builtin_nontemporal_load(pb + clampi(64 * j + lane, K * NH / VEC - 1));
    if (!XBETA0) {
      const V* const pc = reinterpret_cast<const V*>(c + item * (M * N) + h * (M * NH));
#pragma unroll
      for (int j = 0; j < CCH; ++j) rc[j] = __builtin_nontemporal_load(pc + clampi(64 * j + lane, M * NH / VEC - 1));
    }
  };
  auto lds_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  auto store_c = [&](long long item, int h) {
    V* const pc = reinterpret_cast<V*>(c + item * (M * N) + h * (M * NH));
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
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (0 <= prev) { store_c(prev, prevh); lds_sync(); }
      ACC acc[NIH][MI];
      if (!XBETA0) {
#pragma unroll
        for (int j = 0; j < CCH; ++j) {
          const int ch = clampi(64 * j + lane, M * NH / VEC - 1), e = ch * VEC, n = e / M, m = e % M;
          *reinterpret_cast<V*>(Cs + n * CSH + m) = rc[j];
        }
        lds_sync();
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = clampi(16 * ni + NROW(r), NH - 1), m = clampi(16 * mi + l16, M - 1);
              acc[ni][mi][r] = Cs[n * CSH + m];
            }
        lds_sync();
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
      lds_sync();
#if (1 == XKUNROLL)
#pragma unroll 1
#elif (2 == XKUNROLL)
#pragma unroll 2
#else
#pragma unroll
#endif
      for (int ks = 0; ks < KS; ++ks) {
        T af[MI], bf[NIH];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { const int m = clampi(16 * mi + l16, M - 1); af[mi] = As[(4 * ks + lq) * AMS + m]; }
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni) { const int n = clampi(16 * ni + l16, NH - 1); bf[ni] = Bs[n * KSD + 4 * ks + lq]; }
#pragma unroll
        for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MFMA(bf[ni], af[mi], acc[ni][mi]);
      }
      lds_sync();
#pragma unroll
      for (int ni = 0; ni < NIH; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 16 * ni + NROW(r), m = 16 * mi + l16;
            const bool inside = (16 * ni + 15 < NH || n < NH) && (16 * mi + 15 < M || m < M);
            T* const dst = inside ? Cs + n * CSH + m : dummy + lane;
            *dst = acc[ni][mi][r];
          }
      lds_sync();
      prev = item; prevh = h;
    }
    if (next >= batch) break;
    item = next;
  }
  store_c(prev, prevh);
}
#define KERN kern2
#define KLDS LDS2_BYTES
#else
#define KERN kern
#define KLDS LDS_BYTES
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (hipSuccess != e_) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
  const long long batch = (argc > 1) ? atoll(argv[1]) : 200000;
  int bpc = (argc > 2) ? atoi(argv[2]) : 0;
  const size_t na = (size_t)batch * M * K, nb = (size_t)batch * K * N, nc = (size_t)batch * M * N;
  std::vector<T> ha(na), hb(nb), hc(nc), out(nc);
  srand(1);
  for (auto& x : ha) x = (T)(rand() / (double)RAND_MAX - 0.5);
  for (auto& x : hb) x = (T)(rand() / (double)RAND_MAX - 0.5);
  for (auto& x : hc) x = (T)(rand() / (double)RAND_MAX - 0.5);
  T *da, *db, *dc;
  CHECK(hipMalloc(&da, na * sizeof(T))); CHECK(hipMalloc(&db, nb * sizeof(T))); CHECK(hipMalloc(&dc, nc * sizeof(T)));
  CHECK(hipMemcpy(da, ha.data(), na * sizeof(T), hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb.data(), nb * sizeof(T), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dc, hc.data(), nc * sizeof(T), hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KLDS));
  int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, KERN, 64, KLDS));
  if (0 >= bpc) bpc = occ;
  const unsigned blocks = (unsigned)(256 * bpc);
  printf("M=%d N=%d K=%d %s lds=%zu B/wave occupancy=%d waves/CU, using %d\n", M, N, K, XF64 ? "f64" : "f32", (size_t)KLDS, occ, bpc);
  hipLaunchKernelGGL(KERN, dim3(blocks), dim3(64), KLDS, 0, da, db, dc, batch);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(out.data(), dc, nc * sizeof(T), hipMemcpyDeviceToHost));
  long long bad = 0; // fma chain in ascending k, checked on a sample of items
  for (long long it = 0; it < batch; it += (batch / 64 > 0 ? batch / 64 : 1)) {
    for (int n = 0; n < N; ++n) for (int m = 0; m < M; ++m) {
      T acc = XBETA0 ? T(0) : hc[it * M * N + n * M + m];
      for (int k = 0; k < K; ++k) acc = (T)fma(ha[it * M * K + k * M + m], hb[it * K * N + n * K + k], acc);
#if !XF64
      acc = XBETA0 ? 0.f : hc[it * M * N + n * M + m];
      for (int k = 0; k < K; ++k) acc = fmaf(ha[it * M * K + k * M + m], hb[it * K * N + n * K + k], acc);
#endif
      if (acc != out[it * M * N + n * M + m]) ++bad;
    }
  }
  printf("mismatches on the sample: %lld\n", bad);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(KERN, dim3(blocks), dim3(64), KLDS, 0, da, db, dc, batch);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)batch * sizeof(T) * (M * K + K * N + (XBETA0 ? 1 : 2) * M * N);
  printf("%.3f ms  %.0f GB/s  %.1f %% of 8 TB/s\n", ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / 80.0);
  return bad ? 2 : 0;
}
