// sparse.hip -- sparse x dense kernels for gfx950: CSR-A-sparse operator times column panels (fsspmdm,
// libxsmm_create_?csr_reg), spmdm createSparseSlice (dense -> CSR with uint16 indexes) and spmdm compute.
//
// Reference arithmetic:
//  - csr_reg kernel (src/generator_spgemm_csr_asparse_reg.c:227-300): row-major, per row with nnz > 0:
//    acc = beta ? C[m*ldc+n] : 0; for p in row (ascending): acc = fma(val[p], B[col[p]*ldb+n], acc); store.
//    Rows without nnz are skipped (:229,287). The dense fallback of fsspmdm (src/libxsmm_fsspmdm.c:134-142)
//    zeroes such rows when beta == 0 -- `skip_empty_rows` selects between the two.
//  - createSparseSlice (src/template/libxsmm_spmdm_createSparseSlice_fp32_thread.tpl.c:47-141): row scan,
//    ascending column, keep v != 0, uint16 local column index, rowidx[r] = running count.
//  - compute (src/template/libxsmm_spmdm_compute_fp32_thread.tpl.c:81-558): acc = beta*C (beta == 0: C is not
//    read), then fma(val, B[col][n], acc) over the k-blocks in order and the row's nnz in order.
#include "smm_common.cuh"
#include <type_traits>

namespace xsmm {
namespace {

// ---- CSR operator x panels ------------------------------------------------------------------------------------
// The panels sit side by side in one row-major B (K x ldb) / C (M x ldc), so the whole batch is simply
// C[:, 0:ncols] (+)= A_csr * B[:, 0:ncols]: every thread owns VEC adjacent columns, the CSR arrays are
// wave-uniform (scalar loads), B rows are fetched as coalesced row segments (re-reads served by L1/L2;
// the LDS-staged variant lives in csr_panels_lds_kernel).
template<typename T, int VEC>
__global__ __launch_bounds__(256)
void csr_panels_kernel(int M, long long ncols, int ldb, int ldc, int beta0, int skip_empty,
                       const unsigned* __restrict__ rowptr, const unsigned* __restrict__ colidx, const T* __restrict__ values,
                       const T* __restrict__ B, T* __restrict__ C)
{
  const long long ngroups = (ncols + VEC - 1) / VEC;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
    const long long n0 = g * VEC;
    for (int m = 0; m < M; ++m) {
      const unsigned p0 = rowptr[m], p1 = rowptr[m + 1];
      if (p0 == p1 && (0 != skip_empty || 0 == beta0)) continue; // nothing to add; beta==1 keeps C, quirk keeps C
      T acc[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = (0 == beta0 && n0 + v < ncols) ? C[(size_t)m * ldc + n0 + v] : T(0);
      for (unsigned p = p0; p < p1; ++p) {
        const T a = values[p];
        const T* const brow = B + (size_t)colidx[p] * ldb + n0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) if (n0 + v < ncols) acc[v] = xfma(a, brow[v], acc[v]);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) if (n0 + v < ncols) C[(size_t)m * ldc + n0 + v] = acc[v];
    }
  }
}

// ---- spmdm: dense -> CSR ----------------------------------------------------------------------------------------
// One wavefront per slice (batch item, or (kb,mb) block of a large matrix). Rows are scanned in order; within a
// row the 64 lanes test 64 consecutive columns, __ballot gives the keep-mask and the popcount of the lower lanes
// the write position -- the output order is exactly the sequential scan of the reference.
__global__ __launch_bounds__(256)
void spmdm_create_kernel(long long nslices, int nrows_full, int ncols_full, int transa,
                         const float* __restrict__ a, long long a_slice_stride, int ld,
                         // block decomposition of one matrix (mb_count > 0) or batch of whole matrices (mb_count == 0)
                         int mb_count, int bm, int bk, int M, int K,
                         uint16_t* __restrict__ rowidx, uint16_t* __restrict__ colidx, float* __restrict__ values,
                         long long rowidx_stride, long long cap)
{
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (uniform by construction; blockDim.x is a multiple of 64)
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long s = wave; s < nslices; s += nwaves) {
    const float* in; int nrows, ncols;
    if (0 < mb_count) { // slice s = kb*mb_count + mb of one M x K matrix (reference: kb = id / mb, mb = id % mb)
      const int kb = (int)(s / mb_count), mb = (int)(s % mb_count);
      nrows = ((mb + 1) * bm > M) ? (M - mb * bm) : bm;
      ncols = ((kb + 1) * bk > K) ? (K - kb * bk) : bk;
      in = transa ? (a + (size_t)mb * bm + (size_t)kb * bk * M) : (a + (size_t)kb * bk + (size_t)mb * bm * K);
    }
    else { nrows = nrows_full; ncols = ncols_full; in = a + s * a_slice_stride; }
    uint16_t* const ri = rowidx + s * rowidx_stride;
    uint16_t* const ci = colidx + s * cap;
    float* const va = values + s * cap;
    unsigned cnt = 0; // wave-uniform running count
    int r_done = 0;
    if (ncols <= 64) { // one 64-column chunk per row: fetch eight rows at a time, then compact them in order
      const bool in_range = (lane < ncols);
      for (; r_done + 8 <= nrows; r_done += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r_done + u;
          v[u] = in_range ? __builtin_nontemporal_load(transa ? (in + (size_t)lane * ld + r) : (in + (size_t)r * ld + lane)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (0 == lane) ri[r_done + u] = (uint16_t)cnt;
          const bool keep = in_range && !(0.f == v[u]);
          const unsigned long long mask = __ballot(keep);
          if (keep) {
            const unsigned pos = cnt + __popcll(mask & ((1ULL << lane) - 1ULL));
            ci[pos] = (uint16_t)lane; va[pos] = v[u];
          }
          cnt += __popcll(mask);
        }
      }
    }
    for (int r = r_done; r < nrows; ++r) {
      if (0 == lane) ri[r] = (uint16_t)cnt;
      for (int c0 = 0; c0 < ncols; c0 += 64) {
        const int c = c0 + lane;
        float v = 0.f;
        if (c < ncols) v = transa ? in[(size_t)c * ld + r] : in[(size_t)r * ld + c];
        const bool keep = (c < ncols) && !(0.f == v); // LIBXSMM_FEQ(0, v) ? 0 : 1  (-0 is zero, NaN is kept)
        const unsigned long long mask = __ballot(keep);
        if (keep) {
          const unsigned pos = cnt + __popcll(mask & ((1ULL << lane) - 1ULL));
          ci[pos] = (uint16_t)c; va[pos] = v;
        }
        cnt += __popcll(mask);
      }
    }
    if (0 == lane) ri[nrows] = (uint16_t)cnt;
  }
}

// Block form of one large matrix (the reference API, slices of bm <= 512 rows x bk <= 64 columns): a work-group of sixteen
// waves per slice. Wave w owns a contiguous share of at most 32 of the slice's rows and fetches them all at once (a row is
// one load per wave, a lane per column): one memory round trip per slice. The entry counts of the shares (__ballot +
// popcount per row) are exchanged through LDS; then every wave writes rowidx / colidx / values from its registers at the
// positions the sequential scan of the reference gives them.
constexpr int SPB_WAVES = 16, SPB_ROWS = 32;
__global__ __launch_bounds__(64 * SPB_WAVES)
void spmdm_create_block_kernel(int first_slice, int slice_step, int transa, const float* __restrict__ a, int mb_count, int bm, int bk, int M, int K,
                               uint16_t* __restrict__ rowidx, uint16_t* __restrict__ colidx, float* __restrict__ values,
                               long long rowidx_stride, long long cap)
{
  __shared__ unsigned wave_count[SPB_WAVES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = first_slice + (int)blockIdx.x * slice_step;
  const int kb = s / mb_count, mb = s % mb_count; // reference: kb = id / mb, mb = id % mb
  const int nrows = ((mb + 1) * bm > M) ? (M - mb * bm) : bm;  // <= SPB_WAVES * SPB_ROWS (host check)
  const int ncols = ((kb + 1) * bk > K) ? (K - kb * bk) : bk;  // <= 64 (host check)
  const int ld = transa ? M : K;
  const float* const in = transa ? (a + (size_t)mb * bm + (size_t)kb * bk * M) : (a + (size_t)kb * bk + (size_t)mb * bm * K);
  uint16_t* const ri = rowidx + s * rowidx_stride;
  uint16_t* const ci = colidx + s * cap;
  float* const va = values + s * cap;
  const int share = (nrows + SPB_WAVES - 1) / SPB_WAVES;
  const int r0 = wave * share, r1 = (r0 + share < nrows) ? (r0 + share) : nrows;
  const bool in_range = (lane < ncols);
  float v[SPB_ROWS];
#pragma unroll
  for (int u = 0; u < SPB_ROWS; ++u) {
    const int r = r0 + u;
    v[u] = (in_range && r < r1) ? (transa ? in[(size_t)lane * ld + r] : in[(size_t)r * ld + lane]) : 0.f;
  }
  unsigned cnt = 0;
#pragma unroll
  for (int u = 0; u < SPB_ROWS; ++u) cnt += __popcll(__ballot(!(0.f == v[u]))); // LIBXSMM_FEQ(0, v) ? 0 : 1  (-0 is zero, NaN is kept)
  if (0 == lane) wave_count[wave] = cnt;
  __syncthreads();
  cnt = 0;
  for (int w = 0; w < wave; ++w) cnt += wave_count[w];
#pragma unroll
  for (int u = 0; u < SPB_ROWS; ++u) {
    const int r = r0 + u;
    if (r < r1) {
      if (0 == lane) ri[r] = (uint16_t)cnt;
      const bool keep = !(0.f == v[u]);
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        const unsigned pos = cnt + __popcll(mask & ((1ULL << lane) - 1ULL));
        ci[pos] = (uint16_t)lane; va[pos] = v[u];
      }
      cnt += __popcll(mask);
    }
  }
  if (r0 < nrows && r1 == nrows && 0 == lane) ri[nrows] = (uint16_t)cnt; // the wave that owns the last row closes the slice
}

// Batch form for slices of at most 64 columns whose slots are 16-byte aligned (cap % 8 == 0): the compacted entries are
// collected in wave-private LDS and leave as 16-byte vector stores, 1 KiB per wave instruction, instead of the per-row
// 2- and 4-byte scatter of the kernel above (partial lines). Same scan order, same output.
constexpr int SPC_CAP = 1024;  // entries buffered per wave before a flush (+ one row of slack)
__global__ __launch_bounds__(256)
void spmdm_create_staged_kernel(long long nslices, int nrows, int ncols, int transa,
                                const float* __restrict__ a, long long a_slice_stride, int ld,
                                uint16_t* __restrict__ rowidx, uint16_t* __restrict__ colidx, float* __restrict__ values,
                                long long rowidx_stride, long long cap)
{
  __shared__ __align__(16) float lds_vals[4][SPC_CAP + 64];
  __shared__ __align__(16) uint16_t lds_cols[4][SPC_CAP + 64];
  __shared__ __align__(16) uint16_t lds_rows[4][264];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const bv = lds_vals[wv]; uint16_t* const bc = lds_cols[wv]; uint16_t* const br = lds_rows[wv];
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (uniform by construction; blockDim.x is a multiple of 64)
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const bool in_range = (lane < ncols);
  typedef unsigned sp_u32x4c __attribute__((ext_vector_type(4)));
  typedef float sp_f32x4c __attribute__((ext_vector_type(4)));
  for (long long s = wave; s < nslices; s += nwaves) {
    const float* const in = a + s * a_slice_stride;
    uint16_t* const ri = rowidx + s * rowidx_stride;
    uint16_t* const ci = colidx + s * cap;
    float* const va = values + s * cap;
    unsigned cnt = 0;      // entries of the slice so far
    unsigned flushed = 0;  // entries already in HBM (multiple of 8)
    auto flush = [&](unsigned upto) { // entries [flushed, upto) leave the buffer; upto - flushed is a multiple of 8 (or the final tail, padded)
      const unsigned n = upto - flushed;
      wave_lds_sync();
      for (unsigned i = lane; 4 * i < n; i += 64) __builtin_nontemporal_store(*reinterpret_cast<const sp_f32x4c*>(bv + 4 * i), reinterpret_cast<sp_f32x4c*>(va + flushed + 4 * i));
      for (unsigned i = lane; 8 * i < n; i += 64) __builtin_nontemporal_store(*reinterpret_cast<const sp_u32x4c*>(bc + 8 * i), reinterpret_cast<sp_u32x4c*>(ci + flushed + 8 * i));
      wave_lds_sync();
    };
    for (int r0 = 0; r0 < nrows; r0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        v[u] = (in_range && r < nrows) ? __builtin_nontemporal_load(transa ? (in + (size_t)lane * ld + r) : (in + (size_t)r * ld + lane)) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        if (r < nrows) {
          if (0 == lane) br[r] = (uint16_t)cnt;
          const bool keep = in_range && !(0.f == v[u]); // LIBXSMM_FEQ(0, v) ? 0 : 1  (-0 is zero, NaN is kept)
          const unsigned long long mask = __ballot(keep);
          if (keep) {
            const unsigned pos = (cnt - flushed) + __popcll(mask & ((1ULL << lane) - 1ULL));
            bc[pos] = (uint16_t)lane; bv[pos] = v[u];
          }
          cnt += __popcll(mask);
        }
      }
      if (cnt - flushed >= (unsigned)(SPC_CAP - 8 * 64)) { // the next eight rows might not fit: write whole groups of 8 entries out
        const unsigned upto = flushed + ((cnt - flushed) & ~7u);
        const unsigned rest = cnt - upto;
        flush(upto);
        if (lane < rest) { const float tv = bv[(upto - flushed) + lane]; const uint16_t tc = bc[(upto - flushed) + lane]; wave_lds_sync(); bv[lane] = tv; bc[lane] = tc; }
        else wave_lds_sync();
        flushed = upto;
      }
    }
    if (0 == lane) br[nrows] = (uint16_t)cnt;
    flush(cnt + ((8 - ((cnt - flushed) & 7)) & 7)); // tail padded to a whole 16-byte piece (stays inside the slot: cap % 8 == 0)
    for (int i = lane; i <= nrows; i += 64) ri[i] = br[i];
    wave_lds_sync();
  }
}

// ---- spmdm: CSR x dense -----------------------------------------------------------------------------------------
// Generic form (any geometry, transposes, beta): one thread per C element of one item; B is read through the
// caches. The tuned batch kernel for small problems is spmdm_compute_lds_kernel.
__global__ __launch_bounds__(256)
void spmdm_compute_kernel(long long batch, int M, int N, int K, int bm, int bk, int mb_count, int kb_count,
                          int transb, int transc, float beta,
                          const uint16_t* __restrict__ rowidx, const uint16_t* __restrict__ colidx, const float* __restrict__ values,
                          long long rowidx_stride, long long cap, long long slices_per_item,
                          const float* __restrict__ b, float* __restrict__ c, long long b_stride, long long c_stride,
                          int m_begin, int m_end, int n_begin, int n_end)
{
  const int tm = m_end - m_begin, tn = n_end - n_begin;
  const long long per_item = (long long)tm * tn, total = per_item * batch;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long item = e / per_item;
    const int rem = (int)(e % per_item);
    const int m = m_begin + rem / tn, n = n_begin + rem % tn;
    const float* const bi = b + item * b_stride;
    float* const pc = c + item * c_stride + (transc ? ((size_t)n * M + m) : ((size_t)m * N + n));
    float acc = (0.f == beta) ? 0.f : ((1.f == beta) ? *pc : beta * (*pc));
    const int mb = m / bm, ml = m % bm;
    for (int kb = 0; kb < kb_count; ++kb) {
      const long long s = item * slices_per_item + (long long)kb * mb_count + mb;
      const uint16_t* const ri = rowidx + s * rowidx_stride;
      const uint16_t* const ci = colidx + s * cap;
      const float* const va = values + s * cap;
      const unsigned p0 = ri[ml], p1 = ri[ml + 1];
      for (unsigned p = p0; p < p1; ++p) {
        const int kk = kb * bk + ci[p];
        const float bv = transb ? bi[(size_t)n * K + kk] : bi[(size_t)kk * N + n];
        acc = xfma(va[p], bv, acc);
      }
    }
    *pc = acc;
  }
}

// ---- bfloat16 -> float (the reference's EXPAND_BFLOAT16, src/libxsmm_spmdm_begin.h:69-75): bits << 16 ----------------------
__global__ __launch_bounds__(256) void bf16_widen_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long long count)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    dst[i] = __uint_as_float((unsigned)src[i] << 16);
  }
}

// ---- blocked_gemm layout conversions (reference template/libxsmm_blocked_gemm_copy*.tpl.c) -----------------------------
template<typename T>
__global__ __launch_bounds__(256)
void bgemm_copy_kernel(int which, const T* __restrict__ src, int ld, T* __restrict__ dst,
                       int mb, int nb, int kb, int bm, int bn, int bk)
{
  long long total;
  if (0 == which) total = (long long)mb * kb * bk * bm; else if (1 == which || 5 == which) total = (long long)nb * kb * bn * bk; else total = (long long)nb * mb * bn * bm;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    if (0 == which) { // A: dst[mb][kb][bk][bm] = src[(kb*bk+k)*ld + mb*bm+m]
      const int m = (int)(e % bm); long long r = e / bm; const int k = (int)(r % bk); r /= bk; const int ikb = (int)(r % kb); const int imb = (int)(r / kb);
      dst[e] = src[((size_t)ikb * bk + k) * ld + (size_t)imb * bm + m];
    }
    else if (1 == which) { // B: dst[nb][kb][bn][bk] = src[(nb*bn+n)*ld + kb*bk+k]
      const int k = (int)(e % bk); long long r = e / bk; const int n = (int)(r % bn); r /= bn; const int ikb = (int)(r % kb); const int inb = (int)(r / kb);
      dst[e] = src[((size_t)inb * bn + n) * ld + (size_t)ikb * bk + k];
    }
    else if (4 == which) { // convert_b_to_a (template/libxsmm_blocked_gemm_convert_b_to_a.tpl.c:32-46): dst[mb][nb][bn][bm] = src[nb][mb][bn][bm]
      const int m = (int)(e % bm); long long r = e / bm; const int n = (int)(r % bn); r /= bn; const int inb = (int)(r % nb); const int imb = (int)(r / nb);
      dst[e] = src[(((size_t)inb * mb + imb) * bn + n) * bm + m];
    }
    else if (5 == which) { // transpose_b (template/libxsmm_blocked_gemm_transpose_b.tpl.c:32-65): e walks src[kb][nb][bk][bn]
      const int n = (int)(e % bn); long long r = e / bn; const int k = (int)(r % bk); r /= bk; const int inb = (int)(r % nb); const int ikb = (int)(r / nb);
      const int N = nb * bn, K = kb * bk;
      size_t d;
      if (N == K && bn == bk) d = (((size_t)inb * kb + ikb) * bn + n) * bk + k;
      else { // the reference's generic branch: linear position -> (row, col) of a K-wide matrix -> transposed linear position
        const long long job = ((long long)ikb * bk + k) * N + ((long long)inb * bn + n);
        const long long ii = job / K, jj = job % K, jobt = jj * N + ii;
        d = ((((size_t)(jobt / K) / bn) * kb + (size_t)(jobt % K) / bk) * bn + (size_t)(jobt / K) % bn) * bk + (size_t)(jobt % K) % bk;
      }
      dst[d] = src[e];
    }
    else { // C: blocked[nb][mb][bn][bm] <-> plain[(nb*bn+n)*ld + mb*bm+m]
      const int m = (int)(e % bm); long long r = e / bm; const int n = (int)(r % bn); r /= bn; const int imb = (int)(r % mb); const int inb = (int)(r / mb);
      const size_t plain = ((size_t)inb * bn + n) * ld + (size_t)imb * bm + m;
      if (2 == which) dst[e] = src[plain]; else dst[plain] = src[e];
    }
  }
}

typedef float sp_f32x4 __attribute__((ext_vector_type(4)));

// Work-group-per-item form (the one used for BASELINE config 4). 256 threads = 16 row groups of 16 lanes; a lane owns
// four adjacent columns (n = 4*(t&15)..+3), a group one row per round, so 16 rows are in flight per round.
//  LDS: the item's B tile (K x N floats, streamed with 16-byte loads), its rowidx, and a window of CSR entries packed as
//  {float index of the B row, value} (8 bytes) so that one ds_read_b64 feeds one ds_read_b128 + 4 fma.
//  Per C element the chain is acc = beta*C; acc = fma(val_p, B[col_p][n], acc) in row order (compute tpl :321-371).
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));

// CSR entries come out of LDS with hand-placed ds_read_b64: left to the compiler, neighbouring entries are paired into
// ds_read2_b64, which the LDS array serves at half the rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table) -- the
// entry reads were then as expensive as the B-row reads they feed. The loads are issued without a wait; spw_landed()
// is the matching s_waitcnt (LDS operations return in order, so the compiler's own lgkmcnt bookkeeping for the
// ds_read_b128 it issues in between stays conservative).
__device__ __forceinline__ unsigned spw_lds_addr(const void* p)
{
  return (unsigned)(size_t)p; // low half of a flat LDS address is the LDS offset
}
template<int U> struct SpwEntries;
template<> struct SpwEntries<8> {
  sp_f32x2 e[8];
  __device__ __forceinline__ void issue(const float2* p) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                 "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56"
                 : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5]), "=&v"(e[6]), "=&v"(e[7])
                 : "v"(spw_lds_addr(p)) : "memory");
  }
  __device__ __forceinline__ void landed() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : : "memory");
  }
};
template<> struct SpwEntries<4> {
  sp_f32x2 e[4];
  __device__ __forceinline__ void issue(const float2* p) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24"
                 : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]) : "v"(spw_lds_addr(p)) : "memory");
  }
  __device__ __forceinline__ void landed() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]) : : "memory");
  }
};
template<> struct SpwEntries<2> {
  sp_f32x2 e[2];
  __device__ __forceinline__ void issue(const float2* p) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(e[0]), "=&v"(e[1]) : "v"(spw_lds_addr(p)) : "memory");
  }
  __device__ __forceinline__ void landed() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]) : : "memory");
  }
};
template<> struct SpwEntries<1> {
  sp_f32x2 e[1];
  __device__ __forceinline__ void issue(const float2* p) { e[0] = sp_f32x2{ p[0].x, p[0].y }; }
  __device__ __forceinline__ void landed() {}
};

// acc = fma(value_u, B row of entry u, acc) for the U entries in order
template<int U>
__device__ __forceinline__ void spw_apply(const sp_f32x2 (&e)[U], const float* __restrict__ brow0, sp_f32x4& acc)
{
  sp_f32x4 bv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) bv[u] = *reinterpret_cast<const sp_f32x4*>(brow0 + __float_as_int(e[u][0]));
#pragma unroll
  for (int u = 0; u < U; ++u) {
    acc[0] = xfma(e[u][1], bv[u][0], acc[0]); acc[1] = xfma(e[u][1], bv[u][1], acc[1]);
    acc[2] = xfma(e[u][1], bv[u][2], acc[2]); acc[3] = xfma(e[u][1], bv[u][3], acc[3]);
  }
}

template<int U>
__device__ __forceinline__ void spw_fold(const float2* __restrict__ meta, const float* __restrict__ brow0, sp_f32x4& acc)
{
  SpwEntries<U> s;
  s.issue(meta); s.landed();
  spw_apply<U>(s.e, brow0, acc);
}

// Which of the two batch kernels serves a batch: the gather kernel's time grows with the number of entries, the matrix-core
// kernel's does not (measured on config 4: equal at 30 % density, 0.63 vs 0.86 ms at 5 %, 1.38 vs 1.00 ms at 50 %, 2.5 vs
// 1.5 ms at 100 %). Both are launched, each looks at the entry counts of the same 16 items spread over the batch and
// the one that is not wanted returns at once -- no host round trip, and the results do not depend on the choice (same bits).
__device__ __forceinline__ bool spm_batch_is_dense(const uint16_t* __restrict__ rowidx, int rstride, int M, int K, long long batch)
{
  long long total = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) total += rowidx[((batch * j) >> 4) * rstride + M];
  return 100 * total >= 28LL * 16 * M * K;
}

constexpr int SPW_LDB = 64;    // LDS row stride of the B tile in floats (N <= 64)
constexpr int SPW_META = 2560; // CSR entries per window (>= 16 rows * 64 columns... see host check: 16*K <= SPW_META)

// rows [m0, m1) of one item; their CSR entries sit in meta[] starting at entry pbase
// (handing rows out dynamically through an LDS ticket counter was measured: 5% slower than the static stride of 16)
__device__ __forceinline__ void spw_rows(int m0, int m1, int pbase, int g, int n0, bool active_n, int N, float beta,
                                         const unsigned short* __restrict__ ris, const float2* __restrict__ meta,
                                         const float* __restrict__ Bs, float* __restrict__ pc)
{
  if (!active_n) return; // lanes beyond N stay out of the LDS reads altogether (they would only add bank traffic)
  for (int m = m0 + g; m < m1; m += 16) {
    const int p0 = (int)ris[m] - pbase, p1 = (int)ris[m + 1] - pbase;
    sp_f32x4 acc = sp_f32x4{ 0.f, 0.f, 0.f, 0.f };
    if (0.f != beta && active_n) {
      const sp_f32x4 cv = __builtin_nontemporal_load(reinterpret_cast<const sp_f32x4*>(pc + (size_t)m * N + n0));
      acc = (1.f == beta) ? cv : beta * cv;
    }
    // the chain through acc is sequential, but the LDS reads are not: eight (entry, B row) pairs are fetched per step
    int p = p0;
    // no hand pipelining across steps: measured slower (register copies or spills); the other wavefronts of the CU
    // cover the two LDS round trips of a step
    for (; p + 8 <= p1; p += 8) spw_fold<8>(meta + p, Bs + n0, acc);
    const int rem = p1 - p;
    if (rem & 4) { spw_fold<4>(meta + p, Bs + n0, acc); p += 4; }
    if (rem & 2) { spw_fold<2>(meta + p, Bs + n0, acc); p += 2; }
    if (rem & 1) spw_fold<1>(meta + p, Bs + n0, acc);
    if (active_n) __builtin_nontemporal_store(acc, reinterpret_cast<sp_f32x4*>(pc + (size_t)m * N + n0));
  }
}

// Work-group barrier that orders LDS traffic only. __syncthreads() carries a work-group-scope fence, and on gfx9 that
// drains vmcnt: the next item's B tile and CSR entries, requested just before the barrier, would have to arrive before
// this item's rows may start -- the register-staged pipeline would not overlap anything within a work-group.
__device__ __forceinline__ void spw_lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NB = 16-byte pieces of the B tile per thread (tile <= NB * 1024 floats)
template<int NB>
__global__ __launch_bounds__(256, 4) // four work-groups per CU are what the LDS footprint allows: keep VGPRs <= 128
void spmdm_compute_wg_kernel(long long batch, int M, int N, int K, float beta,
                             const uint16_t* __restrict__ rowidx, const uint16_t* __restrict__ colidx, const float* __restrict__ values,
                             int rstride, long long cap, const float* __restrict__ b, float* __restrict__ c, int paired)
{
  extern __shared__ __align__(16) unsigned char spw_raw[];
  if (0 != paired && spm_batch_is_dense(rowidx, rstride, M, K, batch)) return; // the matrix-core kernel launched alongside takes it
  const int tile = K * N;                                   // floats, multiple of 4
  float* const Bs = reinterpret_cast<float*>(spw_raw);      // [K][N]
  // B rows are padded to SPW_LDB = 64 floats in LDS: a row then starts on a 256-byte bank row, so the 16-byte slot of a
  // lane depends on its column group only and the row groups that share a ds_read_b128 phase never collide
  float2* const meta = reinterpret_cast<float2*>(Bs + K * SPW_LDB); // [SPW_META] {bitcast(int row offset), value}
  unsigned short* const ris = reinterpret_cast<unsigned short*>(meta + SPW_META); // [M+1]
  const int t = threadIdx.x, g = t >> 4, n0 = (t & 15) * 4;
  const bool active_n = (n0 < N);
  const int nv4 = tile >> 2;
  long long item = blockIdx.x;
  if (item >= batch) return;
  const long long G = gridDim.x;
  // Register-staged pipeline: while item i is being multiplied out of LDS, the loads of item i+1 (B tile, rowidx and --
  // when its CSR fits the metadata buffer -- its CSR entries) are in flight; entry counts are fetched two
  // items ahead because the CSR loads of item i+1 need nnz(i+1) when they are issued.
  // CSR entries travel two per thread and pass (entries 2*(t + 256*j), +1): the 16-byte {offset, value} x 2 pieces
  // written to LDS are then contiguous over the lanes (a thread owning eight consecutive entries wrote with a 64-byte
  // lane stride: four-way bank conflicts on every ds_write_b128, 10% of all LDS cycles)
  constexpr int NJ = SPW_META / 512;
  sp_f32x4 rb[NB]; unsigned short rix = 0; unsigned cols[NJ]; sp_f32x2 vals[NJ];
  auto fetch = [&](long long it, int nz) {
    const sp_f32x4* const src = reinterpret_cast<const sp_f32x4*>(b + it * tile);
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int i = 256 * j + t; if (i < nv4) rb[j] = __builtin_nontemporal_load(src + i); }
    if (t <= M) rix = rowidx[it * rstride + t]; // M <= 255 on this path (host check)
    if (nz <= SPW_META) {
      const uint16_t* const ci = colidx + it * cap;
      const float* const va = values + it * cap;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int e = 2 * (t + 256 * j);
        if (e < nz) {
          cols[j] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(ci + e));
          vals[j] = __builtin_nontemporal_load(reinterpret_cast<const sp_f32x2*>(va + e));
        }
      }
    }
  };
  int nnz = rowidx[item * rstride + M]; // every thread reads the same word
  int nnz_next = (item + G < batch) ? (int)rowidx[(item + G) * rstride + M] : 0;
  fetch(item, nnz);
  for (; item < batch; item += G) {
    const bool one_window = (nnz <= SPW_META); // the common case: the whole item's CSR fits the metadata buffer
    float* const pc = c + item * (long long)M * N;
    // ---- park this item's registers in LDS
    {
      const int n4 = N >> 2; // 16-byte pieces per B row
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = 256 * j + t;
        if (i < nv4) { const int kr = i / n4, jc = i - kr * n4; *reinterpret_cast<sp_f32x4*>(Bs + kr * SPW_LDB + 4 * jc) = rb[j]; }
      }
    }
    if (t <= M) ris[t] = rix;
    if (one_window) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int e = 2 * (t + 256 * j);
        if (e < nnz) {
          const int o0 = (int)(cols[j] & 0xFFFFu) * SPW_LDB, o1 = (int)(cols[j] >> 16) * SPW_LDB;
          *reinterpret_cast<sp_f32x4*>(meta + e) = sp_f32x4{ __int_as_float(o0), vals[j][0], __int_as_float(o1), vals[j][1] };
        }
      }
    }
    // ---- next item's loads go out now
    const long long next = item + G;
    const int nnz_next2 = (next + G < batch) ? (int)rowidx[(next + G) * rstride + M] : 0;
    if (next < batch) fetch(next, nnz_next);
    spw_lds_barrier();
    if (one_window) {
      spw_rows(0, M, 0, g, n0, active_n, N, beta, ris, meta, Bs, pc);
      spw_lds_barrier();
    }
    else { // dense items: walk the rows in windows of as many 16-row rounds as fit the metadata buffer
      const uint16_t* const ci = colidx + item * cap;
      const float* const va = values + item * cap;
      int m0 = 0;
      while (m0 < M) {
        int m1 = m0 + 16 < M ? m0 + 16 : M;
        const int pbase = ris[m0];
        while (m1 < M) {
          const int m2 = m1 + 16 < M ? m1 + 16 : M;
          if ((int)ris[m2] - pbase > SPW_META) break;
          m1 = m2;
        }
        const int cnt = (int)ris[m1] - pbase;
        for (int e = t; e < cnt; e += 256) {
          const int off = (int)ci[pbase + e] * SPW_LDB;
          meta[e] = float2{ __int_as_float(off), va[pbase + e] };
        }
        spw_lds_barrier();
        spw_rows(m0, m1, pbase, g, n0, active_n, N, beta, ris, meta, Bs, pc);
        spw_lds_barrier();
        m0 = m1;
      }
    }
    nnz = nnz_next; nnz_next = nnz_next2;
  }
}

// ---- spmdm compute on the matrix cores --------------------------------------------------------------------------------
// The LDS kernel above spends its time gathering B rows per non-zero (LDS pipe ~70 % busy at 50 % density). At such
// densities it is cheaper to rebuild the slice as a dense 64 x 64 tile in LDS (zero fill + one scattered write per
// non-zero) and to multiply on the matrix cores: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (one rounding per
// product, no wider accumulation), and fma(0, b, acc) == acc, so each C element still receives exactly the reference's
// chain acc = fma(val_p, B[col_p][n], acc) over its row's entries in ascending column order -- the same bits as the
// sparse kernels. Where B holds inf / NaN against a structural zero of A the reference skips the entry while 0 * inf is NaN:
// the B tile is inspected while it is parked, and an item whose tile holds a non-finite value is multiplied entry by entry
// instead (the gather chain, by the same work-group; tests/test_sparse_gpu.py::test_spmdm_batch_nonfinite_b_under_zeros).
// What remains of the zero-filled multiplication: an accumulator that is exactly -0 may end as +0.
// Roles are swapped (D = B^T-tile x A^T-tile): lane l then holds C[m = l & 15][n = 4 * (l >> 4) .. + 3] of a 16 x 16 tile,
// one 16-byte access per lane. LDS tiles are stored in blocks whose 64 words are exactly one operand fetch of a wave:
//   As[(m >> 4)][k >> 2][m & 15][k & 3]   (B operand: lane (j = m & 15, kq) reads word 4 * j + kq)
//   Bs[(k >> 2)][n >> 4][k & 3][n & 15]   (A operand: lane (i = n & 15, kq) reads word 16 * kq + i)
// so every operand read is a conflict-free ds_read_b32 and there is no padding.
constexpr int SPM_META = 2304; // CSR entries staged through LDS (entries beyond -- slices denser than 56 % -- come from global)

typedef float spm_f32x4 __attribute__((ext_vector_type(4)));

#if defined(SPM_SYNC)
# define SPM_BARRIER() __syncthreads()
#else
# define SPM_BARRIER() spw_lds_barrier()
#endif
// Place of column k inside a row's words of the blocked dense slice: block (k >> 2) * 64, and inside the block the word
// (4 * (row & 15) + (k & 3) + 8 * ((k >> 2) & 3)) mod 64 -- the words of a block are rotated by its index. A block is still exactly one
// operand fetch of a wave (the fetch applies the same rotation), but the scatter of a row's entries -- sixteen lanes, columns in
// ascending order -- now spreads over sixteen banks by (k mod 16) instead of four by (k mod 4): ds_write_b32 is served 32 lanes (two
// rows) at a time over 32 banks, and the two rows' bank sets stay disjoint. Packed: block offset | rotation-and-(k & 3) part (< 64).
__device__ __forceinline__ int spm_word(unsigned col) { return (int)(((col >> 2) << 6) | ((col & 3) + 8 * ((col >> 2) & 3))); }
__device__ __forceinline__ int spm_place(int packed, int r15x4) { return (packed & ~63) + ((r15x4 + (packed & 63)) & 63); }

// FULL: M == K == 64 (the whole geometry is then known at compile time: N = 16 * NT on this path anyway)
template<int NB, int NT, bool FULL>
__global__ __launch_bounds__(256, 3)
void spmdm_compute_mfma_kernel(long long batch, int M_arg, int K_arg, float beta,
                               const uint16_t* __restrict__ rowidx, const uint16_t* __restrict__ colidx, const float* __restrict__ values,
                               int rstride, long long cap, const float* __restrict__ b, float* __restrict__ c, int paired)
{
  extern __shared__ __align__(16) unsigned char spm_raw[];
  if (0 != paired && !spm_batch_is_dense(rowidx, rstride, FULL ? 64 : M_arg, FULL ? 64 : K_arg, batch)) return; // the gather kernel takes it
  float* const As = reinterpret_cast<float*>(spm_raw);                 // 64 x 64, blocked (see above)
  float* const Bs = As + 64 * 64;                                      // K x (16 * NT), blocked
  float2* const meta = reinterpret_cast<float2*>(Bs + 64 * 16 * NT);   // [SPM_META] {bitcast(word offset of the column inside a row of As), value}
  unsigned short* const ris = reinterpret_cast<unsigned short*>(meta + SPM_META); // [M + 1]
  float* const spare = reinterpret_cast<float*>(ris + 72) + (threadIdx.x & 63);     // a word per lane nobody reads (inactive scatter lanes; one shared word would be a 32-way bank conflict)
  int* const nonfinite = reinterpret_cast<int*>(reinterpret_cast<float*>(ris + 72) + 64); // [2]: "this item's B tile holds inf / NaN", one word per register set
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int M = FULL ? 64 : M_arg, K = FULL ? 64 : K_arg;
  constexpr int N = 16 * NT, n4 = 4 * NT;
  const int tile = K * N, nv4 = tile >> 2, ksteps = K >> 2;
  long long item = blockIdx.x;
  if (item >= batch) return;
  const long long G = gridDim.x;
  constexpr int NJ = SPM_META / 512 + 1;
  // Register-staged pipeline, two items deep: while item i is worked on out of LDS the operands of items i + G and i + 2G are
  // in flight (one item ahead leaves the loads less than one pass -- about a microsecond -- to arrive, the memory system
  // under this load answers in two to three). Two register sets, the pass body is instantiated once per set.
  sp_f32x4 rb[2][NB]; unsigned short rix[2] = { 0, 0 }; unsigned cols[2][NJ]; sp_f32x2 vals[2][NJ];
  auto fetch = [&](auto SET, long long it, int nz) {
    constexpr int S = decltype(SET)::value;
    const sp_f32x4* const src = reinterpret_cast<const sp_f32x4*>(b + it * tile);
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int i = 256 * j + t; if (i < nv4) rb[S][j] = __builtin_nontemporal_load(src + i); }
    if (t <= M) rix[S] = rowidx[it * rstride + t];
    const uint16_t* const ci = colidx + it * cap;
    const float* const va = values + it * cap;
    const int staged = nz < SPM_META ? nz : SPM_META;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int e = 2 * (t + 256 * j);
      if (e < staged) {
        cols[S][j] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(ci + e));
        vals[S][j] = __builtin_nontemporal_load(reinterpret_cast<const sp_f32x2*>(va + e));
      }
    }
  };
  // entry counts travel one item further ahead than the operands: the CSR loads of an item need its count when they are issued
  int nnz = rowidx[item * rstride + M];
  int nnz1 = (item + G < batch) ? (int)rowidx[(item + G) * rstride + M] : 0;
  int nnz2 = (item + 2 * G < batch) ? (int)rowidx[(item + 2 * G) * rstride + M] : 0;
  fetch(std::integral_constant<int, 0>(), item, nnz);
  if (item + G < batch) fetch(std::integral_constant<int, 1>(), item + G, nnz1);
  if (t < 2) nonfinite[t] = 0;
  SPM_BARRIER();
  auto pass = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    float* const pc = c + item * (long long)M * N;
    // ---- (1) clear the dense slice, park B / row starts / CSR entries
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<sp_f32x4*>(As)[256 * j + t] = sp_f32x4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = 256 * j + t;
      if (i < nv4) {
        const int kr = i / n4, jc = i - kr * n4;
        *reinterpret_cast<sp_f32x4*>(Bs + (((kr >> 2) * NT + (jc >> 2)) << 6) + ((kr & 3) << 4) + ((jc & 3) << 2)) = rb[S][j];
      }
    }
    if (t <= M) ris[t] = rix[S];
    { // inf / NaN in the tile? (exponent all ones: the bits shifted left by one are >= 0xFF000000)
      unsigned top = 0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (256 * j + t < nv4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { const unsigned u = __float_as_uint(rb[S][j][q]) << 1; top = (u > top) ? u : top; }
        }
      }
      if (top >= 0xFF000000u) nonfinite[S] = 1;
    }
    {
      const int staged = nnz < SPM_META ? nnz : SPM_META;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int e = 2 * (t + 256 * j);
        if (e < staged) { // (an odd count stages one entry too many: within the slot's capacity, never read)
          *reinterpret_cast<sp_f32x4*>(meta + e) = sp_f32x4{ __int_as_float(spm_word(cols[S][j] & 0xFFFFu)), vals[S][j][0], __int_as_float(spm_word(cols[S][j] >> 16)), vals[S][j][1] };
        }
      }
    }
    // ---- the register set is free again: the loads of the item two passes ahead go out now
    const long long ahead = item + 2 * G;
    const int nnz3 = (ahead + G < batch) ? (int)rowidx[(ahead + G) * rstride + M] : 0;
    if (ahead < batch) fetch(SET, ahead, nnz2);
    // C (beta != 0) is requested now and needed after the scatter
    const int mi = lane & 15, kq = lane >> 4, mrow = 16 * wave + mi;
    spm_f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[nt] = spm_f32x4{ 0.f, 0.f, 0.f, 0.f };
      if (0.f != beta && mrow < M) acc[nt] = __builtin_nontemporal_load(reinterpret_cast<const spm_f32x4*>(pc + (size_t)mrow * N + 16 * nt + 4 * kq));
    }
    SPM_BARRIER();
    // ---- (2) scatter the entries into the dense slice: 16 lanes per row, 16 rows per pass, up to 64 entries per row.
    // Branch-free: every lane reads (clamped index) and writes (inactive lanes into a spare word), all reads first -- the
    // compiler cannot tell that meta[] and As[] never overlap, a read-write-read chain would cost an LDS round trip per entry.
    {
      const int q = t & 15;
      float2 ent[2][4]; int pb0[4], cnt[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int r = 16 * ps + (t >> 4), rc = r < M ? r : M - 1;
        pb0[ps] = (int)ris[rc] + q;
        cnt[ps] = (r < M) ? (int)ris[rc + 1] : 0;
        if (cnt[ps] > SPM_META) cnt[ps] = SPM_META; // (entries beyond the staged ones: below)
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) { // (two rounds of 32 rows: all 64 at once costs 16 more registers -- spills)
#pragma unroll
        for (int ps = 2 * half; ps < 2 * half + 2; ++ps) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int p = pb0[ps] + 16 * it;
            ent[ps & 1][it] = meta[p < SPM_META ? p : SPM_META - 1];
          }
        }
#pragma unroll
        for (int ps = 2 * half; ps < 2 * half + 2; ++ps) {
          const int r = 16 * ps + (t >> 4);
          float* const row = As + (((r >> 4) * 16) << 6); // + the entry's place among the row's words (spm_place)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            float* const dst = (pb0[ps] + 16 * it < cnt[ps]) ? (row + spm_place(__float_as_int(ent[ps & 1][it].x), (r & 15) << 2)) : spare;
            *dst = ent[ps & 1][it].y;
          }
        }
      }
      if (nnz > SPM_META) { // slices denser than 56 %: the tail comes straight from global memory
        const uint16_t* const ci = colidx + item * cap;
        const float* const va = values + item * cap;
        for (int r = t >> 4; r < M; r += 16) {
          const int p1 = ris[r + 1];
          float* const row = As + (((r >> 4) * 16) << 6);
          int p = (int)ris[r] + q;
          if (p < SPM_META) p += ((SPM_META - p + 15) >> 4) << 4;
          for (; p < p1; p += 16) row[spm_place(spm_word(ci[p]), (r & 15) << 2)] = va[p];
        }
      }
    }
    SPM_BARRIER();
    const bool exact_gather = (0 != nonfinite[S]);
    if (0 == t) nonfinite[S ^ 1] = 0; // (the other set's word: last read before the previous pass's closing barrier, next written after this one's)
    if (exact_gather) { // ---- (3') B holds inf / NaN: entry by entry, zeros of A skipped as the reference does (rare: speed does not matter)
      const uint16_t* const ci = colidx + item * cap;
      const float* const va = values + item * cap;
      for (int e = t; e < M * N; e += 256) {
        const int m = e / N, n = e - m * N;
        float sum = 0.f;
        if (0.f != beta) { sum = pc[e]; if (1.f != beta) sum = beta * sum; }
        const int p1 = ris[m + 1];
        for (int p = ris[m]; p < p1; ++p) {
          const int kr = ci[p];
          sum = __builtin_fmaf(va[p], Bs[(((kr >> 2) * NT + (n >> 4)) << 6) + ((kr & 3) << 4) + (n & 15)], sum);
        }
        pc[e] = sum;
      }
    }
    // ---- (3) wave w multiplies rows [16 w, 16 w + 16) against all NT column tiles
    else if (16 * wave < M) {
      const int i = mi;
      if (0.f != beta && 1.f != beta) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = beta * acc[nt];
      }
      const float* const pa = As + ((wave * 16) << 6);               // + s * 64 + the lane's word of block s (rotated by 8 * (s & 3))
      const int aw = 4 * i + kq;
      const float* const pb = Bs + 16 * kq + i;                      // + (s * NT + nt) * 64
      float bop = pa[aw], aop[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) aop[nt] = pb[nt << 6];
#pragma unroll
      for (int s = 0; s < ksteps; ++s) {
        const float bcur = bop; float acur[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acur[nt] = aop[nt];
        if (s + 1 < ksteps) { // operands of the next step travel during this step's matrix instructions
          bop = pa[((s + 1) << 6) + ((aw + 8 * ((s + 1) & 3)) & 63)];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) aop[nt] = pb[((s + 1) * NT + nt) << 6];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[nt], bcur, acc[nt], 0, 0, 0);
      }
      // C leaves through LDS: the wave's 16 rows of C are one contiguous piece of memory (16 * N floats), written as whole
      // lines; straight from the accumulators every store would touch sixteen separate 64-byte pieces. The wave's own rows of
      // As -- nobody else reads them -- serve as the buffer.
      float* const Cs = As + ((wave * 16) << 6); // 1024 floats >= 16 * CLD
      constexpr int CLD = (N < 64) ? N + 4 : N;  // rows 4 floats apart from a multiple of 16: the 16 lanes of a ds_write_b128 phase hit 16 different 16-byte slots (N = 64 has no room for it)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<spm_f32x4*>(Cs + i * CLD + 16 * nt + 4 * kq) = acc[nt];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float* const pw = pc + (size_t)(16 * wave) * N;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int piece = 64 * nt + lane, prow = piece / n4, pcol = piece - prow * n4; // 16 * N / 4 = 64 * NT pieces of 16 bytes
        __builtin_nontemporal_store(*reinterpret_cast<const spm_f32x4*>(Cs + prow * CLD + 4 * pcol), reinterpret_cast<spm_f32x4*>(pw + 4 * piece));
      }
    }
    SPM_BARRIER(); // the tiles are overwritten next
    nnz = nnz1; nnz1 = nnz2; nnz2 = nnz3;
  };
  for (;;) {
    pass(std::integral_constant<int, 0>()); item += G; if (item >= batch) break;
    pass(std::integral_constant<int, 1>()); item += G; if (item >= batch) break;
  }
}

// ---- spmdm compute: one large problem (the reference API), tiled ----------------------------------------------------------
// A work-group of sixteen waves owns a 64 x 256 tile of C for the whole sum over k: per slice column block (bk = 64 columns
// of A = 64 rows of B) the B panel [64][256] and the CSR entries of the tile's 64 rows are staged in LDS (loads of the next
// block in flight in registers meanwhile), C stays in registers. All 64 lanes of a wave work on ONE row at a time, a lane
// owning four adjacent columns: per entry one broadcast ds_read_b64 ({B row offset, value}, the same address in every lane)
// and one conflict-free ds_read_b128 of the B row, then four fma. Row r = 16 i + w of the tile belongs to wave w, round i.
// Per C element: acc = beta * C; acc = fma(val_p, B[col_p][n], acc) over the column blocks in order and the row's entries
// in order -- the chain of the reference (compute tpl :321-371); beta == 0 never reads C.
// (Measured alternatives, tools/probe/readlane_cost.hip and A/B builds of this kernel on 2048^3 at 15 %: a row's entries held
// one per lane and handed round with v_readlane -- 7 cycles per value with a run-time lane select, 2.9 with an immediate --
// 0.141 ms against 0.123 ms for the broadcast reads; the B panel brought in by global_load_lds into a second buffer
// (one barrier per column block, no ds_write) 0.163 ms; eight waves per tile 0.167 ms.)
constexpr int SPT_BK = 64, SPT_TN = 256, SPT_CAP_MAX = 1536;
constexpr int SPT_RW = 4;                                         // rows per wave: a tile has 4 * WAVES rows
constexpr size_t SPT_LDS = (size_t)SPT_BK * SPT_TN * 4 + (size_t)SPT_CAP_MAX * 8 + 160;

// entries [0, cnt) of one row (row_meta: wave-uniform) folded into the row's accumulators, in order
__device__ __forceinline__ void spt_row(const float2* __restrict__ row_meta, int cnt, const float* __restrict__ brow, sp_f32x4& acc)
{
  int j = 0;
  for (; j + 8 <= cnt; j += 8) spw_fold<8>(row_meta + j, brow, acc);
  // (the tail of a row as one padded group of 4 or 8 -- one LDS round trip instead of up to three -- was measured 10 % slower:
  // the LDS bandwidth of the padding costs more than the round trips)
  if (cnt & 4) { spw_fold<4>(row_meta + j, brow, acc); j += 4; }
  if (cnt & 2) { spw_fold<2>(row_meta + j, brow, acc); j += 2; }
  if (cnt & 1) spw_fold<1>(row_meta + j, brow, acc);
}

// VEC: N % 4 == 0 (K % 4 == 0 for a transposed B) and 16-byte aligned B and C: 16-byte global accesses
// WAVES: 16 (a 64-row tile) or 4 (a 16-row tile, for calls that cover few tiles: a block call of the reference's per-block interface
// is 64 tiles of 64 rows -- a quarter of the CUs -- but 256 tiles of 16 rows; the column panel of B is then staged four times as often,
// out of the L2 of the XCD that the panel's tiles share)
template<bool VEC, bool TRANSB, int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void spmdm_tiled_kernel(int M, int N, int K, int bm, int mb_count, int kb_count, int transc, float beta,
                        const uint16_t* __restrict__ rowidx, const uint16_t* __restrict__ colidx, const float* __restrict__ values,
                        long long rstride, long long cap, const float* __restrict__ b, float* __restrict__ c,
                        int mb_begin, int mb_n, int n_begin, int n_end)
{
  constexpr int SPT_WAVES = WAVES, SPT_THREADS = 64 * WAVES, SPT_TM = SPT_RW * WAVES, SPT_NB = SPT_BK * SPT_TN / 4 / SPT_THREADS;
  constexpr int SPT_CAP = (2 * SPT_THREADS < SPT_CAP_MAX) ? 2 * SPT_THREADS : SPT_CAP_MAX; // one pair of entries per thread covers a window
  extern __shared__ __align__(16) unsigned char spt_raw[];
  float* const Bs = reinterpret_cast<float*>(spt_raw);                               // [64][256]
  float2* const meta = reinterpret_cast<float2*>(Bs + SPT_BK * SPT_TN);              // [SPT_CAP] {bitcast(float offset of the B row), value}
  unsigned short* const ris = reinterpret_cast<unsigned short*>(meta + SPT_CAP);     // [65]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // ---- which tile: consecutive tiles of one column panel go to one XCD (its L2 then holds that panel of B)
  const int tiles_per_mb = (bm + SPT_TM - 1) / SPT_TM;
  const int tiles_m = mb_n * tiles_per_mb, tiles_n = (n_end - n_begin + SPT_TN - 1) / SPT_TN;
  const int total = tiles_m * tiles_n, per_xcd = (total + 7) / 8;
  const int linear = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || linear >= total) return;
  const int tn = linear / tiles_m, tmi = linear - tn * tiles_m;
  const int mbi = mb_begin + tmi / tiles_per_mb, ml0 = (tmi % tiles_per_mb) * SPT_TM;
  const int nrows_mb = ((mbi + 1) * bm > M) ? (M - mbi * bm) : bm;
  if (ml0 >= nrows_mb) return;
  const int rows = (nrows_mb - ml0 < SPT_TM) ? (nrows_mb - ml0) : SPT_TM;
  const int n0 = n_begin + tn * SPT_TN;
  const int ncols = (n_end - n0 < SPT_TN) ? (n_end - n0) : SPT_TN;
  const int m0 = mbi * bm + ml0;           // first row of the tile in C
  const int nl = 4 * lane;                 // this lane's columns: n0 + nl .. + 3
  const int nvalid = ncols - nl;           // > 0: the lane has columns (>= 4: all four)

  // ---- register stages: B panel (SPT_NB x 16 bytes), row starts, CSR entries of the next column block.
  // Piece j of a thread: (wave-uniform base of the column block and of j) + (one per-thread offset): nothing per piece
  // stays in vector registers between the blocks.
  //   B[k][n]: piece = 4 columns pc of row pr + PRS j               TRANSB, B[n][k]: 4 k's 4 (pr + PRS j) .. + 3 of column pc
  //   (lanes along n also when B is transposed: parked without bank conflicts; the global side is served by L1/L2)
  constexpr int PRS = TRANSB ? (SPT_THREADS / 256) : (SPT_THREADS / 64); // rows resp. groups of four k's the threads cover per piece
  const int pr = TRANSB ? (t >> 8) : (t >> 6), pc = TRANSB ? (t & 255) : ((t & 63) << 2);
  const int voff = TRANSB ? (pc * K + 4 * pr) : (pr * N + pc);
  const bool pc_ok = (pc < ncols);
  sp_f32x4 rb[SPT_NB]; unsigned short rix = 0; unsigned cols = 0; sp_f32x2 vals = sp_f32x2{ 0.f, 0.f };
  auto fetch = [&](int kb, int base, int pend) {
    const int k0 = kb * SPT_BK, kc = (K - k0 < SPT_BK) ? (K - k0) : SPT_BK;
#pragma unroll
    for (int j = 0; j < SPT_NB; ++j) {
      sp_f32x4 v = sp_f32x4{ 0.f, 0.f, 0.f, 0.f };
      if (!TRANSB) {
        const float* const sbase = b + (size_t)(k0 + PRS * j) * N + n0; // wave-uniform
        if (pr + PRS * j < kc && pc_ok) {
          if (VEC) v = *reinterpret_cast<const sp_f32x4*>(sbase + voff);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (pc + q < ncols) v[q] = sbase[voff + q];
          }
        }
      }
      else {
        const float* const sbase = b + (size_t)n0 * K + k0 + 4 * PRS * j;
        if (4 * (pr + PRS * j) < kc && pc_ok) {
          if (VEC) v = *reinterpret_cast<const sp_f32x4*>(sbase + voff);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (4 * (pr + PRS * j) + q < kc) v[q] = sbase[voff + q];
          }
        }
      }
      rb[j] = v;
    }
    const long long s = (long long)kb * mb_count + mbi;
    if (t <= rows) rix = rowidx[s * rstride + ml0 + t];
    if (pend - base <= SPT_CAP) { // one window (the common case): the entries travel through registers as well
      const int e = 2 * t;
      if (e < pend - base) { // (an odd count fetches one entry too many: inside the slice's capacity, never used)
        cols = *reinterpret_cast<const unsigned*>(colidx + s * cap + base + e);
        vals = *reinterpret_cast<const sp_f32x2*>(values + s * cap + base + e);
      }
    }
  };
  // entry range [base, pend) of the tile's rows inside slice (kb, mbi); base is rounded down to an even entry (aligned pairs).
  // The ranges of 64 column blocks are fetched at a time, one block per lane, and picked with v_readlane: no memory round
  // trip on the path of a step.
  int vbase = 0, vpend = 0;
  auto entry_range = [&](int kb, int& base, int& pend) {
    if (0 == (kb & 63) && kb < kb_count) {
      const int kq = kb + lane;
      if (kq < kb_count) {
        const uint16_t* const ri = rowidx + ((long long)kq * mb_count + mbi) * rstride + ml0;
        vbase = (int)ri[0] & ~1; vpend = (int)ri[rows];
      }
    }
    if (kb < kb_count) { base = __builtin_amdgcn_readlane(vbase, kb & 63); pend = __builtin_amdgcn_readlane(vpend, kb & 63); }
    else { base = 0; pend = 0; }
  };

  // ---- C tile: acc[i] = row 16 i + wave
  sp_f32x4 acc[SPT_RW];
#pragma unroll
  for (int i = 0; i < SPT_RW; ++i) {
    acc[i] = sp_f32x4{ 0.f, 0.f, 0.f, 0.f };
    const int r = SPT_WAVES * i + wave;
    if (0.f != beta && r < rows && 0 < nvalid) {
      sp_f32x4 cv = sp_f32x4{ 0.f, 0.f, 0.f, 0.f };
      if (0 == transc) {
        const float* const src = c + (size_t)(m0 + r) * N + n0;
        if (VEC) cv = *reinterpret_cast<const sp_f32x4*>(src + nl);
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (q < nvalid) cv[q] = src[nl + q];
        }
      }
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (q < nvalid) cv[q] = c[(size_t)(n0 + nl + q) * M + m0 + r];
      }
      acc[i] = (1.f == beta) ? cv : beta * cv;
    }
  }

  int base, pend, base1, pend1;
  entry_range(0, base, pend);
  entry_range(1, base1, pend1);
  fetch(0, base, pend);
  const float* const brow = Bs + nl;
  for (int kb = 0; kb < kb_count; ++kb) {
    const bool one_window = (pend - base <= SPT_CAP);
    // ---- park this column block's registers in LDS
#pragma unroll
    for (int j = 0; j < SPT_NB; ++j) {
      if (!TRANSB) *reinterpret_cast<sp_f32x4*>(Bs + (pr + PRS * j) * SPT_TN + pc) = rb[j];
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) Bs[(4 * (pr + PRS * j) + q) * SPT_TN + pc] = rb[j][q];
      }
    }
    if (t <= rows) ris[t] = rix;
    if (one_window) {
      const int e = 2 * t;
      if (e < pend - base) {
        const int o0 = (int)(cols & 0xFFFFu) * SPT_TN, o1 = (int)(cols >> 16) * SPT_TN;
        *reinterpret_cast<sp_f32x4*>(meta + e) = sp_f32x4{ __int_as_float(o0), vals[0], __int_as_float(o1), vals[1] };
      }
    }
    // ---- the next column block's loads go out now
    int base2, pend2;
    entry_range(kb + 2, base2, pend2);
    if (kb + 1 < kb_count) fetch(kb + 1, base1, pend1);
    spw_lds_barrier();
    // ---- rounds of sixteen rows; a window holds the entries of as many consecutive rounds as fit the buffer
    int round0 = 0;
    const int nrounds = (rows + SPT_WAVES - 1) / SPT_WAVES;
    while (round0 < nrounds) {
      int round1 = nrounds, wbase = base;
      if (!one_window) {
        const uint16_t* const ci = colidx + ((long long)kb * mb_count + mbi) * cap;
        const float* const va = values + ((long long)kb * mb_count + mbi) * cap;
        wbase = __builtin_amdgcn_readfirstlane((int)ris[SPT_WAVES * round0]);
        round1 = round0 + 1; // (a round holds at most 16 x 64 entries: always fits)
        while (round1 < nrounds) {
          const int rend = (SPT_WAVES * (round1 + 1) < rows) ? SPT_WAVES * (round1 + 1) : rows;
          if ((int)ris[rend] - wbase > SPT_CAP) break;
          ++round1;
        }
        round1 = __builtin_amdgcn_readfirstlane(round1);
        const int wend = (int)ris[(SPT_WAVES * round1 < rows) ? SPT_WAVES * round1 : rows];
        for (int e = t; e < wend - wbase; e += SPT_THREADS) meta[e] = float2{ __int_as_float((int)ci[wbase + e] * SPT_TN), va[wbase + e] };
        spw_lds_barrier();
      }
      // entry ranges of this wave's rows, one row per lane (rows outside the window or the tile: empty)
      int vp0 = 0, vcnt = 0;
      if (lane < SPT_RW) {
        const int r = SPT_WAVES * lane + wave;
        if (lane >= round0 && lane < round1 && r < rows) { vp0 = (int)ris[r] - wbase; vcnt = (int)ris[r + 1] - wbase - vp0; }
      }
#pragma unroll
      for (int i = 0; i < SPT_RW; ++i) spt_row(meta + __builtin_amdgcn_readlane(vp0, i), __builtin_amdgcn_readlane(vcnt, i), brow, acc[i]);
      spw_lds_barrier(); // the buffers are overwritten next
      round0 = round1;
    }
    base = base1; pend = pend1; base1 = base2; pend1 = pend2;
  }

  // ---- C leaves
#pragma unroll
  for (int i = 0; i < SPT_RW; ++i) {
    const int r = SPT_WAVES * i + wave;
    if (r < rows && 0 < nvalid) {
      if (0 == transc) {
        float* const dst = c + (size_t)(m0 + r) * N + n0;
        if (VEC) *reinterpret_cast<sp_f32x4*>(dst + nl) = acc[i];
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (q < nvalid) dst[nl + q] = acc[i][q];
        }
      }
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (q < nvalid) c[(size_t)(n0 + nl + q) * M + m0 + r] = acc[i][q];
      }
    }
  }
}

unsigned grid_for(long long work, int per_block)
{
  long long blocks = (work + per_block - 1) / per_block;
  if (blocks > 256LL * 32) blocks = 256LL * 32;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

} // namespace

int launch_csr_panels(const CsrPanels& p, void* stream, const char** name)
{
  hipStream_t st = (hipStream_t)stream;
  const long long ncols = (long long)p.n * p.batch;
  if (0 == ncols || 0 == p.m) { *name = "csr_panels_noop"; return 0; }
  if (8 == p.typesize) {
    *name = "fsspmdm_f64_csr_cols";
    hipLaunchKernelGGL((csr_panels_kernel<double, 1>), dim3(grid_for(ncols, 256)), dim3(256), 0, st,
      p.m, ncols, p.ldb, p.ldc, p.beta0, p.skip_empty_rows, p.rowptr, p.colidx, (const double*)p.values, (const double*)p.b, (double*)p.c);
  }
  else {
    *name = "fsspmdm_f32_csr_cols";
    hipLaunchKernelGGL((csr_panels_kernel<float, 1>), dim3(grid_for(ncols, 256)), dim3(256), 0, st,
      p.m, ncols, p.ldb, p.ldc, p.beta0, p.skip_empty_rows, p.rowptr, p.colidx, (const float*)p.values, (const float*)p.b, (float*)p.c);
  }
  return (int)hipGetLastError();
}

int launch_spmdm_create(const SpmdmGeom& g, int transa, const float* a, uint16_t* rowidx, uint16_t* colidx, float* values,
                        void* stream, const char** name)
{ // batch form: every item is one slice of M rows x K columns
  hipStream_t st = (hipStream_t)stream;
  *name = "spmdm_create_slices_wave";
  if (0 == g.batch) return 0;
  static const int staged = []() { const char* e = getenv("XSMM_SPMDM_CREATE_STAGED"); return (nullptr != e && 0 != *e) ? atoi(e) : 1; }();
  if (0 != staged && g.k <= 64 && g.m <= 255 && 0 == (g.cap & 7)) { // entries collected in LDS, 16-byte stores
    *name = "spmdm_create_slices_staged";
    hipLaunchKernelGGL(spmdm_create_staged_kernel, dim3(grid_for(g.batch, 4)), dim3(256), 0, st,
      g.batch, g.m, g.k, transa, a, (long long)g.m * g.k, transa ? g.m : g.k, rowidx, colidx, values, (long long)g.rstride, (long long)g.cap);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(spmdm_create_kernel, dim3(grid_for(g.batch, 4)), dim3(256), 0, st,
    g.batch, g.m, g.k, transa, a, (long long)g.m * g.k, transa ? g.m : g.k, 0, g.m, g.k, g.m, g.k,
    rowidx, colidx, values, (long long)g.rstride, (long long)g.cap);
  return (int)hipGetLastError();
}

int launch_spmdm_create_blocks(int M, int K, int bm, int bk, int mb, int first_slice, int slice_step, int nslices, int transa, const float* a,
                               uint16_t* rowidx, uint16_t* colidx, float* values, void* stream, const char** name)
{ // one matrix decomposed into (kb, mb) slices (slice id = kb * mb_count + mb); slices first_slice + i * slice_step, i < nslices:
  // a work-group per slice
  hipStream_t st = (hipStream_t)stream;
  const long long cap = (long long)bm * bk, rstride = (long long)bm + 1;
  if (0 >= nslices) { *name = "spmdm_create_noop"; return 0; }
  if (bk <= 64 && bm <= SPB_WAVES * SPB_ROWS) {
    *name = "spmdm_create_slice_wg";
    hipLaunchKernelGGL(spmdm_create_block_kernel, dim3((unsigned)nslices), dim3(64 * SPB_WAVES), 0, st,
      first_slice, slice_step, transa, a, mb, bm, bk, M, K, rowidx, colidx, values, rstride, cap);
    return (int)hipGetLastError();
  }
  *name = "spmdm_create_slices_wave"; // other slice shapes (not produced by libxsmm_spmdm_init): a wavefront per slice
  for (int i = 0; i < nslices; ++i) {
    const int s = first_slice + i * slice_step;
    const int kb = s / mb, imb = s % mb;
    const int nrows = ((imb + 1) * bm > M) ? (M - imb * bm) : bm;
    const int ncols = ((kb + 1) * bk > K) ? (K - kb * bk) : bk;
    const float* in = transa ? (a + (size_t)imb * bm + (size_t)kb * bk * M) : (a + (size_t)kb * bk + (size_t)imb * bm * K);
    hipLaunchKernelGGL(spmdm_create_kernel, dim3(1), dim3(64), 0, st,
      1LL, nrows, ncols, transa, in, 0LL, transa ? M : K, 0, bm, bk, M, K,
      rowidx + s * rstride, colidx + s * cap, values + s * cap, rstride, cap);
  }
  return (int)hipGetLastError();
}

// C tile rows of the row blocks [mb_begin, mb_begin + mb_n), columns [n_begin, n_end) of one problem whose slices are
// bm x bk (bk == SPT_BK). -1: the geometry is not served by the tiled kernel.
int launch_spmdm_compute_tiled(int M, int N, int K, int bm, int bk, int mb, int kb, int transb, int transc, float beta,
                               const uint16_t* rowidx, const uint16_t* colidx, const float* values, long long rowidx_stride, long long cap,
                               const float* b, float* c, int mb_begin, int mb_n, int n_begin, int n_end, void* stream, const char** name)
{
  if (SPT_BK != bk || 0 != (cap & 1) || bm > 65535 / SPT_BK) return -1;
  if (0 >= mb_n || n_end <= n_begin) { *name = "spmdm_compute_noop"; return 0; }
  static const bool attr = []() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<true, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<false, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<true, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<false, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<true, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmdm_tiled_kernel<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPT_LDS);
    return true;
  }();
  (void)attr;
  const long long col_tiles = (n_end - n_begin + SPT_TN - 1) / SPT_TN;
  const long long total64 = (long long)mb_n * ((bm + 63) / 64) * col_tiles;
  // Tiles of 16 rows, four waves each, for calls that cover few tiles of 64 rows (a per-block call of the reference's interface: 64 tiles)
  // were measured and are NOT the default: 2048^3 at 15 %, four block calls 0.438 ms against 0.454 ms. A work-group walks the 32 column
  // blocks of A, 3.4 us each. Taking the step apart (builds with the rows' work resp. the fetch compiled out): 64-row tiles, 64 work-groups: barriers + parking B 1.0 us, the
  // panel's fetch + 0.4 us, the rows' entries + 2.1 us (a wave's 4 rows x ~10 entries as ~20 dependent LDS round trips; the FMA issue
  // alone is 1.0 us at 16 waves per CU) -- a CU does the same work per step whether 64 or 256 work-groups run, so a block call takes as
  // long as the whole problem (0.113 vs 0.125 ms). 16-row tiles: 0.9 + 0.7 + 1.8 us, and every panel of B is staged four times as often
  // (2 GB through L2 for the whole problem: 0.265 ms). The caller's remedy is the bracket (xsmm_sparse.cpp:record_block): the block
  // calls of a sweep become one launch, 0.125 ms. XSMM_SPMDM_TILE_ROWS=16: developer knob.
  // Also measured and dropped: a wave's four rows walked side by side, two entries per row and step with the next step's entries requested
  // behind this step's B rows (one LDS round trip per step instead of ~20 dependent ones per block): 0.162 ms for the whole problem against
  // 0.125 (50 % density: 0.61 against 0.35) -- with sixteen waves per CU the round trips of one wave are covered by the others; what the
  // CU runs out of is LDS bandwidth and issue slots, and the B rows fetched for slots past the end of a row (~40 % more) cost exactly those.
  static const int rows_env = []() { const char* e = getenv("XSMM_SPMDM_TILE_ROWS"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
  const bool small = (16 == rows_env);
  (void)total64;
  const int tile_rows = small ? 16 : 64;
  const int tiles_per_mb = (bm + tile_rows - 1) / tile_rows;
  const long long total = (long long)mb_n * tiles_per_mb * col_tiles;
  const unsigned grid = (unsigned)(8 * ((total + 7) / 8));
  const uintptr_t bits = reinterpret_cast<uintptr_t>(b) | (0 == transc ? reinterpret_cast<uintptr_t>(c) : 0); // (a transposed C moves element by element anyway)
  const bool vec = (0 == (bits & 15)) && 0 == (N & 3) && 0 == (n_begin & 3) && (0 == transb || 0 == (K & 3));
  *name = small ? "spmdm_compute_tiled16" : "spmdm_compute_tiled";
#define XSMM_SPT(V, TB, W) hipLaunchKernelGGL((spmdm_tiled_kernel<V, TB, W>), dim3(grid), dim3(64 * W), SPT_LDS, (hipStream_t)stream, \
      M, N, K, bm, mb, kb, transc, beta, rowidx, colidx, values, rowidx_stride, cap, b, c, mb_begin, mb_n, n_begin, n_end)
  if (small) {
    if (vec) { if (0 == transb) XSMM_SPT(true, false, 4); else XSMM_SPT(true, true, 4); }
    else { if (0 == transb) XSMM_SPT(false, false, 4); else XSMM_SPT(false, true, 4); }
  }
  else {
    if (vec) { if (0 == transb) XSMM_SPT(true, false, 16); else XSMM_SPT(true, true, 16); }
    else { if (0 == transb) XSMM_SPT(false, false, 16); else XSMM_SPT(false, true, 16); }
  }
#undef XSMM_SPT
  return (int)hipGetLastError();
}

int launch_spmdm_compute_generic(long long batch, int M, int N, int K, int bm, int bk, int mb, int kb, int transb, int transc, float beta,
                                 const uint16_t* rowidx, const uint16_t* colidx, const float* values, long long rowidx_stride, long long cap,
                                 const float* b, float* c, long long b_stride, long long c_stride,
                                 int m_begin, int m_end, int n_begin, int n_end, void* stream, const char** name)
{
  hipStream_t st = (hipStream_t)stream;
  *name = "spmdm_compute_elem";
  const long long total = (long long)(m_end - m_begin) * (n_end - n_begin) * batch;
  if (0 >= total) return 0;
  hipLaunchKernelGGL(spmdm_compute_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st,
    batch, M, N, K, bm, bk, mb, kb, transb, transc, beta, rowidx, colidx, values, rowidx_stride, cap, (long long)mb * kb,
    b, c, b_stride, c_stride, m_begin, m_end, n_begin, n_end);
  return (int)hipGetLastError();
}

int launch_spmdm_compute(const SpmdmGeom& g, int transb, int transc, float beta, const uint16_t* rowidx, const uint16_t* colidx,
                         const float* values, const float* b, float* c, void* stream, const char** name)
{
  const long long tile = (long long)g.k * g.n;
  // XSMM_SPMDM_MFMA: 0 = gather kernel only, 1 = matrix-core kernel only (where the geometry fits), default = both, chosen on the device
  static const int mfma_env = []() { const char* e = getenv("XSMM_SPMDM_MFMA"); return (nullptr != e && 0 != *e) ? atoi(e) : -1; }();
  const bool gather_fits = (0 == transb && 0 == transc && g.n <= 64 && 0 == (g.n & 3) && 16 * g.k <= SPW_META && (long long)g.k * SPW_LDB * 4 <= 49152 && 0 < g.batch
      && g.m <= 255 && 0 == (g.cap & 7));
  const bool mfma_fits = ((0 <= mfma_env ? 0 != mfma_env : 0 != libxsmm_amd_get_mfma()) && 0 == transb && 0 == transc && 0 < g.batch
      && g.m <= 64 && 0 == (g.m & 15) && g.k <= 64 && 0 == (g.k & 3) && g.n <= 64 && 0 == (g.n & 15) && 0 == (g.cap & 7));
  const int paired = (mfma_fits && gather_fits && 0 > mfma_env) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (gather_fits && (0 != paired || !mfma_fits)) { // work-group-per-item LDS kernel
    const size_t lds = (size_t)g.k * SPW_LDB * 4 + (size_t)SPW_META * 8 + (((size_t)g.m + 1) * 2 + 15) / 16 * 16;
    long long per_cu = (long long)(160 * 1024 / lds); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    const long long want = 256 * per_cu;
    const unsigned grid = (unsigned)(g.batch < want ? g.batch : want);
    const int nb = (int)((tile / 4 + 255) / 256);
    *name = "spmdm_compute_wg_lds";
#define XSMM_SPW(NB) hipLaunchKernelGGL((spmdm_compute_wg_kernel<NB>), dim3(grid), dim3(256), lds, st, \
      g.batch, g.m, g.n, g.k, beta, rowidx, colidx, values, g.rstride, (long long)g.cap, b, c, paired)
    if (nb <= 1) XSMM_SPW(1); else if (nb <= 2) XSMM_SPW(2); else if (nb <= 3) XSMM_SPW(3); else if (nb <= 4) XSMM_SPW(4);
    else if (nb <= 6) XSMM_SPW(6); else if (nb <= 8) XSMM_SPW(8); else XSMM_SPW(12);
#undef XSMM_SPW
    if (0 == paired) return (int)hipGetLastError();
  }
  if (mfma_fits) { // dense slice in LDS, matrix cores
    const int nt = g.n / 16;
    const size_t lds = (size_t)64 * 64 * 4 + (size_t)64 * 16 * nt * 4 + (size_t)SPM_META * 8 + 144 + 256 + 16; // + row starts (<= 65 x 2 bytes) + a spare word per lane + the non-finite flags
    long long per_cu = (long long)(160 * 1024 / lds); if (per_cu > 3) per_cu = 3; if (per_cu < 1) per_cu = 1;
    static const int bpc_env = []() { const char* e = getenv("XSMM_SPMDM_BPC"); return (nullptr != e && 0 != *e) ? atoi(e) : 0; }();
    if (0 < bpc_env) per_cu = bpc_env;
    const long long want = 256 * per_cu;
    const unsigned grid = (unsigned)(g.batch < want ? g.batch : want);
    *name = (0 != paired) ? "spmdm_compute_mfma|wg_lds" : "spmdm_compute_mfma";
#define XSMM_SPM(NT, FULL) hipLaunchKernelGGL((spmdm_compute_mfma_kernel<NT, NT, FULL>), dim3(grid), dim3(256), lds, st, \
      g.batch, g.m, g.k, beta, rowidx, colidx, values, g.rstride, (long long)g.cap, b, c, paired)
    if (64 == g.m && 64 == g.k) { if (1 == nt) XSMM_SPM(1, true); else if (2 == nt) XSMM_SPM(2, true); else if (3 == nt) XSMM_SPM(3, true); else XSMM_SPM(4, true); }
    else { if (1 == nt) XSMM_SPM(1, false); else if (2 == nt) XSMM_SPM(2, false); else if (3 == nt) XSMM_SPM(3, false); else XSMM_SPM(4, false); }
#undef XSMM_SPM
    return (int)hipGetLastError();
  }
  return launch_spmdm_compute_generic(g.batch, g.m, g.n, g.k, g.m, g.k, 1, 1, transb, transc, beta, rowidx, colidx, values,
    (long long)g.rstride, (long long)g.cap, b, c, (long long)g.k * g.n, (long long)g.m * g.n, 0, g.m, 0, g.n, stream, name);
}

int launch_bf16_widen(const unsigned short* src, float* dst, long long count, void* stream)
{
  if (count <= 0) return 0;
  hipLaunchKernelGGL(bf16_widen_kernel, dim3(grid_for(count, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, count);
  return (int)hipGetLastError();
}

int launch_bgemm_copy(const BgemmGeom& g, int which, const void* src, int ld, void* dst, void* stream)
{
  hipStream_t st = (hipStream_t)stream;
  const long long total = (0 == which) ? (long long)g.m * g.k : ((1 == which || 5 == which) ? (long long)g.n * g.k : (long long)g.m * g.n);
  if (8 == g.typesize) {
    hipLaunchKernelGGL((bgemm_copy_kernel<double>), dim3(grid_for(total, 256)), dim3(256), 0, st, which, (const double*)src, ld, (double*)dst,
      g.mb, g.nb, g.kb, g.bm, g.bn, g.bk);
  }
  else {
    hipLaunchKernelGGL((bgemm_copy_kernel<float>), dim3(grid_for(total, 256)), dim3(256), 0, st, which, (const float*)src, ld, (float*)dst,
      g.mb, g.nb, g.kb, g.bm, g.bn, g.bk);
  }
  return (int)hipGetLastError();
}

} // namespace xsmm
