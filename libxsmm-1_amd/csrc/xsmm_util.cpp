// xsmm_util.cpp -- small host services that the hot-path samples link against: allocation, timer, RNG,
// matrix comparison, misc. Thin re-statements of the reference's helpers (include/libxsmm_malloc.h,
// _timer.h, _rng.h, _math.h); none of them is on the device path.
#include "xsmm_internal.hpp"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <unordered_map>

using namespace xsmm;

// ---- allocation ---------------------------------------------------------------------------------------------------
// libxsmm_malloc/aligned_malloc return memory that unchanged callers can read and write on the host. With a
// device present it is pinned host memory (hipHostMalloc), which the GPU can address directly, so samples that
// allocate their operands through the library (samples/spmdm/spmdm.c:205-209) run without staging copies.
namespace {
std::mutex g_alloc_lock;
struct AllocInfo { int kind; void* base; void* context; libxsmm_free_function free_fn; };
// kind 1: pinned (hipHostFree), 0: posix_memalign (free), 2: caller's allocator (base/context/free_fn as of allocation)
std::unordered_map<const void*, AllocInfo> g_allocs;
// custom default allocator (reference include/libxsmm_malloc.h:53-64); both NULL: library default
void* g_alloc_context = nullptr;
libxsmm_malloc_function g_alloc_malloc = { nullptr };
libxsmm_free_function g_alloc_free = { nullptr };
}

LIBXSMM_API int libxsmm_set_default_allocator(void* context, libxsmm_malloc_function malloc_fn, libxsmm_free_function free_fn)
{ // malloc_fn and free_fn must come as a pair; two NULLs restore the built-in (pinned host memory) allocator
  if ((nullptr == malloc_fn.function) != (nullptr == free_fn.function)) {
    static int error_once = 0;
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: allocator setup without malloc or free function!\n");
    return EXIT_FAILURE;
  }
  std::lock_guard<std::mutex> guard(g_alloc_lock);
  g_alloc_context = (nullptr != malloc_fn.function ? context : nullptr);
  g_alloc_malloc = malloc_fn; g_alloc_free = free_fn;
  return EXIT_SUCCESS;
}

LIBXSMM_API int libxsmm_get_default_allocator(void** context, libxsmm_malloc_function* malloc_fn, libxsmm_free_function* free_fn)
{
  if (nullptr == context || nullptr == malloc_fn || nullptr == free_fn) return EXIT_FAILURE;
  std::lock_guard<std::mutex> guard(g_alloc_lock);
  *context = g_alloc_context; *malloc_fn = g_alloc_malloc; *free_fn = g_alloc_free;
  return EXIT_SUCCESS;
}

LIBXSMM_API void* libxsmm_aligned_malloc(size_t size, size_t alignment)
{
  if (0 == size) return nullptr;
  size_t al = (0 == alignment ? (size_t)LIBXSMM_ALIGNMENT : alignment);
  if (al < sizeof(void*)) al = sizeof(void*);
  while (0 != (al & (al - 1))) al &= (al - 1); // round down to a power of two
  void* p = nullptr; AllocInfo info = { 0, nullptr, nullptr, { nullptr } };
  void* ctx; libxsmm_malloc_function mfn; libxsmm_free_function ffn;
  { std::lock_guard<std::mutex> guard(g_alloc_lock); ctx = g_alloc_context; mfn = g_alloc_malloc; ffn = g_alloc_free; }
  if (nullptr != mfn.function) { // the caller's allocator: over-allocate and align inside the block
    void* const base = (nullptr != ctx ? mfn.ctx_form(ctx, size + al) : mfn.function(size + al));
    if (nullptr == base) return nullptr;
    p = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(base) + al - 1) & ~(uintptr_t)(al - 1));
    info.kind = 2; info.base = base; info.context = ctx; info.free_fn = ffn;
  }
  else if (device_ready() && hipSuccess == hipHostMalloc(&p, size, hipHostMallocDefault)) info.kind = 1;
  else {
    (void)hipGetLastError();
    if (0 != posix_memalign(&p, al, size)) p = nullptr;
  }
  if (nullptr != p) { std::lock_guard<std::mutex> guard(g_alloc_lock); g_allocs[p] = info; }
  return p;
}

LIBXSMM_API void* libxsmm_malloc(size_t size) { return libxsmm_aligned_malloc(size, 0); }

LIBXSMM_API void libxsmm_free(const void* memory)
{
  if (nullptr == memory) return;
  int kind = -1; AllocInfo info = { -1, nullptr, nullptr, { nullptr } };
  {
    std::lock_guard<std::mutex> guard(g_alloc_lock);
    auto it = g_allocs.find(memory);
    if (it != g_allocs.end()) { info = it->second; kind = info.kind; g_allocs.erase(it); }
  }
  if (1 == kind) (void)hipHostFree(const_cast<void*>(memory));
  else if (0 == kind) free(const_cast<void*>(memory));
  else if (2 == kind) { // released by the allocator that was active when the buffer was made ("pending buffers")
    if (nullptr != info.context) info.free_fn.ctx_form(info.context, info.base); else info.free_fn.function(info.base);
  }
  else if (nullptr != kernel_from_pointer(memory)) libxsmm_release_kernel(memory); // reference frees csr_reg kernels this way (src/libxsmm_fsspmdm.c:301-306)
  else if (0 != libxsmm_verbosity) fprintf(stderr, "LIBXSMM ERROR: libxsmm_free of unknown memory!\n");
}

// Prints a GEMM call's arguments (reference src/libxsmm_gemm.c:557-650). The reference dumps the operands into MHD image
// files when ostream == NULL; that debugging aid is not part of this engine: the call is then a no-op.
LIBXSMM_API void libxsmm_gemm_print2(void* ostream, libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec,
  const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const void* alpha, const void* a, const libxsmm_blasint* lda, const void* b, const libxsmm_blasint* ldb,
  const void* beta, void* c, const libxsmm_blasint* ldc)
{
  if (nullptr == m) return;
  const libxsmm_blasint nn = *(nullptr != n ? n : m), kk = *(nullptr != k ? k : m);
  const char ta = (nullptr != transa ? *transa : 'n'), tb = (nullptr != transb ? *transb : 'n');
  const libxsmm_blasint ilda = (nullptr != lda ? *lda : (('n' == ta || 'N' == ta) ? *m : kk));
  const libxsmm_blasint ildb = (nullptr != ldb ? *ldb : (('n' == tb || 'N' == tb) ? kk : nn));
  const libxsmm_blasint ildc = *(nullptr != ldc ? ldc : m);
  char sa[64], sb[64], prefix = 0;
  if (LIBXSMM_GEMM_PRECISION_F64 == iprec && iprec == oprec) {
    snprintf(sa, sizeof(sa), "%g", nullptr != alpha ? *static_cast<const double*>(alpha) : 1.0);
    snprintf(sb, sizeof(sb), "%g", nullptr != beta ? *static_cast<const double*>(beta) : 1.0);
    prefix = 'd';
  }
  else if (LIBXSMM_GEMM_PRECISION_F32 == iprec && iprec == oprec) {
    snprintf(sa, sizeof(sa), "%g", nullptr != alpha ? (double)*static_cast<const float*>(alpha) : 1.0);
    snprintf(sb, sizeof(sb), "%g", nullptr != beta ? (double)*static_cast<const float*>(beta) : 1.0);
    prefix = 's';
  }
  else {
    static int error_once = 0;
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: unsupported data-type requested!\n");
    return;
  }
  if (nullptr == ostream) return;
  FILE* const out = static_cast<FILE*>(ostream);
  if (nullptr != a && nullptr != b && nullptr != c) {
    fprintf(out, "%cgemm('%c', '%c', %llu/*m*/, %llu/*n*/, %llu/*k*/,\n  %s/*alpha*/, %p/*a*/, %llu/*lda*/,\n              %p/*b*/, %llu/*ldb*/,\n   %s/*beta*/, %p/*c*/, %llu/*ldc*/)",
      prefix, ta, tb, (unsigned long long)*m, (unsigned long long)nn, (unsigned long long)kk, sa, a, (unsigned long long)ilda,
      b, (unsigned long long)ildb, sb, c, (unsigned long long)ildc);
  }
  else {
    fprintf(out, "%cgemm(trans=%c%c mnk=%llu,%llu,%llu ldx=%llu,%llu,%llu a,b=%s,%s)", prefix, ta, tb,
      (unsigned long long)*m, (unsigned long long)nn, (unsigned long long)kk,
      (unsigned long long)ilda, (unsigned long long)ildb, (unsigned long long)ildc, sa, sb);
  }
}

LIBXSMM_API void libxsmm_gemm_print(void* ostream, libxsmm_gemm_precision precision, const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k, const void* alpha, const void* a, const libxsmm_blasint* lda,
  const void* b, const libxsmm_blasint* ldb, const void* beta, void* c, const libxsmm_blasint* ldc)
{
  libxsmm_gemm_print2(ostream, precision, precision, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}

LIBXSMM_API unsigned char libxsmm_typesize(libxsmm_datatype datatype) { return (unsigned char)LIBXSMM_TYPESIZE(datatype); }

// ---- timer --------------------------------------------------------------------------------------------------------
LIBXSMM_API libxsmm_timer_tickint libxsmm_timer_tick(void)
{
  return (libxsmm_timer_tickint)std::chrono::duration_cast<std::chrono::nanoseconds>(
    std::chrono::steady_clock::now().time_since_epoch()).count();
}
LIBXSMM_API libxsmm_timer_tickint libxsmm_timer_cycles(libxsmm_timer_tickint tick0, libxsmm_timer_tickint tick1)
{ return (tick0 <= tick1) ? (tick1 - tick0) : (tick0 - tick1); }
LIBXSMM_API double libxsmm_timer_duration(libxsmm_timer_tickint tick0, libxsmm_timer_tickint tick1)
{ return 1E-9 * (double)libxsmm_timer_cycles(tick0, tick1); }

// ---- RNG ----------------------------------------------------------------------------------------------------------
// libxsmm_rng_f64 is drand48 and libxsmm_rng_u32 draws from lrand48 on Linux (reference src/libxsmm_rng.c:131,
// 228-259); the POSIX 48-bit LCG is carried explicitly so sequences do not depend on the C library.
namespace {
std::mutex g_rng_lock;
unsigned long long g_rng_x = 0x1234ABCD330EULL;
unsigned long long rng_next() { g_rng_x = (0x5DEECE66DULL * g_rng_x + 0xBULL) & 0xFFFFFFFFFFFFULL; return g_rng_x; }
}
LIBXSMM_API void libxsmm_rng_set_seed(unsigned int seed)
{ std::lock_guard<std::mutex> guard(g_rng_lock); g_rng_x = (((unsigned long long)seed) << 16) | 0x330EULL; }
LIBXSMM_API double libxsmm_rng_f64(void)
{ std::lock_guard<std::mutex> guard(g_rng_lock); return (double)rng_next() / 281474976710656.0; }
LIBXSMM_API unsigned int libxsmm_rng_u32(unsigned int n)
{
  if (0 == n) return 0;
  std::lock_guard<std::mutex> guard(g_rng_lock);
  const unsigned int q = ((1U << 31) / n) * n;
  unsigned int r = (unsigned int)(rng_next() >> 17); // lrand48: upper 31 bits
  if (q != (1U << 31)) while (q <= r) r = (unsigned int)(rng_next() >> 17);
  return r % n;
}
LIBXSMM_API void libxsmm_rng_f32_seq(float* rngs, libxsmm_blasint count)
{ // uniform [0,1); the reference uses xoshiro128+ lanes here -- only the distribution is part of the contract
  if (nullptr == rngs) return;
  std::lock_guard<std::mutex> guard(g_rng_lock);
  for (libxsmm_blasint i = 0; i < count; ++i) rngs[i] = (float)((rng_next() >> 24) * (1.0 / 16777216.0));
}

// ---- math helpers ----------------------------------------------------------------------------------------------------
LIBXSMM_API unsigned int libxsmm_isqrt_u64(unsigned long long x)
{ // floor(sqrt(x)) (include/libxsmm_math.h:102)
  unsigned long long r = (unsigned long long)std::sqrt((long double)x);
  while (r * r > x) --r;
  while ((r + 1) * (r + 1) <= x) ++r;
  return (unsigned int)r;
}

LIBXSMM_API unsigned int libxsmm_isqrt_u32(unsigned int x) { return libxsmm_isqrt_u64(x); } // include/libxsmm_math.h:104

LIBXSMM_API unsigned int libxsmm_icbrt_u64(unsigned long long x)
{ // floor(cbrt(x)) (include/libxsmm_math.h:112)
  unsigned long long r = (unsigned long long)std::cbrt((long double)x);
  while (r * r * r > x) --r;
  while ((r + 1) * (r + 1) * (r + 1) <= x) ++r;
  return (unsigned int)r;
}
LIBXSMM_API unsigned int libxsmm_icbrt_u32(unsigned int x) { return libxsmm_icbrt_u64(x); }

LIBXSMM_API float libxsmm_sexp2(float x) { return std::exp2(x); } // include/libxsmm_math.h:121 (libm path)

LIBXSMM_API size_t libxsmm_shuffle(unsigned int n)
{ // a stride co-prime to n, close to n/2 (used to permute 0..n-1); include/libxsmm_math.h:99
  if (n < 2) return 0;
  auto gcd = [](unsigned a, unsigned b) { while (0 != b) { const unsigned t = a % b; a = b; b = t; } return a; };
  for (unsigned d = 0; d < n; ++d) {
    const unsigned lo = n / 2 - (d < n / 2 ? d : n / 2), hi = n / 2 + d;
    if (0 < lo && 1 == gcd(lo, n)) return lo;
    if (hi < n && 1 == gcd(hi, n)) return hi;
  }
  return 1;
}

// ---- matdiff (include/libxsmm_math.h:40-71; src/template/libxsmm_matdiff.tpl.c) -----------------------------------------------
LIBXSMM_API void libxsmm_matdiff_clear(libxsmm_matdiff_info* info)
{
  if (nullptr == info) return;
  memset(info, 0, sizeof(*info));
  info->min_ref = info->min_tst = std::numeric_limits<double>::infinity();
  info->max_ref = info->max_tst = -std::numeric_limits<double>::infinity();
  info->m = info->n = -1;
}

LIBXSMM_API int libxsmm_matdiff(libxsmm_matdiff_info* info, libxsmm_datatype datatype, libxsmm_blasint m, libxsmm_blasint n,
  const void* ref, const void* tst, const libxsmm_blasint* ldref, const libxsmm_blasint* ldtst)
{
  bool swapped = false; // statistics of a single set: src/libxsmm_math.c:54,161-172
  if (nullptr == ref && nullptr != tst) { ref = tst; tst = nullptr; swapped = true; }
  if (nullptr == info || nullptr == ref || (LIBXSMM_DATATYPE_F64 != datatype && LIBXSMM_DATATYPE_F32 != datatype) || m < 0 || n < 0) return EXIT_FAILURE;
  const libxsmm_blasint ldr = (nullptr != ldref ? *ldref : m), ldt = (nullptr != ldtst ? *ldtst : m);
  auto at = [datatype](const void* p, size_t i) { return LIBXSMM_DATATYPE_F64 == datatype ? static_cast<const double*>(p)[i] : (double)static_cast<const float*>(p)[i]; };
  libxsmm_matdiff_clear(info);
  const double inf = std::numeric_limits<double>::infinity();
  double sumsq_ref = 0, sumsq_d = 0, norm1_ref = 0, normi_ref_acc = 0;
  // column sums (one-norm) and row sums (infinity-norm) of the differences and of the reference
  double* rowsum_d = static_cast<double*>(calloc((size_t)(m ? m : 1), sizeof(double)));
  double* rowsum_r = static_cast<double*>(calloc((size_t)(m ? m : 1), sizeof(double)));
  bool nan = false;
  for (libxsmm_blasint j = 0; j < n && !nan; ++j) {
    double colsum_d = 0, colsum_r = 0;
    for (libxsmm_blasint i = 0; i < m; ++i) {
      const double r = at(ref, (size_t)j * ldr + i), t = (nullptr != tst ? at(tst, (size_t)j * ldt + i) : 0.0);
      if (r < info->min_ref) info->min_ref = r;
      if (r > info->max_ref) info->max_ref = r;
      if (!(t == t) || !(std::fabs(t) < inf)) { info->m = i; info->n = j; nan = true; break; }
      const double d = (nullptr != tst ? std::fabs(r - t) : 0.0);
      if (t < info->min_tst) info->min_tst = t;
      if (t > info->max_tst) info->max_tst = t;
      if (info->linf_abs < d) { info->linf_abs = d; info->m = i; info->n = j; }
      if (0 < std::fabs(r)) { const double dr = d / std::fabs(r); if (info->linf_rel < dr) info->linf_rel = dr; info->l2_rel += dr * dr; }
      info->l1_ref += std::fabs(r); info->l1_tst += std::fabs(t);
      sumsq_ref += r * r; sumsq_d += d * d;
      colsum_d += d; colsum_r += std::fabs(r);
      rowsum_d[i] += d; rowsum_r[i] += std::fabs(r);
    }
    if (info->norm1_abs < colsum_d) info->norm1_abs = colsum_d;
    if (norm1_ref < colsum_r) norm1_ref = colsum_r;
  }
  if (nan) {
    info->norm1_abs = info->norm1_rel = info->normi_abs = info->normi_rel = info->normf_rel = info->linf_abs = info->linf_rel = info->l2_abs = info->l2_rel = inf;
    free(rowsum_d); free(rowsum_r);
    return EXIT_SUCCESS;
  }
  for (libxsmm_blasint i = 0; i < m; ++i) { if (info->normi_abs < rowsum_d[i]) info->normi_abs = rowsum_d[i]; if (normi_ref_acc < rowsum_r[i]) normi_ref_acc = rowsum_r[i]; }
  free(rowsum_d); free(rowsum_r);
  info->norm1_rel = (0 < norm1_ref ? info->norm1_abs / norm1_ref : info->norm1_abs);
  info->normi_rel = (0 < normi_ref_acc ? info->normi_abs / normi_ref_acc : info->normi_abs);
  info->normf_rel = (0 < sumsq_ref ? std::sqrt(sumsq_d / sumsq_ref) : std::sqrt(sumsq_d));
  info->l2_abs = std::sqrt(sumsq_d); info->l2_rel = std::sqrt(info->l2_rel);
  const double cnt = (double)m * n;
  if (0 < cnt) { // src/template/libxsmm_matdiff.tpl.c:153-154,185-196,232-233
    info->avg_ref = info->l1_ref / cnt; info->avg_tst = info->l1_tst / cnt;
    double vr = 0, vt = 0;
    for (libxsmm_blasint j = 0; j < n; ++j) {
      for (libxsmm_blasint i = 0; i < m; ++i) {
        const double r = at(ref, (size_t)j * ldr + i) - info->avg_ref, t = (nullptr != tst ? at(tst, (size_t)j * ldt + i) : 0.0) - info->avg_tst;
        vr += r * r; vt += t * t;
      }
    }
    info->var_ref = vr / cnt; info->var_tst = vt / cnt;
  }
  if (0 > info->m && 0 == info->linf_abs) { info->m = -1; info->n = -1; }
  if (swapped) {
    info->min_tst = info->min_ref; info->min_ref = 0; info->max_tst = info->max_ref; info->max_ref = 0;
    info->avg_tst = info->avg_ref; info->avg_ref = 0; info->var_tst = info->var_ref; info->var_ref = 0;
    info->l1_tst = info->l1_ref; info->l1_ref = 0;
  }
  return EXIT_SUCCESS;
}

LIBXSMM_API void libxsmm_matdiff_reduce(libxsmm_matdiff_info* output, const libxsmm_matdiff_info* input)
{ // keep the larger difference per field (reference src/libxsmm_math.c:196-250)
  if (nullptr == output || nullptr == input) return;
  if (output->linf_abs < input->linf_abs) { output->linf_abs = input->linf_abs; output->m = input->m; output->n = input->n; }
  if (output->norm1_abs < input->norm1_abs) { output->norm1_abs = input->norm1_abs; output->norm1_rel = input->norm1_rel; }
  if (output->normi_abs < input->normi_abs) { output->normi_abs = input->normi_abs; output->normi_rel = input->normi_rel; }
  if (output->normf_rel < input->normf_rel) output->normf_rel = input->normf_rel;
  if (output->linf_rel < input->linf_rel) output->linf_rel = input->linf_rel;
  if (output->l2_abs < input->l2_abs) output->l2_abs = input->l2_abs;
  if (output->l2_rel < input->l2_rel) output->l2_rel = input->l2_rel;
  if (output->var_ref < input->var_ref) output->var_ref = input->var_ref;
  if (output->var_tst < input->var_tst) output->var_tst = input->var_tst;
  output->avg_ref = 0.5 * (output->avg_ref + input->avg_ref); output->avg_tst = 0.5 * (output->avg_tst + input->avg_tst);
  output->l1_ref += input->l1_ref; output->l1_tst += input->l1_tst;
  if (input->min_ref < output->min_ref) output->min_ref = input->min_ref;
  if (input->max_ref > output->max_ref) output->max_ref = input->max_ref;
  if (input->min_tst < output->min_tst) output->min_tst = input->min_tst;
  if (input->max_tst > output->max_tst) output->max_tst = input->max_tst;
}
