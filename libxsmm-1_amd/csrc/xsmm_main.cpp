// xsmm_main.cpp -- library life cycle, GEMM descriptors, kernel registry and dispatch.
//
// Replaces the reference's dispatch layer (src/libxsmm_main.c: libxsmm_init :708-813, internal_find_code
// :1697-1923, libxsmm_build :1246-1683, libxsmm_xmmdispatch :2139-2163) with a back end for gfx950:
// "building" a kernel means validating the descriptor the way the reference's generator does
// (src/generator_gemm.c:211-234) and binding it to a pre-compiled HIP kernel family; the bare function
// pointer handed to the caller is a small executable thunk that carries the kernel record.
#include "xsmm_internal.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace xsmm;

// data symbols of the reference ABI (include/libxsmm_generator.h:279-281)
extern "C" {
LIBXSMM_VISIBILITY unsigned int libxsmm_ninit = 0;
LIBXSMM_VISIBILITY int libxsmm_verbosity = 0;
}

namespace {

struct Key {
  unsigned char bytes[LIBXSMM_DESCRIPTOR_MAXSIZE]; // descriptor padded with zeros (reference pads to 64 B as hash key)
  bool operator==(const Key& o) const { return 0 == memcmp(bytes, o.bytes, sizeof(bytes)); }
};
struct KeyHash {
  size_t operator()(const Key& k) const { // FNV-1a; the reference uses CRC32 (src/libxsmm_hash.c), any hash will do
    uint64_t h = 1469598103934665603ULL;
    for (unsigned char b : k.bytes) { h ^= b; h *= 1099511628211ULL; }
    return (size_t)h;
  }
};

struct Registry {
  std::shared_mutex lock;
  std::unordered_map<Key, Kernel*, KeyHash> by_desc;
  std::unordered_map<const void*, Kernel*> by_thunk;   // registered and caller-owned kernels
  // statistics in the reference's buckets (src/libxsmm_main.c:278, :635-637): MNK^(1/3) <= 13 / 23 / 64 / above
  unsigned long long ntry[2][4] = {{0}}, njit[2][4] = {{0}}, ncol[2][4] = {{0}};
};
Registry& registry() { static Registry* r = new Registry(); return *r; }

std::once_flag g_init_once;
std::atomic<int> g_target_archid{LIBXSMM_AMD_GFX950};
std::atomic<int> g_auto_prefetch{LIBXSMM_GEMM_PREFETCH_NONE};
std::atomic<int> g_mfma{1};

int bucket(unsigned m, unsigned n, unsigned k)
{
  const double s = (double)m * n * k;
  return s <= 13.0 * 13 * 13 ? 0 : (s <= 23.0 * 23 * 23 ? 1 : (s <= 64.0 * 64 * 64 ? 2 : 3));
}

void print_statistic()
{ // LIBXSMM_VERBOSE: per-precision TRY/JIT/COL table (reference src/libxsmm_main.c:278-353)
  Registry& r = registry();
  static const char* const names[2] = { "DP", "SP" };
  for (int p = 0; p < 2; ++p) {
    unsigned long long tot = 0;
    for (int b = 0; b < 4; ++b) tot += r.ntry[p][b];
    if (0 == tot) continue;
    fprintf(stderr, "\nLIBXSMM_TARGET: gfx950 [MI355X]\n%s    TRY    JIT    COL\n", names[p]);
    static const char* const bn[4] = { "sml", "med", "big", "xxx" };
    for (int b = 0; b < 4; ++b) {
      fprintf(stderr, "%s %6llu %6llu %6llu\n", bn[b], r.ntry[p][b], r.njit[p][b], r.ncol[p][b]);
    }
  }
  fprintf(stderr, "Registry: %llu kernels, device launches: %llu\n",
    (unsigned long long)r.by_desc.size(), libxsmm_amd_launch_count());
}

void init_once()
{
  const char* const v = getenv("LIBXSMM_VERBOSE");
  if (nullptr != v && 0 != *v) libxsmm_verbosity = atoi(v);
  const char* const t = getenv("LIBXSMM_TARGET");
  if (nullptr != t && 0 != *t) libxsmm_set_target_arch(t);
  const char* const mf = getenv("LIBXSMM_AMD_MFMA");
  if (nullptr != mf && 0 != *mf) g_mfma.store(atoi(mf));
  const char* const pf = getenv("LIBXSMM_GEMM_PREFETCH");
  if (nullptr != pf && 0 != *pf) g_auto_prefetch.store(atoi(pf));
  atexit([]() { if (0 != libxsmm_verbosity) print_statistic(); });
  ++libxsmm_ninit;
}

// ---- thunks --------------------------------------------------------------------------------------
// A dispatched kernel must be a distinct bare function pointer void(*)(const void*,const void*,void*,...)
// (include/libxsmm_typedefs.h:526-549) with no context argument. Each thunk is 32 bytes of x86-64:
//     mov r10, <Kernel*> ; mov rax, <xsmm_thunk_entry> ; jmp rax
// and xsmm_thunk_entry forwards the six integer argument registers plus r10 (as a 7th, stack argument)
// to xsmm_thunk_dispatch. The reference keeps its JIT code in RWX pages the same way
// (src/libxsmm_main.c:1642-1656). If executable pages cannot be mapped, a fixed pool of pre-compiled
// trampolines is used instead.
extern "C" void xsmm_thunk_entry(void);
extern "C" void xsmm_thunk_dispatch(const void* a, const void* b, void* c, const void* x3, const void* x4,
                                    const void* x5, void* ctx, const void* return_address, const void* x6)
{ // stack at this point: [ctx pushed by xsmm_thunk_entry][the user's return address][the user's 7th argument, if any]:
  // x6 is what i16 -> f32 kernels receive as the scaling factor (kernel(a, b, c, pa, pb, pc, &scf)); for every other
  // kernel it is an unused word of the caller's frame
  (void)x4; (void)x5; (void)return_address;
  call_kernel(static_cast<Kernel*>(ctx), a, b, c, x3, x6);
}

__asm__(
  ".text\n"
  ".globl xsmm_thunk_entry\n"
  ".type xsmm_thunk_entry,@function\n"
  "xsmm_thunk_entry:\n"
  "  pushq %r10\n"
  "  call xsmm_thunk_dispatch@PLT\n"
  "  addq $8, %rsp\n"
  "  ret\n"
  ".size xsmm_thunk_entry, .-xsmm_thunk_entry\n");

constexpr size_t THUNK_SIZE = 32;
struct ThunkPages {
  std::mutex lock;
  std::vector<unsigned char*> pages;
  std::vector<void*> free_list;
  size_t used_in_last = 0;
  bool exec_ok = true;
};
ThunkPages& thunk_pages() { static ThunkPages* t = new ThunkPages(); return *t; }

// fallback pool
constexpr int POOL = 256;
Kernel* g_pool_ctx[POOL];
template<int I> void pool_tramp(const void* a, const void* b, void* c, ...)
{ // the 4th argument (batch-reduce count, or a prefetch pointer that is ignored) is read from the register save
  // area; when the caller passed only three arguments this yields an unused garbage value, never a fault.
  va_list ap; va_start(ap, c);
  const void* const x3 = va_arg(ap, const void*);
  (void)va_arg(ap, const void*); (void)va_arg(ap, const void*);
  const void* const x6 = va_arg(ap, const void*); // 7th argument (first one on the stack): the scaling factor of i16 -> f32 kernels
  va_end(ap);
  call_kernel(g_pool_ctx[I], a, b, c, x3, x6);
}
template<int... I> void fill_pool(void* (&tab)[POOL], std::integer_sequence<int, I...>)
{
  void* t[] = { (void*)&pool_tramp<I>... };
  for (int i = 0; i < POOL; ++i) tab[i] = t[i];
}
void* g_pool_fn[POOL];
bool g_pool_used[POOL];
std::once_flag g_pool_once;

void* pool_alloc(Kernel* k)
{
  std::call_once(g_pool_once, []() { fill_pool(g_pool_fn, std::make_integer_sequence<int, POOL>()); });
  ThunkPages& tp = thunk_pages();
  for (int i = 0; i < POOL; ++i) if (!g_pool_used[i]) { g_pool_used[i] = true; g_pool_ctx[i] = k; (void)tp; return g_pool_fn[i]; }
  return nullptr;
}

} // namespace

namespace xsmm {

int verbosity() { return libxsmm_verbosity; }
bool once(int* flag) { return 1 == __atomic_add_fetch(flag, 1, __ATOMIC_RELAXED); }

void* make_thunk(Kernel* k)
{
  ThunkPages& tp = thunk_pages();
  std::lock_guard<std::mutex> guard(tp.lock);
  unsigned char* slot = nullptr;
  if (tp.exec_ok) {
    if (!tp.free_list.empty()) { slot = (unsigned char*)tp.free_list.back(); tp.free_list.pop_back(); }
    else {
      const size_t page = (size_t)sysconf(_SC_PAGESIZE);
      if (tp.pages.empty() || tp.used_in_last + THUNK_SIZE > page) {
        void* p = mmap(nullptr, page, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (MAP_FAILED == p) { tp.exec_ok = false; }
        else { tp.pages.push_back((unsigned char*)p); tp.used_in_last = 0; }
      }
      if (tp.exec_ok) { slot = tp.pages.back() + tp.used_in_last; tp.used_in_last += THUNK_SIZE; }
    }
  }
  if (nullptr != slot) {
    unsigned char code[THUNK_SIZE];
    memset(code, 0xCC, sizeof(code)); // int3 padding
    const uint64_t ctx = (uint64_t)(uintptr_t)k, entry = (uint64_t)(uintptr_t)&xsmm_thunk_entry;
    code[0] = 0x49; code[1] = 0xBA; memcpy(code + 2, &ctx, 8);      // mov r10, imm64
    code[10] = 0x48; code[11] = 0xB8; memcpy(code + 12, &entry, 8); // mov rax, imm64
    code[20] = 0xFF; code[21] = 0xE0;                               // jmp rax
    memcpy(slot, code, sizeof(code));
    __builtin___clear_cache((char*)slot, (char*)slot + THUNK_SIZE);
    return slot;
  }
  return pool_alloc(k);
}

void free_thunk(void* thunk)
{
  if (nullptr == thunk) return;
  ThunkPages& tp = thunk_pages();
  std::lock_guard<std::mutex> guard(tp.lock);
  for (int i = 0; i < POOL; ++i) if (g_pool_fn[i] == thunk) { g_pool_used[i] = false; g_pool_ctx[i] = nullptr; return; }
  tp.free_list.push_back(thunk);
}

Kernel* kernel_from_pointer(const void* fn)
{
  if (nullptr == fn) return nullptr;
  Registry& r = registry();
  std::shared_lock<std::shared_mutex> guard(r.lock);
  auto it = r.by_thunk.find(fn);
  return it == r.by_thunk.end() ? nullptr : it->second;
}

} // namespace xsmm

// ---- life cycle ----------------------------------------------------------------------------------------
LIBXSMM_API void libxsmm_init(void) { std::call_once(g_init_once, init_once); }

LIBXSMM_API void libxsmm_finalize(void)
{ // kernels stay valid until process exit in this implementation (the reference releases the registry);
  // finalize only drains the stream so that results are complete.
  if (device_ready()) (void)stream_sync();
  jit_async_drain(); // (a process on its way out must not leave the helper thread inside the compiler, see libxsmm_amd_jit_drain)
}

__attribute__((constructor)) static void xsmm_ctor(void) { libxsmm_init(); } // LIBXSMM_ATTRIBUTE_CTOR (src/libxsmm_main.c:708)

LIBXSMM_API int libxsmm_get_target_archid(void) { return g_target_archid.load(); }
LIBXSMM_API void libxsmm_set_target_archid(int id) { g_target_archid.store(id); }
LIBXSMM_API const char* libxsmm_get_target_arch(void)
{
  return LIBXSMM_TARGET_ARCH_GENERIC == g_target_archid.load() ? "generic" : "gfx950";
}
LIBXSMM_API void libxsmm_set_target_arch(const char* arch)
{ // the reference accepts 0|sse|snb|hsw|knl|knm|skx|clx|cpx|generic (src/libxsmm_main.c:1027); "generic" disables
  // JIT and hence dispatch, everything else maps to the one device target.
  if (nullptr != arch && 0 == strcmp(arch, "generic")) g_target_archid.store(LIBXSMM_TARGET_ARCH_GENERIC);
  else g_target_archid.store(LIBXSMM_AMD_GFX950);
}
LIBXSMM_API int libxsmm_get_verbosity(void) { return libxsmm_verbosity; }
LIBXSMM_API void libxsmm_set_verbosity(int level) { libxsmm_verbosity = level; }
LIBXSMM_API libxsmm_gemm_prefetch_type libxsmm_get_gemm_auto_prefetch(void) { return (libxsmm_gemm_prefetch_type)g_auto_prefetch.load(); }
LIBXSMM_API void libxsmm_set_gemm_auto_prefetch(libxsmm_gemm_prefetch_type strategy) { g_auto_prefetch.store((int)strategy); }
LIBXSMM_API libxsmm_gemm_prefetch_type libxsmm_get_gemm_prefetch(int prefetch)
{ // src/libxsmm_gemm.c:478-494: a negative value (LIBXSMM_PREFETCH_AUTO) selects the configured strategy
  return 0 > prefetch ? (libxsmm_gemm_prefetch_type)g_auto_prefetch.load() : (libxsmm_gemm_prefetch_type)prefetch;
}
LIBXSMM_API libxsmm_gemm_prefetch_type libxsmm_get_gemm_xprefetch(const int* prefetch)
{ // src/libxsmm_gemm.c:471-475
  return libxsmm_get_gemm_prefetch(nullptr == prefetch ? g_auto_prefetch.load() : *prefetch);
}
LIBXSMM_API int libxsmm_amd_set_mfma(int mode) { return g_mfma.exchange(0 != mode ? 1 : 0); }
LIBXSMM_API int libxsmm_amd_get_mfma(void) { return g_mfma.load(); }

// ---- descriptors (src/libxsmm_generator.c:47-336) ----------------------------------------------------------
namespace {
libxsmm_gemm_descriptor* desc_init(libxsmm_descriptor_blob* blob, int iprec, int oprec,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  double alpha, double beta, int flags, int prefetch)
{
  // LIBXSMM_GEMM_NO_BYPASS (include/libxsmm_generator.h:36-39): no TRANS_A, alpha == 1, beta in {0, 1}
  if (nullptr == blob || 0 != (flags & LIBXSMM_GEMM_FLAG_TRANS_A) || 1.0 != alpha || (1.0 != beta && 0.0 != beta)) return nullptr;
  if (m < 0 || n < 0 || k < 0 || lda < 0 || ldb < 0 || ldc < 0) return nullptr;
  memset(blob, 0, sizeof(*blob));
  libxsmm_gemm_descriptor* d = reinterpret_cast<libxsmm_gemm_descriptor*>(blob->data);
  d->datatype = (unsigned char)LIBXSMM_GETENUM(iprec, oprec);
  d->flags = (unsigned short)(flags | (0.0 == beta ? LIBXSMM_GEMM_FLAG_BETA_0 : 0)); // src/libxsmm_main.h:117-127
  d->m = (unsigned)m; d->n = (unsigned)n; d->k = (unsigned)k;
  d->lda = (unsigned)lda; d->ldb = (unsigned)ldb; d->ldc = (unsigned)ldc;
  d->prefetch = (unsigned char)prefetch;
  return d;
}
} // namespace

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_dgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  double alpha, double beta, int flags, int prefetch)
{
  return desc_init(blob, LIBXSMM_GEMM_PRECISION_F64, LIBXSMM_GEMM_PRECISION_F64, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch);
}

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_sgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  float alpha, float beta, int flags, int prefetch)
{
  return desc_init(blob, LIBXSMM_GEMM_PRECISION_F32, LIBXSMM_GEMM_PRECISION_F32, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch);
}

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_dinit2(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, double alpha, double beta, int flags, int prefetch)
{ // src/libxsmm_generator.c:194-243; the low-precision cases build a descriptor too (dispatch then returns NULL)
  switch (iprec) {
    case LIBXSMM_GEMM_PRECISION_F64: case LIBXSMM_GEMM_PRECISION_F32:
    case LIBXSMM_GEMM_PRECISION_I16: case LIBXSMM_GEMM_PRECISION_BF16:
      return desc_init(blob, iprec, oprec, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch);
    default: {
      static int error_once = 0;
      if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: GEMM precision is not supported!\n");
      return nullptr;
    }
  }
}

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_dinit(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision precision, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, double alpha, double beta, int flags, int prefetch)
{
  return libxsmm_gemm_descriptor_dinit2(blob, precision, precision, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch);
}

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init3(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta,
  int flags, int prefetch, double* dalpha, double* dbeta)
{ // src/libxsmm_generator.c:268-336: NULL alpha/beta select LIBXSMM_ALPHA/LIBXSMM_BETA; scalars are read in the input precision
  double aa = LIBXSMM_ALPHA, bb = LIBXSMM_BETA;
  switch (iprec) {
    case LIBXSMM_GEMM_PRECISION_F64:
      if (nullptr != alpha) aa = *static_cast<const double*>(alpha);
      if (nullptr != beta) bb = *static_cast<const double*>(beta);
      break;
    case LIBXSMM_GEMM_PRECISION_F32: case LIBXSMM_GEMM_PRECISION_BF16:
      if (nullptr != alpha) aa = *static_cast<const float*>(alpha);
      if (nullptr != beta) bb = *static_cast<const float*>(beta);
      break;
    case LIBXSMM_GEMM_PRECISION_I16:
      if (LIBXSMM_GEMM_PRECISION_I32 == oprec) {
        if (nullptr != alpha) aa = *static_cast<const short*>(alpha);
        if (nullptr != beta) bb = *static_cast<const short*>(beta);
      }
      else {
        if (nullptr != alpha) aa = *static_cast<const float*>(alpha);
        if (nullptr != beta) bb = *static_cast<const float*>(beta);
      }
      break;
    default: {
      static int error_once = 0;
      if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: GEMM precision is not supported!\n");
      return nullptr;
    }
  }
  if (nullptr != dalpha) *dalpha = aa;
  if (nullptr != dbeta) *dbeta = bb;
  return desc_init(blob, iprec, oprec, m, n, k, lda, ldb, ldc, aa, bb, flags, prefetch);
}

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init2(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta, int flags, int prefetch)
{
  return libxsmm_gemm_descriptor_init3(blob, iprec, oprec, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch, nullptr, nullptr);
}

LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision precision, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta, int flags, int prefetch)
{
  return libxsmm_gemm_descriptor_init2(blob, precision, precision, m, n, k, lda, ldb, ldc, alpha, beta, flags, prefetch);
}

// ---- dispatch ---------------------------------------------------------------------------------------------
namespace {

// What the reference's generator rejects (src/generator_gemm.c:211-234) plus what this back end lacks.
// 1 i16->i32, 2 i16->f32, 3 bf16->f32, 4 bf16->bf16 (0: not a low-precision descriptor)
int lowp_kind(const libxsmm_gemm_descriptor& d)
{
  const int ip = LIBXSMM_GETENUM_INP(d.datatype), op = LIBXSMM_GETENUM_OUT(d.datatype);
  if (LIBXSMM_GEMM_PRECISION_I16 == ip) return LIBXSMM_GEMM_PRECISION_I32 == op ? 1 : (LIBXSMM_GEMM_PRECISION_F32 == op ? 2 : 0);
  if (LIBXSMM_GEMM_PRECISION_BF16 == ip) return LIBXSMM_GEMM_PRECISION_F32 == op ? 3 : (LIBXSMM_GEMM_PRECISION_BF16 == op ? 4 : 0);
  return 0;
}

bool desc_buildable(const libxsmm_gemm_descriptor& d)
{
  const int ip = LIBXSMM_GETENUM_INP(d.datatype), op = LIBXSMM_GETENUM_OUT(d.datatype);
  if (0 != lowp_kind(d)) { // constraints of the reference's generator (src/generator_gemm.c:121-147,236-243)
    if (0 != (d.k % 2) || 0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) return false;
    if (0 != (d.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE) && LIBXSMM_GEMM_PRECISION_BF16 != ip) return false; // batch-reduce: bf16 inputs only (src/libxsmm_main.c:2290-2315)
    if (LIBXSMM_GEMM_PRECISION_BF16 == op && 0 != (d.m % 16)) return false;
  }
  else if (!((LIBXSMM_GEMM_PRECISION_F64 == ip && LIBXSMM_GEMM_PRECISION_F64 == op) ||
             (LIBXSMM_GEMM_PRECISION_F32 == ip && LIBXSMM_GEMM_PRECISION_F32 == op))) return false;
  if (0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_A)) return false;
  if (0 == d.m || 0 == d.n || 0 == d.k) return false;
  if (d.lda < d.m) return false;                                                   // LIBXSMM_ERR_LDA
  if (0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B)) { if (d.ldb < d.n) return false; } // LIBXSMM_ERR_LDB_TRANS
  else if (d.ldb < d.k) return false;                                              // LIBXSMM_ERR_LDB
  if (d.ldc < d.m) return false;                                                   // LIBXSMM_ERR_LDC
  return true;
}

Key make_key(const libxsmm_gemm_descriptor& d, int kclass)
{
  Key key; memset(&key, 0, sizeof(key));
  memcpy(key.bytes, &d, sizeof(d));
  key.bytes[sizeof(d)] = (unsigned char)kclass; // reference: libxsmm_descriptor.kind follows the payload
  return key;
}

} // namespace

LIBXSMM_API libxsmm_xmmfunction libxsmm_xmmdispatch(const libxsmm_gemm_descriptor* descriptor)
{
  libxsmm_xmmfunction result; result.xmm = nullptr;
  if (nullptr == descriptor) return result; // quietly accept NULL (src/libxsmm_main.c:2158-2160)
  libxsmm_init();
  if (LIBXSMM_TARGET_ARCH_GENERIC == g_target_archid.load()) return result; // LIBXSMM_TARGET=generic: JIT disabled
  libxsmm_gemm_descriptor d = *descriptor;
  if (0 != (0x80 & d.prefetch)) d.prefetch = (unsigned char)g_auto_prefetch.load(); // "sign bit" => auto (:2146)
  const int kclass = (0 != lowp_kind(d)) ? KC_LOWP : ((0 != (d.flags & LIBXSMM_GEMM_FLAG_BATCH_REDUCE)) ? KC_REDUCE : KC_DENSE);
  const Key key = make_key(d, kclass);
  Registry& r = registry();
  const int p = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(d.datatype)) ? 0 : 1, b = bucket(d.m, d.n, d.k);
  {
    std::shared_lock<std::shared_mutex> guard(r.lock);
    auto it = r.by_desc.find(key);
    if (it != r.by_desc.end()) { result.xmm = reinterpret_cast<void (*)(const void*, const void*, void*, ...)>(it->second->thunk); }
  }
  if (nullptr != result.xmm) { __atomic_add_fetch(&r.ntry[p][b], 1, __ATOMIC_RELAXED); return result; }
  if (!desc_buildable(d)) return result;
  std::unique_lock<std::shared_mutex> guard(r.lock);
  ++r.ntry[p][b];
  auto it = r.by_desc.find(key);
  if (it == r.by_desc.end()) {
    Kernel* k = new Kernel();
    k->desc = d; k->kclass = kclass; k->registered = true;
    k->thunk = make_thunk(k);
    if (nullptr == k->thunk) { delete k; return result; } // registry/thunk capacity exhausted
    r.by_desc.emplace(key, k);
    r.by_thunk.emplace(k->thunk, k);
    ++r.njit[p][b];
    it = r.by_desc.find(key);
  }
  result.xmm = reinterpret_cast<void (*)(const void*, const void*, void*, ...)>(it->second->thunk);
  return result;
}

#define XSMM_DISPATCH_BODY(INIT, ALPHA_T, EXTRA_FLAGS, MEMBER)                                               \
  const int gemm_flags = (nullptr == flags ? LIBXSMM_FLAGS : *flags) | (EXTRA_FLAGS);                        \
  libxsmm_descriptor_blob blob;                                                                              \
  const libxsmm_gemm_descriptor* const desc = INIT(&blob, m, n, k,                                           \
    nullptr != lda ? *lda : (0 == (LIBXSMM_GEMM_FLAG_TRANS_A & gemm_flags) ? m : k),                         \
    nullptr != ldb ? *ldb : (0 == (LIBXSMM_GEMM_FLAG_TRANS_B & gemm_flags) ? k : n),                         \
    nullptr != ldc ? *ldc : m, nullptr != alpha ? *alpha : (ALPHA_T)LIBXSMM_ALPHA,                           \
    nullptr != beta ? *beta : (ALPHA_T)LIBXSMM_BETA, gemm_flags,                                             \
    nullptr == prefetch ? LIBXSMM_GEMM_PREFETCH_NONE : (LIBXSMM_PREFETCH_AUTO == *prefetch ? g_auto_prefetch.load() : *prefetch)); \
  return libxsmm_xmmdispatch(desc).MEMBER

LIBXSMM_API libxsmm_dmmfunction libxsmm_dmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const double* alpha, const double* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_dgemm_descriptor_init, double, 0, dmm); }

LIBXSMM_API libxsmm_smmfunction libxsmm_smmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_sgemm_descriptor_init, float, 0, smm); }

LIBXSMM_API libxsmm_dmmfunction_reducebatch libxsmm_dmmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const double* alpha, const double* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_dgemm_descriptor_init, double, LIBXSMM_GEMM_FLAG_BATCH_REDUCE, dmr); }

LIBXSMM_API libxsmm_smmfunction_reducebatch libxsmm_smmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_sgemm_descriptor_init, float, LIBXSMM_GEMM_FLAG_BATCH_REDUCE, smr); }

// low-precision dispatchers (src/libxsmm_main.c:2198-2259) and their descriptor initialisers (src/libxsmm_generator.c:93-185)
#define XSMM_LOWP_INIT(NAME, AT, IPREC, OPREC)                                                               \
LIBXSMM_API libxsmm_gemm_descriptor* NAME(libxsmm_descriptor_blob* blob,                                     \
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, \
  AT alpha, AT beta, int flags, int prefetch)                                                                \
{ return desc_init(blob, IPREC, OPREC, m, n, k, lda, ldb, ldc, (double)alpha, (double)beta, flags, prefetch); }
XSMM_LOWP_INIT(libxsmm_wigemm_descriptor_init, int, LIBXSMM_GEMM_PRECISION_I16, LIBXSMM_GEMM_PRECISION_I32)
XSMM_LOWP_INIT(libxsmm_wsgemm_descriptor_init, float, LIBXSMM_GEMM_PRECISION_I16, LIBXSMM_GEMM_PRECISION_F32)
XSMM_LOWP_INIT(libxsmm_bsgemm_descriptor_init, float, LIBXSMM_GEMM_PRECISION_BF16, LIBXSMM_GEMM_PRECISION_F32)
XSMM_LOWP_INIT(libxsmm_bgemm_descriptor_init, float, LIBXSMM_GEMM_PRECISION_BF16, LIBXSMM_GEMM_PRECISION_BF16)

LIBXSMM_API libxsmm_wimmfunction libxsmm_wimmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const int* alpha, const int* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_wigemm_descriptor_init, int, 0, wimm); }
LIBXSMM_API libxsmm_wsmmfunction libxsmm_wsmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_wsgemm_descriptor_init, float, 0, wsmm); }
LIBXSMM_API libxsmm_bsmmfunction libxsmm_bsmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_bsgemm_descriptor_init, float, 0, bsmm); }
LIBXSMM_API libxsmm_bmmfunction libxsmm_bmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_bgemm_descriptor_init, float, 0, bmm); }
// batch-reduce twins (src/libxsmm_main.c:2290-2315): kernel(a[], b[], c, &count)
LIBXSMM_API libxsmm_bsmmfunction_reducebatch libxsmm_bsmmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_bsgemm_descriptor_init, float, LIBXSMM_GEMM_FLAG_BATCH_REDUCE, bsmr); }
LIBXSMM_API libxsmm_bmmfunction_reducebatch libxsmm_bmmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch)
{ XSMM_DISPATCH_BODY(libxsmm_bgemm_descriptor_init, float, LIBXSMM_GEMM_FLAG_BATCH_REDUCE, bmr); }

// ---- caller-owned sparse kernels (src/libxsmm_main.c:2523-2582) ----------------------------------------------
namespace {

// unique-value scan of the reference's register kernel (src/generator_spgemm_csr_asparse_reg.c:125-143)
unsigned count_unique(const double* v, unsigned n)
{
  std::vector<double> u;
  for (unsigned i = 0; i < n; ++i) {
    bool hit = false;
    for (double x : u) if (!(x < v[i]) && !(x > v[i])) hit = true;
    if (!hit) u.push_back(v[i]);
  }
  return (unsigned)u.size();
}

void* create_csr_reg(const libxsmm_gemm_descriptor* descriptor, const unsigned* row_ptr, const unsigned* column_idx,
                     const double* dvalues, int typesize)
{
  if (nullptr == descriptor || nullptr == row_ptr || nullptr == column_idx || nullptr == dvalues) return nullptr;
  libxsmm_init();
  if (LIBXSMM_TARGET_ARCH_GENERIC == g_target_archid.load()) return nullptr;
  const libxsmm_gemm_descriptor& d = *descriptor;
  const unsigned nnz = row_ptr[d.m];
  // conditions under which the reference's generator fails: sparse-A form (lda == 0), row-major ld checks
  // (src/generator_spgemm.c:102-113), N == vector length (:187), at most 31 unique values (:146)
  if (0 != d.lda || 0 == d.ldb || 0 == d.ldc || d.ldb < d.n || d.ldc < d.n) return nullptr;
  if (d.n != (8 == typesize ? 8u : 16u)) return nullptr;
  if (0 == nnz || count_unique(dvalues, nnz) > 31) return nullptr;
  if (!device_ready()) { fail_no_device("libxsmm_create_?csr_reg"); return nullptr; }
  Kernel* k = new Kernel();
  k->desc = d; k->kclass = KC_CSR_REG; k->registered = false; k->nnz = nnz;
  k->d_rowptr = (unsigned*)dev_alloc(sizeof(unsigned) * (d.m + 1));
  k->d_colidx = (unsigned*)dev_alloc(sizeof(unsigned) * nnz);
  k->d_values = dev_alloc((size_t)typesize * nnz);
  bool ok = (nullptr != k->d_rowptr && nullptr != k->d_colidx && nullptr != k->d_values);
  if (ok) {
    ok = (0 == h2d(k->d_rowptr, row_ptr, sizeof(unsigned) * (d.m + 1))) && (0 == h2d(k->d_colidx, column_idx, sizeof(unsigned) * nnz));
    if (8 == typesize) ok = ok && (0 == h2d(k->d_values, dvalues, sizeof(double) * nnz));
    else {
      std::vector<float> fv(nnz);
      for (unsigned i = 0; i < nnz; ++i) fv[i] = (float)dvalues[i];
      ok = ok && (0 == h2d(k->d_values, fv.data(), sizeof(float) * nnz)) && (0 == stream_sync());
    }
    ok = ok && (0 == stream_sync());
  }
  if (ok) k->thunk = make_thunk(k);
  if (!ok || nullptr == k->thunk) { dev_free(k->d_rowptr); dev_free(k->d_colidx); dev_free(k->d_values); delete k; return nullptr; }
  Registry& r = registry();
  std::unique_lock<std::shared_mutex> guard(r.lock);
  r.by_thunk.emplace(k->thunk, k);
  return k->thunk;
}

} // namespace

namespace xsmm {
void* adopt_kernel(Kernel* k)
{
  if (nullptr == k) return nullptr;
  k->registered = false;
  k->thunk = make_thunk(k);
  if (nullptr == k->thunk) return nullptr;
  Registry& r = registry();
  std::unique_lock<std::shared_mutex> guard(r.lock);
  r.by_thunk.emplace(k->thunk, k);
  return k->thunk;
}
}

LIBXSMM_API libxsmm_dmmfunction libxsmm_create_dcsr_reg(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* row_ptr, const unsigned int* column_idx, const double* values)
{
  return reinterpret_cast<libxsmm_dmmfunction>(create_csr_reg(descriptor, row_ptr, column_idx, values, 8));
}

LIBXSMM_API libxsmm_smmfunction libxsmm_create_scsr_reg(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* row_ptr, const unsigned int* column_idx, const float* values)
{ // values are widened to double for the de-duplication and narrowed again (src/libxsmm_main.c:2557-2563)
  if (nullptr == descriptor || nullptr == row_ptr || nullptr == values) return nullptr;
  const unsigned n = row_ptr[descriptor->m];
  std::vector<double> dv(n);
  for (unsigned i = 0; i < n; ++i) dv[i] = (double)values[i];
  return reinterpret_cast<libxsmm_smmfunction>(create_csr_reg(descriptor, row_ptr, column_idx, dv.data(), 4));
}

LIBXSMM_API void libxsmm_release_kernel(const void* jit_kernel)
{ // src/libxsmm_main.c:2585-2619: caller-owned kernels are freed; registered kernels stay (de-registration is
  // compiled out in the reference's default build, which only warns)
  if (nullptr == jit_kernel) return;
  static int error_once = 0;
  Registry& r = registry();
  Kernel* k = nullptr;
  {
    std::unique_lock<std::shared_mutex> guard(r.lock);
    auto it = r.by_thunk.find(jit_kernel);
    if (it != r.by_thunk.end()) {
      k = it->second;
      if (k->registered) k = nullptr, (void)0;
      else r.by_thunk.erase(it);
      if (nullptr == k) {
        if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM WARNING: attempt to unregister a JIT-kernel!\n");
        return;
      }
    }
  }
  if (nullptr == k) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: failed to release kernel!\n");
    return;
  }
  if (device_ready()) (void)stream_sync();
  dev_free(k->d_rowptr); dev_free(k->d_colidx); dev_free(k->d_values);
  if (nullptr != k->text) text_kernel_destroy(k->text);
  free_thunk(k->thunk);
  delete k;
}

// ---- introspection (src/libxsmm_main.c:1926-2130) ---------------------------------------------------------------
LIBXSMM_API int libxsmm_get_kernel_kind(const void* kernel, libxsmm_kernel_kind* kind)
{
  if (nullptr == kernel || nullptr == kind) return EXIT_FAILURE;
  const Kernel* const k = kernel_from_pointer(kernel);
  *kind = (nullptr != k ? LIBXSMM_KERNEL_KIND_MATMUL : LIBXSMM_KERNEL_KIND_INVALID);
  return nullptr != k ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_get_mmkernel_info(libxsmm_xmmfunction kernel, libxsmm_mmkernel_info* info, size_t* code_size)
{
  const Kernel* const k = kernel_from_pointer(reinterpret_cast<const void*>(kernel.xmm));
  if (nullptr == k || (nullptr == info && nullptr == code_size)) {
    static int error_once = 0;
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: invalid argument!\n");
    return EXIT_FAILURE;
  }
  if (nullptr != info) {
    info->iprecision = (libxsmm_gemm_precision)LIBXSMM_GETENUM_INP(k->desc.datatype);
    info->oprecision = (libxsmm_gemm_precision)LIBXSMM_GETENUM_OUT(k->desc.datatype);
    info->prefetch = (libxsmm_gemm_prefetch_type)k->desc.prefetch;
    info->flags = k->desc.flags;
    info->lda = k->desc.lda; info->ldb = k->desc.ldb; info->ldc = k->desc.ldc;
    info->m = k->desc.m; info->n = k->desc.n; info->k = k->desc.k;
  }
  if (nullptr != code_size) *code_size = THUNK_SIZE;
  return EXIT_SUCCESS;
}

LIBXSMM_API int libxsmm_get_registry_info(libxsmm_registry_info* info)
{
  if (nullptr == info) return EXIT_FAILURE;
  Registry& r = registry();
  std::shared_lock<std::shared_mutex> guard(r.lock);
  info->capacity = 131072; // LIBXSMM_CAPACITY_REGISTRY of the reference; this registry grows on demand
  info->size = r.by_desc.size();
  info->nbytes = r.by_desc.size() * (THUNK_SIZE + sizeof(Kernel));
  info->nstatic = 0;
  info->ncache = 0;
  return EXIT_SUCCESS;
}
