// xsmm_blocked.cpp -- libxsmm_blocked_gemm_*: GEMM on block-major operands with C-block accumulation.
//
// Reference: src/libxsmm_blocked_gemm.c:47-568 and src/template/libxsmm_blocked_gemm*.tpl.c. Layouts
// (tpl :37-39): A[mb][kb][bk][bm], B[nb][kb][bn][bk], C[nb][mb][bn][bm]; source matrices are column-major.
// The reference splits (i,j,k) work items over threads, accumulates consecutive items of one C block in a
// thread-local buffer and adds it to C under a per-block lock (tpl :93-111,139-160). On the GPU every C block
// is owned by one work-group which walks its k blocks in order (the SYNC_RUNS batch mode): no locks, no
// thread-local copies, deterministic result. The block order argument therefore only permutes the launch.
#include "xsmm_internal.hpp"

#include <cstring>
#include <vector>

namespace xsmm { int launch_smm_generic(const SmmBatch& s, void* stream, const char** name);
                 int launch_smm_special(const SmmBatch& s, void* stream, const char** name); }
using namespace xsmm;

struct libxsmm_blocked_gemm_handle {
  libxsmm_gemm_precision iprec, oprec;
  libxsmm_blocked_gemm_order order;
  libxsmm_blasint m, n, k, bm, bn, bk;
  libxsmm_blasint b_m1, b_n1, b_k1, b_k2;
  libxsmm_blasint mb, nb, kb;
  int nthreads, typesize, flags;
  int* d_ia; int* d_ib; int* d_ic; long long nitems; // device index arrays: item (j,i,k) -> element offsets of its blocks
};

LIBXSMM_API libxsmm_blocked_gemm_handle* libxsmm_blocked_gemm_handle_create(/*unsigned*/int nthreads,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* bm, const libxsmm_blasint* bn, const libxsmm_blasint* bk,
  const libxsmm_blasint* b_m1, const libxsmm_blasint* b_n1, const libxsmm_blasint* b_k1, const libxsmm_blasint* b_k2,
  const void* alpha, const void* beta, const int* gemm_flags, const libxsmm_gemm_prefetch_type* prefetch,
  const libxsmm_blocked_gemm_order* order)
{
  (void)prefetch;
  static int error_once = 0;
  libxsmm_init();
  // block sizes: env overrides and clamping as in the reference (:55-58)
  const char* const env_m = getenv("LIBXSMM_BLOCKED_GEMM_M"); const char* const env_n = getenv("LIBXSMM_BLOCKED_GEMM_N");
  const char* const env_k = getenv("LIBXSMM_BLOCKED_GEMM_K");
  const libxsmm_blasint mm = LIBXSMM_MIN(nullptr == bm ? ((nullptr == env_m || 0 == *env_m) ? 32 : atoi(env_m)) : *bm, m);
  const libxsmm_blasint kk = LIBXSMM_MIN(nullptr == bk ? ((nullptr == env_k || 0 == *env_k) ? mm : atoi(env_k)) : *bk, k);
  const libxsmm_blasint nn = LIBXSMM_MIN(nullptr == bn ? ((nullptr == env_n || 0 == *env_n) ? kk : atoi(env_n)) : *bn, n);
  const libxsmm_blasint m1 = (nullptr != b_m1 ? *b_m1 : 1), n1 = (nullptr != b_n1 ? *b_n1 : 1), k1 = (nullptr != b_k1 ? *b_k1 : 1), k2 = (nullptr != b_k2 ? *b_k2 : 1);
  if (!(0 < m && 0 < n && 0 < k && 0 < mm && 0 < nn && 0 < kk && 0 < nthreads && 0 < m1 && 0 < n1 && 0 < k1 && 0 < k2)) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: invalid arguments for libxsmm_blocked_gemm_handle_create!\n");
    return nullptr;
  }
  if (!(0 == (m % mm) && 0 == (n % nn) && 0 == (k % kk) && 0 == (m % m1) && 0 == (n % n1) && 0 == (k % k1) &&
        0 == ((k / k1 / k2) % kk) && 0 == ((n / n1) % nn) && 0 == ((m / m1) % mm))) { // :65-67
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: BGEMM block-size is invalid!\n");
    return nullptr;
  }
  if (iprec != oprec || (LIBXSMM_GEMM_PRECISION_F64 != iprec && LIBXSMM_GEMM_PRECISION_F32 != iprec)) {
    // (the reference also has a 16-bit integer blocked GEMM, src/libxsmm_blocked_gemm.c:536-550; the copy and compute kernels
    // here move 4- and 8-byte elements only)
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: BGEMM precision is not supported!\n");
    return nullptr;
  }
  libxsmm_descriptor_blob blob;
  const libxsmm_gemm_descriptor* const desc = libxsmm_gemm_descriptor_init2(&blob, iprec, oprec, mm, nn, kk, mm, kk, mm,
    alpha, beta, nullptr == gemm_flags ? LIBXSMM_GEMM_FLAG_NONE : *gemm_flags, LIBXSMM_GEMM_PREFETCH_NONE);
  if (nullptr == desc || nullptr == libxsmm_xmmdispatch(desc).xmm) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: unsupported BGEMM kernel requested!\n");
    return nullptr;
  }
  if (!device_ready()) { fail_no_device("libxsmm_blocked_gemm_handle_create"); return nullptr; }
  libxsmm_blocked_gemm_handle* h = static_cast<libxsmm_blocked_gemm_handle*>(calloc(1, sizeof(*h)));
  if (nullptr == h) return nullptr;
  h->iprec = iprec; h->oprec = oprec; h->order = (nullptr == order ? LIBXSMM_BLOCKED_GEMM_ORDER_JIK : *order);
  h->m = m; h->n = n; h->k = k; h->bm = mm; h->bn = nn; h->bk = kk; h->b_m1 = m1; h->b_n1 = n1; h->b_k1 = k1; h->b_k2 = k2;
  h->mb = m / mm; h->nb = n / nn; h->kb = k / kk; h->nthreads = nthreads;
  h->typesize = (LIBXSMM_GEMM_PRECISION_F64 == iprec) ? 8 : 4;
  h->flags = (nullptr == gemm_flags ? 0 : (*gemm_flags & LIBXSMM_GEMM_FLAG_TRANS_B));
  // work list: for every C block (j,i) its kb products in ascending k => equal C offsets are consecutive
  h->nitems = (long long)h->mb * h->nb * h->kb;
  std::vector<int> ia((size_t)h->nitems), ib((size_t)h->nitems), ic((size_t)h->nitems);
  size_t w = 0;
  for (int j = 0; j < h->nb; ++j) for (int i = 0; i < h->mb; ++i) for (int kq = 0; kq < h->kb; ++kq, ++w) {
    ia[w] = (int)((((size_t)i * h->kb + kq) * h->bk) * h->bm);
    ib[w] = (int)((((size_t)j * h->kb + kq) * h->bn) * h->bk);
    ic[w] = (int)((((size_t)j * h->mb + i) * h->bn) * h->bm);
  }
  h->d_ia = static_cast<int*>(dev_alloc(sizeof(int) * ia.size())); h->d_ib = static_cast<int*>(dev_alloc(sizeof(int) * ib.size()));
  h->d_ic = static_cast<int*>(dev_alloc(sizeof(int) * ic.size()));
  bool ok = (nullptr != h->d_ia && nullptr != h->d_ib && nullptr != h->d_ic);
  ok = ok && 0 == h2d(h->d_ia, ia.data(), sizeof(int) * ia.size()) && 0 == h2d(h->d_ib, ib.data(), sizeof(int) * ib.size())
          && 0 == h2d(h->d_ic, ic.data(), sizeof(int) * ic.size()) && 0 == stream_sync();
  if (!ok) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: BGEMM handle allocation failed!\n");
    libxsmm_blocked_gemm_handle_destroy(h); return nullptr;
  }
  return h;
}

LIBXSMM_API void libxsmm_blocked_gemm_handle_destroy(const libxsmm_blocked_gemm_handle* handle)
{
  if (nullptr == handle) return;
  if (device_ready()) (void)stream_sync();
  dev_free(handle->d_ia); dev_free(handle->d_ib); dev_free(handle->d_ic);
  free(const_cast<libxsmm_blocked_gemm_handle*>(handle));
}

namespace {
int bgemm_copy(const libxsmm_blocked_gemm_handle* h, int which, const void* src, const libxsmm_blasint* ld, void* dst)
{
  static int error_once = 0;
  if (nullptr == h) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: BGEMM-handle cannot be NULL!\n");
    return EXIT_FAILURE;
  }
  if (nullptr == src || nullptr == dst) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_blocked_gemm_copy*"); return EXIT_FAILURE; }
  BgemmGeom g; g.typesize = h->typesize; g.m = h->m; g.n = h->n; g.k = h->k; g.bm = h->bm; g.bn = h->bn; g.bk = h->bk; g.mb = h->mb; g.nb = h->nb; g.kb = h->kb;
  const int rows = (1 == which ? h->k : h->m), cols = (0 == which ? h->k : h->n); // plain matrix: rows x cols, column-major
  const int ldv = (nullptr != ld ? *ld : rows);
  const size_t plain = ((size_t)(cols - 1) * ldv + rows) * h->typesize, blocked = (size_t)rows * cols * h->typesize;
  const bool out = (3 == which);
  const size_t src_bytes = out ? blocked : plain, dst_bytes = out ? plain : blocked;
  const void* ds = src; void* dd = dst;
  const bool src_host = !is_device_ptr(src), dst_host = !is_device_ptr(dst);
  if (src_host) { void* t = scratch(3, src_bytes); if (nullptr == t || 0 != h2d(t, src, src_bytes)) return EXIT_FAILURE; ds = t; }
  if (dst_host) { void* t = scratch(4, dst_bytes); if (nullptr == t) return EXIT_FAILURE; if (out && ldv != rows && 0 != h2d(t, dst, dst_bytes)) return EXIT_FAILURE; dd = t; }
  const int e = launch_bgemm_copy(g, which, ds, ldv, dd, device().stream); note_launch("bgemm_copy");
  if (0 != e) return EXIT_FAILURE;
  if (dst_host) return 0 == d2h(dst, dd, dst_bytes) ? EXIT_SUCCESS : EXIT_FAILURE;
  if (src_host) return 0 == stream_sync() ? EXIT_SUCCESS : EXIT_FAILURE;
  settle(src, dst);
  return EXIT_SUCCESS;
}
}

namespace {
// blocked -> blocked permutations (reference src/libxsmm_blocked_gemm.c:369-466; like there, `ld` is ignored)
int bgemm_permute(const libxsmm_blocked_gemm_handle* h, int which, const void* src, void* dst)
{
  static int error_once = 0;
  if (nullptr == h) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: BGEMM-handle cannot be NULL!\n");
    return EXIT_FAILURE;
  }
  if (nullptr == src || nullptr == dst) return EXIT_FAILURE;
  if (!device_ready()) { fail_no_device("libxsmm_blocked_gemm_convert/transpose"); return EXIT_FAILURE; }
  BgemmGeom g; g.typesize = h->typesize; g.m = h->m; g.n = h->n; g.k = h->k; g.bm = h->bm; g.bn = h->bn; g.bk = h->bk; g.mb = h->mb; g.nb = h->nb; g.kb = h->kb;
  const size_t bytes = (size_t)(4 == which ? (size_t)h->m * h->n : (size_t)h->k * h->n) * h->typesize;
  const void* ds = src; void* dd = dst;
  const bool src_host = !is_device_ptr(src), dst_host = !is_device_ptr(dst);
  if (src_host) { void* t = scratch(3, bytes); if (nullptr == t || 0 != h2d(t, src, bytes)) return EXIT_FAILURE; ds = t; }
  if (dst_host) { void* t = scratch(4, bytes); if (nullptr == t) return EXIT_FAILURE; dd = t; }
  const int e = launch_bgemm_copy(g, which, ds, 0, dd, device().stream); note_launch(4 == which ? "bgemm_convert_b_to_a" : "bgemm_transpose_b");
  if (0 != e) return EXIT_FAILURE;
  if (dst_host) return 0 == d2h(dst, dd, bytes) ? EXIT_SUCCESS : EXIT_FAILURE;
  if (src_host) return 0 == stream_sync() ? EXIT_SUCCESS : EXIT_FAILURE;
  settle(src, dst);
  return EXIT_SUCCESS;
}
}

LIBXSMM_API int libxsmm_blocked_gemm_convert_b_to_a(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst) { (void)ld; return bgemm_permute(handle, 4, src, dst); }
LIBXSMM_API int libxsmm_blocked_gemm_transpose_b(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst) { (void)ld; return bgemm_permute(handle, 5, src, dst); }
LIBXSMM_API int libxsmm_blocked_gemm_copyin_a(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst) { return bgemm_copy(handle, 0, src, ld, dst); }
LIBXSMM_API int libxsmm_blocked_gemm_copyin_b(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst) { return bgemm_copy(handle, 1, src, ld, dst); }
LIBXSMM_API int libxsmm_blocked_gemm_copyin_c(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst) { return bgemm_copy(handle, 2, src, ld, dst); }
LIBXSMM_API int libxsmm_blocked_gemm_copyout_c(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst) { return bgemm_copy(handle, 3, src, ld, dst); }

namespace {
void bgemm_run(const libxsmm_blocked_gemm_handle* h, const void* a, const void* b, void* c)
{
  if (nullptr == h || nullptr == a || nullptr == b || nullptr == c) return;
  if (!device_ready()) { fail_no_device("libxsmm_blocked_gemm"); return; }
  const int ts = h->typesize;
  const size_t ea = (size_t)h->m * h->k * ts, eb = (size_t)h->k * h->n * ts, ec = (size_t)h->m * h->n * ts;
  const void* da = a; const void* db = b; void* dc = c;
  const bool c_host = !is_device_ptr(c);
  if (!is_device_ptr(a)) { void* t = scratch(3, ea); if (nullptr == t || 0 != h2d(t, a, ea)) return; da = t; }
  if (!is_device_ptr(b)) { void* t = scratch(4, eb); if (nullptr == t || 0 != h2d(t, b, eb)) return; db = t; }
  if (c_host) { void* t = scratch(5, ec); if (nullptr == t || 0 != h2d(t, c, ec)) return; dc = t; }
  SmmBatch s; memset(&s, 0, sizeof(s));
  s.typesize = ts; s.m = h->bm; s.n = h->bn; s.k = h->bk; s.lda = h->bm; s.ldb = h->bk; s.ldc = h->bm;
  s.flags = 0; // C blocks always accumulate (tpl :101,149: real_c += l_out)
  s.mode = ADDR_INDEX; s.a = da; s.b = db; s.c = dc; s.ia = h->d_ia; s.ib = h->d_ib; s.ic = h->d_ic;
  s.index_base = 0; s.index_stride = (int)sizeof(int); s.batch = h->nitems; s.sync = SYNC_RUNS;
  s.use_mfma = libxsmm_amd_get_mfma(); s.alpha = 1; s.beta = 1;
  s.uniform_run = h->kb; // every C block's k blocks follow each other in the work list
  const char* name = "";
  int e = launch_smm_special(s, device().stream, &name);
  if (e < 0) {
    // The work list is a sequence of runs (all k blocks of a C block follow each other). The reference sums a run into a
    // thread-local block and adds that to C under a lock (tpl :95-103): the order of the partial sums is open, so the
    // shape-specialised run kernels may cut few long runs into segments; the verdict on the runs is
    // taken on the device like for any index batch.
    int* const d_flags = flag_slot();
    SmmBatch j = s;
    if (nullptr != d_flags && 0 == launch_c_order_check(j, d_flags, device().stream)) {
      j.sync = SYNC_DEVICE; j.devflags = d_flags; j.relaxed = 1; j.c_atomics = is_host_visible(dc) ? 0 : 1;
      j.uniform_run = h->kb; // every C block's k blocks follow each other in the work list
      if (smm_jit_eligible(j)) e = launch_smm_jit(j, device().stream, &name);
    }
    flag_slot_commit();
  }
  if (e < 0) e = launch_smm_generic(s, device().stream, &name);
  note_launch(name);
  if (0 != e) { fprintf(stderr, "LIBXSMM-AMD ERROR: kernel launch failed (%s, hip error %d)\n", name, e); return; }
  if (c_host) (void)d2h(c, dc, ec);
  else if (da != a || db != b) (void)stream_sync();
  else settle(a, b, c);
}
}

LIBXSMM_API void libxsmm_blocked_gemm_st(const libxsmm_blocked_gemm_handle* handle, const void* a, const void* b, void* c,
  /*unsigned*/int start_thread, /*unsigned*/int tid)
{ // the reference expects every thread of the team to call this (barrier at entry/exit, :517-519,560-562); the device grid
  // does the whole multiplication, so only the team's first thread launches and the others return immediately.
  static int error_once = 0;
  if (nullptr == handle || nullptr == a || nullptr == b || nullptr == c || start_thread > tid || 0 > tid) {
    if (0 != libxsmm_verbosity && once(&error_once)) fprintf(stderr, "LIBXSMM ERROR: invalid arguments for libxsmm_blocked_gemm!\n");
    return;
  }
  if (tid == start_thread) bgemm_run(handle, a, b, c);
}

LIBXSMM_APIEXT void libxsmm_blocked_gemm_omp(const libxsmm_blocked_gemm_handle* handle,
  const void* a, const void* b, void* c, /*unsigned*/int count)
{ // src/libxsmm_ext_blocked_gemm.c:47-73: `count` repetitions of the same multiplication
  for (int i = 0; i < count; ++i) bgemm_run(handle, a, b, c);
}
