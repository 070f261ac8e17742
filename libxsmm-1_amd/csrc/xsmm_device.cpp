// xsmm_device.cpp -- HIP device context of the engine: device probe, stream, pointer classification,
// scratch memory and staging copies. Host-only translation unit (HIP runtime API, no kernels).
#include "xsmm_internal.hpp"

#include <dlfcn.h>

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace xsmm {

namespace {
std::once_flag g_probe_once;
Device g_device;
std::atomic<unsigned long long> g_launches{0};
thread_local const char* tl_last_kernel = "";

struct Scratch { void* ptr = nullptr; size_t size = 0; void* stream = nullptr; bool used = false; };
thread_local Scratch tl_scratch[8];
}

// The engine's current stream is a per-thread setting (every entry point may be called from any thread, as in the
// reference; a thread that never calls libxsmm_amd_set_stream launches on the default stream).
// (Whoever asks for the stream is about to queue work or to wait for it: a burst of deferred per-call kernels that is still
// open on this thread is sealed first, so that everything stays in the order of the calls -- xsmm_defer.cpp)
Device& device_raw() { thread_local Device tl_device; tl_device.count = g_device.count; return tl_device; }
Device& device() { if (tl_defer_open || tl_spmdm_open) defer_flush(); return device_raw(); }

bool device_ready()
{
  std::call_once(g_probe_once, []() {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    g_device.count = (hipSuccess == e ? n : 0);
  });
  return 0 < g_device.count;
}

void fail_no_device(const char* what)
{ // the product has no CPU compute path: say so loudly, independent of the verbosity level
  fprintf(stderr, "LIBXSMM-AMD FATAL: %s requires a HIP device (gfx950) but none is usable; "
                  "there is no CPU fallback in this library.\n", what);
}

int pointer_kind(const void* p)
{ // one driver query: bit 0 = the GPU reaches it (device, managed, pinned host), bit 1 = the CPU addresses it as well (pinned, managed)
  if (nullptr == p || !device_ready()) return 0;
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipSuccess != hipPointerGetAttributes(&attr, p)) { (void)hipGetLastError(); return 0; } // plain host memory on older runtimes
  const bool reach = (hipMemoryTypeDevice == attr.type || hipMemoryTypeManaged == attr.type || hipMemoryTypeHost == attr.type || hipMemoryTypeArray == attr.type);
  const bool host = (hipMemoryTypeHost == attr.type || hipMemoryTypeManaged == attr.type);
  return (reach ? 1 : 0) | (host ? 2 : 0);
}

bool is_device_ptr(const void* p)
{
  if (nullptr == p || !device_ready()) return false;
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  const hipError_t e = hipPointerGetAttributes(&attr, p);
  if (hipSuccess != e) { (void)hipGetLastError(); return false; } // plain host memory on older runtimes
  return hipMemoryTypeDevice == attr.type || hipMemoryTypeManaged == attr.type || hipMemoryTypeHost == attr.type
      || hipMemoryTypeArray == attr.type;
}

bool is_host_visible(const void* p)
{ // pinned host memory (libxsmm_malloc, hipHostMalloc) and managed memory: the GPU works on it in place, the CPU may read it
  if (nullptr == p || !device_ready()) return false;
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipSuccess != hipPointerGetAttributes(&attr, p)) { (void)hipGetLastError(); return false; }
  return hipMemoryTypeHost == attr.type || hipMemoryTypeManaged == attr.type;
}

void settle(const void* p0, const void* p1, const void* p2)
{ // Work on device memory stays asynchronous (the caller needs a copy or a synchronisation to touch it anyway). Operands
  // in memory the CPU addresses directly (libxsmm_malloc / pinned / managed) must be done with when the call returns: an
  // unchanged CPU caller reads the result, or overwrites an input, next.
  if (is_host_visible(p0) || is_host_visible(p1) || is_host_visible(p2)) (void)stream_sync();
}

void* dev_alloc(size_t bytes)
{
  void* p = nullptr;
  if (!device_ready()) return nullptr;
  if (hipSuccess != hipMalloc(&p, 0 != bytes ? bytes : 1)) { (void)hipGetLastError(); return nullptr; }
  return p;
}

void dev_free(void* p) { if (nullptr != p) (void)hipFree(p); }

int h2d(void* dst, const void* src, size_t bytes)
{
  if (0 == bytes) return 0;
  return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)device().stream);
}

int d2h(void* dst, const void* src, size_t bytes)
{
  if (0 == bytes) return 0;
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)device().stream);
  if (hipSuccess == e) e = hipStreamSynchronize((hipStream_t)device().stream);
  return (int)e;
}

int stream_sync() { return (int)hipStreamSynchronize((hipStream_t)device().stream); }

void note_launch(const char* name)
{
  tl_last_kernel = (nullptr != name ? name : "");
  g_launches.fetch_add(1, std::memory_order_relaxed);
}

// ---- library GEMM for products far outside the SMM domain ---------------------------------------------------------------
// libxsmm_?gemm (and a BLAS caller relinked with --wrap) may be handed one large product. The reference passes those to
// the BLAS it is linked with (src/libxsmm_gemm.c: libxsmm_blas_?gemm); here rocBLAS plays that role: loaded on first use
// (no link-time dependency), one handle per calling thread, the engine's stream. -1: not available (the tiled generic
// kernel serves the call then).
namespace {
struct RocBlas {
  void* lib = nullptr;
  int (*create)(void**) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  int (*dgemm)(void*, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*, double*, int) = nullptr;
  int (*sgemm)(void*, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*, float*, int) = nullptr;
  bool ok = false;
};
RocBlas& rocblas()
{
  static RocBlas r;
  static std::once_flag once;
  std::call_once(once, []() {
    const char* const env = getenv("LIBXSMM_AMD_BLAS"); // 0: never use the library GEMM
    if (nullptr != env && 0 == atoi(env)) return;
    r.lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (nullptr == r.lib) r.lib = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (nullptr == r.lib) return;
    r.create = reinterpret_cast<decltype(r.create)>(dlsym(r.lib, "rocblas_create_handle"));
    r.set_stream = reinterpret_cast<decltype(r.set_stream)>(dlsym(r.lib, "rocblas_set_stream"));
    r.dgemm = reinterpret_cast<decltype(r.dgemm)>(dlsym(r.lib, "rocblas_dgemm"));
    r.sgemm = reinterpret_cast<decltype(r.sgemm)>(dlsym(r.lib, "rocblas_sgemm"));
    r.ok = (nullptr != r.create && nullptr != r.set_stream && nullptr != r.dgemm && nullptr != r.sgemm);
  });
  return r;
}
thread_local void* tl_rocblas_handle = nullptr;
}

int library_gemm(int typesize, int transa, int transb, int m, int n, int k, double alpha, const void* a, int lda,
                 const void* b, int ldb, double beta, void* c, int ldc)
{
  RocBlas& r = rocblas();
  if (!r.ok) return -1;
  if (nullptr == tl_rocblas_handle && 0 != r.create(&tl_rocblas_handle)) { tl_rocblas_handle = nullptr; return -1; }
  if (0 != r.set_stream(tl_rocblas_handle, (hipStream_t)device().stream)) return -1;
  const int ta = transa ? 112 : 111, tb = transb ? 112 : 111; // rocblas_operation_transpose / _none
  if (8 == typesize) return 0 == r.dgemm(tl_rocblas_handle, ta, tb, m, n, k, &alpha, static_cast<const double*>(a), lda, static_cast<const double*>(b), ldb, &beta, static_cast<double*>(c), ldc) ? 0 : 1;
  const float al = (float)alpha, be = (float)beta;
  return 0 == r.sgemm(tl_rocblas_handle, ta, tb, m, n, k, &al, static_cast<const float*>(a), lda, static_cast<const float*>(b), ldb, &be, static_cast<float*>(c), ldc) ? 0 : 1;
}

// Slots in device memory, one per batch call whose C ordering is inspected on the device: int[4] {equal pairs, decreasing
// pairs, ticket counter, -} followed by a pair of counts per block of the check kernel. A slot is written (check kernel)
// and read (compute kernels) by launches of one stream, in order; nothing has to be cleared between uses (the ticket
// counter wraps back to zero, see c_order_kernel), only once at allocation. Every thread owns a ring of slots; a slot is
// handed out again only after the launches that read it have completed (event recorded by flag_slot_commit) -- calls in
// flight on several streams, or more calls in flight than the ring is long, never share a slot.
namespace {
constexpr unsigned FLAG_RING = 256, FLAG_INTS = 4 + 2 * FLAG_SLOT_BLOCKS;
// A call's slots share ONE event (recorded by flag_slot_commit): an event record is a barrier packet of its own on the queue, and
// one per slot -- 27 for a grouped CP2K call -- left the GPU idle for 100 us behind every call (profiles/r3_cp2k_trace.txt). The
// ring of commit events is as long as the ring of slots and every commit takes at least one slot, so an event is recorded
// again only after every slot that referred to its previous recording has been handed out -- and waited for -- again.
struct FlagSlot { int commit = -1; void* stream = nullptr; int state = 0; /* 1: handed out, 2: committed */ };
struct FlagRing { int* mem = nullptr; FlagSlot slot[FLAG_RING]; hipEvent_t commits[FLAG_RING] = {}; unsigned next = 0, next_commit = 0; };
// rings outlive their threads: a thread that ends hands its ring to the next thread that needs one (no HIP call at thread exit)
std::mutex g_flag_rings_lock;
std::vector<FlagRing*> g_flag_rings_idle;
struct FlagRingHolder {
  FlagRing* ring = nullptr;
  ~FlagRingHolder() { if (nullptr != ring) { std::lock_guard<std::mutex> guard(g_flag_rings_lock); g_flag_rings_idle.push_back(ring); } }
};
thread_local FlagRingHolder tl_flags;
FlagRing* flag_ring()
{
  if (nullptr != tl_flags.ring) return tl_flags.ring;
  {
    std::lock_guard<std::mutex> guard(g_flag_rings_lock);
    if (!g_flag_rings_idle.empty()) { tl_flags.ring = g_flag_rings_idle.back(); g_flag_rings_idle.pop_back(); return tl_flags.ring; }
  }
  // a new ring: cleared once, and the clearing is complete before the first slot is handed out (a check kernel on another
  // stream must never run ahead of it). Cleared with a synchronous memset that involves no stream of the caller's: the first
  // batch call of a thread may happen while its stream is being captured (nothing may be queued on, or waited for, there).
  void* p = nullptr;
  const size_t bytes = (size_t)FLAG_RING * FLAG_INTS * sizeof(int);
  if (hipSuccess != hipMalloc(&p, bytes)) { (void)hipGetLastError(); return nullptr; }
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed; // (a blocking memset is a "potentially unsafe" call for captures of other threads in global mode)
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  const hipError_t e = hipMemset(p, 0, bytes);
  const hipError_t e2 = (hipSuccess == e ? hipDeviceSynchronize() : e);
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (hipSuccess != e || hipSuccess != e2) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }
  FlagRing* const r = new FlagRing();
  r->mem = static_cast<int*>(p);
  tl_flags.ring = r;
  return r;
}
}

int* flag_slot()
{
  FlagRing* const r = flag_ring();
  if (nullptr == r) return nullptr;
  const unsigned i = r->next++ % FLAG_RING;
  FlagSlot& f = r->slot[i];
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipSuccess != hipStreamIsCapturing((hipStream_t)device().stream, &capturing)) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
  if (hipStreamCaptureStatusNone == capturing) { // (inside a stream capture nothing may be waited for: the slot's last use lies a whole ring back)
    if (2 == f.state && 0 <= f.commit) { if (hipSuccess != hipEventSynchronize(r->commits[f.commit])) (void)hipGetLastError(); } // (an event recorded inside a capture cannot be waited for: treated as done)
    else if (1 == f.state) (void)hipStreamSynchronize((hipStream_t)f.stream); // handed out but never committed (an error path)
  }
  f.stream = device().stream; f.state = 1; f.commit = -1;
  return r->mem + (size_t)FLAG_INTS * i;
}

void flag_slot_commit()
{
  FlagRing* const r = tl_flags.ring;
  if (nullptr == r) return;
  // one event per stream that holds handed-out slots (one stream, unless the caller switched streams inside a call)
  for (unsigned i = 0; i < FLAG_RING; ++i) {
    if (1 != r->slot[i].state) continue;
    void* const stream = r->slot[i].stream;
    const int c = (int)(r->next_commit++ % FLAG_RING);
    bool ok = (nullptr != r->commits[c]) || hipSuccess == hipEventCreateWithFlags(&r->commits[c], hipEventDisableTiming);
    if (ok) ok = (hipSuccess == hipEventRecord(r->commits[c], (hipStream_t)stream));
    if (!ok) { (void)hipGetLastError(); continue; } // (the slots stay "handed out": their next use waits for the stream)
    for (unsigned j = i; j < FLAG_RING; ++j) {
      FlagSlot& f = r->slot[j];
      if (1 == f.state && f.stream == stream) { f.state = 2; f.commit = c; }
    }
  }
}

int flag_slot_set(int* slot, int equal_pairs, int decreasing_pairs)
{ // a verdict fixed by the host, written in stream order like the check kernel's
  hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(slot), equal_pairs, 1, (hipStream_t)device().stream);
  if (hipSuccess == e) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(slot + 1), decreasing_pairs, 1, (hipStream_t)device().stream);
  return (int)e;
}

// Index arrays of a batch call that live in host memory (an unchanged caller's stride arrays): staged through a per-thread ring
// of pinned buffers and copied with the copy engine, asynchronously -- the call does not wait for the GPU. A ring entry is
// reused INDEX_RING calls later; by then the launch that read it (event recorded by index_upload_commit) is normally long done.
namespace {
struct IndexStage { void* host = nullptr; void* dev = nullptr; size_t size = 0; int commit = -1; void* stream = nullptr; int state = 0; /* 1: filled, 2: committed */ };
constexpr int INDEX_RING = 256; // (a grouped call stages three arrays per group plus its table)
thread_local IndexStage tl_index_ring[INDEX_RING];
thread_local unsigned tl_index_next = 0;
// (one event per commit, shared by the entries a call has filled: see the flag ring above)
thread_local hipEvent_t tl_index_commits[INDEX_RING] = {};
thread_local unsigned tl_index_next_commit = 0;
}

void* index_upload(const void* src, size_t bytes)
{
  IndexStage& e = tl_index_ring[tl_index_next++ % INDEX_RING];
  if (2 == e.state && 0 <= e.commit) (void)hipEventSynchronize(tl_index_commits[e.commit]);
  else if (1 == e.state) (void)hipStreamSynchronize((hipStream_t)e.stream); // filled but never committed (an error path)
  e.state = 0; e.commit = -1;
  if (e.size < bytes) {
    if (nullptr != e.host) (void)hipHostFree(e.host);
    if (nullptr != e.dev) (void)hipFree(e.dev);
    e.host = e.dev = nullptr; e.size = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipSuccess != hipHostMalloc(&e.host, want, hipHostMallocDefault) || hipSuccess != hipMalloc(&e.dev, want)) {
      (void)hipGetLastError();
      if (nullptr != e.host) (void)hipHostFree(e.host);
      e.host = nullptr; e.dev = nullptr;
      return nullptr;
    }
    e.size = want;
  }
  memcpy(e.host, src, bytes);
  e.stream = device().stream;
  if (hipSuccess != hipMemcpyAsync(e.dev, e.host, bytes, hipMemcpyHostToDevice, (hipStream_t)e.stream)) { (void)hipGetLastError(); return nullptr; }
  e.state = 1;
  return e.dev;
}

void index_upload_commit()
{ // the launches that read the staged arrays are queued: mark the point after which the entries may be overwritten
  for (int i = 0; i < INDEX_RING; ++i) {
    if (1 != tl_index_ring[i].state) continue;
    void* const stream = tl_index_ring[i].stream;
    const int c = (int)(tl_index_next_commit++ % INDEX_RING);
    bool ok = (nullptr != tl_index_commits[c]) || hipSuccess == hipEventCreateWithFlags(&tl_index_commits[c], hipEventDisableTiming);
    if (ok) ok = (hipSuccess == hipEventRecord(tl_index_commits[c], (hipStream_t)stream));
    if (!ok) { (void)hipGetLastError(); continue; }
    for (int j = i; j < INDEX_RING; ++j) {
      IndexStage& e = tl_index_ring[j];
      if (1 == e.state && e.stream == stream) { e.state = 2; e.commit = c; }
    }
  }
}

void* scratch(int slot, size_t bytes)
{
  Scratch& s = tl_scratch[slot & 7];
  // Launches are asynchronous: whatever was queued on the stream that used this buffer last may still be reading it.
  // (Scratch only serves the staging of host-resident operands -- the compatibility path -- so the wait costs nothing
  // on the device-resident path.)
  // (a new user on the same stream is ordered behind the old one by the stream itself; only another stream, or a buffer that has to
  // grow -- hipFree does wait, but on everything -- needs the wait)
  void* const now = device().stream;
  if (s.used && (s.stream != now || s.size < bytes)) (void)hipStreamSynchronize((hipStream_t)s.stream);
  s.stream = now; s.used = true;
  if (s.size < bytes) {
    if (nullptr != s.ptr) (void)hipFree(s.ptr);
    s.ptr = nullptr; s.size = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipSuccess == hipMalloc(&s.ptr, want)) s.size = want; else { (void)hipGetLastError(); s.ptr = nullptr; }
  }
  return s.ptr;
}

} // namespace xsmm

using namespace xsmm;

LIBXSMM_API int libxsmm_amd_device_count(void) { return device_ready() ? device().count : 0; }
LIBXSMM_API void libxsmm_amd_set_stream(void* hip_stream) { device().stream = hip_stream; }
LIBXSMM_API void* libxsmm_amd_get_stream(void) { return device().stream; }

LIBXSMM_API int libxsmm_amd_synchronize(void)
{
  if (!device_ready()) return EXIT_FAILURE;
  return 0 == stream_sync() ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API void* libxsmm_amd_device_malloc(size_t size) { return dev_alloc(size); }
LIBXSMM_API void libxsmm_amd_device_free(void* ptr) { dev_free(ptr); }

LIBXSMM_API int libxsmm_amd_memcpy_h2d(void* dst_device, const void* src_host, size_t size)
{
  if (!device_ready()) return EXIT_FAILURE;
  if (0 != h2d(dst_device, src_host, size)) return EXIT_FAILURE;
  return 0 == stream_sync() ? EXIT_SUCCESS : EXIT_FAILURE; // the host buffer may be reused right away
}

LIBXSMM_API int libxsmm_amd_memcpy_d2h(void* dst_host, const void* src_device, size_t size)
{
  if (!device_ready()) return EXIT_FAILURE;
  return 0 == d2h(dst_host, src_device, size) ? EXIT_SUCCESS : EXIT_FAILURE;
}

LIBXSMM_API int libxsmm_amd_is_device_pointer(const void* ptr) { return is_device_ptr(ptr) ? 1 : 0; }
LIBXSMM_API const char* libxsmm_amd_last_kernel(void) { return tl_last_kernel; }
LIBXSMM_API unsigned long long libxsmm_amd_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
