// xsmm_defer.cpp -- per-call kernel invocations on device-resident operands without a launch per call (opt-in).
//
// The reference's canonical caller loops over its products and calls the dispatched kernel once per product
// (samples/smm/specialized.cpp:172-190, samples/cp2k/cp2k.cpp:341-346); its deferral concept is the explicit
// libxsmm_mmbatch_begin/end bracket (src/libxsmm_ext_gemm.c:1016-1135). On the GPU a launch per product costs 6-8 us, a
// hundred times the product itself.
//
// DEFAULT: every call is a launch of its own on the calling thread's stream -- stream order is call order, whatever the
// caller queues on that stream between two calls (its own kernels, hipMemcpyAsync, torch operations) is ordered between
// them. That is the only rule that is safe for a caller the library knows nothing about: HIP offers no way to see
// whether a stream has received foreign work since the library's last submission.
//
// OPT-IN (libxsmm_amd_defer_begin/end bracket on the calling thread, or LIBXSMM_AMD_DEFER=1 for the whole process): the
// caller promises that between two kernel calls it queues nothing of its own on the engine's stream that touches
// operands of the recorded calls -- or calls libxsmm_amd_flush() first. Consecutive calls of one kernel on device
// memory then form a *burst*:
//   * the first call of a burst queues two launches on the caller's stream: a gate (one lane that waits until the burst
//     is sealed) and the batch kernel behind it, which takes its item count from the gate and its operand pointers from a
//     ring in pinned host memory;
//   * every further call only appends {a, b, c} to that ring (a hundred nanoseconds, no driver call) -- it RUNS AT THE
//     STREAM POSITION OF THE BURST'S FIRST CALL, which is why foreign work in between needs the flush;
//   * the burst is sealed by whatever comes first: libxsmm_amd_flush / libxsmm_amd_defer_end, another entry point of the
//     library on this thread (it asks for the stream: device()), a call that must not run beside the recorded ones (other
//     kernel, operands that overlap a C of the burst, a C that repeats but not consecutively, a full ring), or a helper
//     thread once the calls have stopped coming for a few microseconds.
// Everything the burst does is already queued on the stream when the first call returns, so whatever the caller queues
// or waits for AFTER the last call of the burst -- hipMemcpy, hipStreamSynchronize, hipDeviceSynchronize -- is ordered
// behind it like behind any asynchronous call. Consecutive calls with the same C are a run (summed in call order by one
// unit of the batch kernel: the sequential chain, bit for bit).
// Never deferred: operands the CPU addresses (results must be there on return), calls while a stream is being captured.
#include "xsmm_internal.hpp"

#include <hip/hip_runtime_api.h>
#include <sys/prctl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace xsmm {

thread_local bool tl_defer_open = false;

namespace {

constexpr unsigned long long SEALED = 1ULL << 63;
constexpr int DEFER_CAP = 8192;    // calls per burst
constexpr int DEFER_SLOTS = 8;     // bursts of a thread that may be in flight on the GPU
constexpr long long IDLE_NS = 6000;  // the helper seals a burst that has not grown for this long (a loop in C issues a call every 0.05 us)

struct Entry { const void* a; const void* b; void* c; };
struct Range { uintptr_t lo = 0, hi = 0; bool has(uintptr_t p, size_t bytes) const { return lo <= p && p + bytes <= hi; } };

struct Slot {
  std::atomic<unsigned long long>* word = nullptr; // pinned: bit 63 sealed, low 32 bits number of calls
  Entry* entries = nullptr;                        // pinned [DEFER_CAP]
  unsigned long long* count = nullptr;             // device: {calls, gate gave up}
  hipEvent_t done = nullptr; bool pending = false; void* stream = nullptr;
};

struct Ring { // one per thread; outlives its thread (handed on), never freed: the helper may look at it at any time
  Slot slot[DEFER_SLOTS];
  std::atomic<int> open_slot{-1};          // slot of the open burst (-1: none): what the helper looks at
  std::atomic<long long> last_ns{0};       // time of the last call
  // caller-side state of the open burst
  int cur = 0, mine = -1;                  // next slot; slot of the open burst as the owner knows it
  Kernel* kernel = nullptr; void* stream = nullptr; int ncalls = 0;
  const void* last_c = nullptr;
  uintptr_t c_lo = 0, c_hi = 0, a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0; // address ranges written / read by the burst
  Range known[3];                                       // device allocations the operands were found in (valid within the burst)
  // a burst of per-panel operator calls (defer_panels): panel i of the burst is at b0 / c0 + i * step
  const void* panel_handle = nullptr; uintptr_t b0 = 0, c0 = 0; size_t step = 0; int max_panels = 0;
  bool ok = false;
};

// (never destroyed: the helper thread waits on them for as long as the process lives)
std::mutex& g_rings_lock = *new std::mutex;
std::vector<Ring*>& g_rings = *new std::vector<Ring*>; std::vector<Ring*>& g_rings_idle = *new std::vector<Ring*>;
std::condition_variable& g_helper_wake = *new std::condition_variable;
std::atomic<int> g_open_bursts{0};
bool g_helper_started = false;

long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void helper_loop()
{
  // (sleeps of a few microseconds: the default timer slack of a thread, 50 us, would add that much to the latency of a caller who
  // issues one call and waits for the device)
  (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
  for (;;) {
    {
      std::unique_lock<std::mutex> guard(g_rings_lock);
      g_helper_wake.wait(guard, []() { return 0 < g_open_bursts.load(std::memory_order_acquire); });
    }
    // bursts are open somewhere: look at them every few microseconds until none is left
    while (0 < g_open_bursts.load(std::memory_order_acquire)) {
      const long long t = now_ns();
      {
        std::lock_guard<std::mutex> guard(g_rings_lock);
        for (Ring* r : g_rings) {
          const int s = r->open_slot.load(std::memory_order_acquire);
          if (0 <= s && t - r->last_ns.load(std::memory_order_relaxed) > IDLE_NS) {
            std::atomic<unsigned long long>& w = *r->slot[s].word;
            unsigned long long v = w.load(std::memory_order_relaxed);
            while (0 == (v & SEALED) && !w.compare_exchange_weak(v, v | SEALED, std::memory_order_release, std::memory_order_relaxed)) {}
            int expect = s; // (the owner finds the bit at its next call, or in defer_flush)
            if (r->open_slot.compare_exchange_strong(expect, -1, std::memory_order_acq_rel)) g_open_bursts.fetch_sub(1, std::memory_order_acq_rel);
          }
        }
      }
      std::this_thread::sleep_for(std::chrono::microseconds(2));
    }
  }
}

struct RingHolder {
  Ring* ring = nullptr;
  ~RingHolder() {
    if (nullptr == ring) return;
    if (tl_defer_open) defer_flush();
    std::lock_guard<std::mutex> guard(g_rings_lock);
    g_rings_idle.push_back(ring);
  }
};
thread_local RingHolder tl_ring;

Ring* my_ring()
{
  if (nullptr != tl_ring.ring) return tl_ring.ring->ok ? tl_ring.ring : nullptr;
  {
    std::lock_guard<std::mutex> guard(g_rings_lock);
    if (!g_rings_idle.empty()) { tl_ring.ring = g_rings_idle.back(); g_rings_idle.pop_back(); return tl_ring.ring->ok ? tl_ring.ring : nullptr; }
  }
  Ring* const r = new Ring();
  bool ok = true;
  for (int i = 0; i < DEFER_SLOTS && ok; ++i) {
    void* w = nullptr; void* e = nullptr; void* c = nullptr;
    ok = hipSuccess == hipHostMalloc(&w, 64, hipHostMallocDefault) && hipSuccess == hipHostMalloc(&e, sizeof(Entry) * DEFER_CAP, hipHostMallocDefault)
      && hipSuccess == hipMalloc(&c, 2 * sizeof(unsigned long long)) && hipSuccess == hipEventCreateWithFlags(&r->slot[i].done, hipEventDisableTiming);
    if (!ok) { (void)hipGetLastError(); break; }
    r->slot[i].word = new (w) std::atomic<unsigned long long>(0);
    r->slot[i].entries = static_cast<Entry*>(e); r->slot[i].count = static_cast<unsigned long long*>(c);
  }
  r->ok = ok;
  tl_ring.ring = r;
  {
    std::lock_guard<std::mutex> guard(g_rings_lock);
    g_rings.push_back(r);
    if (ok && !g_helper_started) {
      g_helper_started = true; std::thread(helper_loop).detach();
      (void)atexit([]() { // no gate is left waiting when the process goes
        std::lock_guard<std::mutex> guard2(g_rings_lock);
        for (Ring* q : g_rings) for (int i = 0; i < DEFER_SLOTS; ++i) if (nullptr != q->slot[i].word) q->slot[i].word->fetch_or(SEALED, std::memory_order_release);
      });
    }
  }
  return ok ? r : nullptr;
}

// a stream that is being captured takes no part in bursts (a call appended to a burst queued before the capture began would run
// outside the graph)
bool capturing_now(void* stream)
{
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipSuccess != hipStreamIsCapturing((hipStream_t)stream, &st)) { (void)hipGetLastError(); return true; }
  return hipStreamCaptureStatusNone != st;
}

// Off unless the caller opted in: the bracket of the calling thread (libxsmm_amd_defer_begin/end) or LIBXSMM_AMD_DEFER=1
thread_local int tl_defer_bracket = 0;
bool defer_enabled()
{
  static const int env = []() { const char* e = getenv("LIBXSMM_AMD_DEFER"); return (nullptr == e || 0 == *e) ? 0 : atoi(e); }();
  return 0 != env || 0 < tl_defer_bracket;
}

// the device allocation p lies in (empty: not pure device memory)
Range device_range(const void* p)
{
  Range r;
  if (1 != pointer_kind(p)) return r; // host memory, or memory the CPU addresses as well (results are expected on return)
  hipDeviceptr_t base = nullptr; size_t size = 0;
  if (hipSuccess != hipMemGetAddressRange(&base, &size, const_cast<void*>(p))) { (void)hipGetLastError(); return r; }
  r.lo = reinterpret_cast<uintptr_t>(base); r.hi = r.lo + size;
  return r;
}

void close_burst(Ring& r)
{ // caller side: the burst takes no more calls
  if (0 <= r.mine) {
    std::atomic<unsigned long long>& w = *r.slot[r.mine].word;
    unsigned long long v = w.load(std::memory_order_relaxed);
    while (0 == (v & SEALED) && !w.compare_exchange_weak(v, v | SEALED, std::memory_order_release, std::memory_order_relaxed)) {}
    int expect = r.mine; // (whoever takes the burst off the helper's list lowers the count)
    if (r.open_slot.compare_exchange_strong(expect, -1, std::memory_order_acq_rel)) g_open_bursts.fetch_sub(1, std::memory_order_acq_rel);
  }
  tl_defer_open = false;
  r.mine = -1; r.kernel = nullptr; r.ncalls = 0; r.last_c = nullptr; r.panel_handle = nullptr;
}

// queue gate + batch kernel of a new burst on the caller's stream
bool open_burst(Ring& r, Kernel* k)
{
  Device& dev = device_raw();
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipSuccess != hipStreamIsCapturing((hipStream_t)dev.stream, &capturing)) { (void)hipGetLastError(); return false; }
  if (hipStreamCaptureStatusNone != capturing) return false; // a captured gate would wait for a seal that a replay never gets
  const int s = r.cur;
  Slot& sl = r.slot[s];
  if (sl.pending) { (void)hipEventSynchronize(sl.done); sl.pending = false; } // the slot's previous burst (DEFER_SLOTS bursts ago)
  sl.word->store(0, std::memory_order_release);
  SmmBatch b; memset(&b, 0, sizeof(b));
  const libxsmm_gemm_descriptor& d = k->desc;
  b.typesize = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(d.datatype)) ? 8 : 4;
  b.m = (int)d.m; b.n = (int)d.n; b.k = (int)d.k; b.lda = (int)d.lda; b.ldb = (int)d.ldb; b.ldc = (int)d.ldc;
  b.flags = d.flags & (LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_BETA_0);
  b.alpha = 1.0; b.beta = (0 != (d.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? 0.0 : 1.0;
  b.mode = ADDR_POINTER; // the ring is an array of {a, b, c}: three pointer arrays with a stride of one entry
  b.a = &sl.entries[0].a; b.b = &sl.entries[0].b; b.c = &sl.entries[0].c; b.sa = b.sb = b.sc = (long long)sizeof(Entry);
  b.batch = DEFER_CAP; b.batch_ptr = sl.count;
  b.sync = (0 != (b.flags & LIBXSMM_GEMM_FLAG_BETA_0)) ? SYNC_NONE : SYNC_RUNS; // consecutive calls with one C: a run, in call order
  if (0 != launch_defer_gate(reinterpret_cast<unsigned long long*>(sl.word), sl.count, dev.stream)) return false;
  const char* name = "";
  const int e = launch_smm_generic(b, dev.stream, &name);
  if (0 != e) { // (the gate is queued already: let it through with nothing recorded)
    sl.word->store(SEALED, std::memory_order_release);
    fprintf(stderr, "LIBXSMM-AMD ERROR: kernel launch failed (%s, hip error %d)\n", name, e);
    return false;
  }
  note_launch("smm_deferred_calls");
  if (hipSuccess == hipEventRecord(sl.done, (hipStream_t)dev.stream)) sl.pending = true; else (void)hipGetLastError();
  sl.stream = dev.stream;
  r.cur = (s + 1) % DEFER_SLOTS;
  r.kernel = k; r.stream = dev.stream; r.ncalls = 0; r.last_c = nullptr; r.mine = s;
  r.c_lo = r.a_lo = r.b_lo = ~(uintptr_t)0; r.c_hi = r.a_hi = r.b_hi = 0;
  r.last_ns.store(now_ns(), std::memory_order_relaxed);
  r.open_slot.store(s, std::memory_order_release);
  tl_defer_open = true;
  if (0 == g_open_bursts.fetch_add(1, std::memory_order_acq_rel)) { std::lock_guard<std::mutex> guard(g_rings_lock); g_helper_wake.notify_one(); }
  return true;
}

} // namespace

bool defer_bracket_open() { return 0 < tl_defer_bracket; }

void defer_flush()
{
  if (tl_spmdm_open) spmdm_flush_record();
  if (!tl_defer_open || nullptr == tl_ring.ring) { tl_defer_open = false; return; }
  close_burst(*tl_ring.ring);
}

bool defer_call(Kernel* k, const void* a, const void* b, void* c)
{
  if (tl_spmdm_open) spmdm_flush_record(); // (recorded spmdm block calls come first: a burst runs at the stream position of its first call)
  if (!defer_enabled() || nullptr == k || KC_DENSE != k->kclass) return false;
  // small products only (the reference's own JIT domain, LIBXSMM_MAX_MNK = 64^3): a large product is a launch -- or a library
  // GEMM -- of its own that spreads over the chip
  if ((long long)k->desc.m * k->desc.n * k->desc.k > 64LL * 64 * 64 || k->desc.m > 128 || k->desc.n > 128) return false;
  Ring* const rp = my_ring();
  if (nullptr == rp) return false;
  Ring& r = *rp;
  const libxsmm_gemm_descriptor& d = k->desc;
  const size_t ts = (LIBXSMM_GEMM_PRECISION_F64 == LIBXSMM_GETENUM_INP(d.datatype)) ? 8 : 4;
  const bool tb = 0 != (d.flags & LIBXSMM_GEMM_FLAG_TRANS_B);
  const size_t bytes_a = ((size_t)(d.k - 1) * d.lda + d.m) * ts, bytes_b = (tb ? ((size_t)(d.k - 1) * d.ldb + d.n) : ((size_t)(d.n - 1) * d.ldb + d.k)) * ts,
               bytes_c = ((size_t)(d.n - 1) * d.ldc + d.m) * ts;
  const uintptr_t pa = reinterpret_cast<uintptr_t>(a), pb = reinterpret_cast<uintptr_t>(b), pc = reinterpret_cast<uintptr_t>(c);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tl_defer_open) {
      bool fits = (r.kernel == k && r.stream == device_raw().stream && r.ncalls < DEFER_CAP && !capturing_now(r.stream));
      // operands inside device allocations already seen in this burst? (anything else is looked up, below, in a new burst)
      auto known = [&](uintptr_t p, size_t n) { return r.known[0].has(p, n) || r.known[1].has(p, n) || r.known[2].has(p, n); };
      fits = fits && known(pa, bytes_a) && known(pb, bytes_b) && known(pc, bytes_c);
      // the recorded calls run side by side: this one must not read what they write or write what they read or write --
      // except a C equal to the previous call's C, which continues its run. (Hulls of the addresses: a loop that walks its
      // arrays in one direction never touches them, anything else starts a new burst.)
      // (a kernel that overwrites C has no runs: the later call alone must remain)
      if (fits && (c != r.last_c || 0 != (d.flags & LIBXSMM_GEMM_FLAG_BETA_0))) fits = !(pc < r.c_hi && r.c_lo < pc + bytes_c);
      fits = fits && !(pa < pc + bytes_c && pc < pa + bytes_a) && !(pb < pc + bytes_c && pc < pb + bytes_b);
      fits = fits && !(pa < r.c_hi && r.c_lo < pa + bytes_a) && !(pb < r.c_hi && r.c_lo < pb + bytes_b) && !(pc < r.a_hi && r.a_lo < pc + bytes_c) && !(pc < r.b_hi && r.b_lo < pc + bytes_c);
      if (fits) {
        Slot& sl = r.slot[r.mine];
        sl.entries[r.ncalls] = Entry{ a, b, c };
        unsigned long long expect = (unsigned long long)r.ncalls;
        if (sl.word->compare_exchange_strong(expect, expect + 1, std::memory_order_release, std::memory_order_relaxed)) {
          ++r.ncalls;
          r.last_c = c;
          if (pc < r.c_lo) r.c_lo = pc;
          if (pc + bytes_c > r.c_hi) r.c_hi = pc + bytes_c;
          if (pa < r.a_lo) r.a_lo = pa;
          if (pa + bytes_a > r.a_hi) r.a_hi = pa + bytes_a;
          if (pb < r.b_lo) r.b_lo = pb;
          if (pb + bytes_b > r.b_hi) r.b_hi = pb + bytes_b;
          r.last_ns.store(now_ns(), std::memory_order_relaxed);
          return true;
        }
        // the helper has sealed the burst in the meantime
      }
      close_burst(r);
    }
    // a new burst: where do the operands live? (driver queries: once per burst)
    if (0 != attempt) break;
    r.known[0] = device_range(a);
    if (0 == r.known[0].hi) return false;
    r.known[1] = r.known[0].has(pb, bytes_b) ? r.known[0] : device_range(b);
    if (0 == r.known[1].hi) return false;
    r.known[2] = r.known[0].has(pc, bytes_c) ? r.known[0] : (r.known[1].has(pc, bytes_c) ? r.known[1] : device_range(c));
    if (0 == r.known[2].hi) return false;
    if (!r.known[0].has(pa, bytes_a) || !r.known[1].has(pb, bytes_b) || !r.known[2].has(pc, bytes_c)) return false;
    if (pa < pc + bytes_c && pc < pa + bytes_a) return false; // (a call whose own operands overlap: left to the ordinary path)
    if (pb < pc + bytes_c && pc < pb + bytes_b) return false;
    if (!open_burst(r, k)) return false;
  }
  return false;
}

// Per-panel calls of a fixed operator: libxsmm_?fsspmdm_execute(handle, B + i * N, C + i * N) for i = 0, 1, ... (the PyFR driver,
// samples/pyfr/pyfr_driver_asp_reg.c:300-308) -- a panel of N columns is a few kilobytes, a launch per panel is a hundred times
// its work. The same scheme as above: the first call queues the gate and the operator kernel (which reads the number of panels
// from device memory), the following calls only count up as long as they continue the walk along the rows.
bool defer_panels(const void* handle, JitKernel* jit, const void* B, void* C, int typesize, int M, int N, int K, long long ldb, long long ldc, int vec)
{
  if (tl_spmdm_open) spmdm_flush_record();
  if (!defer_enabled() || nullptr == handle || nullptr == jit || nullptr == B || nullptr == C) return false;
  Ring* const rp = my_ring();
  if (nullptr == rp) return false;
  Ring& r = *rp;
  const uintptr_t pb = reinterpret_cast<uintptr_t>(B), pc = reinterpret_cast<uintptr_t>(C);
  const size_t ts = (size_t)typesize, step = (size_t)N * ts;
  if (tl_defer_open) {
    if (r.panel_handle == handle && r.stream == device_raw().stream && r.ncalls < r.max_panels
      && pb == r.b0 + (size_t)r.ncalls * step && pc == r.c0 + (size_t)r.ncalls * step && !capturing_now(r.stream))
    {
      Slot& sl = r.slot[r.mine];
      unsigned long long expect = (unsigned long long)r.ncalls;
      if (sl.word->compare_exchange_strong(expect, expect + 1, std::memory_order_release, std::memory_order_relaxed)) {
        ++r.ncalls;
        r.last_ns.store(now_ns(), std::memory_order_relaxed);
        return true;
      }
    }
    close_burst(r);
  }
  // a new burst: both panels in device memory, aligned for the kernel's vector width; how far may the walk go?
  if (0 != ((pb | pc) & (ts * (size_t)vec - 1))) return false;
  const Range rb = device_range(B), rc = device_range(C);
  if (0 == rb.hi || 0 == rc.hi) return false;
  const size_t rows_b = (size_t)(K - 1) * (size_t)ldb * ts, rows_c = (size_t)(M - 1) * (size_t)ldc * ts; // offset of a panel's last row
  if (pb + rows_b + step > rb.hi || pc + rows_c + step > rc.hi) return false;
  long long maxp = DEFER_CAP;
  const long long by_b = (long long)((rb.hi - pb - rows_b) / step), by_c = (long long)((rc.hi - pc - rows_c) / step);
  if (maxp > by_b) maxp = by_b;
  if (maxp > by_c) maxp = by_c;
  const long long by_row = (ldb < ldc ? ldb : ldc) / N; // a walk stays inside one row of B and C
  if (maxp > by_row) maxp = by_row;
  if (maxp < 1) return false;
  { // the panels of B that are read and the panels of C that are written must not meet
    const uintptr_t be = pb + rows_b + (size_t)maxp * step, ce = pc + rows_c + (size_t)maxp * step;
    if (pb < ce && pc < be) return false;
  }
  Device& dev = device_raw();
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipSuccess != hipStreamIsCapturing((hipStream_t)dev.stream, &capturing)) { (void)hipGetLastError(); return false; }
  if (hipStreamCaptureStatusNone != capturing) return false;
  const int s = r.cur;
  Slot& sl = r.slot[s];
  if (sl.pending) { (void)hipEventSynchronize(sl.done); sl.pending = false; }
  sl.word->store(1, std::memory_order_release); // this call is the burst's first panel
  if (0 != launch_defer_gate(reinterpret_cast<unsigned long long*>(sl.word), sl.count, dev.stream)) return false;
  const int e = jit_launch_panels(jit, B, C, maxp * N, ldb, ldc, vec, dev.stream, sl.count, (long long)N);
  if (0 != e) {
    sl.word->store(SEALED, std::memory_order_release); // (the gate is queued already: let it through)
    fprintf(stderr, "LIBXSMM-AMD ERROR: kernel launch failed (fsspmdm operator, hip error %d)\n", e);
    return false;
  }
  note_launch(8 == typesize ? "fsspmdm_f64_jit_operator_deferred" : "fsspmdm_f32_jit_operator_deferred");
  if (hipSuccess == hipEventRecord(sl.done, (hipStream_t)dev.stream)) sl.pending = true; else (void)hipGetLastError();
  sl.stream = dev.stream;
  r.cur = (s + 1) % DEFER_SLOTS;
  r.kernel = nullptr; r.panel_handle = handle; r.stream = dev.stream; r.ncalls = 1; r.mine = s;
  r.b0 = pb; r.c0 = pc; r.step = step; r.max_panels = (int)maxp;
  r.last_ns.store(now_ns(), std::memory_order_relaxed);
  r.open_slot.store(s, std::memory_order_release);
  tl_defer_open = true;
  if (0 == g_open_bursts.fetch_add(1, std::memory_order_acq_rel)) { std::lock_guard<std::mutex> guard(g_rings_lock); g_helper_wake.notify_one(); }
  return true;
}

} // namespace xsmm

using namespace xsmm;

LIBXSMM_API void libxsmm_amd_flush(void) { defer_flush(); }
LIBXSMM_API void libxsmm_amd_defer_begin(void) { ++tl_defer_bracket; }
LIBXSMM_API void libxsmm_amd_defer_end(void)
{
  if (0 < tl_defer_bracket) --tl_defer_bracket;
  defer_flush(); // whatever was recorded is complete now: the caller may queue its own work behind it
}
LIBXSMM_API int libxsmm_amd_defer_active(void) { return defer_enabled() ? 1 : 0; }
