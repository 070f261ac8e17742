// xsmm_internal.hpp -- shared declarations between the host runtime (.cpp, built with g++) and the
// device launchers (.hip, built with hipcc). Nothing here is part of the public C-ABI.
#ifndef XSMM_INTERNAL_HPP
#define XSMM_INTERNAL_HPP

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>

#include "../../include/libxsmm.h"

// ---------------------------------------------------------------------------------------------------
// The GEMM descriptor. Layout follows the reference's packed POD (src/libxsmm_main.h:157-168): it is the
// registry key and what libxsmm_get_mmkernel_info decodes, so callers that memcpy/hash it keep working.
// ---------------------------------------------------------------------------------------------------
#pragma pack(push, 1)
struct libxsmm_gemm_descriptor {
  unsigned char datatype;   // inp | out << 4  (include/libxsmm_typedefs.h:91-95)
  unsigned short flags;     // libxsmm_gemm_flags
  unsigned int m, n, k;
  unsigned int lda, ldb, ldc;
  unsigned char prefetch;
};
#pragma pack(pop)
static_assert(sizeof(libxsmm_gemm_descriptor) == 28, "descriptor must stay packed (28 bytes)");

namespace xsmm {

// ---- device launch descriptions ------------------------------------------------------------------
enum AddrMode : int {
  ADDR_STRIDED = 0,   // item i: base + i*stride (elements)
  ADDR_INDEX = 1,     // item i: base + (idx[i*index_stride bytes] - index_base) elements; NULL idx => shared
  ADDR_POINTER = 2    // item i: *(T**)((char*)ptrs + i*stride bytes)
};

enum SyncMode : int {
  SYNC_NONE = 0,      // every item owns its C (or beta == 0)
  SYNC_RUNS = 1,      // equal C only in consecutive runs: one work-group walks a whole run in batch order
  SYNC_ATOMIC = 2,    // arbitrary duplicates: product from zero, then atomic add into C
  // Decided on the device, without a host round trip: a check kernel leaves {#equal neighbours, #out-of-order repeats}
  // in SmmBatch::devflags and the compute kernels read them -- the call stays asynchronous (and graph-capturable).
  SYNC_DEVICE = 3           // pick NONE / RUNS / ATOMIC from the flags (generic kernel); run kernels: runs in batch order, or
                            // segments whose sums join C with atomics ([1] != 0, or few long runs under a relaxed order)
};

struct SmmBatch {
  int typesize;             // 8: f64, 4: f32
  int m, n, k, lda, ldb, ldc;
  int flags;                // LIBXSMM_GEMM_FLAG_TRANS_B | LIBXSMM_GEMM_FLAG_BETA_0 (others ignored)
  int mode;                 // AddrMode
  const void* a; const void* b; void* c;
  const int* ia; const int* ib; const int* ic;  // ADDR_INDEX (device arrays)
  int index_base, index_stride;                 // ADDR_INDEX: base and byte step through the index arrays
  long long sa, sb, sc;     // ADDR_STRIDED: element strides; ADDR_POINTER: byte distance between pointers
  long long batch;
  int sync;                 // SyncMode
  const int* devflags;      // SYNC_DEVICE*: device int[2] written by the check kernel earlier on the same stream
  int c_atomics;            // SYNC_DEVICE: != 0 if floating-point atomics reach C (device memory, not host memory the GPU maps)
  int lowp; float scf;      // low-precision kernels (kernels/smm_lowp.hip): 1 i16->i32, 2 i16->f32 (times scf), 3 bf16->f32, 4 bf16->bf16; 0: f32/f64
  int shared_across_calls;  // != 0: other tasks of the same libxsmm_mmbatch update the same C blocks concurrently (ntasks > 1): atomics
  int tasks;                // number of tasks of the libxsmm_mmbatch call this slice belongs to (0/1: the whole batch)
  long long uniform_run;    // > 0: the batch consists of runs of exactly this many consecutive items per C (blocked GEMM work lists)
  int jit_always;           // != 0: specialise with hiprtc whatever the batch size (batch-reduce kernels: short batches, called over and over)
  int relaxed;              // != 0: sums into a shared C may be formed in any order (the caller's reference path is multi-threaded)
  int use_mfma;             // policy bit (0: scalar FMA only)
  const unsigned long long* batch_ptr; // != NULL: the number of items is read from here by the kernel (deferred per-call kernels); `batch` is the capacity
  // general form used by the BLAS-like fallback (libxsmm_?gemm with alpha/beta/trans outside the SMM domain)
  double alpha, beta; int general;              // general != 0: C = alpha*op(A)*op(B) + beta*C, flags may hold TRANS_A
};

// returns hipError_t as int (0 == success); *name receives a static string naming the kernel variant
int launch_smm_batch(const SmmBatch& args, void* stream, const char** name);
int launch_smm_lowp(const SmmBatch& args, void* stream, const char** name);  // args.lowp != 0; independent C operands
int launch_smm_lowp_reduce(const SmmBatch& args, void* stream, const char** name); // bf16 batch-reduce: one C, pointer arrays of A and B

constexpr int FLAG_SLOT_BLOCKS = 512; // work-groups of the C ordering check (each leaves a pair of counts in the flag slot)
// detects how C operands alias across the batch: out[0] = number of i with c_i == c_{i-1},
// out[1] = number of i with c_i < c_{i-1}. d_out is a slot of flag_slot().
int launch_c_order_check(const SmmBatch& args, int* d_out, void* stream);
int launch_defer_gate(unsigned long long* word, unsigned long long* count_out, void* stream); // see xsmm_defer.cpp
int launch_smm_generic(const SmmBatch& s, void* stream, const char** name);
// deferred per-call kernels (xsmm_defer.cpp)
struct Kernel;
bool defer_call(Kernel* k, const void* a, const void* b, void* c);  // true: recorded (runs later, in stream order)
void defer_flush();
struct JitKernel;
// per-panel calls of a fixed operator (libxsmm_?fsspmdm_execute) that walk along the rows of B and C: recorded like per-product calls
bool defer_panels(const void* handle, JitKernel* jit, const void* B, void* C, int typesize, int M, int N, int K, long long ldb, long long ldc, int vec);                                                  // seal the calling thread's open burst
extern thread_local bool tl_defer_open;
extern thread_local bool tl_spmdm_open;   // spmdm block calls recorded inside a bracket (xsmm_sparse.cpp)
void spmdm_flush_record();                // launches them
bool defer_bracket_open();                // the calling thread is inside libxsmm_amd_defer_begin/end

// CSR "register" kernel family (fsspmdm sparse path, libxsmm_create_?csr_reg): row-major
// C[m*ldc+n] = (beta? C:0) + sum_p val[p]*B[col[p]*ldb+n], rows without nnz untouched.
struct CsrPanels {
  int typesize;
  int m, k;                  // operator shape
  int n;                     // panel width per item (multiple of the reference's chunk; any n >= 1 here)
  int ldb, ldc;
  int beta0;
  int skip_empty_rows;       // 1: reference's csr_reg quirk (rows without nnz are not written even when beta == 0)
  const unsigned* rowptr; const unsigned* colidx; const void* values; // device
  unsigned nnz;
  const void* b; void* c;    // device; panel i starts at column i*n
  long long batch;
};
int launch_csr_panels(const CsrPanels& args, void* stream, const char** name);

// run-time specialised operator kernels (xsmm_jit.cpp)
struct JitKernel;
std::string gen_csr_panels_source(int typesize, int M, int K, const unsigned* rowptr, const unsigned* colidx, const double* values,
                                  int beta0, int skip_empty_rows, int vec, const char* fname, bool burst_args = false);
JitKernel* jit_compile(const std::string& src, const char* fname, std::string* log);
JitKernel* jit_from_cache(const std::string& src, const char* fname); // only if the code-object cache on disk holds it
int jit_build_offline(const std::string& src, std::string* log);      // compile into the cache on disk (no device needed); 0: there
bool jit_async_enabled();                                             // LIBXSMM_AMD_JIT_ASYNC (default on)
void jit_async(std::function<void()> job);                            // run on the compiler thread
void jit_async_wait();
void jit_async_drain(); // drop queued compile jobs, wait for the running one                                                // until the compiler thread has nothing left to do
int jit_check_source(const std::string& src, std::string* log);
void jit_release(JitKernel* k);
int jit_launch_panels(JitKernel* k, const void* B, void* C, long long ncols, long long ldb, long long ldc, int vec, void* stream,
                      const unsigned long long* npanels = nullptr, long long panel = 0);
int jit_blocks_per_cu(JitKernel* k, int threads); // occupancy of a generated kernel (0: unknown)
int jit_launch_args(JitKernel* k, unsigned blocks, unsigned threads, void** args, void* stream);
int jit_launch_raw(JitKernel* k, unsigned blocks, unsigned threads, void* arg0, size_t arg0_size, void* arg1, void* stream);
// dense SMM kernels specialised per shape (xsmm_jit_smm.cpp)
enum { SMM_JIT_SCALAR = 1, SMM_JIT_RUNS = 2, SMM_JIT_WGRUNS = 4, SMM_JIT_HASWG = 8, SMM_JIT_BIG = 16, SMM_JIT_SPLIT = 32,
       SMM_JIT_MFMA = 64 /* matrix-core work-group kernel (kernels/smm_mfma_wg.inc) with the shape baked in; + 128: tight fp32 operands as 16-byte chunks */, SMM_JIT_MFMA_TIGHT = 128, SMM_JIT_MFMA_TIGHTC = 8192 /* fp32: C as a contiguous array through LDS */,
       SMM_JIT_MFMA_WAVE = 16384 /* matrix-core kernel with one wave per item (16x16x4 tiles) */,
       SMM_JIT_MFMA_WAVE2 = 32768 /* ... the columns of C in two halves against one image of A (fp64 56^3: the images of a whole item leave no room for four waves per CU) */,
       SMM_JIT_MFMA_RUNS = 65536 /* run form on the matrix cores: a wave per run, A fragments straight from memory, B through LDS (M, N <= 32) */,
       SMM_JIT_DEEP = 131072 /* grouped bodies of a batch split into tiles of C: few waves on the chip, as many products in flight per wave as the wait counter allows */ }; // variant bits of the generated dense kernel
std::string gen_smm_source(int typesize, int m, int n, int k, int flags, int variant, int lda = 0, int ldb = 0, int ldc = 0); // (0: tight)
bool smm_jit_eligible(const SmmBatch& s);
int launch_smm_jit(const SmmBatch& s, void* stream, const char** name); // -1: not available
int launch_smm_jit_mfma(const SmmBatch& s, void* stream, const char** name); // matrix-core work-group kernels specialised per descriptor; -1: not available
int launch_smm_jit_lowp(const SmmBatch& s, void* stream, const char** name); // 16-bit inputs on the specialised streaming form; -1: not applicable
bool smm_jit_grouped_eligible(const SmmBatch& s);
int smm_jit_prebuild(const SmmBatch* shapes, int nshapes, int grouped, int* built); // code objects into the cache on disk; returns failures
std::string gen_smm_grouped_source_for(const SmmBatch* groups, int ngroups, bool tiles = false);
int launch_smm_jit_grouped(const SmmBatch* groups, int ngroups, void* stream, const char** name); // several batches, one launch; -1: not available
int launch_c_order_check_groups(const SmmBatch* groups, int ngroups, void* stream); // one check launch for up to 32 batches (each with its devflags slot)
int jit_launch_dyn(JitKernel* k, unsigned blocks, unsigned threads, unsigned lds_bytes, void** args, void* stream);

// spmdm batch
struct SpmdmGeom {
  int m, n, k; long long batch;
  int cap;      // colidx/values capacity per item (m*k rounded up to even: slots stay dword-aligned)
  int rstride;  // rowidx entries per item (m+1 rounded up to even)
};
int launch_spmdm_create(const SpmdmGeom& g, int transa, const float* a, uint16_t* rowidx, uint16_t* colidx, float* values,
                        void* stream, const char** name);
int launch_spmdm_compute(const SpmdmGeom& g, int transb, int transc, float beta, const uint16_t* rowidx, const uint16_t* colidx,
                         const float* values, const float* b, float* c, void* stream, const char** name);

// blocked_gemm helpers (layouts of template/libxsmm_blocked_gemm_copy*.tpl.c)
int launch_bf16_widen(const unsigned short* src, float* dst, long long count, void* stream); // dst[i] = float(bits(src[i]) << 16)
struct BgemmGeom { int typesize, m, n, k, bm, bn, bk, mb, nb, kb; };
int launch_bgemm_copy(const BgemmGeom& g, int which /*0:A 1:B 2:C-in 3:C-out 4:convert_b_to_a 5:transpose_b (blocked -> blocked)*/, const void* src, int ld, void* dst, void* stream);
int launch_bgemm_compute(const BgemmGeom& g, int beta0, const void* a, const void* b, void* c, void* stream, const char** name);

// ---- host runtime ------------------------------------------------------------------------------------
struct Device {
  int count = -1;           // -1: not probed
  void* stream = nullptr;   // hipStream_t
};
Device& device();
Device& device_raw();                     // the same without sealing an open burst of deferred calls
bool device_ready();                      // probes once; false if no HIP device
void fail_no_device(const char* what);    // prints a loud error (always) -- the product has no CPU compute path
bool is_device_ptr(const void* p);
int pointer_kind(const void* p);         // bit 0: the GPU reaches it; bit 1: pinned host / managed memory (the CPU addresses it as well)
bool is_host_visible(const void* p);      // pinned host or managed memory (processed in place, but the CPU reads it directly)
void settle(const void* p0, const void* p1 = nullptr, const void* p2 = nullptr); // wait for the stream if an operand is host-visible
int flag_slot_set(int* slot, int equal_pairs, int decreasing_pairs); // the verdict without a check kernel (0: ok)
void* index_upload(const void* host_array, size_t bytes); // async copy of a host index array to the device (nullptr: failed)
void index_upload_commit();                                // after the launches that read uploaded arrays were queued
int library_gemm(int typesize, int transa, int transb, int m, int n, int k, double alpha, const void* a, int lda,
                 const void* b, int ldb, double beta, void* c, int ldc); // rocBLAS on the engine's stream; -1: not available
int* flag_slot();                         // device int[4] for one batch call's C-ordering verdict (nullptr: out of memory)
void flag_slot_commit();                  // after the launches that read the calling thread's latest slot were queued
void* dev_alloc(size_t bytes);
void dev_free(void* p);
int h2d(void* dst, const void* src, size_t bytes);
int d2h(void* dst, const void* src, size_t bytes);
int stream_sync();
void note_launch(const char* name);

// grow-only device scratch, one per thread-local slot id
void* scratch(int slot, size_t bytes);

enum KernelClass : int { KC_DENSE = 0, KC_REDUCE = 1, KC_CSR_REG = 2, KC_TEXT = 3 /* pattern kernel compiled from generated text (SOA family) */,
  KC_LOWP = 4 /* i16 / bf16 inputs (kernels/smm_lowp.hip) */ };

struct Kernel {                 // what a dispatched function pointer stands for
  libxsmm_gemm_descriptor desc;
  int kclass;
  bool registered;
  void* thunk;                  // the bare function pointer handed to the caller
  // KC_CSR_REG payload
  unsigned nnz = 0;
  unsigned* d_rowptr = nullptr; unsigned* d_colidx = nullptr; void* d_values = nullptr;
  // KC_TEXT payload: libxsmm_amd_spgemm* (xsmm_generator.cpp)
  void* text = nullptr;
};

Kernel* kernel_from_pointer(const void* fn);           // NULL if fn is not one of ours
void* adopt_kernel(Kernel* k);                         // caller-owned kernel: make its thunk and index it; NULL on failure
int text_kernel_execute(void* text, const void* a, const void* b, void* c, long long stride_dense, long long stride_c, long long batch);
void text_kernel_destroy(void* text);
void* make_thunk(Kernel* k);                           // executable stub carrying k
void free_thunk(void* thunk);
void call_kernel(Kernel* k, const void* a, const void* b, void* c, const void* x3, const void* x6); // what a thunk does (x3, x6: the 4th and 7th argument of the call)

int verbosity();
bool once(int* flag);   // true the first time

} // namespace xsmm

#endif
