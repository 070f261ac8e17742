/*
 * libxsmm.h -- C-ABI of the MI355X-native small/sparse-GEMM engine.
 *
 * This is the drop-in boundary: the hot-path subset of LIBXSMM 1.12's public interface, re-declared
 * with identical names, argument meaning, enum values and struct layouts, so that callers written
 * against the reference (samples/smm, samples/spmdm, samples/pyfr, CP2K-style batched SMM) compile
 * and link against libxsmm.so built from libxsmm-1_amd/csrc. Every declaration cites the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Differences a caller can observe (all documented in DESIGN.md / INTEGRATION.md):
 *  - operands may be device (hipMalloc) or host pointers; device operands are processed in place and
 *    asynchronously on the engine's HIP stream (see libxsmm_amd.h), host operands are staged over PCIe.
 *  - "JIT": hand-written gfx950 kernels for the headline shapes, and kernels generated as HIP text per descriptor and compiled
 *    with hiprtc (on a helper thread, code objects kept on disk) for everything else up to 64 x 64 x K; a pre-compiled
 *    kernel for any descriptor serves until then. Kernel pointers are host thunks (callable like the reference's).
 */
#ifndef LIBXSMM_H
#define LIBXSMM_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>

/* ---------------------------------------------------------------------------------------------
 * configuration (reference: generated include/libxsmm_config.h; defaults per SURVEY.md Appendix C)
 * --------------------------------------------------------------------------------------------- */
#define LIBXSMM_VERSION "1.12-amd"
#define LIBXSMM_VERSION_MAJOR 1
#define LIBXSMM_VERSION_MINOR 12
#define LIBXSMM_CACHELINE 64
#define LIBXSMM_ALIGNMENT 64
#define LIBXSMM_ILP64 0
#define LIBXSMM_SYNC 1
#define LIBXSMM_JIT 1
#if !defined(LIBXSMM_PREFETCH)
# define LIBXSMM_PREFETCH (-1) /* auto => samples call kernels with six arguments */
#endif
#define LIBXSMM_MAX_MNK 262144
#define LIBXSMM_MAX_DIM 64
#define LIBXSMM_MAX_M 64
#define LIBXSMM_MAX_N 64
#define LIBXSMM_MAX_K 64
#define LIBXSMM_FLAGS 0
#define LIBXSMM_ALPHA 1
#define LIBXSMM_BETA 1

#if defined(__cplusplus)
# define LIBXSMM_EXTERN_C extern "C"
#else
# define LIBXSMM_EXTERN_C
#endif
#if defined(__GNUC__)
# define LIBXSMM_VISIBILITY __attribute__((visibility("default")))
#else
# define LIBXSMM_VISIBILITY
#endif
#define LIBXSMM_API LIBXSMM_EXTERN_C LIBXSMM_VISIBILITY
#define LIBXSMM_APIEXT LIBXSMM_API /* reference: symbols of libxsmmext (OpenMP layer) */
#if defined(__cplusplus)
# define LIBXSMM_APIVAR(DECL) extern "C" LIBXSMM_VISIBILITY DECL
#else
# define LIBXSMM_APIVAR(DECL) extern LIBXSMM_VISIBILITY DECL
#endif
#define LIBXSMM_RETARGETABLE
#define LIBXSMM_INLINE static inline

/* small helpers that the reference's samples rely on (include/libxsmm_macros.h) */
#define LIBXSMM_STRINGIFY2(X) #X
#define LIBXSMM_STRINGIFY(X) LIBXSMM_STRINGIFY2(X)
#define LIBXSMM_CONCATENATE2(A, B) A##B
#define LIBXSMM_CONCATENATE(A, B) LIBXSMM_CONCATENATE2(A, B)
#define LIBXSMM_MIN(A, B) ((A) < (B) ? (A) : (B))
#define LIBXSMM_MAX(A, B) ((A) < (B) ? (B) : (A))
#define LIBXSMM_ABS(A) (0 <= (A) ? (A) : -(A))
#define LIBXSMM_CLMP(V, LO, HI) ((LO) < (V) ? ((V) <= (HI) ? (V) : (HI)) : (LO))
#define LIBXSMM_UP2(N, NPOT) (((N) + ((NPOT) - 1)) & ~((NPOT) - 1))
#define LIBXSMM_UP(N, UP) ((((N) + (UP) - 1) / (UP)) * (UP))
#define LIBXSMM_MOD2(A, NPOT) ((A) & ((NPOT) - 1))
#define LIBXSMM_FEQ(A, B) ((A) == (B))
#define LIBXSMM_NEQ(A, B) ((A) != (B))
#define LIBXSMM_ISNAN(A) LIBXSMM_NEQ(A, A)
#define LIBXSMM_NOTNAN(A) LIBXSMM_FEQ(A, A)
#define LIBXSMM_UNUSED(VAR) (void)(VAR)
#define LIBXSMM_ALIGN(POINTER, ALIGNMENT) ((POINTER) + (LIBXSMM_UP2((uintptr_t)(POINTER), ALIGNMENT) - ((uintptr_t)(POINTER))) / sizeof(*(POINTER)))
#define LIBXSMM_ASSERT(EXPR) ((void)0)
#define LIBXSMM_LD(M, N) (M) /* column-major library */
#define LIBXSMM_MNK_SIZE(M, N, K) (((size_t)(M)) * ((size_t)(N)) * ((size_t)(K)))
#define LIBXSMM_SIZE(M, N, K, S) (((size_t)(M) * (size_t)(K)) + ((size_t)(K) * (size_t)(N)) + ((size_t)(S) * (size_t)(M) * (size_t)(N)))
/* include/libxsmm_frontend.h:348 -- suitability of an SMM by arithmetic intensity */
#define LIBXSMM_SMM_AI(M, N, K, S, TYPESIZE) ((LIBXSMM_MNK_SIZE(M, N, K) * 2) <= ((size_t)(TYPESIZE) * 4 * LIBXSMM_SIZE(M, N, K, S)))
#define LIBXSMM_SMM(M, N, K, S, TYPESIZE) (LIBXSMM_MNK_SIZE(M, N, K) <= (LIBXSMM_MAX_MNK))

/* ---------------------------------------------------------------------------------------------
 * basic types (include/libxsmm_typedefs.h)
 * --------------------------------------------------------------------------------------------- */
typedef int libxsmm_blasint;                 /* :42-48,131 LP64 */
typedef unsigned short libxsmm_bfloat16;     /* :119 */
typedef union libxsmm_bfloat16_hp { libxsmm_bfloat16 i[2]; float f; } libxsmm_bfloat16_hp; /* :121-124: i[1] is the bf16 of f (little endian) */
typedef unsigned long long libxsmm_timer_tickint; /* include/libxsmm_timer.h:42 */

#define LIBXSMM_DESCRIPTOR_MAXSIZE 64        /* :109-111 */
typedef struct libxsmm_descriptor_blob { char data[LIBXSMM_DESCRIPTOR_MAXSIZE]; } libxsmm_descriptor_blob; /* :138-140 */
typedef struct libxsmm_gemm_blob { char data[128]; } libxsmm_gemm_blob;                                     /* :134 */
typedef struct libxsmm_gemm_descriptor libxsmm_gemm_descriptor; /* opaque; layout in src/libxsmm_main.h:157-168 */

typedef enum libxsmm_datatype { /* :158-167 */
  LIBXSMM_DATATYPE_F64 = 0, LIBXSMM_DATATYPE_F32 = 1, LIBXSMM_DATATYPE_BF16 = 2, LIBXSMM_DATATYPE_I64 = 3,
  LIBXSMM_DATATYPE_I32 = 4, LIBXSMM_DATATYPE_I16 = 5, LIBXSMM_DATATYPE_I8 = 6, LIBXSMM_DATATYPE_UNSUPPORTED = 7
} libxsmm_datatype;

typedef enum libxsmm_gemm_precision { /* :170-177 */
  LIBXSMM_GEMM_PRECISION_F64 = LIBXSMM_DATATYPE_F64, LIBXSMM_GEMM_PRECISION_F32 = LIBXSMM_DATATYPE_F32,
  LIBXSMM_GEMM_PRECISION_BF16 = LIBXSMM_DATATYPE_BF16, LIBXSMM_GEMM_PRECISION_I32 = LIBXSMM_DATATYPE_I32,
  LIBXSMM_GEMM_PRECISION_I16 = LIBXSMM_DATATYPE_I16, LIBXSMM_GEMM_PRECISION_I8 = LIBXSMM_DATATYPE_I8
} libxsmm_gemm_precision;

typedef enum libxsmm_gemm_flags { /* :180-213 */
  LIBXSMM_GEMM_FLAG_NONE = 0,
  LIBXSMM_GEMM_FLAG_TRANS_A = 1,
  LIBXSMM_GEMM_FLAG_TRANS_B = 2,
  LIBXSMM_GEMM_FLAG_TRANS_AB = 3,
  LIBXSMM_GEMM_FLAG_BETA_0 = 16,
  LIBXSMM_GEMM_FLAG_ALIGN_A = 64,
  LIBXSMM_GEMM_FLAG_ALIGN_C = 128,
  LIBXSMM_GEMM_FLAG_BATCH_REDUCE = 256,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT = 640,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BATCH_REDUCE = 896,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BETA_0 = 656,
  LIBXSMM_GEMM_FLAG_ALIGN_C_NTS_HINT_BETA_0_BATCH_REDUCE = 912,
  LIBXSMM_GEMM_FLAG_INVALID = 1024
} libxsmm_gemm_flags;

typedef enum libxsmm_mmbatch_flags { /* :224-233 */
  LIBXSMM_MMBATCH_FLAG_DEFAULT = 0, LIBXSMM_MMBATCH_FLAG_SYNCHRONIZED = 1024,
  LIBXSMM_MMBATCH_FLAG_SEQUENTIAL = 2048, LIBXSMM_MMBATCH_FLAG_STATISTIC = 4096
} libxsmm_mmbatch_flags;

#define LIBXSMM_PREFETCH_NONE 0
#define LIBXSMM_PREFETCH_SIGONLY 1
#define LIBXSMM_PREFETCH_AUTO (-1)
typedef enum libxsmm_gemm_prefetch_type { /* :236-262; accepted and recorded, no effect on gfx950 */
  LIBXSMM_GEMM_PREFETCH_NONE = 0, LIBXSMM_GEMM_PREFETCH_SIGONLY = 1, LIBXSMM_GEMM_PREFETCH_AL2 = 2,
  LIBXSMM_GEMM_PREFETCH_AL2_JPST = 4, LIBXSMM_GEMM_PREFETCH_BL2_VIA_C = 8, LIBXSMM_GEMM_PREFETCH_AL2_AHEAD = 16,
  LIBXSMM_GEMM_PREFETCH_AL2BL2_VIA_C = 10, LIBXSMM_GEMM_PREFETCH_AL2BL2_VIA_C_JPST = 12,
  LIBXSMM_GEMM_PREFETCH_AL2BL2_VIA_C_AHEAD = 24, LIBXSMM_GEMM_PREFETCH_AL1 = 32, LIBXSMM_GEMM_PREFETCH_BL1 = 64,
  LIBXSMM_GEMM_PREFETCH_CL1 = 128, LIBXSMM_GEMM_PREFETCH_AL1_BL1 = 96, LIBXSMM_GEMM_PREFETCH_BL1_CL1 = 192,
  LIBXSMM_GEMM_PREFETCH_AL1_CL1 = 160, LIBXSMM_GEMM_PREFETCH_AL1_BL1_CL1 = 224,
  LIBXSMM_PREFETCH_AL2CL2BL2_VIA_C = 10
} libxsmm_gemm_prefetch_type;

typedef enum libxsmm_kernel_kind { /* :552-569 */
  LIBXSMM_KERNEL_KIND_MATMUL = 0, LIBXSMM_KERNEL_KIND_MCOPY = 1, LIBXSMM_KERNEL_KIND_TRANS = 2,
  LIBXSMM_KERNEL_KIND_PGEMM = 3, LIBXSMM_KERNEL_KIND_GETRF = 4, LIBXSMM_KERNEL_KIND_TRMM = 5,
  LIBXSMM_KERNEL_KIND_TRSM = 6, LIBXSMM_KERNEL_KIND_INVALID = 7
} libxsmm_kernel_kind;

/* type helpers (:60-107) */
#define LIBXSMM_TYPESIZE(ENUM) ( \
  ((int)(ENUM)) == LIBXSMM_DATATYPE_F64 ? 8 : (((int)(ENUM)) == LIBXSMM_DATATYPE_F32 ? 4 : ( \
  ((int)(ENUM)) == LIBXSMM_DATATYPE_BF16 ? 2 : (((int)(ENUM)) == LIBXSMM_DATATYPE_I32 ? 4 : ( \
  ((int)(ENUM)) == LIBXSMM_DATATYPE_I16 ? 2 : (((int)(ENUM)) == LIBXSMM_DATATYPE_I8 ? 1 : 0))))))
#define LIBXSMM_GETENUM_INP(SRC) ((SRC) & 0x0F)
#define LIBXSMM_GETENUM_OUT(SRC) (0 == ((SRC) >> 4) ? LIBXSMM_GETENUM_INP(SRC) : ((SRC) >> 4))
#define LIBXSMM_GETENUM(INP, OUT) (((INP) == (OUT)) ? (INP) : ((INP) | ((OUT) << 4)))
#define LIBXSMM_TYPESYMBOL_double F64
#define LIBXSMM_TYPESYMBOL_float F32
#define LIBXSMM_TYPESYMBOL_libxsmm_bfloat16 BF16
#define LIBXSMM_TYPESYMBOL_int I32
#define LIBXSMM_TYPESYMBOL_short I16
#define LIBXSMM_TYPESYMBOL_char I8
#define LIBXSMM_TYPESYMBOL(TYPE) LIBXSMM_CONCATENATE(LIBXSMM_TYPESYMBOL_, TYPE)
#define LIBXSMM_DATATYPE(TYPE) LIBXSMM_CONCATENATE(LIBXSMM_DATATYPE_, LIBXSMM_TYPESYMBOL(TYPE))
#define LIBXSMM_GEMM_PRECISION(TYPE) LIBXSMM_CONCATENATE(LIBXSMM_GEMM_PRECISION_, LIBXSMM_TYPESYMBOL(TYPE))
#define LIBXSMM_TYPENAME_double f64
#define LIBXSMM_TYPENAME_float f32
#define LIBXSMM_TYPENAME(TYPE) LIBXSMM_STRINGIFY(LIBXSMM_CONCATENATE(LIBXSMM_TYPENAME_, TYPE))

/* kernel function types (:526-549). A dispatched kernel is a bare function pointer; the optional trailing
 * arguments are the reference's prefetch locations (pa, pb, pc) and are ignored here. */
typedef void (*libxsmm_dmmfunction)(const double* a, const double* b, double* c, ...);
typedef void (*libxsmm_smmfunction)(const float* a, const float* b, float* c, ...);
typedef void (*libxsmm_wimmfunction)(const short* a, const short* b, int* c, ...);
typedef void (*libxsmm_wsmmfunction)(const short* a, const short* b, float* c, ...);
typedef void (*libxsmm_bsmmfunction)(const libxsmm_bfloat16* a, const libxsmm_bfloat16* b, float* c, ...);
typedef void (*libxsmm_bmmfunction)(const libxsmm_bfloat16* a, const libxsmm_bfloat16* b, libxsmm_bfloat16* c, ...);
typedef void (*libxsmm_dmmfunction_reducebatch)(const double** a, const double** b, double* c, const unsigned long long* count, ...);
typedef void (*libxsmm_smmfunction_reducebatch)(const float** a, const float** b, float* c, const unsigned long long* count, ...);
typedef void (*libxsmm_bsmmfunction_reducebatch)(const libxsmm_bfloat16** a, const libxsmm_bfloat16** b, float* c, const unsigned long long* count, ...);
typedef void (*libxsmm_bmmfunction_reducebatch)(const libxsmm_bfloat16** a, const libxsmm_bfloat16** b, libxsmm_bfloat16* c, const unsigned long long* count, ...);

typedef union libxsmm_xmmfunction { /* :544-549 */
  void (*xmm)(const void* a, const void* b, void* c, ...);
  void (*xbm)(const void** a, const void** b, void* c, const unsigned long long* count, ...);
  libxsmm_dmmfunction dmm; libxsmm_smmfunction smm; libxsmm_wimmfunction wimm; libxsmm_wsmmfunction wsmm;
  libxsmm_bsmmfunction bsmm; libxsmm_bmmfunction bmm;
  libxsmm_dmmfunction_reducebatch dmr; libxsmm_smmfunction_reducebatch smr;
  libxsmm_bsmmfunction_reducebatch bsmr; libxsmm_bmmfunction_reducebatch bmr;
} libxsmm_xmmfunction;

typedef struct libxsmm_mmkernel_info { /* :596-607 */
  libxsmm_gemm_precision iprecision, oprecision;
  libxsmm_gemm_prefetch_type prefetch;
  unsigned int lda, ldb, ldc;
  unsigned int m, n, k;
  int flags;
} libxsmm_mmkernel_info;

typedef struct libxsmm_registry_info { size_t capacity, size, nbytes, nstatic, ncache; } libxsmm_registry_info; /* :630-632 */

/* ---------------------------------------------------------------------------------------------
 * front-end macros (include/libxsmm_frontend.h)
 * --------------------------------------------------------------------------------------------- */
/* :202-210 -- anything but 'N'/'n' counts as transposed */
#define LIBXSMM_GEMM_FLAGS(TRANSA, TRANSB) \
  ((('n' == (TRANSA) || 'N' == (TRANSA)) ? LIBXSMM_GEMM_FLAG_NONE : LIBXSMM_GEMM_FLAG_TRANS_A) \
 | (('n' == (TRANSB) || 'N' == (TRANSB)) ? LIBXSMM_GEMM_FLAG_NONE : LIBXSMM_GEMM_FLAG_TRANS_B))
#define LIBXSMM_GEMM_PFLAGS(TRANSA, TRANSB, DEFAULT) (LIBXSMM_GEMM_FLAGS( \
  NULL != ((const void*)(TRANSA)) ? (*(const char*)(TRANSA)) : (0 == (LIBXSMM_GEMM_FLAG_TRANS_A & (DEFAULT)) ? 'n' : 't'), \
  NULL != ((const void*)(TRANSB)) ? (*(const char*)(TRANSB)) : (0 == (LIBXSMM_GEMM_FLAG_TRANS_B & (DEFAULT)) ? 'n' : 't')) \
  | (~(LIBXSMM_GEMM_FLAG_TRANS_A | LIBXSMM_GEMM_FLAG_TRANS_B) & (DEFAULT)))

/* :36-70 -- with LIBXSMM_PREFETCH != 0 the samples pass prefetch pointers; they are accepted and ignored */
#if (0 != LIBXSMM_PREFETCH)
# define LIBXSMM_GEMM_PREFETCH_A(EXPR) (EXPR)
# define LIBXSMM_GEMM_PREFETCH_B(EXPR) (EXPR)
# define LIBXSMM_GEMM_PREFETCH_C(EXPR) (EXPR)
#else
# define LIBXSMM_GEMM_PREFETCH_A(EXPR) 0
# define LIBXSMM_GEMM_PREFETCH_B(EXPR) 0
# define LIBXSMM_GEMM_PREFETCH_C(EXPR) 0
#endif
/* :322-341 */
#define LIBXSMM_MMCALL_ABC(FN, A, B, C) FN(A, B, C)
#define LIBXSMM_MMCALL_PRF(FN, A, B, C, PA, PB, PC) \
  FN(A, B, C, LIBXSMM_GEMM_PREFETCH_A(PA), LIBXSMM_GEMM_PREFETCH_B(PB), LIBXSMM_GEMM_PREFETCH_C(PC))
#if (0 == LIBXSMM_PREFETCH)
# define LIBXSMM_MMCALL_LDX(FN, A, B, C, M, N, K, LDA, LDB, LDC) LIBXSMM_MMCALL_ABC(FN, A, B, C)
#else
# define LIBXSMM_MMCALL_LDX(FN, A, B, C, M, N, K, LDA, LDB, LDC) \
  LIBXSMM_MMCALL_PRF(FN, A, B, C, (A) + ((size_t)LDA) * (K), (B) + ((size_t)LDB) * (N), (C) + ((size_t)LDC) * (N))
#endif
#define LIBXSMM_MMCALL(FN, A, B, C, M, N, K) LIBXSMM_MMCALL_LDX(FN, A, B, C, M, N, K, M, K, M)
#define LIBXSMM_USEOMP(FUNCTION) LIBXSMM_CONCATENATE(FUNCTION, _omp)

/* LIBXSMM_MATINIT (:414-446): dst[i*ld+j] = scale*(seed+1)/(1+i*ld+j) for j<nrows, padding rows = seed.
 * Only the seed != 0 formula is provided (the shuffle-based seed==0 variant is not used by the hot-path samples). */
#define LIBXSMM_MATINIT(TYPE, SEED, DST, NROWS, NCOLS, LD, SCALE) do { \
  const double libxsmm_mi_s_ = (double)(SCALE) * (double)(SEED) + (double)(SCALE); \
  const libxsmm_blasint libxsmm_mi_ld_ = (libxsmm_blasint)(LD); \
  libxsmm_blasint libxsmm_mi_c_, libxsmm_mi_r_; \
  for (libxsmm_mi_c_ = 0; libxsmm_mi_c_ < (libxsmm_blasint)(NCOLS); ++libxsmm_mi_c_) { \
    for (libxsmm_mi_r_ = 0; libxsmm_mi_r_ < libxsmm_mi_ld_; ++libxsmm_mi_r_) { \
      const libxsmm_blasint libxsmm_mi_k_ = libxsmm_mi_c_ * libxsmm_mi_ld_ + libxsmm_mi_r_; \
      (DST)[libxsmm_mi_k_] = (libxsmm_mi_r_ < (libxsmm_blasint)(NROWS)) \
        ? (TYPE)(libxsmm_mi_s_ / (1.0 + libxsmm_mi_k_)) : (TYPE)(SEED); \
    } \
  } } while (0)
#define LIBXSMM_MATINIT_SEQ LIBXSMM_MATINIT
#define LIBXSMM_MATINIT_OMP LIBXSMM_MATINIT

/* ---------------------------------------------------------------------------------------------
 * library control (src/template/libxsmm.h:73-104)
 * --------------------------------------------------------------------------------------------- */
LIBXSMM_API void libxsmm_init(void);                               /* :73 */
LIBXSMM_API void libxsmm_finalize(void);                           /* :75 */
LIBXSMM_API int libxsmm_get_target_archid(void);                   /* :81 */
LIBXSMM_API void libxsmm_set_target_archid(int id);                /* :83 */
LIBXSMM_API const char* libxsmm_get_target_arch(void);             /* :89 */
LIBXSMM_API void libxsmm_set_target_arch(const char* arch);        /* :91 */
LIBXSMM_API int libxsmm_get_verbosity(void);                       /* :94 */
LIBXSMM_API void libxsmm_set_verbosity(int level);                 /* :99 */
LIBXSMM_API libxsmm_gemm_prefetch_type libxsmm_get_gemm_auto_prefetch(void);            /* :102 */
LIBXSMM_API void libxsmm_set_gemm_auto_prefetch(libxsmm_gemm_prefetch_type strategy);   /* :104 */
/* prefetch strategy a dispatcher uses for a caller's request (src/libxsmm_gemm.c:471-494): negative = the configured default */
LIBXSMM_API libxsmm_gemm_prefetch_type libxsmm_get_gemm_prefetch(int prefetch);
LIBXSMM_API libxsmm_gemm_prefetch_type libxsmm_get_gemm_xprefetch(const int* prefetch);
/* arch ids (include/libxsmm_cpuid.h:40-52); LIBXSMM_TARGET_ARCH_GENERIC disables dispatch as in the reference */
#define LIBXSMM_TARGET_ARCH_UNKNOWN 0
#define LIBXSMM_TARGET_ARCH_GENERIC 1
#define LIBXSMM_AMD_GFX950 9500
/* data symbols (include/libxsmm_generator.h:279-281) */
LIBXSMM_APIVAR(unsigned int libxsmm_ninit);
LIBXSMM_APIVAR(int libxsmm_verbosity);

/* ---------------------------------------------------------------------------------------------
 * descriptors (include/libxsmm_generator.h:43-96; src/libxsmm_generator.c:47-336).
 * NULL is returned (quietly) unless alpha==1, beta in {0,1} and no TRANS_A (include/libxsmm_generator.h:36-39).
 * --------------------------------------------------------------------------------------------- */
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_dgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  double alpha, double beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_sgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  float alpha, float beta, int flags, int prefetch);
/* low-precision descriptors (reference src/libxsmm_generator.c:93-185) */
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_wigemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  int alpha, int beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_wsgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  float alpha, float beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_bsgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  float alpha, float beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_bgemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
  float alpha, float beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_dinit(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision precision, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, double alpha, double beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_dinit2(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, double alpha, double beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision precision, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init2(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta, int flags, int prefetch);
LIBXSMM_API libxsmm_gemm_descriptor* libxsmm_gemm_descriptor_init3(libxsmm_descriptor_blob* blob,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const void* alpha, const void* beta,
  int flags, int prefetch, double* dalpha, double* dbeta);

/* ---------------------------------------------------------------------------------------------
 * dispatch (src/template/libxsmm.h:124-176; src/libxsmm_main.c:2139-2315)
 * --------------------------------------------------------------------------------------------- */
LIBXSMM_API libxsmm_xmmfunction libxsmm_xmmdispatch(const libxsmm_gemm_descriptor* descriptor);
LIBXSMM_API libxsmm_dmmfunction libxsmm_dmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const double* alpha, const double* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_smmfunction libxsmm_smmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch);
/* Low-precision dispatchers (reference src/libxsmm_main.c:2198-2259): i16 -> i32, i16 -> f32 (the kernel is called as
 * kernel(a, b, c, pa, pb, pc, &scf): the 7th argument points to the scaling factor), bf16 -> f32, bf16 -> bf16. As in the
 * reference's generator (src/generator_gemm.c:121-147,236-243) k must be even, TRANS_B is not available and a bf16 result
 * needs m % 16 == 0 (otherwise NULL); A is stored in pairs of k (a[(k/2)*lda*2 + m*2 + k%2]), B and C column-major; the
 * arithmetic is that of the gold loops in samples/xgemm/kernel.c (:915-927, :1007-1021, :1104-1123, :1207-1229). Batches
 * (libxsmm_mmbatch_kernel) take operands the GPU can reach and independent C operands. */
LIBXSMM_API libxsmm_wimmfunction libxsmm_wimmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const int* alpha, const int* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_wsmmfunction libxsmm_wsmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_bsmmfunction libxsmm_bsmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_bmmfunction libxsmm_bmmdispatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_dmmfunction_reducebatch libxsmm_dmmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const double* alpha, const double* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_smmfunction_reducebatch libxsmm_smmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const float* alpha, const float* beta, const int* flags, const int* prefetch);
/** Low-precision batch-reduce dispatchers (src/libxsmm_main.c:2290-2315): bf16 inputs, fp32 or bf16 result; kernel(a[], b[], c, &count).
 *  The sums stay in fp32 across the batch; a bf16 C is read once and truncated once. */
LIBXSMM_API libxsmm_bsmmfunction_reducebatch libxsmm_bsmmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc, const float* alpha, const float* beta, const int* flags, const int* prefetch);
LIBXSMM_API libxsmm_bmmfunction_reducebatch libxsmm_bmmdispatch_reducebatch(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc, const float* alpha, const float* beta, const int* flags, const int* prefetch);

/* caller-owned sparse kernels (src/template/libxsmm.h:303-319; src/libxsmm_main.c:2523-2582) */
LIBXSMM_API libxsmm_dmmfunction libxsmm_create_dcsr_reg(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* row_ptr, const unsigned int* column_idx, const double* values);
LIBXSMM_API libxsmm_smmfunction libxsmm_create_scsr_reg(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* row_ptr, const unsigned int* column_idx, const float* values);
/* SOA kernels for fused runs (EDGE/SeisSol; src/template/libxsmm.h:386-414, src/libxsmm_main.c:2423-2520): sparse or dense
 * operator times operands stored [row][col][v] with the run index v innermost; v = 8 (fp64) / 16 (fp32), the widths of
 * the reference's AVX-512 kernels (libxsmm_amd_soa_width). Caller-owned: release with libxsmm_release_kernel.
 *   xcsr_soa, lda == 0: C[m][n][v] (+)= A_csr(m,k) * B[k][n][v]      call kernel(values of A, B, C); rows without entries
 *                                                                    are not touched (also for beta == 0)
 *   xcsr_soa/xcsc_soa, ldb == 0: C[m][n][v] (+)= A[m][k][v] * B(k,n) call kernel(A, values of B, C)
 *   rm_ac_soa: C[m][n][v] (+)= A[m][k][v] * B[k*ldb+n]               rm_bc_soa: C[m][n][v] (+)= A[m*lda+k] * B[k][n][v]
 * One call is one small product; many products that share the operator go through libxsmm_amd_kernel_execute_batch. */
LIBXSMM_API libxsmm_xmmfunction libxsmm_create_xcsr_soa(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* row_ptr, const unsigned int* column_idx, const void* values);
LIBXSMM_API libxsmm_xmmfunction libxsmm_create_xcsc_soa(const libxsmm_gemm_descriptor* descriptor,
  const unsigned int* column_ptr, const unsigned int* row_idx, const void* values);
LIBXSMM_API libxsmm_xmmfunction libxsmm_create_rm_ac_soa(const libxsmm_gemm_descriptor* descriptor);
LIBXSMM_API libxsmm_xmmfunction libxsmm_create_rm_bc_soa(const libxsmm_gemm_descriptor* descriptor);
LIBXSMM_API void libxsmm_release_kernel(const void* jit_kernel);   /* src/template/libxsmm.h:325 */

/* introspection (src/template/libxsmm.h:107-121) */
LIBXSMM_API int libxsmm_get_kernel_kind(const void* kernel, libxsmm_kernel_kind* kind);
LIBXSMM_API int libxsmm_get_mmkernel_info(libxsmm_xmmfunction kernel, libxsmm_mmkernel_info* info, size_t* code_size);
LIBXSMM_API int libxsmm_get_registry_info(libxsmm_registry_info* info);

/* ---------------------------------------------------------------------------------------------
 * batched SMM (src/template/libxsmm.h:178-260; src/libxsmm_gemm.c:1315-1888; src/libxsmm_ext_gemm.c:758-1013)
 *  index_stride != 0: a,b,c point to elements, stride_* are arrays of element indexes walked with a byte
 *                     step of index_stride (index_base is subtracted);
 *  index_stride == 0: a,b,c are arrays of pointers, *stride_* is the byte distance between two pointers;
 *  a NULL stride array means the operand is shared by all multiplications;
 *  batchsize < 0: the caller guarantees that no two multiplications update the same C concurrently.
 * --------------------------------------------------------------------------------------------- */
LIBXSMM_API void libxsmm_mmbatch(libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec,
  const char* transa, const char* transb, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const void* alpha, const void* a, const libxsmm_blasint* lda, const void* b, const libxsmm_blasint* ldb,
  const void* beta, void* c, const libxsmm_blasint* ldc, libxsmm_blasint index_base, libxsmm_blasint index_stride,
  const libxsmm_blasint stride_a[], const libxsmm_blasint stride_b[], const libxsmm_blasint stride_c[],
  libxsmm_blasint batchsize, /*unsigned*/int tid, /*unsigned*/int nthreads);
LIBXSMM_API int libxsmm_mmbatch_kernel(libxsmm_xmmfunction kernel, libxsmm_blasint index_base,
  libxsmm_blasint index_stride, const libxsmm_blasint stride_a[], const libxsmm_blasint stride_b[], const libxsmm_blasint stride_c[],
  const void* a, const void* b, void* c, libxsmm_blasint batchsize, /*unsigned*/int tid, /*unsigned*/int ntasks,
  unsigned char itypesize, unsigned char otypesize, int flags);       /* src/libxsmm_gemm.h:200 */
LIBXSMM_API int libxsmm_mmbatch_blas(libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec,
  const char* transa, const char* transb, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const void* alpha, const void* a, const libxsmm_blasint* lda, const void* b, const libxsmm_blasint* ldb,
  const void* beta, void* c, const libxsmm_blasint* ldc, libxsmm_blasint index_base, libxsmm_blasint index_stride,
  const libxsmm_blasint stride_a[], const libxsmm_blasint stride_b[], const libxsmm_blasint stride_c[],
  libxsmm_blasint batchsize);                                          /* src/libxsmm_gemm.h:207 */
LIBXSMM_API void libxsmm_gemm_batch(libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec,
  const char* transa, const char* transb, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const void* alpha, const void* a, const libxsmm_blasint* lda, const void* b, const libxsmm_blasint* ldb,
  const void* beta, void* c, const libxsmm_blasint* ldc, libxsmm_blasint index_base, libxsmm_blasint index_stride,
  const libxsmm_blasint stride_a[], const libxsmm_blasint stride_b[], const libxsmm_blasint stride_c[],
  libxsmm_blasint batchsize);
LIBXSMM_APIEXT void libxsmm_gemm_batch_omp(libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec,
  const char* transa, const char* transb, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const void* alpha, const void* a, const libxsmm_blasint* lda, const void* b, const libxsmm_blasint* ldb,
  const void* beta, void* c, const libxsmm_blasint* ldc, libxsmm_blasint index_base, libxsmm_blasint index_stride,
  const libxsmm_blasint stride_a[], const libxsmm_blasint stride_b[], const libxsmm_blasint stride_c[],
  libxsmm_blasint batchsize);
/* groups of homogeneous batches, BLAS-style pointer arrays (CP2K mixed shapes) */
LIBXSMM_API void libxsmm_dgemm_batch(const char transa_array[], const char transb_array[],
  const libxsmm_blasint m_array[], const libxsmm_blasint n_array[], const libxsmm_blasint k_array[],
  const double alpha_array[], const double* a_array[], const libxsmm_blasint lda_array[],
  const double* b_array[], const libxsmm_blasint ldb_array[],
  const double beta_array[], double* c_array[], const libxsmm_blasint ldc_array[],
  const libxsmm_blasint* group_count, const libxsmm_blasint group_size[]);
LIBXSMM_API void libxsmm_sgemm_batch(const char transa_array[], const char transb_array[],
  const libxsmm_blasint m_array[], const libxsmm_blasint n_array[], const libxsmm_blasint k_array[],
  const float alpha_array[], const float* a_array[], const libxsmm_blasint lda_array[],
  const float* b_array[], const libxsmm_blasint ldb_array[],
  const float beta_array[], float* c_array[], const libxsmm_blasint ldc_array[],
  const libxsmm_blasint* group_count, const libxsmm_blasint group_size[]);
LIBXSMM_APIEXT void libxsmm_dgemm_batch_omp(const char transa_array[], const char transb_array[],
  const libxsmm_blasint m_array[], const libxsmm_blasint n_array[], const libxsmm_blasint k_array[],
  const double alpha_array[], const double* a_array[], const libxsmm_blasint lda_array[],
  const double* b_array[], const libxsmm_blasint ldb_array[],
  const double beta_array[], double* c_array[], const libxsmm_blasint ldc_array[],
  const libxsmm_blasint* group_count, const libxsmm_blasint group_size[]);
LIBXSMM_APIEXT void libxsmm_sgemm_batch_omp(const char transa_array[], const char transb_array[],
  const libxsmm_blasint m_array[], const libxsmm_blasint n_array[], const libxsmm_blasint k_array[],
  const float alpha_array[], const float* a_array[], const libxsmm_blasint lda_array[],
  const float* b_array[], const libxsmm_blasint ldb_array[],
  const float beta_array[], float* c_array[], const libxsmm_blasint ldc_array[],
  const libxsmm_blasint* group_count, const libxsmm_blasint group_size[]);
/* auto-batch recording (src/libxsmm_ext_gemm.c:1016-1135): between begin/end, matching libxsmm_?gemm calls are
 * recorded instead of executed and flushed as one device batch by libxsmm_mmbatch_end. */
LIBXSMM_APIEXT void libxsmm_mmbatch_begin(libxsmm_gemm_precision precision, const int* flags,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const libxsmm_blasint* lda, const libxsmm_blasint* ldb, const libxsmm_blasint* ldc,
  const void* alpha, const void* beta);
LIBXSMM_APIEXT void libxsmm_mmbatch_end(void);

/* single GEMM, BLAS-like (src/template/libxsmm.h:389-400; src/libxsmm_gemm.c:1265-1290). Any alpha/beta/trans. */
LIBXSMM_API void libxsmm_dgemm(const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const double* alpha, const double* a, const libxsmm_blasint* lda, const double* b, const libxsmm_blasint* ldb,
  const double* beta, double* c, const libxsmm_blasint* ldc);
LIBXSMM_API void libxsmm_sgemm(const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const float* alpha, const float* a, const libxsmm_blasint* lda, const float* b, const libxsmm_blasint* ldb,
  const float* beta, float* c, const libxsmm_blasint* ldc);
/* BLAS call wrapper (reference src/libxsmm_ext_gemm.c:256-660, documentation/libxsmm_mm.md "Call Wrapper"): relink an
 * application that calls the Fortran BLAS symbols with -Wl,--wrap=dgemm_,--wrap=sgemm_ (optionally also
 * --wrap=dgemm_batch_,--wrap=sgemm_batch_,--wrap=dgemm_batch,--wrap=sgemm_batch) and its calls arrive here. Every call is
 * served by the device path (general alpha/beta/transposes included): no __real_ symbol, i.e. no BLAS library, is needed. */
LIBXSMM_APIEXT void __wrap_dgemm_(const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const double* alpha, const double* a, const libxsmm_blasint* lda, const double* b, const libxsmm_blasint* ldb,
  const double* beta, double* c, const libxsmm_blasint* ldc);
LIBXSMM_APIEXT void __wrap_sgemm_(const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const float* alpha, const float* a, const libxsmm_blasint* lda, const float* b, const libxsmm_blasint* ldb,
  const float* beta, float* c, const libxsmm_blasint* ldc);
#define LIBXSMM_AMD_WRAP_BATCH_DECL(NAME, T) LIBXSMM_APIEXT void NAME(const char transa_array[], const char transb_array[], \
  const libxsmm_blasint m_array[], const libxsmm_blasint n_array[], const libxsmm_blasint k_array[], \
  const T alpha_array[], const T* a_array[], const libxsmm_blasint lda_array[], const T* b_array[], const libxsmm_blasint ldb_array[], \
  const T beta_array[], T* c_array[], const libxsmm_blasint ldc_array[], const libxsmm_blasint* group_count, const libxsmm_blasint group_size[])
LIBXSMM_AMD_WRAP_BATCH_DECL(__wrap_dgemm_batch_, double);
LIBXSMM_AMD_WRAP_BATCH_DECL(__wrap_sgemm_batch_, float);
LIBXSMM_AMD_WRAP_BATCH_DECL(__wrap_dgemm_batch, double);
LIBXSMM_AMD_WRAP_BATCH_DECL(__wrap_sgemm_batch, float);
/* the library's BLAS fallback entry points (src/template/libxsmm.h:402-414); here: the engine's general-form kernel */
LIBXSMM_API void libxsmm_blas_dgemm(const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const double* alpha, const double* a, const libxsmm_blasint* lda, const double* b, const libxsmm_blasint* ldb,
  const double* beta, double* c, const libxsmm_blasint* ldc);
LIBXSMM_API void libxsmm_blas_sgemm(const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const float* alpha, const float* a, const libxsmm_blasint* lda, const float* b, const libxsmm_blasint* ldb,
  const float* beta, float* c, const libxsmm_blasint* ldc);

/* ---------------------------------------------------------------------------------------------
 * fsspmdm -- fixed-sparsity operator times dense panels (include/libxsmm_fsspmdm.h:37-58)
 * --------------------------------------------------------------------------------------------- */
typedef struct libxsmm_dfsspmdm libxsmm_dfsspmdm;
typedef struct libxsmm_sfsspmdm libxsmm_sfsspmdm;
LIBXSMM_API libxsmm_dfsspmdm* libxsmm_dfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const double alpha, const double beta, const double* a_dense);
LIBXSMM_API void libxsmm_dfsspmdm_execute(const libxsmm_dfsspmdm* handle, const double* B, double* C);
LIBXSMM_API void libxsmm_dfsspmdm_destroy(libxsmm_dfsspmdm* handle);
LIBXSMM_API libxsmm_sfsspmdm* libxsmm_sfsspmdm_create(libxsmm_blasint M, libxsmm_blasint N, libxsmm_blasint K,
  libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, const float alpha, const float beta, const float* a_dense);
LIBXSMM_API void libxsmm_sfsspmdm_execute(const libxsmm_sfsspmdm* handle, const float* B, float* C);
LIBXSMM_API void libxsmm_sfsspmdm_destroy(libxsmm_sfsspmdm* handle);

/* ---------------------------------------------------------------------------------------------
 * spmdm -- dense-in, sparse-compute (include/libxsmm_spmdm.h:37-133). Handle and slice are caller-visible.
 * --------------------------------------------------------------------------------------------- */
typedef enum libxsmm_spmdm_datatype { LIBXSMM_SPMDM_DATATYPE_F32, LIBXSMM_SPMDM_DATATYPE_BFLOAT16 } libxsmm_spmdm_datatype; /* :37-40 */
typedef struct libxsmm_spmdm_handle { /* :42-61 */
  int m, n, k;
  int bm, bn, bk;
  int mb, nb, kb;
  libxsmm_spmdm_datatype datatype;
  char* base_ptr_scratch_A;
  char* base_ptr_scratch_B_scratch_C;
  int memory_for_scratch_per_thread;
} libxsmm_spmdm_handle;
typedef struct libxsmm_CSR_sparseslice { uint16_t* rowidx; uint16_t* colidx; float* values; } libxsmm_CSR_sparseslice; /* :67-72 */
LIBXSMM_API void libxsmm_spmdm_init(int M, int N, int K, int max_threads,
  libxsmm_spmdm_handle* handle, libxsmm_CSR_sparseslice** libxsmm_output_csr);
LIBXSMM_API void libxsmm_spmdm_destroy(libxsmm_spmdm_handle* handle);
LIBXSMM_API int libxsmm_spmdm_get_num_createSparseSlice_blocks(const libxsmm_spmdm_handle* handle);
LIBXSMM_API int libxsmm_spmdm_get_num_compute_blocks(const libxsmm_spmdm_handle* handle);
LIBXSMM_API void libxsmm_spmdm_createSparseSlice_fp32_thread(const libxsmm_spmdm_handle* handle, char transa,
  const float* a, libxsmm_CSR_sparseslice* libxsmm_output_csr_a, int block_id, int tid, int nthreads);
LIBXSMM_API void libxsmm_spmdm_compute_fp32_thread(const libxsmm_spmdm_handle* handle, char transa, char transb,
  const float* alpha, libxsmm_CSR_sparseslice* a_sparse, const float* b, char transc, const float* beta, float* c,
  int block_id, int tid, int nthreads);
/* bfloat16 twins (include/libxsmm_spmdm.h:98-133): A resp. B hold the upper halves of IEEE floats; slices, sums and C
 * are float. As in the reference template, `*beta` is used as a number without widening it: the 16-bit pattern is the
 * factor (pattern 0: beta = 0, pattern 1: beta = 1; the bf16 encoding of 1.0, 0x3F80, scales C by 16256). */
LIBXSMM_API void libxsmm_spmdm_createSparseSlice_bfloat16_thread(const libxsmm_spmdm_handle* handle, char transa,
  const libxsmm_bfloat16* a, libxsmm_CSR_sparseslice* libxsmm_output_csr_a, int block_id, int tid, int nthreads);
LIBXSMM_API void libxsmm_spmdm_compute_bfloat16_thread(const libxsmm_spmdm_handle* handle, char transa, char transb,
  const libxsmm_bfloat16* alpha, libxsmm_CSR_sparseslice* a_sparse, const libxsmm_bfloat16* b, char transc,
  const libxsmm_bfloat16* beta, float* c, int block_id, int tid, int nthreads);

/* ---------------------------------------------------------------------------------------------
 * blocked_gemm (include/libxsmm_blocked_gemm.h:37-97)
 * --------------------------------------------------------------------------------------------- */
typedef enum libxsmm_blocked_gemm_order { /* :37-44 */
  LIBXSMM_BLOCKED_GEMM_ORDER_JIK = 0, LIBXSMM_BLOCKED_GEMM_ORDER_IJK = 1, LIBXSMM_BLOCKED_GEMM_ORDER_JKI = 2,
  LIBXSMM_BLOCKED_GEMM_ORDER_IKJ = 3, LIBXSMM_BLOCKED_GEMM_ORDER_KJI = 4, LIBXSMM_BLOCKED_GEMM_ORDER_KIJ = 5
} libxsmm_blocked_gemm_order;
typedef struct libxsmm_blocked_gemm_handle libxsmm_blocked_gemm_handle;
LIBXSMM_API libxsmm_blocked_gemm_handle* libxsmm_blocked_gemm_handle_create(/*unsigned*/int nthreads,
  libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
  const libxsmm_blasint* bm, const libxsmm_blasint* bn, const libxsmm_blasint* bk,
  const libxsmm_blasint* b_m1, const libxsmm_blasint* b_n1, const libxsmm_blasint* b_k1, const libxsmm_blasint* b_k2,
  const void* alpha, const void* beta, const int* gemm_flags, const libxsmm_gemm_prefetch_type* prefetch,
  const libxsmm_blocked_gemm_order* order);
LIBXSMM_API void libxsmm_blocked_gemm_handle_destroy(const libxsmm_blocked_gemm_handle* handle);
LIBXSMM_API int libxsmm_blocked_gemm_copyin_a(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst);
LIBXSMM_API int libxsmm_blocked_gemm_copyin_b(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst);
LIBXSMM_API int libxsmm_blocked_gemm_copyin_c(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst);
LIBXSMM_API int libxsmm_blocked_gemm_copyout_c(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst);
/* blocked -> blocked permutations (include/libxsmm_blocked_gemm.h:77-79); `ld` is ignored, as in the reference */
LIBXSMM_API int libxsmm_blocked_gemm_convert_b_to_a(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst);
LIBXSMM_API int libxsmm_blocked_gemm_transpose_b(const libxsmm_blocked_gemm_handle* handle, const void* src, const libxsmm_blasint* ld, void* dst);
LIBXSMM_API void libxsmm_blocked_gemm_st(const libxsmm_blocked_gemm_handle* handle, const void* a, const void* b, void* c,
  /*unsigned*/int start_thread, /*unsigned*/int tid);
LIBXSMM_APIEXT void libxsmm_blocked_gemm_omp(const libxsmm_blocked_gemm_handle* handle,
  const void* a, const void* b, void* c, /*unsigned*/int count);

/* ---------------------------------------------------------------------------------------------
 * helpers the hot-path samples link (include/libxsmm_malloc.h, _timer.h, _rng.h, _math.h)
 * --------------------------------------------------------------------------------------------- */
/* custom default allocator (include/libxsmm_malloc.h:36-64): malloc_fn/free_fn come as a pair, two NULLs restore the
 * built-in allocator (pinned host memory that the GPU addresses directly); buffers are released by the allocator that
 * made them */
typedef void* (*libxsmm_malloc_ctx)(void* context, size_t size);
typedef void* (*libxsmm_malloc_fun)(size_t size);
typedef union libxsmm_malloc_function { libxsmm_malloc_ctx ctx_form; libxsmm_malloc_fun function; } libxsmm_malloc_function;
typedef void (*libxsmm_free_ctx)(void* context, void* buffer);
typedef void (*libxsmm_free_fun)(void* buffer);
typedef union libxsmm_free_function { libxsmm_free_ctx ctx_form; libxsmm_free_fun function; } libxsmm_free_function;
LIBXSMM_API int libxsmm_set_default_allocator(void* context, libxsmm_malloc_function malloc_fn, libxsmm_free_function free_fn);
LIBXSMM_API int libxsmm_get_default_allocator(void** context, libxsmm_malloc_function* malloc_fn, libxsmm_free_function* free_fn);
LIBXSMM_API void* libxsmm_malloc(size_t size);                                /* include/libxsmm_malloc.h:73 */
LIBXSMM_API void* libxsmm_aligned_malloc(size_t size, size_t alignment);      /* :67 */
LIBXSMM_API void libxsmm_free(const void* memory);                            /* :89 */
LIBXSMM_API unsigned char libxsmm_typesize(libxsmm_datatype datatype);        /* src/template/libxsmm.h */
LIBXSMM_API libxsmm_timer_tickint libxsmm_timer_tick(void);                   /* include/libxsmm_timer.h:45 */
LIBXSMM_API libxsmm_timer_tickint libxsmm_timer_cycles(libxsmm_timer_tickint tick0, libxsmm_timer_tickint tick1); /* :48 */
LIBXSMM_API double libxsmm_timer_duration(libxsmm_timer_tickint tick0, libxsmm_timer_tickint tick1);            /* :51 */
LIBXSMM_API void libxsmm_rng_set_seed(unsigned int seed);                     /* include/libxsmm_rng.h:40 */
LIBXSMM_API double libxsmm_rng_f64(void);                                     /* :58 */
LIBXSMM_API unsigned int libxsmm_rng_u32(unsigned int n);                     /* :53 */
LIBXSMM_API void libxsmm_rng_f32_seq(float* rngs, libxsmm_blasint count);     /* :47 */
LIBXSMM_API size_t libxsmm_shuffle(unsigned int n);                           /* include/libxsmm_math.h:99 */
LIBXSMM_API unsigned int libxsmm_isqrt_u64(unsigned long long x);             /* :102 */
LIBXSMM_API unsigned int libxsmm_isqrt_u32(unsigned int x);                   /* :104 */
LIBXSMM_API unsigned int libxsmm_icbrt_u64(unsigned long long x);             /* :112 floor(cbrt(x)) */
LIBXSMM_API unsigned int libxsmm_icbrt_u32(unsigned int x);                   /* :114 */
LIBXSMM_API float libxsmm_sexp2(float x);                                      /* :121 2^x */

/* ---------------------------------------------------------------------------------------------
 * text generators (include/libxsmm_generator.h:120-215, src/generator_spgemm.c, src/generator_gemm.c). On this target
 * the text is HIP source for gfx950: the sparsity pattern (and, for the csr_reg form, the values) become code, one
 * statement per non-zero, exactly as in the reference's C text -- but the emitted kernel walks a batch
 * (arguments A, B, C as in the reference, then stride_dense, stride_c, batch in elements/items).
 * Error codes are the reference's (src/generator_common.h:267-320: 90007 lda, 90008 ldb, 90009 ldc, 90010 "which operand
 * is sparse?", 90011-90015/90035-90039 reader errors, 90049 datatype); libxsmm_strerror translates them.
 * --------------------------------------------------------------------------------------------- */
typedef struct libxsmm_generated_code { /* include/libxsmm_generator.h:248-265 */
  void* generated_code;       /* malloc'ed, NUL-terminated text (the caller frees it) */
  unsigned int buffer_size;   /* bytes allocated */
  unsigned int code_size;     /* bytes used */
  unsigned int code_type;     /* 0: source text (the only form generated here); > 1 is rejected */
  unsigned int last_error;    /* 0, or an error code for libxsmm_strerror */
} libxsmm_generated_code;
LIBXSMM_API const char* libxsmm_strerror(unsigned int i_error_code);          /* include/libxsmm_generator.h:271 */
/* dense kernel: complete translation unit, kernel name "xsmm_smm_op" / i_routine_name */
LIBXSMM_API void libxsmm_generator_gemm_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc, const char* i_arch);
LIBXSMM_API void libxsmm_generator_gemm_inlineasm(const char* i_file_out, const char* i_routine_name, const libxsmm_gemm_descriptor* i_xgemm_desc, const char* i_arch);
LIBXSMM_API void libxsmm_generator_gemm_directasm(const char* i_file_out, const char* i_routine_name, const libxsmm_gemm_descriptor* i_xgemm_desc, const char* i_arch);
/* sparse kernels: lda == 0 marks A as the sparse operand, ldb == 0 marks B (src/generator_spgemm.c:55-145). The *_kernel
 * entry points append the statements of the kernel body (csr_reg: a complete translation unit);
 * libxsmm_generator_spgemm reads a MatrixMarket file (i_is_csr: 0 CSC, 1 CSR, 3 CSR with baked-in values), adds the
 * signature and appends the kernel to i_file_out; like the reference it terminates the process on errors. */
LIBXSMM_API void libxsmm_generator_spgemm(const char* i_file_out, const char* i_routine_name, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const char* i_file_in, const int i_is_csr);
LIBXSMM_API void libxsmm_generator_spgemm_csc_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const double* i_values);
LIBXSMM_API void libxsmm_generator_spgemm_csr_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const double* i_values);
LIBXSMM_API void libxsmm_generator_spgemm_csr_reg_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const double* i_values);
/* SOA forms (operands [row][col][v], v = libxsmm_amd_soa_width innermost): CSR with A or B sparse, CSC with B sparse */
LIBXSMM_API void libxsmm_generator_spgemm_csr_soa_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const void* i_values);
LIBXSMM_API void libxsmm_generator_spgemm_csc_soa_kernel(libxsmm_generated_code* io_generated_code, const libxsmm_gemm_descriptor* i_xgemm_desc,
  const char* i_arch, const unsigned int* i_row_idx, const unsigned int* i_column_idx, const void* i_values);

typedef struct libxsmm_matdiff_info { /* include/libxsmm_math.h:40-55 */
  double norm1_abs, norm1_rel;
  double normi_abs, normi_rel;
  double normf_rel;
  double linf_abs, linf_rel, l2_abs, l2_rel;
  double l1_ref, min_ref, max_ref, avg_ref, var_ref;
  double l1_tst, min_tst, max_tst, avg_tst, var_tst;
  libxsmm_blasint m, n;
} libxsmm_matdiff_info;
LIBXSMM_API int libxsmm_matdiff(libxsmm_matdiff_info* info, libxsmm_datatype datatype, libxsmm_blasint m, libxsmm_blasint n,
  const void* ref, const void* tst, const libxsmm_blasint* ldref, const libxsmm_blasint* ldtst); /* :62 */
LIBXSMM_API void libxsmm_matdiff_reduce(libxsmm_matdiff_info* output, const libxsmm_matdiff_info* input); /* :69 */
LIBXSMM_API void libxsmm_matdiff_clear(libxsmm_matdiff_info* info);                                       /* :71 */
/* prints a GEMM call's arguments to `ostream` (a FILE*), include/libxsmm_frontend.h:471-482; ostream == NULL (the
 * reference's MHD image dump of the operands) is a no-op here */
LIBXSMM_API void libxsmm_gemm_print(void* ostream, libxsmm_gemm_precision precision, const char* transa, const char* transb,
  const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k, const void* alpha, const void* a, const libxsmm_blasint* lda,
  const void* b, const libxsmm_blasint* ldb, const void* beta, void* c, const libxsmm_blasint* ldc);
LIBXSMM_API void libxsmm_gemm_print2(void* ostream, libxsmm_gemm_precision iprec, libxsmm_gemm_precision oprec,
  const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k,
  const void* alpha, const void* a, const libxsmm_blasint* lda, const void* b, const libxsmm_blasint* ldb,
  const void* beta, void* c, const libxsmm_blasint* ldc);

/* ---------------------------------------------------------------------------------------------
 * Frontend conveniences the reference's sample programs lean on (include/libxsmm_macros.h, include/libxsmm_frontend.h,
 * include/libxsmm_cpuid.h:40-52, include/libxsmm_timer.h:36): type-name pasting, pragmas, multi-dimensional views.
 * Own definitions with the reference's names and meaning.
 * --------------------------------------------------------------------------------------------- */
#define LIBXSMM_X86_GENERIC 1002
#define LIBXSMM_X86_SSE3 1003
#define LIBXSMM_X86_SSE4 1004
#define LIBXSMM_X86_AVX 1005
#define LIBXSMM_X86_AVX2 1006
#define LIBXSMM_X86_AVX512 1007
#define LIBXSMM_X86_AVX512_MIC 1010
#define LIBXSMM_X86_AVX512_KNM 1011
#define LIBXSMM_X86_AVX512_CORE 1020
#define LIBXSMM_X86_AVX512_CLX 1021
#define LIBXSMM_X86_AVX512_CPX 1022
#define libxsmm_timer_diff(TICK0, TICK1) libxsmm_timer_cycles(TICK0, TICK1)
#define LIBXSMM_GEMM_CONST const
#define LIBXSMM_PRAGMA(DIRECTIVE) _Pragma(LIBXSMM_STRINGIFY(DIRECTIVE))
#define LIBXSMM_PRAGMA_SIMD
#define LIBXSMM_PRAGMA_UNROLL
#define LIBXSMM_PRAGMA_UNROLL_N(N)
#define LIBXSMM_PRAGMA_LOOP_COUNT(MIN, MAX, AVG)
#define LIBXSMM_PRAGMA_NONTEMPORAL
#define LIBXSMM_OPENMP_COLLAPSE(N) collapse(N)
#define LIBXSMM_OMP_VAR(A) LIBXSMM_UNUSED(A)
#define LIBXSMM_VERSION2(MAJOR, MINOR) ((MAJOR) * 10000 + (MINOR) * 100)
#define LIBXSMM_VERSION3(MAJOR, MINOR, UPDATE) (LIBXSMM_VERSION2(MAJOR, MINOR) + (UPDATE))
/* d/s prefixes and type predicates: LIBXSMM_MMFUNCTION_TYPE(double) -> libxsmm_dmmfunction, LIBXSMM_EQUAL(float, float) -> 1 */
#define LIBXSMM_TPREFIX_double d
#define LIBXSMM_TPREFIX_float s
#define LIBXSMM_TPREFIX(TYPE, FUNCTION) LIBXSMM_CONCATENATE(LIBXSMM_CONCATENATE(LIBXSMM_TPREFIX_, TYPE), FUNCTION)
#define LIBXSMM_MMFUNCTION_TYPE(TYPE) LIBXSMM_CONCATENATE(libxsmm_, LIBXSMM_TPREFIX(TYPE, mmfunction))
#define LIBXSMM_MMDISPATCH_SYMBOL(TYPE) LIBXSMM_CONCATENATE(libxsmm_, LIBXSMM_TPREFIX(TYPE, mmdispatch))
#define LIBXSMM_XGEMM_SYMBOL(TYPE) LIBXSMM_CONCATENATE(libxsmm_, LIBXSMM_TPREFIX(TYPE, gemm))
#define LIBXSMM_XBLAS_SYMBOL(TYPE) LIBXSMM_CONCATENATE(libxsmm_blas_, LIBXSMM_TPREFIX(TYPE, gemm))
#define LIBXSMM_TPREFIX_doubledouble d
#define LIBXSMM_TPREFIX_floatfloat s
#define LIBXSMM_TPREFIX_shortfloat ws
#define LIBXSMM_TPREFIX_shortint wi
#define LIBXSMM_TPREFIX2(ITYPE, OTYPE, FUNCTION) LIBXSMM_CONCATENATE(LIBXSMM_CONCATENATE(LIBXSMM_TPREFIX_, LIBXSMM_CONCATENATE(ITYPE, OTYPE)), FUNCTION)
#define LIBXSMM_MMFUNCTION_TYPE2(ITYPE, OTYPE) LIBXSMM_CONCATENATE(libxsmm_, LIBXSMM_TPREFIX2(ITYPE, OTYPE, mmfunction))
#define LIBXSMM_MMDISPATCH_SYMBOL2(ITYPE, OTYPE) LIBXSMM_CONCATENATE(libxsmm_, LIBXSMM_TPREFIX2(ITYPE, OTYPE, mmdispatch))
#if defined(__cplusplus)
# define LIBXSMM_EXTERN extern "C"
#else
# define LIBXSMM_EXTERN extern
#endif
#define LIBXSMM_BLAS_INIT /* (the reference pins its BLAS to one thread here) */
/* include/libxsmm_generator.h:36-41: what the SMM kernels cover (everything else is the BLAS domain of libxsmm_?gemm) */
#define LIBXSMM_GEMM_NO_BYPASS(FLAGS, ALPHA, BETA) ( \
  0 == ((FLAGS) & (LIBXSMM_GEMM_FLAG_TRANS_A)) && (LIBXSMM_FEQ(1, ALPHA) /*|| LIBXSMM_FEQ(-1, ALPHA)*/) && \
  (LIBXSMM_FEQ(1, BETA) || LIBXSMM_FEQ(0, BETA)))
#define LIBXSMM_GEMM_NO_BYPASS_DIMS(M, N, K) (0x7FFFFFFF >= (M) && 0x7FFFFFFF >= (N) && 0x7FFFFFFF >= (K))
/* include/libxsmm_macros.h:407-440 */
#define LIBXSMM_DELTA(T0, T1) ((T0) < (T1) ? ((T1) - (T0)) : ((T0) - (T1)))
#define LIBXSMM_ROUNDX(TYPE, A) ((TYPE)((long long)(0 <= (A) ? ((double)(A) + 0.5) : ((double)(A) - 0.5))))
#define LIBXSMM_ROUND(A) round(A)
#define LIBXSMM_ROUNDF(A) roundf(A)
#define LIBXSMM_EXP2F(A) exp2f(A)
#define LIBXSMM_SQRTF(A) sqrtf(A)
#define LIBXSMM_TANHF(A) tanhf(A)
#define LIBXSMM_YGEMM_SYMBOL(TYPE) LIBXSMM_XGEMM_SYMBOL(TYPE) /* (the reference appends _omp when OpenMP is on: one device path here) */
#define LIBXSMM_FSYMBOL(SYMBOL) LIBXSMM_CONCATENATE(SYMBOL, _)
#define LIBXSMM_BLAS_SYMBOL(TYPE, KIND) LIBXSMM_FSYMBOL(LIBXSMM_TPREFIX(TYPE, KIND))
#define LIBXSMM_EQUAL_doubledouble 1
#define LIBXSMM_EQUAL_floatfloat 1
#define LIBXSMM_EQUAL_doublefloat 0
#define LIBXSMM_EQUAL_floatdouble 0
#define LIBXSMM_EQUAL(T1, T2) LIBXSMM_CONCATENATE(LIBXSMM_CONCATENATE(LIBXSMM_EQUAL_, T1), T2)
#define LIBXSMM_TYPEINFO_FP_double 1
#define LIBXSMM_TYPEINFO_FP_float 1
#define LIBXSMM_TYPEINFO_FP_int 0
#define LIBXSMM_TYPEINFO_FP_short 0
#define LIBXSMM_TYPEINFO_FP_char 0
#define LIBXSMM_TYPEINFO(TYPE, INFO) LIBXSMM_CONCATENATE(LIBXSMM_CONCATENATE(LIBXSMM_CONCATENATE(LIBXSMM_TYPEINFO_, INFO), _), TYPE)
/* LIBXSMM_INLINE_XGEMM / LIBXSMM_XGEMM / LIBXSMM_BLAS_XGEMM: one GEMM for a type given as a token (the reference inlines
 * CPU loops for the first; there is no CPU compute path here, all three reach the engine) */
#define LIBXSMM_XGEMM(ITYPE, OTYPE, TRANSA, TRANSB, M, N, K, ALPHA, A, LDA, B, LDB, BETA, C, LDC) \
  LIBXSMM_XGEMM_SYMBOL(ITYPE)(TRANSA, TRANSB, M, N, K, ALPHA, A, LDA, B, LDB, BETA, C, LDC)
#define LIBXSMM_INLINE_XGEMM LIBXSMM_XGEMM
#define LIBXSMM_BLAS_XGEMM(ITYPE, OTYPE, TRANSA, TRANSB, M, N, K, ALPHA, A, LDA, B, LDB, BETA, C, LDC) \
  LIBXSMM_XBLAS_SYMBOL(ITYPE)(TRANSA, TRANSB, M, N, K, ALPHA, A, LDA, B, LDB, BETA, C, LDC)
/* Fortran BLAS symbols some samples use for their gold results (a BLAS library must then be linked by the sample) */
#define LIBXSMM_GEMM_SYMBOL(TYPE) LIBXSMM_CONCATENATE(LIBXSMM_TPREFIX(TYPE, gemm), _)
#define LIBXSMM_BLAS_SYMBOL_DECL(TYPE, KIND) LIBXSMM_EXTERN_C void LIBXSMM_CONCATENATE(LIBXSMM_TPREFIX(TYPE, KIND), _)( \
  const char*, const char*, const libxsmm_blasint*, const libxsmm_blasint*, const libxsmm_blasint*, const TYPE*, const TYPE*, \
  const libxsmm_blasint*, const TYPE*, const libxsmm_blasint*, const TYPE*, TYPE*, const libxsmm_blasint*);
/* Views of a flat array as an N-dimensional one (N <= 5): LIBXSMM_VLA_DECL(3, T, v, ptr, d1, d2) then
 * LIBXSMM_VLA_ACCESS(3, v, i0, i1, i2, d1, d2) is ptr[(i0*d1 + i1)*d2 + i2] (row-major, the leading extent is implied). */
#define LIBXSMM_VLA_POSTFIX _
#define LIBXSMM_VLA_DECL(NDIMS, ELEMENT_TYPE, ARRAY_VAR, ...) \
  ELEMENT_TYPE *LIBXSMM_RESTRICT LIBXSMM_CONCATENATE(ARRAY_VAR, LIBXSMM_VLA_POSTFIX) = LIBXSMM_VLA_INIT(__VA_ARGS__, 0)
#define LIBXSMM_VLA_INIT(INIT, ...) INIT
#define LIBXSMM_VLA_INDEX_1(I0) ((size_t)(I0))
#define LIBXSMM_VLA_INDEX_2(I0, I1, S1) ((size_t)(I0) * (S1) + (I1))
#define LIBXSMM_VLA_INDEX_3(I0, I1, I2, S1, S2) (LIBXSMM_VLA_INDEX_2(I0, I1, S1) * (S2) + (I2))
#define LIBXSMM_VLA_INDEX_4(I0, I1, I2, I3, S1, S2, S3) (LIBXSMM_VLA_INDEX_3(I0, I1, I2, S1, S2) * (S3) + (I3))
#define LIBXSMM_VLA_INDEX_5(I0, I1, I2, I3, I4, S1, S2, S3, S4) (LIBXSMM_VLA_INDEX_4(I0, I1, I2, I3, S1, S2, S3) * (S4) + (I4))
#define LIBXSMM_VLA_ACCESS(NDIMS, ARRAY, ...) \
  LIBXSMM_CONCATENATE(ARRAY, LIBXSMM_VLA_POSTFIX)[LIBXSMM_CONCATENATE(LIBXSMM_VLA_INDEX_, NDIMS)(__VA_ARGS__)]
#if defined(__cplusplus)
# define LIBXSMM_RESTRICT __restrict__
#else
# define LIBXSMM_RESTRICT restrict
#endif

#if defined(__cplusplus)
/* ---------------------------------------------------------------------------------------------
 * C++ convenience layer used by the C++ programs under samples/smm (src/template/libxsmm.h:416-511): precision traits and the
 * libxsmm_mmfunction<> functor (dispatch in the constructor, call through operator()).
 * --------------------------------------------------------------------------------------------- */
template<typename T> struct libxsmm_gemm_precision_enum { static const libxsmm_gemm_precision value = static_cast<libxsmm_gemm_precision>(LIBXSMM_DATATYPE_UNSUPPORTED); };
template<> struct libxsmm_gemm_precision_enum<double> { static const libxsmm_gemm_precision value = LIBXSMM_GEMM_PRECISION_F64; };
template<> struct libxsmm_gemm_precision_enum<float> { static const libxsmm_gemm_precision value = LIBXSMM_GEMM_PRECISION_F32; };
template<> struct libxsmm_gemm_precision_enum<int> { static const libxsmm_gemm_precision value = LIBXSMM_GEMM_PRECISION_I32; };
template<> struct libxsmm_gemm_precision_enum<short> { static const libxsmm_gemm_precision value = LIBXSMM_GEMM_PRECISION_I16; };
template<> struct libxsmm_gemm_precision_enum<libxsmm_bfloat16> { static const libxsmm_gemm_precision value = LIBXSMM_GEMM_PRECISION_BF16; };
template<typename INP_TYPE> struct libxsmm_gemm_default_output { typedef INP_TYPE type; };
template<> struct libxsmm_gemm_default_output<short> { typedef int type; };

template<typename INP_TYPE, typename OUT_TYPE = typename libxsmm_gemm_default_output<INP_TYPE>::type>
class libxsmm_mmfunction {
  mutable libxsmm_xmmfunction m_function;
  void dispatch(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc,
                const OUT_TYPE* alpha, const OUT_TYPE* beta, int prefetch) {
    libxsmm_descriptor_blob blob;
    const int strategy = (0 > prefetch ? (int)libxsmm_get_gemm_auto_prefetch() : prefetch); /* recorded, no effect on gfx950 */
    const libxsmm_gemm_descriptor* const desc = libxsmm_gemm_descriptor_init2(&blob,
      libxsmm_gemm_precision_enum<INP_TYPE>::value, libxsmm_gemm_precision_enum<OUT_TYPE>::value,
      m, n, k, lda, ldb, ldc, alpha, beta, flags, strategy);
    m_function.xmm = (0 != desc ? libxsmm_xmmdispatch(desc).xmm : 0);
  }
public:
  typedef INP_TYPE itype;
  typedef OUT_TYPE otype;
  libxsmm_mmfunction() { m_function.xmm = 0; }
  libxsmm_mmfunction(libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, int flags = LIBXSMM_FLAGS) { dispatch(flags, m, n, k, m, k, m, 0, 0, LIBXSMM_PREFETCH); }
  libxsmm_mmfunction(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, int prefetch) { dispatch(flags, m, n, k, m, k, m, 0, 0, prefetch); }
  libxsmm_mmfunction(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, otype alpha, otype beta) { dispatch(flags, m, n, k, m, k, m, &alpha, &beta, LIBXSMM_PREFETCH); }
  libxsmm_mmfunction(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, otype alpha, otype beta, int prefetch) { dispatch(flags, m, n, k, m, k, m, &alpha, &beta, prefetch); }
  libxsmm_mmfunction(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
    libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, int prefetch) { dispatch(flags, m, n, k, lda, ldb, ldc, 0, 0, prefetch); }
  libxsmm_mmfunction(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
    libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, otype alpha, otype beta) { dispatch(flags, m, n, k, lda, ldb, ldc, &alpha, &beta, LIBXSMM_PREFETCH); }
  libxsmm_mmfunction(int flags, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k,
    libxsmm_blasint lda, libxsmm_blasint ldb, libxsmm_blasint ldc, otype alpha, otype beta, int prefetch) { dispatch(flags, m, n, k, lda, ldb, ldc, &alpha, &beta, prefetch); }
  const libxsmm_xmmfunction& kernel() const { return m_function; }
  operator const void*() const { return 0 != m_function.xmm ? this : 0; }
  void operator()(const itype* a, const itype* b, otype* c) const { LIBXSMM_MMCALL_ABC(m_function.xmm, a, b, c); }
  void operator()(const itype* a, const itype* b, otype* c, const itype* pa, const itype* pb, const otype* pc) const { LIBXSMM_MMCALL_PRF(m_function.xmm, a, b, c, pa, pb, pc); }
};
/* overloads by element type, m/n/k by pointer or by value (src/template/libxsmm.h:619-720) */
#define LIBXSMM_CXX_GEMM(NAME, TYPE, TARGET) \
inline void NAME(const char* transa, const char* transb, const libxsmm_blasint* m, const libxsmm_blasint* n, const libxsmm_blasint* k, \
  const TYPE* alpha, const TYPE* a, const libxsmm_blasint* lda, const TYPE* b, const libxsmm_blasint* ldb, const TYPE* beta, TYPE* c, const libxsmm_blasint* ldc) \
{ TARGET(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc); } \
inline void NAME(const char* transa, const char* transb, libxsmm_blasint m, libxsmm_blasint n, libxsmm_blasint k, \
  const TYPE* alpha, const TYPE* a, const libxsmm_blasint* lda, const TYPE* b, const libxsmm_blasint* ldb, const TYPE* beta, TYPE* c, const libxsmm_blasint* ldc) \
{ TARGET(transa, transb, &m, &n, &k, alpha, a, lda, b, ldb, beta, c, ldc); }
LIBXSMM_CXX_GEMM(libxsmm_gemm, double, libxsmm_dgemm)
LIBXSMM_CXX_GEMM(libxsmm_gemm, float, libxsmm_sgemm)
LIBXSMM_CXX_GEMM(libxsmm_blas_gemm, double, libxsmm_blas_dgemm)
LIBXSMM_CXX_GEMM(libxsmm_blas_gemm, float, libxsmm_blas_sgemm)
#endif /* __cplusplus */

#include "libxsmm_amd.h"

#endif /* LIBXSMM_H */
