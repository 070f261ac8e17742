/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_blocked_gemm.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_BLOCKED_GEMM_H_COMPAT
#define LIBXSMM_BLOCKED_GEMM_H_COMPAT
#include "libxsmm.h"
#endif
