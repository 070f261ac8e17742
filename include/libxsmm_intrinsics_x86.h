/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_intrinsics_x86.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_INTRINSICS_X86_H_COMPAT
#define LIBXSMM_INTRINSICS_X86_H_COMPAT
#include "libxsmm.h"
#endif
