/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_typedefs.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_TYPEDEFS_H_COMPAT
#define LIBXSMM_TYPEDEFS_H_COMPAT
#include "libxsmm.h"
#endif
