/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_malloc.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_MALLOC_H_COMPAT
#define LIBXSMM_MALLOC_H_COMPAT
#include "libxsmm.h"
#endif
