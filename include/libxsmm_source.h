/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_source.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_SOURCE_H_COMPAT
#define LIBXSMM_SOURCE_H_COMPAT
#include "libxsmm.h"
#endif
