/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_spmdm.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_SPMDM_H_COMPAT
#define LIBXSMM_SPMDM_H_COMPAT
#include "libxsmm.h"
#endif
