/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_frontend.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_FRONTEND_H_COMPAT
#define LIBXSMM_FRONTEND_H_COMPAT
#include "libxsmm.h"
#endif
