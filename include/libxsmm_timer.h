/* Compatibility header: the reference splits its interface over several headers (include/libxsmm_timer.h there); everything
 * this engine provides is declared in libxsmm.h. */
#ifndef LIBXSMM_TIMER_H_COMPAT
#define LIBXSMM_TIMER_H_COMPAT
#include "libxsmm.h"
#endif
