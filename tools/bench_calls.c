/* Per-call cost of a dispatched kernel on device operands (the reference's canonical usage, samples/smm/specialized.cpp):
 * calls/s of kernel(a_i, b_i, c_i) issued back to back, then one synchronisation. Default: a launch per call (stream order is
 * call order whatever the caller queues in between); LIBXSMM_AMD_DEFER=1 or libxsmm_amd_defer_begin/end: recorded into bursts.
 * usage: bench_calls [m n k [calls]]
 * Build: gcc -O2 -I include tools/bench_calls.c -o /tmp/bench_calls -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib */
#include <libxsmm.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char* argv[])
{
  const int m = (1 < argc ? atoi(argv[1]) : 23), n = (2 < argc ? atoi(argv[2]) : m), k = (3 < argc ? atoi(argv[3]) : m), s = (4 < argc ? atoi(argv[4]) : 20000);
  const size_t asz = (size_t)m * k, bsz = (size_t)k * n, csz = (size_t)m * n;
  double *a, *b, *c;
  libxsmm_dmmfunction kernel;
  libxsmm_timer_tickint t0;
  double dt;
  int i, rep;
  libxsmm_init();
  a = (double*)libxsmm_amd_device_malloc(sizeof(double) * asz * s);
  b = (double*)libxsmm_amd_device_malloc(sizeof(double) * bsz * s);
  c = (double*)libxsmm_amd_device_malloc(sizeof(double) * csz * s);
  kernel = libxsmm_dmmdispatch(m, n, k, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
  if (NULL == a || NULL == b || NULL == c || NULL == kernel) return 1;
  kernel(a, b, c); libxsmm_amd_synchronize(); /* one-time costs (code object load, staging rings) are not per-call costs */
  for (rep = 0; rep < 3; ++rep) {
    t0 = libxsmm_timer_tick();
    for (i = 0; i < s; ++i) kernel(a + i * asz, b + i * bsz, c + i * csz);
    dt = libxsmm_timer_duration(t0, libxsmm_timer_tick());
    libxsmm_amd_synchronize();
    printf("%dx%dx%d fp64, %d calls issued in %.1f ms: %.2f us per call (%.1f ms until the GPU is done)%s\n", m, n, k, s, dt * 1e3, dt * 1e6 / s,
      libxsmm_timer_duration(t0, libxsmm_timer_tick()) * 1e3, libxsmm_amd_defer_active() ? " [recorded into bursts]" : " [a launch per call]");
  }
  libxsmm_amd_device_free(a); libxsmm_amd_device_free(b); libxsmm_amd_device_free(c);
  libxsmm_finalize();
  return 0;
}
