/* The PyFR driver's loop on device memory (samples/pyfr/pyfr_driver_asp_reg.c:300-308): libxsmm_dfsspmdm_execute once per 48-column
 * panel, then one wait. Build: gcc -O2 -I include tools/bench_panels.c -o /tmp/bench_panels -L libxsmm-1_amd/lib -lxsmm -Wl,-rpath,$PWD/libxsmm-1_amd/lib */
#include <libxsmm.h>
#include <stdio.h>
#include <stdlib.h>

int main(void)
{
  const int m = 35, k = 35, nblock = 48, panels = 3125;
  const long long n = (long long)nblock * panels;
  double *a = (double*)calloc((size_t)m * k, sizeof(double)), *b, *c, dt;
  libxsmm_dfsspmdm* h;
  libxsmm_timer_tickint t0;
  long long z; int i, rep;
  libxsmm_init();
  srand(1);
  for (i = 0; i < m * k; ++i) if (rand() % 100 < 15) a[i] = (double)(rand() % 7 + 1) * 0.25;
  b = (double*)libxsmm_amd_device_malloc(sizeof(double) * k * n); c = (double*)libxsmm_amd_device_malloc(sizeof(double) * m * n);
  h = libxsmm_dfsspmdm_create(m, nblock, k, k, (libxsmm_blasint)n, (libxsmm_blasint)n, 1.0, 1.0, a);
  if (NULL == b || NULL == c || NULL == h) return 1;
  libxsmm_dfsspmdm_execute(h, b, c); libxsmm_amd_synchronize();
  for (rep = 0; rep < 3; ++rep) {
    t0 = libxsmm_timer_tick();
    for (z = 0; z < n; z += nblock) libxsmm_dfsspmdm_execute(h, b + z, c + z);
    dt = libxsmm_timer_duration(t0, libxsmm_timer_tick());
    libxsmm_amd_synchronize();
    printf("%d panels of %d columns issued in %.2f ms: %.2f us per call (%.2f ms until the GPU is done)\n", panels, nblock, dt * 1e3, dt * 1e6 / panels,
      libxsmm_timer_duration(t0, libxsmm_timer_tick()) * 1e3);
  }
  t0 = libxsmm_timer_tick();
  libxsmm_amd_dfsspmdm_execute_batch(h, b, c, panels); libxsmm_amd_synchronize();
  printf("libxsmm_amd_dfsspmdm_execute_batch (all panels, one call): %.2f ms\n", libxsmm_timer_duration(t0, libxsmm_timer_tick()) * 1e3);
  libxsmm_dfsspmdm_destroy(h);
  libxsmm_finalize();
  return 0;
}
