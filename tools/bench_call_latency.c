/* Latency of ONE dispatched-kernel call followed by a wait for the device (the opposite of tools/bench_calls.c): what a burst
 * of one call costs on top of a plain launch. Build: gcc -O2 -I include tools/bench_call_latency.c -I/opt/rocm/include -o /tmp/bench_call_latency -L libxsmm-1_amd/lib -lxsmm -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/libxsmm-1_amd/lib */
#define __HIP_PLATFORM_AMD__
#include <hip/hip_runtime_api.h>
#include <libxsmm.h>
#include <stdio.h>

int main(void)
{
  const int m = 23, n = 23, k = 23, reps = 2000;
  double *a, *b, *c, dt;
  libxsmm_dmmfunction kernel;
  libxsmm_timer_tickint t0;
  int i;
  libxsmm_init();
  a = (double*)libxsmm_amd_device_malloc(sizeof(double) * m * k); b = (double*)libxsmm_amd_device_malloc(sizeof(double) * k * n);
  c = (double*)libxsmm_amd_device_malloc(sizeof(double) * m * n);
  kernel = libxsmm_dmmdispatch(m, n, k, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
  if (NULL == a || NULL == b || NULL == c || NULL == kernel) return 1;
  kernel(a, b, c); libxsmm_amd_synchronize();
  t0 = libxsmm_timer_tick();
  for (i = 0; i < reps; ++i) { kernel(a, b, c); libxsmm_amd_synchronize(); }
  dt = libxsmm_timer_duration(t0, libxsmm_timer_tick());
  printf("call + libxsmm_amd_synchronize: %.1f us per iteration\n", dt * 1e6 / reps);
  t0 = libxsmm_timer_tick();
  for (i = 0; i < reps; ++i) { kernel(a, b, c); (void)hipDeviceSynchronize(); }
  dt = libxsmm_timer_duration(t0, libxsmm_timer_tick());
  printf("call + hipDeviceSynchronize (no library call in between): %.1f us per iteration\n", dt * 1e6 / reps);
  libxsmm_finalize();
  return 0;
}
