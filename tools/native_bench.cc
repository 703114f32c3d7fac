// tools/native_bench.cc -- times twx_step_random of a library from a plain HIP process (no torch):
// is a difference against tools/ab_bench.py a property of the process (allocation placement, clocks)?
//   g++ -O2 -D__HIP_PLATFORM_AMD__ -I include -I/opt/rocm/include tools/native_bench.cc -o tools/bin/native_bench -ldl -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
//   tools/bin/native_bench lib.so [N=24] [steps=200]
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef struct twx_env twx_env;
int main(int argc, char **argv) {
  void *h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("%s\n", dlerror()); return 1; }
  auto create = (int (*)(int, int64_t, int, uint64_t, twx_env **))dlsym(h, "twx_create");
  auto step = (int (*)(twx_env *, uint32_t, int, int32_t *, uint8_t *, void *))dlsym(h, "twx_step_random");
  auto destroy = (int (*)(twx_env *))dlsym(h, "twx_destroy");
  const int n = argc > 2 ? atoi(argv[2]) : 24, steps = argc > 3 ? atoi(argv[3]) : 200;
  for (int rep = 0; rep < 3; rep++) {
    twx_env *e = nullptr;
    if (create(n, 1 << 20, 0, 0, &e)) { printf("create failed\n"); return 1; }
    for (int i = 0; i < 720; i++) step(e, 1, 1, nullptr, nullptr, nullptr);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceSynchronize();
    hipEventRecord(a, nullptr);
    for (int i = 0; i < steps; i++) step(e, 1, 1, nullptr, nullptr, nullptr);
    hipEventRecord(b, nullptr);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s N=%d: %.4f ms per step\n", argv[1], n, ms / steps);
    destroy(e);
  }
  return 0;
}
