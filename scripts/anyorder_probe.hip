// developer probe: do two kernels launched on ONE stream with hipExtAnyOrderLaunch overlap on gfx950?
// (hip_ext.h says the flag is not supported on GFX9xx boards for hipExtModuleLaunchKernel)
//   hipcc --offload-arch=gfx950 -O2 scripts/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(long long ticks, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}

static double run(int flags, int n, hipStream_t st, int* out) {
  hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n; ++k) hipExtLaunchKernelGGL(spin_kernel, dim3(4), dim3(64), 0, st, nullptr, nullptr, flags, 2000000LL /* 20 ms at 100 MHz */, out);
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  int* out = nullptr;
  hipMalloc(&out, 64);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  run(0, 1, st, out);
  printf("one kernel: %.1f ms\n", run(0, 1, st, out));
  printf("4 kernels, in order:  %.1f ms\n", run(0, 4, st, out));
  printf("4 kernels, any order: %.1f ms\n", run(hipExtAnyOrderLaunch, 4, st, out));
  hipEvent_t ev[4];
  for (auto& e : ev) hipEventCreate(&e);
  hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 4; ++k) hipExtLaunchKernelGGL(spin_kernel, dim3(4), dim3(64), 0, st, nullptr, ev[k], hipExtAnyOrderLaunch, (long long)(500000 * (4 - k)), out);
  for (int k = 3; k >= 0; --k) {
    hipEventSynchronize(ev[k]);
    printf("  stop event of kernel %d (%d ms of work) after %.1f ms\n", k, 5 * (4 - k), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  return 0;
}
