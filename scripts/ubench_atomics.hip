// Developer microbenchmark: latency/throughput of returning global atomics vs loads from ONE workgroup
// (the regime of the per-query A* search).  hipcc --offload-arch=gfx950 -O3 ubench_atomics.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void chain_atomic(int* g, const int* idx, int steps, long long* out) {
  // each lane: dependent chain of returning atomics on scattered addresses
  int p = idx[threadIdx.x];
  long long t0 = wall_clock64();
  int acc = 0;
  for (int s = 0; s < steps; ++s) {
    int old = atomicMin(&g[p], 1000000 - s);
    acc += old;
    p = (p * 1103515245 + 12345 + (old & 1)) & ((1 << 24) - 1);
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

__global__ void chain_load(int* g, const int* idx, int steps, long long* out) {
  int p = idx[threadIdx.x];
  long long t0 = wall_clock64();
  int acc = 0;
  for (int s = 0; s < steps; ++s) {
    int v = __hip_atomic_load(&g[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += v;
    p = (p * 1103515245 + 12345 + (v & 1)) & ((1 << 24) - 1);
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

template <int K, bool ATOMIC>
__global__ void burst(int* g, const int* idx, int steps, long long* out) {
  // each lane: K independent ops per step, then wait (the A* expansion pattern)
  int p = idx[threadIdx.x];
  long long t0 = wall_clock64();
  int acc = 0;
  for (int s = 0; s < steps; ++s) {
    int r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int a = (p + k * 4099) & ((1 << 24) - 1);
      if (ATOMIC) r[k] = atomicMin(&g[a], 1000000 - s);
      else r[k] = __hip_atomic_load(&g[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int x = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) x += r[k];
    acc += x;
    p = (p * 1103515245 + 12345 + (x & 1)) & ((1 << 24) - 1);
    __syncthreads();
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

int main() {
  const size_t N = 1 << 24;
  int* g; hipMalloc(&g, N * sizeof(int)); hipMemset(g, 0x7f, N * sizeof(int));
  std::vector<int> h(1024);
  srand(1);
  for (auto& v : h) v = rand() & (N - 1);
  int* idx; hipMalloc(&idx, 4096); hipMemcpy(idx, h.data(), 4096, hipMemcpyHostToDevice);
  long long* out; hipMalloc(&out, 16);
  long long ho[2];
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", clk_khz);
  const double ns_per_tick = 1e6 / clk_khz;
  const int steps = 2000;
  for (int threads : {64, 256, 1024}) {
    hipLaunchKernelGGL(chain_atomic, dim3(1), dim3(threads), 0, 0, g, idx, steps, out); hipDeviceSynchronize();
    hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
    printf("threads=%4d dependent atomicMin(ret): %.0f ns/step\n", threads, ho[0] * ns_per_tick / steps);
    hipLaunchKernelGGL(chain_load, dim3(1), dim3(threads), 0, 0, g, idx, steps, out); hipDeviceSynchronize();
    hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
    printf("threads=%4d dependent load sc1      : %.0f ns/step\n", threads, ho[0] * ns_per_tick / steps);
    hipLaunchKernelGGL((burst<8, true>), dim3(1), dim3(threads), 0, 0, g, idx, steps, out); hipDeviceSynchronize();
    hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
    printf("threads=%4d burst 8 atomics + barrier: %.0f ns/step (%.1f ns per atomic)\n", threads, ho[0] * ns_per_tick / steps, ho[0] * ns_per_tick / steps / (8.0 * threads));
    hipLaunchKernelGGL((burst<8, false>), dim3(1), dim3(threads), 0, 0, g, idx, steps, out); hipDeviceSynchronize();
    hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost);
    printf("threads=%4d burst 8 loads   + barrier: %.0f ns/step (%.1f ns per load)\n", threads, ho[0] * ns_per_tick / steps, ho[0] * ns_per_tick / steps / (8.0 * threads));
  }
  return 0;
}
