#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  int* d; hipMalloc(&d, 1024 * 4);
  hipLaunchKernelGGL(k, dim3(512), dim3(1024), 0, 0, d);
  int h[512]; hipMemcpy(h, d, 512 * 4, hipMemcpyDeviceToHost);
  int cnt[16] = {0};
  for (int i = 0; i < 512; ++i) cnt[h[i] & 15]++;
  printf("raw first 16:"); for (int i = 0; i < 16; ++i) printf(" %x", h[i]); printf("\ncounts by (x&15):");
  for (int i = 0; i < 16; ++i) printf(" %d", cnt[i]); printf("\n");
  return 0;
}
