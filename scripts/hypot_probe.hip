// Developer probe: is gridmath.hpp's glibc_hypot on the device bit-identical to the host libm's hypot()?
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I ros_navigation_amd/csrc -I include scripts/hypot_probe.hip -o gpurun_out/hypot_probe
#include "gridmath.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace rna;
__global__ void k(const double* x, const double* y, double* h, double* s, double* q, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { h[i] = glibc_hypot(x[i], y[i]); s[i] = sqrt(x[i] * x[i] + y[i] * y[i]); q[i] = x[i] / (y[i] == 0 ? 1.0 : y[i]); }
}
int main() {
  const int n = 1 << 24;
  std::vector<double> x(n), y(n), h(n), s(n), q(n);
  srand(3);
  for (int i = 0; i < n; ++i) {
    const int m = i % 3;
    if (m == 0) { x[i] = (rand() % 600 - 300) * 0.05 - (rand() % 600 - 300) * 0.05; y[i] = (rand() % 200) * 0.05 - 2.3 - ((rand() % 200) * 0.05 - 2.3); }
    else if (m == 1) { x[i] = (double)rand() / RAND_MAX * 60 - 30; y[i] = (double)rand() / RAND_MAX * 60 - 30; }
    else { x[i] = (rand() % 300 - 150) * 0.2 - (rand() % 300 - 150) * 0.2; y[i] = (rand() % 300) * 0.2 - 7.7 - ((rand() % 300) * 0.2 - 7.7); }
  }
  double *dx, *dy, *dh, *ds, *dq;
  hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dh, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dq, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dy, dh, ds, dq, n);
  hipMemcpy(h.data(), dh, n * 8, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
  long bh = 0, bs = 0, bq = 0;
  for (int i = 0; i < n; ++i) {
    const double hh = hypot(x[i], y[i]);
    volatile double t = x[i] * x[i]; volatile double u = y[i] * y[i]; volatile double w = t + u;
    const double ss = sqrt(w), qq = x[i] / (y[i] == 0 ? 1.0 : y[i]);
    if (memcmp(&hh, &h[i], 8)) { if (bh < 3) printf("hypot %.17g %.17g host %.17g dev %.17g\n", x[i], y[i], hh, h[i]); bh++; }
    if (memcmp(&ss, &s[i], 8)) { if (bs < 3) printf("sqrt  %.17g %.17g host %.17g dev %.17g\n", x[i], y[i], ss, s[i]); bs++; }
    if (memcmp(&qq, &q[i], 8)) bq++;
  }
  printf("of %d: hypot differs %ld, sqrt(x*x+y*y) differs %ld, division differs %ld\n", n, bh, bs, bq);
  return 0;
}
