// Developer microbenchmark: how many wave64 VALU instructions one gfx950 SIMD issues per cycle -- the ceiling the
// register-resident A* sweeps (astar_tile.hip) are priced against.  Every wavefront runs a loop of
// independent integer VALU instructions of the kinds the sweeps are made of and times it with s_memtime (shader
// cycles); W wavefronts share a SIMD, so the SIMD's rate is W x instructions / cycles.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o ubench_valu && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

// 16 independent chains of one VALU instruction each per macro; 64 per loop iteration
#define PLAIN_MAX(r) "v_max_i32 %" #r ", %" #r ", %16\n"
#define BODY_MAX PLAIN_MAX(0) PLAIN_MAX(1) PLAIN_MAX(2) PLAIN_MAX(3) PLAIN_MAX(4) PLAIN_MAX(5) PLAIN_MAX(6) PLAIN_MAX(7) \
                 PLAIN_MAX(8) PLAIN_MAX(9) PLAIN_MAX(10) PLAIN_MAX(11) PLAIN_MAX(12) PLAIN_MAX(13) PLAIN_MAX(14) PLAIN_MAX(15)
#define DPP_ADD(r) "v_add_u32_dpp %" #r ", %" #r ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define BODY_DPP DPP_ADD(0) DPP_ADD(1) DPP_ADD(2) DPP_ADD(3) DPP_ADD(4) DPP_ADD(5) DPP_ADD(6) DPP_ADD(7) \
                 DPP_ADD(8) DPP_ADD(9) DPP_ADD(10) DPP_ADD(11) DPP_ADD(12) DPP_ADD(13) DPP_ADD(14) DPP_ADD(15)
#define ROWDPP_MAX(r) "v_max_i32_dpp %" #r ", %" #r ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define BODY_ROWDPP ROWDPP_MAX(0) ROWDPP_MAX(1) ROWDPP_MAX(2) ROWDPP_MAX(3) ROWDPP_MAX(4) ROWDPP_MAX(5) ROWDPP_MAX(6) ROWDPP_MAX(7) \
                    ROWDPP_MAX(8) ROWDPP_MAX(9) ROWDPP_MAX(10) ROWDPP_MAX(11) ROWDPP_MAX(12) ROWDPP_MAX(13) ROWDPP_MAX(14) ROWDPP_MAX(15)
#define MAX3(r) "v_max3_i32 %" #r ", %" #r ", %16, %17\n"
#define BODY_MAX3 MAX3(0) MAX3(1) MAX3(2) MAX3(3) MAX3(4) MAX3(5) MAX3(6) MAX3(7) MAX3(8) MAX3(9) MAX3(10) MAX3(11) MAX3(12) MAX3(13) MAX3(14) MAX3(15)
#define BFE(r) "v_bfe_i32 %" #r ", %" #r ", 3, 1\n"
#define BODY_BFE BFE(0) BFE(1) BFE(2) BFE(3) BFE(4) BFE(5) BFE(6) BFE(7) BFE(8) BFE(9) BFE(10) BFE(11) BFE(12) BFE(13) BFE(14) BFE(15)
#define CMPSEL(r) "v_cmp_gt_i32 vcc, %16, %" #r "\nv_cndmask_b32 %" #r ", %" #r ", %17, vcc\n"
#define BODY_CMPSEL CMPSEL(0) CMPSEL(1) CMPSEL(2) CMPSEL(3) CMPSEL(4) CMPSEL(5) CMPSEL(6) CMPSEL(7)
#define MULLO(r) "v_mul_lo_u32 %" #r ", %" #r ", %16\n"
#define BODY_MULLO MULLO(0) MULLO(1) MULLO(2) MULLO(3) MULLO(4) MULLO(5) MULLO(6) MULLO(7) MULLO(8) MULLO(9) MULLO(10) MULLO(11) MULLO(12) MULLO(13) MULLO(14) MULLO(15)
// VALU with a scalar instruction after each (the sweeps' flag algebra): does the SALU take VALU issue slots of the same wave?
#define VS(r) "v_max_i32 %" #r ", %" #r ", %16\ns_add_u32 s40, s40, 1\n"
#define BODY_VS VS(0) VS(1) VS(2) VS(3) VS(4) VS(5) VS(6) VS(7) VS(8) VS(9) VS(10) VS(11) VS(12) VS(13) VS(14) VS(15)
// scalar only: four independent chains (the scalar unit is shared by the four SIMDs of a CU)
#define SS4 "s_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\ns_add_u32 s43, s43, 1\n"
#define BODY_S SS4 SS4 SS4 SS4
// 64-bit lane-mask algebra as in the sweeps
#define SB4 "s_or_b64 s[40:41], s[40:41], s[44:45]\ns_and_b64 s[42:43], s[42:43], s[46:47]\ns_or_b64 s[48:49], s[48:49], s[44:45]\ns_andn2_b64 s[50:51], s[50:51], s[46:47]\n"
#define BODY_SB SB4 SB4 SB4 SB4
// one VALU per three SALU
#define V3S(r) "v_max_i32 %" #r ", %" #r ", %16\ns_add_u32 s40, s40, 1\ns_add_u32 s41, s41, 1\ns_add_u32 s42, s42, 1\n"
#define BODY_V3S V3S(0) V3S(1) V3S(2) V3S(3)

#define KERNEL(name, BODY, PER_ITER)                                                                                  \
  __global__ void __launch_bounds__(256) name(int iters, int* sink, long long* cyc) {                                   \
    int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;    \
    int r8 = r0 + 8, r9 = r0 + 9, r10 = r0 + 10, r11 = r0 + 11, r12 = r0 + 12, r13 = r0 + 13, r14 = r0 + 14, r15 = r0 + 15; \
    int a = blockIdx.x, b = blockIdx.x * 3;                                                                             \
    int s = 0;                                                                                                          \
    __syncthreads();                                                                                                    \
    const long long t0 = __builtin_amdgcn_s_memtime();                                                                  \
    for (int i = 0; i < iters; ++i) {                                                                                   \
      asm volatile(REP4(BODY)                                                                                           \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(r8), "+v"(r9), \
                     "+v"(r10), "+v"(r11), "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15)                                   \
                   : "v"(a), "v"(b), "s"(s)                                                                             \
                   : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");                                                                                            \
    }                                                                                                                   \
    const long long t1 = __builtin_amdgcn_s_memtime();                                                                  \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                    \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13 + r14 + r15; \
  }                                                                                                                     \
  static const int name##_per_iter = PER_ITER;

// the "s" operand is read-write inside BODY_VS; declared as input only because the value is never used afterwards
KERNEL(k_max, BODY_MAX, 64)
KERNEL(k_dpp, BODY_DPP, 64)
KERNEL(k_rowdpp, BODY_ROWDPP, 64)
KERNEL(k_max3, BODY_MAX3, 64)
KERNEL(k_bfe, BODY_BFE, 64)
KERNEL(k_cmpsel, BODY_CMPSEL, 64)   // 8 x (cmp + cndmask) x 4 = 64 VALU
KERNEL(k_mullo, BODY_MULLO, 64)
KERNEL(k_vs, BODY_VS, 64)           // 64 VALU + 64 SALU
KERNEL(k_s, BODY_S, 64)             // 64 SALU
KERNEL(k_sb, BODY_SB, 64)           // 64 SALU (64-bit)
KERNEL(k_v3s, BODY_V3S, 64)         // 16 VALU + 48 SALU

template <class K>
static void run(const char* what, K kernel, int per_iter, int waves_per_simd, int* sink, long long* cyc, double extra_salu = 0.0) {
  // 256-thread blocks = one wave per SIMD each; W blocks per CU -> W waves per SIMD.  256 CUs.
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, 10, sink, cyc);   // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, iters, sink, cyc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double instr = (double)iters * per_iter;
  // s_memtime counts at a fixed 100 MHz on some parts and at the shader clock on others: report both readings
  printf("%-34s W=%d  per wave: %.0f ticks for %.0f instr  ->  SIMD rate %.3f instr/tick (x W)  | wall %.3f ms -> %.3f instr/ns/SIMD%s\n", what,
         waves_per_simd, med, instr, waves_per_simd * instr / med, ms, (double)blocks * 4 * instr / (ms * 1e6) / (256.0 * 4.0),
         extra_salu > 0 ? "  (+ as many SALU)" : "");
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  int* sink;
  long long* cyc;
  hipMalloc(&sink, 256 * 8 * 256 * sizeof(int));
  hipMalloc(&cyc, 256 * 8 * 4 * sizeof(long long));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clockRate %d kHz; wall figures assume every SIMD of 256 CUs busy\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 2, 4, 8}) {
    run("v_max_i32", k_max, k_max_per_iter, w, sink, cyc);
    run("v_add_u32_dpp wave_shr:1", k_dpp, k_dpp_per_iter, w, sink, cyc);
    run("v_max_i32_dpp row_shr:1", k_rowdpp, k_rowdpp_per_iter, w, sink, cyc);
    run("v_max3_i32", k_max3, k_max3_per_iter, w, sink, cyc);
    run("v_bfe_i32", k_bfe, k_bfe_per_iter, w, sink, cyc);
    run("v_cmp + v_cndmask", k_cmpsel, k_cmpsel_per_iter, w, sink, cyc);
    run("v_mul_lo_u32", k_mullo, k_mullo_per_iter, w, sink, cyc);
    run("v_max_i32 ; s_add_u32 alternating (instr = 64 VALU + 64 SALU)", k_vs, k_vs_per_iter, w, sink, cyc, 1.0);
    run("s_add_u32 only (instr = SALU)", k_s, k_s_per_iter, w, sink, cyc);
    run("s_or/and_b64 only (instr = SALU)", k_sb, k_sb_per_iter, w, sink, cyc);
    run("1 v_max : 3 s_add (instr = all 64)", k_v3s, k_v3s_per_iter, w, sink, cyc);
  }
  return 0;
}
