// vfh.hip -- batched VFH+ steps on gfx950: Steerer::getRangesFromSubmap + VFH::Update_VFH for many
// independent robots per launch.
//
// Reference path replaced (mc/ = move_control in the reference):
//   Steerer::getRangesFromSubmap           mc/src/steerer.cpp:147-191
//   MapProvider::getSubMap / getSubmap     mc/src/map_provider.cpp:93-100, gmc/src/GridMap.cpp:287-339
//   VFH::Init tables                       mc/src/vfh.cpp:237-416,144-166
//   VFH::Update_VFH and its stages         mc/src/vfh.cpp:480-605,986-1261
//
// Layout: one 128-thread workgroup (two wavefronts) per robot.  The 1.5 m window (<= 31 x 31 cells)
// is read straight from the master layer with lanes along Index(0) (contiguous in the column-major
// layer); the 361-bin min-range scan lives in LDS as IEEE-double bit patterns reduced with LDS
// atomicMin (all distances are positive, so the unsigned order is the numeric order); cell
// magnitudes and a non-zero bit mask are staged in LDS; every histogram sector is owned by one lane
// that adds its cells in the reference's (y outer, x inner) order through a precomputed
// sector-major membership bit mask, which makes OriginHist bit-identical to the sequential CPU sum.
// phi_left/phi_right of the masked histogram are order-free min/max reductions (see comment there).
// All tables that need libm (atanf, asinf, pow, tan, hypotf) are built on the host at init with the
// same calls the reference makes, so the device never re-derives them.
#include "engine.hpp"

#include <cmath>
#include <cstring>
#include <vector>

using namespace rna;

namespace {

constexpr int VFH_THREADS = 128;
constexpr int MAX_W = 64;
constexpr int MAX_NQ = (MAX_W / 2 + 1) * MAX_W;   // 2112
[[maybe_unused]] constexpr int MAX_NW = (MAX_NQ + 31) / 32;        // 66 (the kernel sizes its LDS by the actual window)
constexpr int MAX_H = 128;
constexpr int VFH_OCC_CAP = 1024;

struct VfhK {
  int W, H, T, CX, CY, YH, NQ, NQF, NW, max_speed, sector_angle;
  float cell_width, robot_radius, sd0, sd1, bl0, bh0, bl1, bh1, U1, U2;
  int current_max_speed, max_speed_narrow, max_speed_wide, max_accel, mt0, mt1;
  const float *cell_dist, *cell_base_mag, *cell_dir;
  const int* range_idx;
  const unsigned *memb, *in_circle;
  const int* mtr;
  const float* bcr;
  float *last_binary, *hist, *origin, *picked, *last_picked, *blocked_radius;
  int *last_chosen_speed, *max_speed_picked;
};

// ---- small pieces of vfh.cpp that are pure arithmetic (usable on host and device) --------------
RNA_HD int k_max_turnrate(int mt0, int mt1, int speed) {  // vfh.cpp:130-138
  int val = (mt0 - (int)(speed * (mt0 - mt1) / 1000.0));
  return val < 0 ? 0 : val;
}
RNA_HD int k_safety_dist(float sd0, float sd1, int speed) {  // vfh.cpp:194-205
  int val = (int)(sd0 + (int)(speed * (sd1 - sd0) / 1000.0));
  return val < 0 ? 0 : val;
}
RNA_HD float k_delta_angle(float a1, float a2) {  // vfh.cpp:673-686
  const float diff = a2 - a1;
  // (both corrections are computed and one value is picked: as an if / else-if the device compiler keeps two nested
  // branches with their exec-mask bookkeeping at every one of the kernel's twenty call sites)
  const float down = diff - 360, up = diff + 360;
  return diff > 180 ? down : (diff < -180 ? up : diff);
}
// glibc 2.35 hypotf() is (float)sqrt((double)x*x + (double)y*y); restated so host tables and the
// device agree bit for bit (checked against libm in tests/test_host_tables.py)
RNA_HD float k_hypotf(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }

// angles::normalize_angle_positive (ROS `angles` package)
// fmod is exact, and for y <= |x| < 2y its value is |x| - y with the sign of x, itself exact (Sterbenz): with |a| < 4 pi --
// any yaw within two turns -- both fmod calls are a compare and a subtraction, and the library's loop is the rare path.
__device__ __forceinline__ double fmod_two_pi(double x) {
  const double y = 2.0 * M_PI, ax = fabs(x);
  if (ax < y) return x;
  if (ax < 2.0 * y) return copysign(ax - y, x);
  return fmod(x, y);
}
__device__ __forceinline__ double normalize_angle_positive(double a) {
  return fmod_two_pi(fmod_two_pi(a) + 2.0 * M_PI);
}

// wrap_index (gridmath.hpp) for the values the kernel meets -- an index of the map plus or minus the buffer's start, in
// [-size, 2 size) -- without its two integer divisions; anything else takes the original.
__device__ __forceinline__ int wrap_near(int v, int size) {
  const int w = v < 0 ? v + size : (v >= size ? v - size : v);
  return (unsigned)w < (unsigned)size ? w : wrap_index(v, size);
}

// ONE axis of submap_information (gridmath.hpp: getSubmapInformation, gmc/src/GridMapMath.cpp:246-296, followed by
// GridMap::setGeometry(SubmapGeometry), gmc/src/GridMap.cpp:51-75), operation by operation in the same order.  What the
// kernel needs of it: does the submap exist, the unwrapped index of its top-left cell, its size, and the position of that
// cell's centre (`off`: pos + (len/2 - res/2) of the SUBMAP, the origin of GridMapIterator's positions).
// setGeometry's size = round(len / res) with len = sz * res is sz itself: RN(RN(sz res) / res) = sz (1 + d1)(1 + d2),
// |d| <= 2^-53, is within 2^-51 sz of sz, less than 1/2 for any size a map can have, so the division and the round are
// not evaluated; its len = size * res is then the same product again.
struct SubmapAxis { int ok, tl_u, size; double off; };
__device__ __forceinline__ SubmapAxis submap_axis(const Geom& g, int a, double p, double req_len) {
  SubmapAxis o;
  o.ok = 0; o.tl_u = 0; o.size = 0; o.off = 0.0;
  const double len = g.len[a], pos = g.pos[a], res = g.res;
  const int size = g.size[a], start = g.start[a];
  const bool unmoved = g.start[0] == 0 && g.start[1] == 0;
  auto limit = [&](double x) {   // limit_position_to_range
    const double vto = 0.5 * len;
    double shifted = (x - pos) + vto;
    double eps = 10.0 * DBL_EPSILON;
    if (fabs(x) > 1.0) eps *= fabs(x);
    if (shifted <= 0) shifted = eps;
    else if (shifted >= len) shifted = len - eps;
    return (shifted + pos) - vto;
  };
  auto unwrapped_index_of = [&](double x, int& u_out) -> bool {   // index_from_position, then unwrap_index of its result
    const double t = -((x - pos) - 0.5 * len);
    if (!(t >= 0.0 && t < len)) return false;
    const int u = -(int)(((x - 0.5 * len) - pos) / res);
    if (unmoved) {
      if ((unsigned)u >= (unsigned)size) return false;
      u_out = u;
    } else {
      u_out = wrap_near(wrap_near(u + start, size) - start, size);
    }
    return true;
  };
  int tl_u, br_u;
  if (!unwrapped_index_of(limit(p - (-0.5 * req_len)), tl_u)) return o;
  if (!unwrapped_index_of(limit(p + (-0.5 * req_len)), br_u)) return o;
  double corner = (pos + (0.5 * len - 0.5 * res)) + res * (double)(-tl_u);   // position_from_index(top_left)
  corner = corner - (-(0.5 * res));
  const int sz = br_u - tl_u + 1;
  const double len_s = (double)sz * res;
  const double pos_s = corner - 0.5 * len_s;
  const double t = -((p - pos_s) - 0.5 * len_s);   // the requested position has to lie inside the submap
  if (!(t >= 0.0 && t < len_s)) return o;
  o.ok = 1; o.tl_u = tl_u; o.size = sz;
  o.off = pos_s + (0.5 * len_s - 0.5 * res);
  return o;
}

// Wavefronts per SIMD the register allocation aims at: six (80 registers, nothing spilled).  At 16 384 poses the kernel is bound
// by instruction issue and a sixth wavefront fills the gaps of the others' waits (five: +5 %); seven (72 registers) spill three
// and cost 1.5 us at 1 024 poses (profiles/r06_vfh_rework_ab.txt).
#ifndef RNA_VFH_WAVES
#define RNA_VFH_WAVES 6
#endif
#define VFH_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(RNA_VFH_WAVES, RNA_VFH_WAVES)))
#ifdef RNA_VFH_SKIPS
// developer build (scripts/pmc_vfh_sq.sh): phases switched off by RNA_VFH_SKIP's bits, to read the instruction counters of the
// rest.  1 the obstacle cells' trigonometry, 2 the window's cell loads, 4 the cell magnitudes, 8 the sector sums, 16 the
// masked histogram's cell walk, 32 Select_Direction.  Results are wrong by design.
__device__ int d_vfh_skip;
#define VFH_SKIP(bit) (__builtin_amdgcn_readfirstlane(d_vfh_skip) & (bit))
// RNA_VFH_EXIT=n: every thread leaves at the n-th mark (the counters of the kernel's prefix, on live data)
__device__ int d_vfh_exit;
#define VFH_EXIT(n) if (__builtin_amdgcn_readfirstlane(d_vfh_exit) == (n)) return
#else
#define VFH_SKIP(bit) 0
#define VFH_EXIT(n)
#endif
__global__ void __launch_bounds__(VFH_THREADS) VFH_WAVES_ATTR
vfh_step_kernel(VfhK K, Geom g, const float* __restrict__ master, const rna_pose* __restrict__ poses,
                const double* __restrict__ ext_ranges, rna_vfh_out* __restrict__ out,
                float* __restrict__ origin_out, float* __restrict__ hist_out) {
  __shared__ unsigned long long rng[361];
  // cell magnitudes and their non-zero bits: sized by the launch for THIS window (NQ floats, then NW words).  As static arrays
  // for the largest window the header allows (64 x 33 cells: 8.7 KB) a workgroup took 16 KB of LDS and ten fitted a CU; with
  // the Steerer's 30 x 30 window it takes 9.5 KB and all sixteen that the CU's wave slots hold are resident (config 2's
  // batches of thousands of poses run in waves of resident workgroups: 16 384 poses 147 -> see profiles/r05_vfh_probe.txt)
  extern __shared__ float vfh_dyn[];
  float* const mag = vfh_dyn;
  unsigned* const nz = reinterpret_cast<unsigned*>(vfh_dyn + ((K.NQ + 31) & ~31));
  __shared__ float s_hist[MAX_H];
  __shared__ unsigned s_phi_right, s_phi_left;
  __shared__ int s_emergency, s_nocc, s_cant;
  __shared__ int occ[VFH_OCC_CAP];   // submap cells that hold an obstacle: (j << 16) | i
  __shared__ int s_ax_ok[2], s_ax_tl[2], s_ax_size[2];   // the submap's geometry, one axis from each wavefront
  __shared__ double s_ax_off[2];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
#ifdef RNA_VFH_STATS
  unsigned long long ts_[8];
  int nts_ = 0;
#define VFH_STAMP() ts_[nts_++] = wall_clock64()
#else
#define VFH_STAMP()
#endif
  VFH_STAMP();
  const rna_pose pose = poses[b];
  // the robot's state of the last step, needed by the serial tail only: loaded here so the round trip is long over by then
  float st_picked = 0.0f, st_last_picked = 0.0f, st_blocked_radius = 0.0f;
  int st_max_speed_picked = 0;
  const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;   // wave-uniform: the serial tail runs as scalar control flow
  if (wave0) {
    st_picked = K.picked[b];
    st_last_picked = K.last_picked[b];
    st_max_speed_picked = K.max_speed_picked[b];
  }
  st_blocked_radius = K.blocked_radius[b];   // (both: wavefront 1 answers Cant_Turn_To_Goal)
  // (round 5: every load whose address is known here goes out here -- the kernel is a chain of phases, each of which used to
  // start with a trip to L2 of its own: the robot's last speed before the cell magnitudes, its last binary histogram inside
  // the histogram step)
  const int last_speed = K.last_chosen_speed[b];
  float st_last_binary = 0.0f;
  if (tid < K.H) st_last_binary = K.last_binary[(size_t)b * K.H + tid];

  // ---------------- Update_VFH prologue (vfh.cpp:490-515) ----------------
  const float desired_angle = pose.goal_direction;
  const float dist_to_goal = pose.goal_distance;
  const float goal_tol = pose.goal_tolerance;
  int speed = pose.current_speed < 0 ? 0 : pose.current_speed;
  if (speed < last_speed) speed = last_speed;
  const int tspeed = speed > K.max_speed ? K.max_speed : speed;  // table index (reference: OOB read)
  const float bcr_tspeed = K.bcr[tspeed];   // (for the tail: on its way while the histograms are built)

  // The cell tables Calculate_Cells_Mag reads: distance, range index, base magnitude, direction, the two blocked-circle words.
  // (They depend on the thread and the speed alone, and asking for them a phase early -- with the window's map cells -- was
  // tried: 24 registers held through the obstacle cells' trigonometry cost more than the trip saved, 13.7 -> 14.1 us for one
  // pose at five wavefronts per SIMD, 83.7 -> 92 us for 16 384; profiles/r06_vfh_rework_ab.txt.)
  float cd4[4], bm4[4], dir4[4];
  int ri4[4];
  unsigned inr4[4], inl4[4];
  const unsigned* inr = K.in_circle + ((size_t)tspeed * 2 + 0) * K.NW;
  const unsigned* inl = K.in_circle + ((size_t)tspeed * 2 + 1) * K.NW;
  auto load_cell_tables = [&](int q0) {
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int q = q0 + k4 * VFH_THREADS + tid;
      cd4[k4] = 0.0f; bm4[k4] = 0.0f; ri4[k4] = 0; dir4[k4] = 0.0f; inr4[k4] = 0u; inl4[k4] = 0u;
      if (q < K.NQF) {   // (unsigned indices: a scalar base and ONE 32-bit offset register serve the four cell tables)
        const unsigned uq = (unsigned)q, uw = uq >> 5;
        cd4[k4] = K.cell_dist[uq]; ri4[k4] = K.range_idx[uq]; bm4[k4] = K.cell_base_mag[uq];
        dir4[k4] = K.cell_dir[uq]; inr4[k4] = inr[uw]; inl4[k4] = inl[uw];
      }
    }
  };

  // ---------------- ranges (steerer.cpp:147-191) ----------------
  for (int i = tid; i < 361; i += VFH_THREADS)
    rng[i] = ext_ranges ? (unsigned long long)__double_as_longlong(ext_ranges[((size_t)b * 361 + i) * 2])
                        : (unsigned long long)__double_as_longlong(5000.0);
  for (int i = tid; i < K.NW; i += VFH_THREADS) nz[i] = 0;
  if (tid == 0) { s_phi_right = 0u; s_phi_left = __float_as_uint(180.0f); s_emergency = 0; s_nocc = 0; }
  VFH_EXIT(1);   // loads of the state, ranges preset
  // The 1.5 m submap's geometry is the same for all 128 threads and all f64 (six divisions when both axes are done by
  // everyone: 312 of the kernel's 1 608 vector instructions per wavefront, profiles/r06_vfh_phase_counters.txt).  The axes
  // share nothing but the verdict: wavefront 0 takes x, wavefront 1 takes y, and they meet at the barrier the preset needs.
  if (!ext_ranges) {
    const int a = wave0 ? 0 : 1;
    const SubmapAxis ax = submap_axis(g, a, a == 0 ? pose.x : pose.y, 1.5);
    if ((tid & 63) == 0) { s_ax_ok[a] = ax.ok; s_ax_tl[a] = ax.tl_u; s_ax_size[a] = ax.size; s_ax_off[a] = ax.off; }
  }
  __syncthreads();

  VFH_EXIT(2);   // + the submap's geometry
  if (!ext_ranges) {
    if (s_ax_ok[0] && s_ax_ok[1]) {  // getSubMap failure leaves every range at 5000
      const int tl_u[2] = {s_ax_tl[0], s_ax_tl[1]};
      const int sr = s_ax_size[0], sc = s_ax_size[1];
      const double offx = s_ax_off[0], offy = s_ax_off[1];
      // One obstacle cell costs an f64 atan2, a division and a sqrt (~2 us), a free one a load and two compares, and 2 % of
      // the cells are obstacles: they are collected first and then shared out one per lane, so the wavefronts run the
      // trigonometry once instead of once per loop iteration that happens to hold an obstacle (6-8 us -> 3-4 us).
      // atomicMin makes the result independent of the order.
      auto obstacle = [&](int i, int j) {   // cell (i, j) of the submap: GridMapIterator's Index(0), Index(1)
        if (VFH_SKIP(1)) return;
        const double px = offx + g.res * (double)(-i);
        const double py = offy + g.res * (double)(-j);
        const double angle = atan2(py - pose.y, px - pose.x);
        const double deg = normalize_angle_positive(angle - pose.yaw + 3.14 / 2) * 180.0 / M_PI;
        if (deg > 180) return;
        const int fl = (int)floor(deg), ce = (int)ceil(deg);
        const double dx = pose.x - px, dy = pose.y - py;
        const double distance = sqrt(dx * dx + dy * dy) * 1000.0;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(distance);
        atomicMin(&rng[fl * 2], bits);
        atomicMin(&rng[ce * 2], bits);
      };
      // getBufferIndexFromIndex (gridmath.hpp buffer_index), then the matrix.  A cell of the submap has an unwrapped index in
      // [0, size) and the buffer's start lies in [0, size) (launch_step checks it): wrap_index of their sum is one compare,
      // and on a map that never moved (start 0) the same code never takes it -- one path, no divisions, no branches.
      const int base0 = tl_u[0] + g.start[0], base1 = tl_u[1] + g.start[1];
      auto cell_address = [&](int i, int j) -> const float* {
        int b0 = base0 + i, b1 = base1 + j;
        b0 -= b0 >= g.size[0] ? g.size[0] : 0;
        b1 -= b1 >= g.size[1] ? g.size[1] : 0;
        return master + ((size_t)b1 * g.size[0] + (size_t)b0);
      };
      // Eight cells per thread and trip, all eight reads issued before the first is looked at: the 31 x 31 window is ONE trip
      // to L2 (round 5).  Round 6: a thread keeps its column and steps through the rows -- lanes run along Index(0), a row of
      // the window is one coalesced read, and no cell's (i, j) needs a division (lin % sr, lin / sr cost 15 instructions a
      // cell: with the addressing 357 of the 1 608).  A thread outside the window reads the window's last row / column
      // and drops the value: eight loads without a branch between them.  The order of the cells is free: atomicMin.
      const bool listed = sr <= VFH_THREADS && sc < 32768 && sr > 0 && sc > 0;   // (not: a resolution below 1.2 cm, rows longer than the workgroup)
      if (listed) {
        const int cwl = sr <= 32 ? 5 : sr <= 64 ? 6 : 7;       // columns of the thread grid: the power of two that holds a row
        const int i = tid & ((1 << cwl) - 1), j0 = tid >> cwl, rpp = VFH_THREADS >> cwl;
        const int ic = i < sr ? i : sr - 1;
        for (int jb = 0; jb < sc && !VFH_SKIP(2); jb += 8 * rpp) {
          float val[8];
#pragma unroll
          for (int k8 = 0; k8 < 8; ++k8) {
            const int j = jb + k8 * rpp + j0;
            val[k8] = *cell_address(ic, j < sc ? j : sc - 1);
          }
          // the thread's obstacle cells of this trip as a mask, ONE reservation in the list for all of them (most threads
          // have none: 2 % of the cells are obstacles)
          unsigned found = 0u;
#pragma unroll
          for (int k8 = 0; k8 < 8; ++k8)   // (NaN compares false, as the reference's `value != value ||` makes it)
            found |= (val[k8] > 3 && jb + k8 * rpp + j0 < sc ? 1u : 0u) << k8;
          if (i < sr && found) {
            int k = atomicAdd(&s_nocc, __popc(found));
            while (found) {
              const int k8 = __ffs(found) - 1;
              found &= found - 1;
              if (k < VFH_OCC_CAP) occ[k] = ((jb + k8 * rpp + j0) << 16) | i;
              ++k;
            }
          }
        }
      }
      __syncthreads();
      VFH_EXIT(3);   // + the window's cells
      // the listed obstacle cells, one per lane; a window the list does not hold (more than 1 024 obstacle cells, or not
      // listed at all) is walked cell by cell instead -- a cell met twice changes nothing (atomicMin)
      const bool use_list = listed && s_nocc <= VFH_OCC_CAP;
      const int n_todo = use_list ? s_nocc : sr * sc;
      for (int k = tid; k < n_todo; k += VFH_THREADS) {
        int i, j;
        if (use_list) {
          i = occ[k] & 0xffff; j = occ[k] >> 16;
        } else {
          i = k % sr; j = k / sr;
          const float value = *cell_address(i, j);
          if (value != value || value <= 3) continue;
        }
        obstacle(i, j);
      }
    }
  }
  __syncthreads();
  VFH_STAMP();
  VFH_EXIT(4);   // + the obstacle cells' ranges

  int speed_index = (int)floorf(((float)speed / (float)K.current_max_speed) * K.T);  // vfh.cpp:175-186
  if (speed_index >= K.T) speed_index = K.T - 1;

  // ---------------- Calculate_Cells_Mag (vfh.cpp:986-1049) ----------------
  // with the occupied front cells' part of Build_Masked_Polar_Histogram (vfh.cpp:1131-1213): the reference walks the occupied
  // front cells sequentially, raising phi_right / lowering phi_left.  Front cells have directions in (0,180): a cell updates
  // phi_right iff it lies right of straight-ahead (Delta(dir,90) > 0), inside the right blocked circle and dir >= phi_right,
  // so the final phi_right is the MAX direction of those cells (and symmetrically phi_left the MIN on the left side) --
  // independent of the visiting order, and so the thread that finds a cell occupied here can say it right away (its
  // direction and circle words come with the trip that brings the cell's distance: the walk used to be a pass of its own
  // over the magnitudes, with a trip to L2 of its own).  Circle membership is a host table.  After an emergency stop the
  // two angles are not read.
  const float r_safe = K.robot_radius + (float)k_safety_dist(K.sd0, K.sd1, speed);
  const int q_centre = K.CY * K.W + K.CX;   // (x == CX && y == CY of q = y W + x, without the division)
  for (int q0 = 0; q0 < K.NQ && !VFH_SKIP(4); q0 += 4 * VFH_THREADS) {   // (four cells per thread and trip, the table reads first)
    load_cell_tables(q0);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int q = q0 + k4 * VFH_THREADS + tid;
      if (q >= K.NQ) continue;
      float m = 0.0f;
      if (q < K.NQF) {
        const float cd = cd4[k4];
        const double range = __longlong_as_double((long long)rng[ri4[k4]]);
        if ((cd + K.cell_width / 2.0) > range) {
          if (cd < r_safe && q != q_centre) s_emergency = 1;
          m = bm4[k4];
        }
      }
      mag[q] = m;
      if (m != 0.0f) {
        atomicOr(&nz[q >> 5], 1u << (q & 31));
        if (!VFH_SKIP(16)) {
          const float dir = dir4[k4];
          const unsigned bit = 1u << (q & 31);
          if (k_delta_angle(dir, 90.0f) > 0) {
            if ((inr4[k4] & bit) && dir >= 0.0f && dir < 90.0f) atomicMax(&s_phi_right, __float_as_uint(dir));
          } else {
            if ((inl4[k4] & bit) && dir <= 180.0f) atomicMin(&s_phi_left, __float_as_uint(dir));
          }
        }
      }
    }
  }
  // The sector's membership words (Build_Primary_Polar_Histogram below) depend on the speed alone: all sixteen leave here,
  // before the barrier, in one trip (they used to be fetched eight at a time after it: two trips to L2 at the head of the
  // histogram step).  Sixteen reads whatever NW is, without a test between them: the words past NW belong to the next
  // sector -- the table ends with sixteen spare words -- and the sums below stop at NW.  (A window of more than 512 cells
  // keeps the old loop.)
  const bool memb_in_regs = K.NW <= 16;
  unsigned mbw[16];
  {
    const unsigned* mb = K.memb + ((size_t)speed_index * K.H + (tid < K.H ? tid : K.H - 1)) * K.NW;
#pragma unroll
    for (int k = 0; k < 16; ++k) mbw[k] = mb[k];
  }
  __syncthreads();
  VFH_STAMP();
  VFH_EXIT(5);   // + cell magnitudes
  const bool emergency = s_emergency != 0;

  float* origin = K.origin + (size_t)b * K.H;
  float* hist = K.hist + (size_t)b * K.H;
  float* last_binary = K.last_binary + (size_t)b * K.H;

  if (emergency) {
    // vfh.cpp:1070-1077 and :533-540 : OriginHist all 1, Hist untouched
    if (tid < K.H) origin[tid] = 1.0f;
  } else {
    // ---------------- Build_Primary_Polar_Histogram (vfh.cpp:1057-1095) ----------------
    float sum = 0.0f;
    if (memb_in_regs) {
      // The window's non-zero words: ONE read, lane l holds word l, and the loop takes word w from lane w with v_readlane -- the
      // same for every sector, so a scalar, and a word without an occupied cell is skipped by a scalar branch (reading nz[w]
      // in the loop was an LDS round trip at the head of each of the fifteen turns).  EVERY lane runs the loop, so that the
      // lane v_readlane asks is awake: the second wavefront holds eight sectors, its other lanes repeat the last one's sum
      // (`mbw` above was read for sector H - 1 there) and drop it.
      {
        const int wl = tid & 63;
        const unsigned nzv = nz[wl < K.NW ? wl : 0];
        for (int w = 0; w < K.NW && !VFH_SKIP(8); ++w) {   // (mbw[w]: a register picked by the loop counter, s_set_gpr_idx)
          const unsigned nzw = __builtin_amdgcn_readlane(nzv, w);
          if (nzw == 0u) continue;
          unsigned bits = mbw[w] & nzw;
          // ascending q == the reference's (y outer, x inner) order.  Two cells a turn: both magnitudes are asked for
          // before the first is added (the LDS round trip was the turn's length); a turn with one cell left adds 0.0f,
          // which leaves a sum that started at +0 as it is.
          while (bits) {
            const int bit0 = __ffs(bits) - 1;
            bits &= bits - 1;
            const bool two = bits != 0u;
            const int bit1 = two ? __ffs(bits) - 1 : bit0;
            bits &= bits - 1;   // (0 & 0xffffffff stays 0)
            const float m0 = mag[w * 32 + bit0], m1 = mag[w * 32 + bit1];
            sum += m0;
            sum += two ? m1 : 0.0f;
          }
        }
      }
    }
    if (tid < K.H) {
      if (!memb_in_regs) {
        const unsigned* mb = K.memb + ((size_t)speed_index * K.H + tid) * K.NW;
        for (int w0 = 0; w0 < K.NW && !VFH_SKIP(8); w0 += 8) {   // (the sector's membership words eight at a time: one trip to L2 per eight)
          unsigned mb8[8];
#pragma unroll
          for (int k8 = 0; k8 < 8; ++k8) mb8[k8] = w0 + k8 < K.NW ? mb[w0 + k8] : 0u;
#pragma unroll
          for (int k8 = 0; k8 < 8; ++k8) {
            const int w = w0 + k8;
            unsigned bits = w < K.NW ? (mb8[k8] & nz[w]) : 0u;
            while (bits) {
              const int bit = __ffs(bits) - 1;
              bits &= bits - 1;
              sum += mag[w * 32 + bit];
            }
          }
        }
      }
      origin[tid] = sum;
      VFH_EXIT(6);   // + sector sums (leaves inside a divergent region: wave 1's upper lanes go on to the next mark)
      // ---------------- Build_Binary_Polar_Histogram (vfh.cpp:1102-1121, 214-231) ----------------
      const float hi = (float)(K.bh0 - (speed * (K.bh0 - K.bh1) / 1000.0));
      const float lo = (float)(K.bl0 - (speed * (K.bl0 - K.bl1) / 1000.0));
      float h;
      if (sum > hi) h = 1.0f;
      else if (sum < lo) h = 0.0f;
      else h = st_last_binary;
      last_binary[tid] = h;
      s_hist[tid] = h;
    }
    // ---------------- Build_Masked_Polar_Histogram (vfh.cpp:1131-1213): phi_left / phi_right were found with the magnitudes ----------------
    __syncthreads();
    if (tid < K.H) {
      const float phi_right = __uint_as_float(s_phi_right), phi_left = __uint_as_float(s_phi_left);
      const float angle = (float)(tid * K.sector_angle);
      const float h = s_hist[tid];
      // (the reference's short-circuit expression, every term evaluated: none has a side effect)
      const float d_right = k_delta_angle(angle, phi_right), d_left = k_delta_angle(angle, phi_left);
      const float d_ahead = k_delta_angle(angle, 90.0f);
      const bool open = (h == 0) & (((d_right <= 0) & (d_ahead >= 0)) | ((d_left >= 0) & (d_ahead <= 0)));
      const float nh = open ? 0.0f : 1.0f;
      s_hist[tid] = nh;
      hist[tid] = nh;
    }
  }
  __syncthreads();
  VFH_STAMP();
  VFH_EXIT(7);   // + binary and masked histograms

  // ---------------- serial tail on one lane ----------------
  // the masked histogram as two 64-bit lane masks (H <= 128), so the valley search below tests bits in registers
  // instead of making 73 dependent LDS reads
  unsigned long long hbits_lo = 0ull, hbits_hi = 0ull;
  if (tid < 64) {
    hbits_lo = __ballot(tid < K.H && s_hist[tid] == 1);
    hbits_hi = __ballot(tid + 64 < K.H && s_hist[(tid + 64) & (MAX_H - 1)] == 1);
  }
  float picked = st_picked;
  float last_picked = st_last_picked;
  int max_speed_for_picked = st_max_speed_picked;
  const float blocked_radius = emergency ? st_blocked_radius : bcr_tspeed;   // (an emergency stop leaves the last step's)
  if (!wave0) {
    // Cant_Turn_To_Goal (vfh.cpp:612-654) needs the goal and the blocked circle, not the picked direction: the second
    // wavefront, which has nothing else to do in the tail, answers it while the first selects the direction (an f64 cos
    // and sin and two hypotf: 0.55 us of the 2.5 us tail, and 178 instructions off the first wavefront's path).
    const float goal_x = (float)(dist_to_goal * cos(desired_angle * M_PI / 180));
    const float goal_y = (float)(dist_to_goal * sin(desired_angle * M_PI / 180));
    bool cant = false;
    float dc = k_hypotf(goal_x - blocked_radius, goal_y);
    if (dc + goal_tol < blocked_radius) cant = true;
    if (!cant) {
      dc = k_hypotf(-goal_x - blocked_radius, goal_y);
      if (dc + goal_tol < blocked_radius) cant = true;
    }
    if ((tid & 63) == 0) s_cant = cant ? 1 : 0;
  }
  if (wave0) {   // every lane of the first wavefront computes the same values; lane 0 stores them
    const int H = K.H, SA = K.sector_angle;
    auto hist_at = [&](int i) -> float { return ((i < 64 ? hbits_lo >> i : hbits_hi >> (i - 64)) & 1ull) ? 1.0f : 0.0f; };

    if (emergency) {
      picked = last_picked;
      max_speed_for_picked = 0;
      last_picked = picked;
    } else {
      // ---------------- Select_Direction (vfh.cpp:755-870) ----------------
      // The reference walks the ring of sectors once from the first blocked sector of the front half, opens a valley
      // at every 1 -> 0 step, closes it at the next 0 -> 1 step, and weighs the valley's candidate angles as they come;
      // Select_Candidate_Angle keeps the FIRST minimum (strict <, vfh.cpp:715-749).  Here every lane owns the sectors
      // lane and lane + 64: a lane whose sector starts a valley finds the valley's end with a bit search in the ring
      // rotated to its own position, weighs the valley's (at most four) candidates in the reference's order, and the
      // wavefront takes the minimum by (weight, position of the valley in the reference's walk, candidate number) --
      // the same winner, bit for bit, in ~150 instructions instead of 73 dependent loop turns of ~25.
      const int half = H / 2;
      const unsigned long long front = half >= 64 ? hbits_lo : (hbits_lo & ((1ull << half) - 1ull));
      int ncand = 0;
      float best_angle = 90.0f;
      int best_speed = max_speed_for_picked;
      if (front == 0ull || VFH_SKIP(32)) {
        picked = desired_angle;
        last_picked = picked;
        max_speed_for_picked = K.current_max_speed;
      } else {
        const int start = __builtin_ctzll(front);
        const int cms = K.current_max_speed;
        const int sp_narrow = cms < K.max_speed_narrow ? cms : K.max_speed_narrow;
        const int sp_wide = cms < K.max_speed_wide ? cms : K.max_speed_wide;
        // the ring as a 128-bit value (bits >= H are zero)
        const unsigned long long r_lo = hbits_lo, r_hi = hbits_hi;
        float my_w = 10000000.0f;    // Select_Candidate_Angle's starting minimum: a candidate has to beat it
        int my_ord = 0x7fffffff;     // (position of the valley in the walk) * 4 + candidate number
        float my_a = 90.0f;
        int my_sp = 0, my_n = 0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int sct = tid + 64 * pass;
          if (sct >= H) continue;
          const int prev = sct == 0 ? H - 1 : sct - 1;
          if (!(hist_at(sct) == 0 && hist_at(prev) == 1)) continue;   // not the first sector of a valley
          // first blocked sector after sct, going round (the ring holds a 1: sector `start`, in its lower word): the lowest set
          // bit above sct, or else the lowest of all.  `pass` is a constant of the unrolled loop: one 64-bit shift and two
          // bit searches (round 6; the ring used to be rotated to the lane's position as a 128-bit value, two five-way
          // branches on the lane's shift count).
          int next;
          if (pass == 0) {
            const unsigned long long above = sct >= 63 ? 0ull : (r_lo & (~0ull << (sct + 1)));
            next = above ? __builtin_ctzll(above) : (r_hi ? 64 + __builtin_ctzll(r_hi) : __builtin_ctzll(r_lo));
          } else {
            const unsigned long long above = sct >= 127 ? 0ull : (r_hi & (~0ull << (sct - 63)));
            next = above ? 64 + __builtin_ctzll(above) : (r_lo ? __builtin_ctzll(r_lo) : 64 + __builtin_ctzll(r_hi));
          }
          const int last = next == 0 ? H - 1 : next - 1;   // the valley's last free sector = (im - 1) of the closing step
          const int b1 = sct * SA;
          const int b2 = last * SA;                    // (im - 1) * SA, + 360 when im == 0: the same number, H * SA == 360
          const float angle = k_delta_angle((float)b1, (float)b2);
          if (fabsf(angle) < 10) continue;
          int pos = sct - start;                 // the walk reaches this valley after `pos` steps
          if (pos < 0) pos += H;
          int cno = 0;
          auto consider = [&](float a, int sp) {
            const float weight = K.U1 * fabsf(k_delta_angle(desired_angle, a)) +
                                 K.U2 * fabsf(k_delta_angle(last_picked, a));
            const int ord = pos * 4 + cno;
            if (weight < my_w || (my_ord != 0x7fffffff && weight == my_w && ord < my_ord)) { my_w = weight; my_ord = ord; my_a = a; my_sp = sp; }
            ++cno;
            ++my_n;
          };
          if (fabsf(angle) < 80) {
            consider((float)(b1 + (b2 - b1) / 2.0), sp_narrow);
          } else {
            consider((float)(b1 + (b2 - b1) / 2.0), cms);
            const float c2 = (float)((b1 + 40) % 360);
            consider(c2, sp_wide);
            float c3 = (float)(b2 - 40);
            if (c3 < 0) c3 += 360;
            consider(c3, sp_wide);
            if ((k_delta_angle(desired_angle, c2) < 0) && (k_delta_angle(desired_angle, c3) > 0))
              consider(desired_angle, sp_wide);
          }
        }
        // the wavefront's minimum by (weight, order); a candidate only counts if it beats the starting minimum.  A ring has a
        // handful of valleys: the lanes that hold a candidate are read one after the other (v_readlane, a scalar loop of two
        // or three turns) -- the butterfly of 18 ds_bpermute round trips this replaces was a third of the step's 1.5 us.
        {
          unsigned long long holders = __ballot(my_ord != 0x7fffffff);
          float w = 10000000.0f;
          int ord = 0x7fffffff;
          while (holders) {
            const int l = __builtin_ctzll(holders);
            holders &= holders - 1ull;
            const float cw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_w), l));
            const int co = __builtin_amdgcn_readlane(my_ord, l);
            if (cw < w || (cw == w && co < ord)) {
              w = cw; ord = co;
              best_angle = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_a), l));
              best_speed = __builtin_amdgcn_readlane(my_sp, l);
            }
          }
          ncand = __ballot(my_n != 0) != 0ull ? 1 : 0;   // (only "was any candidate looked at" is asked below)
        }
        if (ncand == 0) {
          picked = last_picked;
          max_speed_for_picked = 0;
          last_picked = picked;
        } else {
          picked = best_angle;
          max_speed_for_picked = best_speed;
          last_picked = picked;
        }
      }
    }

    VFH_STAMP();   // (developer build: lane 0's stamps 4.. subdivide the tail)
    VFH_EXIT(8);   // + Select_Direction
  }
  __syncthreads();
  if (wave0) {
    // ---------------- speed (vfh.cpp:571-599) ----------------
    int speed_incr;
    if ((pose.dt > 0.3) || (pose.dt < 0)) speed_incr = 10;
    else speed_incr = (int)(K.max_accel * pose.dt);
    if (s_cant) speed_incr = -speed_incr;
    VFH_STAMP();
    VFH_EXIT(9);   // + Cant_Turn_To_Goal
    int chosen_speed = last_speed + speed_incr;
    if (max_speed_for_picked < chosen_speed) chosen_speed = max_speed_for_picked;

    // ---------------- Set_Motion (vfh.cpp:1222-1261) ----------------
    const int maxturn = k_max_turnrate(K.mt0, K.mt1, speed);
    int turnrate;
    if (chosen_speed <= 0) {
      turnrate = maxturn;
      chosen_speed = 0;
    } else if ((picked > 270) && (picked < 360)) {
      turnrate = -1 * maxturn;
    } else if ((picked < 270) && (picked > 180)) {
      turnrate = maxturn;
    } else {
      turnrate = (int)rint(((float)(picked - 90) / 75.0) * maxturn);
      if (turnrate > maxturn) turnrate = maxturn;
      else if (turnrate < (-1 * maxturn)) turnrate = -1 * maxturn;
    }

    if (tid == 0) {
      K.picked[b] = picked;
      K.last_picked[b] = last_picked;
      K.max_speed_picked[b] = max_speed_for_picked;
      K.blocked_radius[b] = blocked_radius;
      K.last_chosen_speed[b] = chosen_speed;
      rna_vfh_out o;
      o.chosen_speed = chosen_speed;
      o.chosen_turnrate = turnrate;
      o.picked_angle = picked;
      o.emergency = emergency ? 1 : 0;
      out[b] = o;
    }
  }
  __syncthreads();
  VFH_STAMP();
#ifdef RNA_VFH_STATS
  if (tid == 0 && b == 0)
    printf("[vfh stats] us: ranges %.2f cells_mag %.2f histograms %.2f tail: select %.2f cant_turn %.2f rest %.2f\n", (ts_[1] - ts_[0]) * 0.01,
           (ts_[2] - ts_[1]) * 0.01, (ts_[3] - ts_[2]) * 0.01, (ts_[4] - ts_[3]) * 0.01, (ts_[5] - ts_[4]) * 0.01, (ts_[6] - ts_[5]) * 0.01);
#endif
  // (the caller's copies of the two histograms.  Writing them where the values are made, from registers, instead of reading
  // back here what the same thread stored above, was tried: +0.6 us at 1 024 poses, +4 % at 16 384 -- r06_vfh_rework_ab.txt)
  if (tid < K.H) {
    if (origin_out) origin_out[(size_t)b * K.H + tid] = origin[tid];
    if (hist_out) hist_out[(size_t)b * K.H + tid] = hist[tid];
  }
}

__global__ void vfh_reset_kernel(VfhK K, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * K.H) {  // vfh.cpp:258-262
    K.hist[i] = 0.0f;
    K.origin[i] = 0.0f;
    K.last_binary[i] = 1.0f;
  }
  if (i < n) {  // vfh.cpp:92-95
    K.picked[i] = 90.0f;
    K.last_picked[i] = 90.0f;
    K.blocked_radius[i] = 0.0f;   // uninitialised in the reference until the first normal step
    K.last_chosen_speed[i] = 0;
    K.max_speed_picked[i] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// host: VFH::VFH + VFH::Init tables (vfh.cpp:53-110,144-166,237-416)
// ------------------------------------------------------------------------------------------------
struct HostTables {
  int W, H, T, CX, CY, YH, NQ, NQF, NW, max_speed;
  std::vector<float> cell_dist, cell_base_mag, cell_dir;
  std::vector<int> range_idx;
  std::vector<unsigned> memb, in_circle;
  std::vector<int> mtr;
  std::vector<float> bcr;
};

float sector_to_dir(float sector, float dir) {  // the four wrap-aware differences of vfh.cpp:343-381
  if ((sector - dir) > 180) return dir - (sector - 360);
  if ((dir - sector) > 180) return sector - (dir + 360);
  return dir - sector;
}

bool build_tables(const rna_vfh_params& p, HostTables& t) {
  const float CELL_WIDTH = (float)p.cell_size;
  const float SD0 = (float)p.safety_dist_0ms, SD1 = (float)p.safety_dist_1ms;
  const float ROBOT_RADIUS = (float)p.robot_radius;
  const int W = p.window_diameter, SA = p.sector_angle, MAX_SPEED = p.max_speed;
  if (W < 2 || W > MAX_W || SA < 1 || MAX_SPEED < 1 || MAX_SPEED > 100000) return false;
  t.W = W;
  t.T = (SD0 == SD1) ? 1 : 20;  // vfh.cpp:100-109
  t.CX = t.CY = (int)floor(W / 2.0);
  t.H = (int)rint(360.0 / SA);
  if (t.H > MAX_H || t.H != 360 / SA) return false;
  t.YH = (int)ceil(W / 2.0);
  t.NQ = (t.YH + 1) * W;
  t.NQF = t.YH * W;
  t.NW = (t.NQ + 31) / 32;
  t.max_speed = MAX_SPEED;
  if (t.YH + 1 > W) return false;
  const int H = t.H, T = t.T, CX = t.CX, CY = t.CY;

  // Min_Turning_Radius (SetCurrentMaxSpeed, vfh.cpp:144-166) and per-speed blocked circles
  t.mtr.resize(MAX_SPEED + 1);
  t.bcr.resize(MAX_SPEED + 1);
  for (int x = 0; x <= MAX_SPEED; x++) {
    const double dx = (double)x / 1e6;
    const double dtheta = ((M_PI / 180) * (double)(k_max_turnrate(p.max_turnrate_0ms, p.max_turnrate_1ms, x))) / 1000.0;
    t.mtr[x] = (int)(((dx / tan(dtheta)) * 1000.0) * p.min_turn_radius_safety_factor);
    t.bcr[x] = t.mtr[x] + ROBOT_RADIUS + k_safety_dist(SD0, SD1, x);  // vfh.cpp:1148
  }

  t.cell_dist.assign(t.NQ, 0.f);
  t.cell_base_mag.assign(t.NQ, 0.f);
  t.cell_dir.assign(t.NQ, 0.f);
  t.range_idx.assign(t.NQ, 0);
  t.memb.assign((size_t)T * H * t.NW + 16, 0u);   // (+16: the kernel reads sixteen words from a row's start whatever NW is)
  t.in_circle.assign((size_t)(MAX_SPEED + 1) * 2 * t.NW, 0u);

  for (int x = 0; x < W; x++) {
    for (int y = 0; y <= t.YH; y++) {
      const int q = y * W + x;
      const float dist = (float)(sqrt(pow((double)(CX - x), 2.0) + pow((double)(CY - y), 2.0)) * CELL_WIDTH);
      const float base = (float)(15 * pow((3000.0 - dist), 4.0) / 100000000.0);
      float d = 0.0f;
      if (x < CX) {
        if (y < CY) { d = atanf((float)(CY - y) / (float)(CX - x)); d = (float)(d * (360.0 / 6.28)); d = (float)(180.0 - d); }
        else if (y == CY) d = 180.0f;
        else { d = atanf((float)(y - CY) / (float)(CX - x)); d = (float)(d * (360.0 / 6.28)); d = (float)(180.0 + d); }
      } else if (x == CX) {
        d = (y < CY) ? 90.0f : (y == CY ? -1.0f : 270.0f);
      } else {
        if (y < CY) { d = atanf((float)(CY - y) / (float)(x - CX)); d = (float)(d * (360.0 / 6.28)); }
        else if (y == CY) d = 0.0f;
        else { d = atanf((float)(y - CY) / (float)(x - CX)); d = (float)(d * (360.0 / 6.28)); d = (float)(360.0 - d); }
      }
      t.cell_dist[q] = dist;
      t.cell_base_mag[q] = base;
      t.cell_dir[q] = d;
      if (y < t.YH) {
        const int ri = (int)rint(d * 2.0);  // vfh.cpp:1018
        if (ri < 0 || ri > 360) return false;  // cannot happen for front cells (y < CY or odd-W centre row)
        t.range_idx[q] = ri;
      }
      for (int tb = 0; tb < T; tb++) {
        const int max_speed_this_table = (int)(((float)(tb + 1) / (float)T) * (float)MAX_SPEED);
        float enlarge;
        if (dist > 0) {
          const float r = ROBOT_RADIUS + k_safety_dist(SD0, SD1, max_speed_this_table);
          enlarge = (float)((float)asinf(r / dist) * (180 / M_PI));
        } else {
          enlarge = 0;
        }
        const float plus_dir = d + enlarge, neg_dir = d - enlarge;
        for (int i = 0; i < (360 / SA); i++) {
          const float plus_sector = (i + 1) * (float)SA, neg_sector = i * (float)SA;
          const float nn = sector_to_dir(neg_sector, neg_dir), pn = sector_to_dir(plus_sector, neg_dir);
          const float pp = sector_to_dir(plus_sector, plus_dir), np = sector_to_dir(neg_sector, plus_dir);
          bool plus_dir_bw = false, neg_dir_bw = false, around = false;
          if ((nn >= 0) && (pn <= 0)) neg_dir_bw = true;
          if ((np >= 0) && (pp <= 0)) plus_dir_bw = true;
          if ((nn <= 0) && (np >= 0)) around = true;
          if ((pn <= 0) && (pp >= 0)) plus_dir_bw = true;
          if (plus_dir_bw || neg_dir_bw || around) t.memb[((size_t)tb * H + i) * t.NW + (q >> 5)] |= 1u << (q & 31);
        }
      }
      if (y < t.YH) {  // blocked-circle membership per speed (vfh.cpp:1140-1189)
        for (int s = 0; s <= MAX_SPEED; s++) {
          const float cxr = CX + (t.mtr[s] / (float)CELL_WIDTH);
          const float cxl = CX - (t.mtr[s] / (float)CELL_WIDTH);
          const float cy = (float)CY;
          const float dr = k_hypotf(cxr - x, cy - y) * CELL_WIDTH;
          const float dl = k_hypotf(cxl - x, cy - y) * CELL_WIDTH;
          if (dr < t.bcr[s]) t.in_circle[((size_t)s * 2 + 0) * t.NW + (q >> 5)] |= 1u << (q & 31);
          if (dl < t.bcr[s]) t.in_circle[((size_t)s * 2 + 1) * t.NW + (q >> 5)] |= 1u << (q & 31);
        }
      }
    }
  }
  return true;
}

template <typename T>
int upload(rna_engine* e, T** dst, const std::vector<T>& src) {
  int rc = dev_alloc(e, dst, src.size());
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, e->stream));
  return RNA_OK;
}

VfhK make_k(const rna_engine* e) {
  const VfhDevice& v = e->vfh;
  VfhK K{};
  K.W = v.W; K.H = v.H; K.T = v.T; K.CX = v.CX; K.CY = v.CY; K.YH = v.NQF / v.W; K.NQ = v.NQ; K.NQF = v.NQF;
  K.NW = v.NW; K.max_speed = v.max_speed; K.sector_angle = v.p.sector_angle;
  K.cell_width = (float)v.p.cell_size; K.robot_radius = (float)v.p.robot_radius;
  K.sd0 = (float)v.p.safety_dist_0ms; K.sd1 = (float)v.p.safety_dist_1ms;
  K.bl0 = (float)v.p.free_space_cutoff_0ms; K.bh0 = (float)v.p.obs_cutoff_0ms;
  K.bl1 = (float)v.p.free_space_cutoff_1ms; K.bh1 = (float)v.p.obs_cutoff_1ms;
  K.U1 = (float)v.p.weight_desired_dir; K.U2 = (float)v.p.weight_current_dir;
  K.current_max_speed = v.p.max_speed; K.max_speed_narrow = v.p.max_speed_narrow_opening;
  K.max_speed_wide = v.p.max_speed_wide_opening; K.max_accel = v.p.max_acceleration;
  K.mt0 = v.p.max_turnrate_0ms; K.mt1 = v.p.max_turnrate_1ms;
  K.cell_dist = v.cell_dist; K.cell_base_mag = v.cell_base_mag; K.cell_dir = v.cell_dir; K.range_idx = v.range_idx;
  K.memb = v.memb; K.in_circle = v.in_circle; K.mtr = v.min_turning_radius; K.bcr = v.blocked_radius + v.n_robots;
  K.last_binary = v.last_binary; K.hist = v.hist; K.origin = v.origin; K.picked = v.picked;
  K.last_picked = v.last_picked; K.blocked_radius = v.blocked_radius; K.last_chosen_speed = v.last_chosen_speed;
  K.max_speed_picked = v.last_chosen_speed + v.n_robots;
  return K;
}

int ensure_staging(rna_engine* e, bool ranges) {
  VfhDevice& v = e->vfh;
  int rc;
  if (!v.poses_dev) {
    if ((rc = dev_alloc(e, &v.poses_dev, (size_t)v.n_robots)) != RNA_OK) return rc;
    if ((rc = dev_alloc(e, &v.out_dev, (size_t)v.n_robots)) != RNA_OK) return rc;
  }
  if (ranges && !v.ranges_dev)
    if ((rc = dev_alloc(e, &v.ranges_dev, (size_t)v.n_robots * 361 * 2)) != RNA_OK) return rc;
  return RNA_OK;
}

// `side`: on the engine's VFH+ stream, behind an event of the engine stream (rna_engine::vfh_stream); the map update that
// follows only joins it where it writes the master layer
int launch_step(rna_engine* e, const rna_pose* poses_dev, const double* ranges_dev, int n, rna_vfh_out* out_dev,
                float* origin_dev, float* hist_dev, bool side = false) {
  hipStream_t st = e->stream;
  if (side) {
    { hipStream_t ss; const int rc = side_stream(e, &ss); if (rc != RNA_OK) return rc; }
    RNA_HIP(e, hipEventRecord(e->ev_vfh_go, e->stream));
    RNA_HIP(e, hipStreamWaitEvent(e->vfh_stream, e->ev_vfh_go, 0));
    st = e->vfh_stream;
  }
  {
    // (the kernel wraps a window cell's buffer index with one compare: the start index rna_move leaves is in [0, size))
    for (int a = 0; a < 2; ++a)
      if (e->geom.start[a] < 0 || e->geom.start[a] >= e->geom.size[a]) return fail(e, RNA_EINVAL, "vfh: buffer start index outside the map");
    KernelTimer kt(e, RNA_K_VFH_STEP, st);
    const VfhK K = make_k(e);
    const size_t lds = (size_t)(((K.NQ + 31) & ~31) + K.NW) * sizeof(float);   // mag | nz
#ifdef RNA_VFH_SKIPS
    { const char* sk = getenv("RNA_VFH_SKIP"); const int v = sk ? atoi(sk) : 0; RNA_HIP(e, hipMemcpyToSymbol(HIP_SYMBOL(d_vfh_skip), &v, sizeof v)); }
    { const char* sk = getenv("RNA_VFH_EXIT"); const int v = sk ? atoi(sk) : 0; RNA_HIP(e, hipMemcpyToSymbol(HIP_SYMBOL(d_vfh_exit), &v, sizeof v)); }
#endif
    hipLaunchKernelGGL(vfh_step_kernel, dim3(n), dim3(VFH_THREADS), lds, st, K, e->geom,
                       e->layer[RNA_LAYER_MASTER], poses_dev, ranges_dev, out_dev, origin_dev, hist_dev);
    RNA_HIP(e, hipGetLastError());
  }
  if (side) {
    RNA_HIP(e, hipEventRecord(e->ev_vfh_done, e->vfh_stream));
    e->vfh_pending = true;
  }
  return RNA_OK;
}

}  // namespace

namespace rna {
int vfh_release(rna_engine* e) {
  VfhDevice& v = e->vfh;
  dev_free(&v.cell_dist); dev_free(&v.cell_base_mag); dev_free(&v.cell_dir); dev_free(&v.range_idx);
  dev_free(&v.memb); dev_free(&v.in_circle); dev_free(&v.min_turning_radius);
  dev_free(&v.last_binary); dev_free(&v.hist); dev_free(&v.origin); dev_free(&v.picked); dev_free(&v.last_picked);
  dev_free(&v.blocked_radius); dev_free(&v.last_chosen_speed);
  dev_free(&v.poses_dev); dev_free(&v.out_dev); dev_free(&v.ranges_dev);
  v.ready = false;
  return RNA_OK;
}
}  // namespace rna

extern "C" void rna_vfh_default_params(rna_vfh_params* p) {  // Steerer::initVfh, steerer.cpp:69-121
  if (!p) return;
  p->cell_size = 100; p->window_diameter = 30; p->sector_angle = 5;
  p->safety_dist_0ms = 10; p->safety_dist_1ms = 50;
  p->max_speed = 200; p->max_speed_narrow_opening = 200; p->max_speed_wide_opening = 300;
  p->max_acceleration = 200; p->min_turnrate = 40; p->max_turnrate_0ms = 40; p->max_turnrate_1ms = 40;
  p->min_turn_radius_safety_factor = 1.0;
  p->free_space_cutoff_0ms = 2000000.0; p->obs_cutoff_0ms = 4000000.0;
  p->free_space_cutoff_1ms = 2000000.0; p->obs_cutoff_1ms = 4000000.0;
  p->weight_desired_dir = 10.0; p->weight_current_dir = 1.0;
  p->robot_radius = 178.0;
}

extern "C" int rna_vfh_init(rna_engine* e, const rna_vfh_params* p, int n_robots) {
  if (!e || !p || n_robots <= 0) return RNA_EINVAL;
  RNA_ENTER(e);
  HostTables t;
  if (!build_tables(*p, t)) return fail(e, RNA_EINVAL, "rna_vfh_init: unsupported VFH parameters");
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  vfh_release(e);
  VfhDevice& v = e->vfh;
  v.p = *p;
  v.n_robots = n_robots;
  v.W = t.W; v.H = t.H; v.T = t.T; v.CX = t.CX; v.CY = t.CY; v.NQ = t.NQ; v.NQF = t.NQF; v.NW = t.NW;
  v.max_speed = t.max_speed;
  int rc;
  if ((rc = upload(e, &v.cell_dist, t.cell_dist)) != RNA_OK) return rc;
  if ((rc = upload(e, &v.cell_base_mag, t.cell_base_mag)) != RNA_OK) return rc;
  if ((rc = upload(e, &v.cell_dir, t.cell_dir)) != RNA_OK) return rc;
  if ((rc = upload(e, &v.range_idx, t.range_idx)) != RNA_OK) return rc;
  if ((rc = upload(e, &v.memb, t.memb)) != RNA_OK) return rc;
  if ((rc = upload(e, &v.in_circle, t.in_circle)) != RNA_OK) return rc;
  if ((rc = upload(e, &v.min_turning_radius, t.mtr)) != RNA_OK) return rc;
  const size_t nh = (size_t)n_robots * t.H;
  if ((rc = dev_alloc(e, &v.last_binary, nh)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &v.hist, nh)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &v.origin, nh)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &v.picked, (size_t)n_robots)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &v.last_picked, (size_t)n_robots)) != RNA_OK) return rc;
  // blocked_radius: [n_robots] per-robot state followed by the [max_speed+1] per-speed table
  if ((rc = dev_alloc(e, &v.blocked_radius, (size_t)n_robots + t.bcr.size())) != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(v.blocked_radius + n_robots, t.bcr.data(), t.bcr.size() * sizeof(float),
                            hipMemcpyHostToDevice, e->stream));
  // last_chosen_speed: [n_robots] followed by Max_Speed_For_Picked_Angle [n_robots]
  if ((rc = dev_alloc(e, &v.last_chosen_speed, (size_t)2 * n_robots)) != RNA_OK) return rc;
  RNA_HIP(e, hipStreamSynchronize(e->stream));  // host vectors go out of scope
  v.ready = true;
  return rna_vfh_reset(e);
}

extern "C" int rna_vfh_reset(rna_engine* e) {
  if (!e) return RNA_EINVAL;
  if (!e->vfh.ready) return fail(e, RNA_ESTATE, "rna_vfh_reset before rna_vfh_init");
  RNA_ENTER(e);
  const int n = e->vfh.n_robots;
  const int threads = n * e->vfh.H;
  hipLaunchKernelGGL(vfh_reset_kernel, dim3((threads + 255) / 256), dim3(256), 0, e->stream, make_k(e), n);
  RNA_HIP(e, hipGetLastError());
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_vfh_hist_size(const rna_engine* e) { return (e && e->vfh.ready) ? e->vfh.H : RNA_ESTATE; }

extern "C" int rna_vfh_step_batch_device(rna_engine* e, const rna_pose* poses, int n, rna_vfh_out* out,
                                         float* origin_hist, float* hist) {
  if (!e || !poses || !out || n < 0) return RNA_EINVAL;
  if (!e->vfh.ready) return fail(e, RNA_ESTATE, "rna_vfh_step_batch before rna_vfh_init");
  if (n > e->vfh.n_robots) return fail(e, RNA_EINVAL, "more poses than VFH instances");
  if (n == 0) return RNA_OK;
  RNA_ENTER_NOJOIN(e);   // (steps on the VFH+ stream follow one another there)
  // next to pipelined searches the step runs on its own stream: it only reads the master layer and the robots' state
  return launch_step(e, poses, nullptr, n, out, origin_hist, hist, e->astar.depth > 1 && !getenv("RNA_VFH_INLINE"));
}

static int step_host(rna_engine* e, const double* ranges_host, const rna_pose* poses_host, int n,
                     rna_vfh_out* out_host, float* origin_host, float* hist_host) {
  if (!e || !poses_host || !out_host || n < 0) return RNA_EINVAL;
  if (!e->vfh.ready) return fail(e, RNA_ESTATE, "VFH step before rna_vfh_init");
  VfhDevice& v = e->vfh;
  if (n > v.n_robots) return fail(e, RNA_EINVAL, "more poses than VFH instances");
  if (n == 0) return RNA_OK;
  RNA_ENTER(e);
  int rc = ensure_staging(e, ranges_host != nullptr);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(v.poses_dev, poses_host, (size_t)n * sizeof(rna_pose), hipMemcpyHostToDevice, e->stream));
  if (ranges_host)
    RNA_HIP(e, hipMemcpyAsync(v.ranges_dev, ranges_host, (size_t)n * 361 * 2 * sizeof(double), hipMemcpyHostToDevice,
                              e->stream));
  rc = launch_step(e, v.poses_dev, ranges_host ? v.ranges_dev : nullptr, n, v.out_dev, nullptr, nullptr);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(out_host, v.out_dev, (size_t)n * sizeof(rna_vfh_out), hipMemcpyDeviceToHost, e->stream));
  if (origin_host)
    RNA_HIP(e, hipMemcpyAsync(origin_host, v.origin, (size_t)n * v.H * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (hist_host)
    RNA_HIP(e, hipMemcpyAsync(hist_host, v.hist, (size_t)n * v.H * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_vfh_step_batch(rna_engine* e, const rna_pose* poses_host, int n, rna_vfh_out* out_host,
                                  float* origin_hist_host, float* hist_host) {
  return step_host(e, nullptr, poses_host, n, out_host, origin_hist_host, hist_host);
}

extern "C" int rna_vfh_update_batch(rna_engine* e, const double* ranges_host, const rna_pose* poses_host, int n,
                                    rna_vfh_out* out_host, float* origin_hist_host, float* hist_host) {
  if (!ranges_host) return RNA_EINVAL;
  return step_host(e, ranges_host, poses_host, n, out_host, origin_hist_host, hist_host);
}
