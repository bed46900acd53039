// scan.hip -- laser ingestion as a device pre-pass: sensor_msgs/LaserScan
// batches go straight into rna_ray records, in scan order then beam order, ready for
// rna_update_map_device.  Replaces LaserMapUpdater::bufferIncomingMsg
// (move_control/src/laser_map_updater.cpp:37-75): simplifyLaserScan (:118-143),
// laser_geometry::LaserProjection::transformLaserScanToPointCloud (:78-99) -- for a planar sensor pose
// (rna_laser_scan: x, y, yaw) and for the full transform tf reports (rna_laser_scan_tf: translation + quaternion, a
// tilted or raised mount) -- and the ray origin from tf::transformPoint (:101-116).  laser_geometry / tf are external ROS packages
// (move_control/package.xml:16-29, not version-pinned): see oracle/scan.c for the restated algorithm
// and the reference quirks that are kept.
#include "engine.hpp"

using namespace rna;

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_MAX_BEAMS = 8192;   // simplified beams per scan held in LDS

// The transform of beam i of n (ratio = i / (n - 1)) applied to the projected float32 point (px, py, 0): map-frame x, y.
// Planar pose: position linearly, yaw along the shortest arc; a constant pose (end == start) reduces to one rotation.
__device__ __forceinline__ void beam_to_map(const rna_laser_scan& sc, double ratio, float px, float py, double& gx, double& gy) {
  double dyaw = fmod(sc.yaw_end - sc.yaw, 2.0 * M_PI);
  if (dyaw > M_PI) dyaw -= 2.0 * M_PI;
  if (dyaw < -M_PI) dyaw += 2.0 * M_PI;
  const double keep = 1.0 - ratio;
  const double yaw_i = sc.yaw + ratio * dyaw;
  const double cy = cos(yaw_i), sy = sin(yaw_i);
  const double tx = keep * sc.x + ratio * sc.x_end, ty = keep * sc.y + ratio * sc.y_end;
  gx = cy * (double)px - sy * (double)py + tx;
  gy = sy * (double)px + cy * (double)py + ty;
}
__device__ __forceinline__ void scan_origin(const rna_laser_scan& sc, double& ox, double& oy) { ox = sc.x; oy = sc.y; }

// Full transform, as tf's LinearMath does it in double (tf/LinearMath/{Quaternion,Matrix3x3,Transform}.h; oracle/scan.c
// restates the same lines): Vector3::setInterpolate3, Quaternion::slerp through angleShortestPath, Matrix3x3::setRotation,
// Transform::operator*.
__device__ __forceinline__ double quat_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ __forceinline__ void beam_to_map(const rna_laser_scan_tf& sc, double ratio, float px, float py, double& gx, double& gy) {
  const double* q0 = sc.q;
  const double* q1 = sc.q_end;
  // v.setInterpolate3(origin_start, origin_end, ratio)
  const double keep = 1.0 - ratio;
  const double tx = keep * sc.t[0] + ratio * sc.t_end[0], ty = keep * sc.t[1] + ratio * sc.t_end[1];
  // slerp(quat_start, quat_end, ratio)
  double q[4] = {q0[0], q0[1], q0[2], q0[3]};
  {
    const double sl = sqrt(quat_dot(q0, q0) * quat_dot(q1, q1));
    const double dt = quat_dot(q0, q1);
    // tfAcos (tf/LinearMath/Scalar.h) clamps its argument to [-1, 1]: for two orientations a few 1e-9 rad apart the
    // quotient rounds to 1 + ulp in a fifth of the cases, and an unclamped acos made every beam of the scan NaN
    double ca = (dt < 0 ? -dt : dt) / sl;
    ca = ca < -1.0 ? -1.0 : (ca > 1.0 ? 1.0 : ca);
    const double theta = (acos(ca) * 2.0) / 2.0;   // angleShortestPath(q) / 2
    if (theta != 0.0) {
      const double d = 1.0 / sin(theta);
      const double s0 = sin((1.0 - ratio) * theta);
      const double s1 = sin(ratio * theta);
      if (dt < 0) { for (int k = 0; k < 4; ++k) q[k] = (q0[k] * s0 + -q1[k] * s1) * d; }
      else { for (int k = 0; k < 4; ++k) q[k] = (q0[k] * s0 + q1[k] * s1) * d; }
    }
  }
  // Matrix3x3::setRotation(q), rows 0 and 1
  const double dd = quat_dot(q, q);
  const double s = 2.0 / dd;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  const double x = (double)px, y = (double)py, z = 0.0;
  gx = ((1.0 - (yy + zz)) * x + (xy - wz) * y + (xz + wy) * z) + tx;
  gy = ((xy + wz) * x + (1.0 - (xx + zz)) * y + (yz - wx) * z) + ty;
}
__device__ __forceinline__ void scan_origin(const rna_laser_scan_tf& sc, double& ox, double& oy) { ox = sc.t[0]; oy = sc.t[1]; }

// One workgroup per scan.  Thread 0 replays the decimation (a sequential float accumulation), then all
// threads project, filter and transform the beams; an ordered block-wide compaction keeps beam order.
template <class Scan>
__global__ void __launch_bounds__(SCAN_THREADS)
scan_to_rays_kernel(const Scan* __restrict__ scans, const float* __restrict__ ranges, int max_rays_per_scan,
                    rna_ray* __restrict__ staged, int* __restrict__ counts) {
  __shared__ unsigned short s_sel[SCAN_MAX_BEAMS];
  __shared__ int s_n;
  __shared__ float s_inc;
  __shared__ int s_wave_cnt[SCAN_THREADS / 64];
  __shared__ int s_base;
  const Scan sc = scans[blockIdx.x];
  const float* r = ranges + sc.ranges_offset;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool simplify = sc.angle_increment < 0.017;   // laser_map_updater.cpp:82
  if (tid == 0) {
    int m = sc.n_ranges;
    float inc = sc.angle_increment;
    if (simplify) {                                   // laser_map_updater.cpp:118-143
      m = 0;
      inc = 0.0f;
      float increment = 0.0f;
      if (sc.n_ranges > 0) { s_sel[0] = 0; m = 1; }
      for (int i = 0; i < sc.n_ranges; ++i) {
        increment += sc.angle_increment;
        if (increment >= 0.017) {
          inc = increment;
          increment = 0.0f;
          if (m < SCAN_MAX_BEAMS) s_sel[m] = (unsigned short)i;
          ++m;
        }
      }
    }
    s_n = m < SCAN_MAX_BEAMS ? m : SCAN_MAX_BEAMS;
    s_inc = inc;
    s_base = 0;
  }
  __syncthreads();
  const int n = s_n;
  const float inc = s_inc;
  const double range_cutoff = sc.range_max;
  // tf's start / end transforms interpolated per beam (laser_geometry's high-fidelity projection)
  const double ranges_norm = n > 1 ? 1.0 / ((double)n - 1.0) : 0.0;
  double ox, oy;
  scan_origin(sc, ox, oy);
  rna_ray* out = staged + (size_t)blockIdx.x * max_rays_per_scan;
  for (int i0 = 0; i0 < n; i0 += SCAN_THREADS) {
    const int i = i0 + tid;
    bool valid = false;
    rna_ray ray;
    if (i < n) {
      const float range = r[simplify ? (int)s_sel[i] : i];
      valid = range < range_cutoff && range >= sc.range_min;
      if (valid) {
        const double a = sc.angle_min + (double)i * inc;
        const float px = (float)(range * cos(a)), py = (float)(range * sin(a));   // projectLaser_: float32 point
        double gx, gy;
        beam_to_map(sc, (double)i * ranges_norm, px, py, gx, gy);
        ray.sx = ox; ray.sy = oy;
        ray.ex = (double)(float)gx;                                               // tf transform, float32 again
        ray.ey = (double)(float)gy;
        const float orig = r[i < sc.n_ranges ? i : sc.n_ranges - 1];              // ORIGINAL ranges[index], :62-69
        ray.clear_end = (isinf(orig) || orig == sc.range_max) ? 1 : 0;
        ray._pad = 0;
      }
    }
    const unsigned long long m = __ballot(valid);
    if (lane == 0) s_wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wv; ++w) before += s_wave_cnt[w];
    const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
    if (valid && pos < max_rays_per_scan) out[pos] = ray;
    __syncthreads();
    if (tid == 0) {
      int t = s_base;
      for (int w = 0; w < SCAN_THREADS / 64; ++w) t += s_wave_cnt[w];
      s_base = t;
    }
    __syncthreads();
  }
  if (tid == 0) counts[blockIdx.x] = s_base < max_rays_per_scan ? s_base : max_rays_per_scan;
}

// pack the per-scan segments into one contiguous, ordered ray array (single workgroup: n_scans is small)
__global__ void __launch_bounds__(256) scan_pack_kernel(const rna_ray* __restrict__ staged, const int* __restrict__ counts,
                                                        int n_scans, int max_rays_per_scan, rna_ray* __restrict__ out,
                                                        int max_rays, int* __restrict__ n_out) {
  __shared__ int s_off;
  if (threadIdx.x == 0) s_off = 0;
  __syncthreads();
  for (int s = 0; s < n_scans; ++s) {
    const int c = counts[s], off = s_off;
    for (int k = threadIdx.x; k < c; k += blockDim.x)
      if (off + k < max_rays) out[off + k] = staged[(size_t)s * max_rays_per_scan + k];
    __syncthreads();
    if (threadIdx.x == 0) s_off = off + c;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = s_off;   // may exceed max_rays: the caller sees that the output was truncated
}

}  // namespace

extern "C" int rna_scan_projected_beams(int n_ranges, float angle_increment) {
  if (n_ranges <= 0) return 0;
  if (!(angle_increment < 0.017)) return n_ranges;   // laser_map_updater.cpp:82
  int m = 1;                                          // simplifyLaserScan (:118-143)
  float increment = 0.0f;
  for (int i = 0; i < n_ranges; ++i) {
    increment += angle_increment;
    if (increment >= 0.017) { increment = 0.0f; ++m; }
  }
  return m;
}

template <class Scan>
static int scan_to_rays_device_t(rna_engine* e, const Scan* scans_device, int n_scans,
                                       const float* ranges_device, int max_beams_per_scan, rna_ray* rays_device, int max_rays,
                                       int* n_rays_device) {
  if (!e || n_scans < 0 || max_rays < 0 || max_beams_per_scan <= 0 || !n_rays_device) return RNA_EINVAL;
  if (n_scans > 0 && (!scans_device || !ranges_device || !rays_device)) return RNA_EINVAL;
  if (max_beams_per_scan > SCAN_MAX_BEAMS) return fail(e, RNA_EINVAL, "rna_scan_to_rays: more than 8192 beams per scan");
  RNA_ENTER(e);
  if (n_scans == 0) {
    RNA_HIP(e, hipMemsetAsync(n_rays_device, 0, sizeof(int), e->stream));
    return RNA_OK;
  }
  rna_ray* staged = nullptr;
  int* counts = nullptr;
  int rc = dev_alloc(e, &staged, (size_t)n_scans * max_beams_per_scan);
  if (rc == RNA_OK) rc = dev_alloc(e, &counts, (size_t)n_scans);
  if (rc == RNA_OK) {
    hipLaunchKernelGGL(scan_to_rays_kernel<Scan>, dim3(n_scans), dim3(SCAN_THREADS), 0, e->stream, scans_device, ranges_device,
                       max_beams_per_scan, staged, counts);
    hipLaunchKernelGGL(scan_pack_kernel, dim3(1), dim3(256), 0, e->stream, staged, counts, n_scans, max_beams_per_scan,
                       rays_device, max_rays, n_rays_device);
    if (hipGetLastError() != hipSuccess) rc = fail(e, RNA_EHIP, "rna_scan_to_rays: launch failed");
  }
  // the staging buffers are freed once the kernels are done (hipFree synchronises)
  hipError_t st = hipStreamSynchronize(e->stream);
  dev_free(&staged);
  dev_free(&counts);
  if (rc != RNA_OK) return rc;
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  return RNA_OK;
}

template <class Scan>
static int scan_to_rays_host_t(rna_engine* e, const Scan* scans_host, int n_scans, const float* ranges_host,
                                size_t n_ranges_total, rna_ray* rays_host, int max_rays, int* n_rays) {
  if (!e || n_scans < 0 || max_rays < 0 || !n_rays) return RNA_EINVAL;
  *n_rays = 0;
  if (n_scans == 0) return RNA_OK;
  if (!scans_host || !ranges_host || (max_rays > 0 && !rays_host)) return RNA_EINVAL;
  int max_beams = 1;
  for (int s = 0; s < n_scans; ++s) {
    const Scan& sc = scans_host[s];
    if (sc.n_ranges < 0 || sc.ranges_offset < 0 || (size_t)sc.ranges_offset + (size_t)sc.n_ranges > n_ranges_total)
      return fail(e, RNA_EINVAL, "rna_scan_to_rays: scan ranges outside the ranges array");
    if (sc.n_ranges > max_beams) max_beams = sc.n_ranges;
  }
  if (max_beams > SCAN_MAX_BEAMS) return fail(e, RNA_EINVAL, "rna_scan_to_rays: more than 8192 beams per scan");
  RNA_ENTER(e);
  Scan* d_scans = nullptr;
  float* d_ranges = nullptr;
  rna_ray* d_rays = nullptr;
  int* d_n = nullptr;
  int rc = dev_alloc(e, &d_scans, (size_t)n_scans);
  if (rc == RNA_OK) rc = dev_alloc(e, &d_ranges, n_ranges_total ? n_ranges_total : 1);
  if (rc == RNA_OK) rc = dev_alloc(e, &d_rays, (size_t)(max_rays ? max_rays : 1));
  if (rc == RNA_OK) rc = dev_alloc(e, &d_n, (size_t)1);
  hipError_t st = hipSuccess;
  if (rc == RNA_OK) {
    st = hipMemcpyAsync(d_scans, scans_host, sizeof(Scan) * n_scans, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess && n_ranges_total)
      st = hipMemcpyAsync(d_ranges, ranges_host, sizeof(float) * n_ranges_total, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess) rc = scan_to_rays_device_t<Scan>(e, d_scans, n_scans, d_ranges, max_beams, d_rays, max_rays, d_n);
    if (st == hipSuccess && rc == RNA_OK) st = hipMemcpyAsync(n_rays, d_n, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess && rc == RNA_OK) st = hipStreamSynchronize(e->stream);
    if (st == hipSuccess && rc == RNA_OK) {
      const int have = *n_rays < max_rays ? *n_rays : max_rays;
      if (have > 0) st = hipMemcpy(rays_host, d_rays, sizeof(rna_ray) * have, hipMemcpyDeviceToHost);
    }
  }
  dev_free(&d_scans); dev_free(&d_ranges); dev_free(&d_rays); dev_free(&d_n);
  if (rc != RNA_OK) return rc;
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  if (*n_rays > max_rays) return fail(e, RNA_ECAPACITY, "rna_scan_to_rays: max_rays too small (n_rays holds the required count)");
  return RNA_OK;
}

extern "C" int rna_scan_to_rays_device(rna_engine* e, const rna_laser_scan* scans_device, int n_scans,
                                       const float* ranges_device, int max_beams_per_scan, rna_ray* rays_device, int max_rays,
                                       int* n_rays_device) {
  return scan_to_rays_device_t(e, scans_device, n_scans, ranges_device, max_beams_per_scan, rays_device, max_rays, n_rays_device);
}
extern "C" int rna_scan_to_rays(rna_engine* e, const rna_laser_scan* scans_host, int n_scans, const float* ranges_host,
                                size_t n_ranges_total, rna_ray* rays_host, int max_rays, int* n_rays) {
  return scan_to_rays_host_t(e, scans_host, n_scans, ranges_host, n_ranges_total, rays_host, max_rays, n_rays);
}
extern "C" int rna_scan_to_rays_tf_device(rna_engine* e, const rna_laser_scan_tf* scans_device, int n_scans,
                                          const float* ranges_device, int max_beams_per_scan, rna_ray* rays_device, int max_rays,
                                          int* n_rays_device) {
  return scan_to_rays_device_t(e, scans_device, n_scans, ranges_device, max_beams_per_scan, rays_device, max_rays, n_rays_device);
}
extern "C" int rna_scan_to_rays_tf(rna_engine* e, const rna_laser_scan_tf* scans_host, int n_scans, const float* ranges_host,
                                   size_t n_ranges_total, rna_ray* rays_host, int max_rays, int* n_rays) {
  return scan_to_rays_host_t(e, scans_host, n_scans, ranges_host, n_ranges_total, rays_host, max_rays, n_rays);
}

// RangeMapUpdater::bufferIncomingMsg (mc/src/range_map_updater.cpp:38-76): two tf::transformPoint calls per sonar
// reading, in double, for a planar sensor pose.  Host only (five sonars per cycle): cos / sin are the host libm's.
extern "C" int rna_range_to_rays(const rna_range_reading* readings, int n, rna_ray* rays) {
  if (n < 0 || (n > 0 && (!readings || !rays))) return RNA_EINVAL;
  for (int k = 0; k < n; ++k) {
    const rna_range_reading& m = readings[k];
    double c, s;
    sincos(m.yaw, &s, &c);                         // one libm call for both, stated explicitly (host glibc)
    const double r = (double)m.range;              // in.point.x = msg->range
    rays[k].sx = c * 0.0 - s * 0.0 + m.x;
    rays[k].sy = s * 0.0 + c * 0.0 + m.y;
    rays[k].ex = c * r - s * 0.0 + m.x;
    rays[k].ey = s * r + c * 0.0 + m.y;
    rays[k].clear_end = (m.range < m.max_range) ? 0 : 1;
    rays[k]._pad = 0;
  }
  return RNA_OK;
}

// ... and for the sensor's full pose: tf::Transform(q, t) * point in double, x and y kept (Matrix3x3::setRotation, rows 0 / 1)
extern "C" int rna_range_to_rays_tf(const rna_range_reading_tf* readings, int n, rna_ray* rays) {
  if (n < 0 || (n > 0 && (!readings || !rays))) return RNA_EINVAL;
  for (int k = 0; k < n; ++k) {
    const rna_range_reading_tf& m = readings[k];
    const double* q = m.q;
    const double dd = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double s = 2.0 / dd;
    const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
    const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
    const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
    const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
    const double m00 = 1.0 - (yy + zz), m01 = xy - wz, m02 = xz + wy;
    const double m10 = xy + wz, m11 = 1.0 - (xx + zz), m12 = yz - wx;
    const double r = (double)m.range;
    rays[k].sx = (m00 * 0.0 + m01 * 0.0 + m02 * 0.0) + m.t[0];
    rays[k].sy = (m10 * 0.0 + m11 * 0.0 + m12 * 0.0) + m.t[1];
    rays[k].ex = (m00 * r + m01 * 0.0 + m02 * 0.0) + m.t[0];
    rays[k].ey = (m10 * r + m11 * 0.0 + m12 * 0.0) + m.t[1];
    rays[k].clear_end = (m.range < m.max_range) ? 0 : 1;
    rays[k]._pad = 0;
  }
  return RNA_OK;
}
