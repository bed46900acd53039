// scan.hip -- laser ingestion as a device pre-pass: sensor_msgs/LaserScan
// batches go straight into rna_ray records, in scan order then beam order, ready for
// rna_update_map_device.  Replaces LaserMapUpdater::bufferIncomingMsg
// (move_control/src/laser_map_updater.cpp:37-75): simplifyLaserScan (:118-143), the planar,
// time-constant case of laser_geometry::LaserProjection::transformLaserScanToPointCloud (:78-99) and
// the ray origin from tf::transformPoint (:101-116).  laser_geometry / tf are external ROS packages
// (move_control/package.xml:16-29, not version-pinned): see oracle/scan.c for the restated algorithm
// and the reference quirks that are kept.
#include "engine.hpp"

using namespace rna;

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_MAX_BEAMS = 8192;   // simplified beams per scan held in LDS

// One workgroup per scan.  Thread 0 replays the decimation (a sequential float accumulation), then all
// threads project, filter and transform the beams; an ordered block-wide compaction keeps beam order.
__global__ void __launch_bounds__(SCAN_THREADS)
scan_to_rays_kernel(const rna_laser_scan* __restrict__ scans, const float* __restrict__ ranges, int max_rays_per_scan,
                    rna_ray* __restrict__ staged, int* __restrict__ counts) {
  __shared__ unsigned short s_sel[SCAN_MAX_BEAMS];
  __shared__ int s_n;
  __shared__ float s_inc;
  __shared__ int s_wave_cnt[SCAN_THREADS / 64];
  __shared__ int s_base;
  const rna_laser_scan sc = scans[blockIdx.x];
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
  // tf's start / end transforms interpolated per beam (laser_geometry's high-fidelity projection): position linearly,
  // yaw along the shortest arc; a constant pose (end == start) reduces to one rotation
  double dyaw = fmod(sc.yaw_end - sc.yaw, 2.0 * M_PI);
  if (dyaw > M_PI) dyaw -= 2.0 * M_PI;
  if (dyaw < -M_PI) dyaw += 2.0 * M_PI;
  const double ranges_norm = n > 1 ? 1.0 / ((double)n - 1.0) : 0.0;
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
        const double ratio = (double)i * ranges_norm, keep = 1.0 - ratio;
        const double yaw_i = sc.yaw + ratio * dyaw;
        const double cy = cos(yaw_i), sy = sin(yaw_i);
        const double tx = keep * sc.x + ratio * sc.x_end, ty = keep * sc.y + ratio * sc.y_end;
        ray.sx = sc.x; ray.sy = sc.y;
        ray.ex = (double)(float)(cy * (double)px - sy * (double)py + tx);         // tf transform, float32 again
        ray.ey = (double)(float)(sy * (double)px + cy * (double)py + ty);
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

extern "C" int rna_scan_to_rays_device(rna_engine* e, const rna_laser_scan* scans_device, int n_scans,
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
    hipLaunchKernelGGL(scan_to_rays_kernel, dim3(n_scans), dim3(SCAN_THREADS), 0, e->stream, scans_device, ranges_device,
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

extern "C" int rna_scan_to_rays(rna_engine* e, const rna_laser_scan* scans_host, int n_scans, const float* ranges_host,
                                size_t n_ranges_total, rna_ray* rays_host, int max_rays, int* n_rays) {
  if (!e || n_scans < 0 || max_rays < 0 || !n_rays) return RNA_EINVAL;
  *n_rays = 0;
  if (n_scans == 0) return RNA_OK;
  if (!scans_host || !ranges_host || (max_rays > 0 && !rays_host)) return RNA_EINVAL;
  int max_beams = 1;
  for (int s = 0; s < n_scans; ++s) {
    const rna_laser_scan& sc = scans_host[s];
    if (sc.n_ranges < 0 || sc.ranges_offset < 0 || (size_t)sc.ranges_offset + (size_t)sc.n_ranges > n_ranges_total)
      return fail(e, RNA_EINVAL, "rna_scan_to_rays: scan ranges outside the ranges array");
    if (sc.n_ranges > max_beams) max_beams = sc.n_ranges;
  }
  if (max_beams > SCAN_MAX_BEAMS) return fail(e, RNA_EINVAL, "rna_scan_to_rays: more than 8192 beams per scan");
  RNA_ENTER(e);
  rna_laser_scan* d_scans = nullptr;
  float* d_ranges = nullptr;
  rna_ray* d_rays = nullptr;
  int* d_n = nullptr;
  int rc = dev_alloc(e, &d_scans, (size_t)n_scans);
  if (rc == RNA_OK) rc = dev_alloc(e, &d_ranges, n_ranges_total ? n_ranges_total : 1);
  if (rc == RNA_OK) rc = dev_alloc(e, &d_rays, (size_t)(max_rays ? max_rays : 1));
  if (rc == RNA_OK) rc = dev_alloc(e, &d_n, (size_t)1);
  hipError_t st = hipSuccess;
  if (rc == RNA_OK) {
    st = hipMemcpyAsync(d_scans, scans_host, sizeof(rna_laser_scan) * n_scans, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess && n_ranges_total)
      st = hipMemcpyAsync(d_ranges, ranges_host, sizeof(float) * n_ranges_total, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess) rc = rna_scan_to_rays_device(e, d_scans, n_scans, d_ranges, max_beams, d_rays, max_rays, d_n);
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
