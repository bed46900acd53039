/*
 * oracle/scan.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
 *
 * Laser ingestion: LaserMapUpdater::bufferIncomingMsg turns one
 * sensor_msgs/LaserScan into RangeSamples (move_control/src/laser_map_updater.cpp:37-75) via
 *   simplifyLaserScan            (laser_map_updater.cpp:118-143)   decimation to >= 0.017 rad,
 *   laser_geometry::LaserProjection::transformLaserScanToPointCloud (laser_map_updater.cpp:78-99),
 *   tf::TransformListener::transformPoint for the ray origin       (laser_map_updater.cpp:101-116).
 * laser_geometry and tf are ROS packages that are NOT under /root/reference and are not
 * version-pinned by it (move_control/package.xml:16-29 lists them without versions; ROS Indigo
 * ships laser_geometry 1.6.x, tf 1.11.x).  Their published algorithm is restated here for the case
 * the engine supports: planar sensor poses (x, y, yaw) in the map frame at the scan's start time (header.stamp)
 * and end time (stamp + (beams - 1) * time_increment, beams of the PROJECTED scan), as
 * transformLaserScanToPointCloud looks them up:
 *   projectLaser_:  x = r*cos(angle_min + i*angle_increment), y = r*sin(...) in double, a point is
 *                   emitted iff r < range_max && r >= range_min, stored as float32 with its index;
 *   per point:      ratio = index / (beams - 1); origin = (1 - ratio) * t_start + ratio * t_end
 *                   (tf::Vector3::setInterpolate3), rotation = slerp(q_start, q_end, ratio) -- for rotations about z
 *                   the yaw interpolated along the shortest arc;
 *   transform:      p' = R(yaw_i) p + t_i in double on the float32 point, stored as float32 again.
 * (A scan of one beam divides by zero there; defined here: ratio = 0.)
 * PARITY UNPINNED: no reference test covers this path and the dependencies cannot be built here.
 * Reference quirks kept: the point's index refers to the SIMPLIFIED scan but ifClearEnd looks it
 * up in the ORIGINAL ranges (laser_map_updater.cpp:62-69); the simplified scan starts with
 * ranges[0] twice over (index 0 is pushed, then the loop starts at i = 0).
 */
#define _GNU_SOURCE   /* sincos */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "rna_oracle.h"

/* laser_map_updater.cpp:118-143; sel[k] = index into the original ranges of simplified beam k */
int og_simplify_scan(int n, float angle_increment, int* sel, int cap, float* out_increment) {
  int m = 0;
  float increment = 0.0f;
  *out_increment = 0.0f;               /* a fresh LaserScan message is zero-initialised */
  if (n <= 0) return 0;
  if (m < cap) sel[m] = 0;
  ++m;
  for (int i = 0; i < n; ++i) {
    increment += angle_increment;
    if (increment >= 0.017) {
      *out_increment = increment;
      increment = 0.0f;
      if (m < cap) sel[m] = i;
      ++m;
    }
  }
  return m;
}

int og_scan_to_rays(const og_scan* s, const float* ranges, og_ray* out, int cap) {
  const float* r = ranges + s->ranges_offset;
  int n = s->n_ranges;
  float inc = s->angle_increment;
  static int sel_buf[1 << 16];
  int* sel = NULL;
  if (s->angle_increment < 0.017) {   /* laser_map_updater.cpp:82 */
    n = og_simplify_scan(s->n_ranges, s->angle_increment, sel_buf, 1 << 16, &inc);
    if (n > (1 << 16)) n = 1 << 16;
    sel = sel_buf;
  }
  const double range_cutoff = s->range_max;
  double dyaw = fmod(s->yaw_end - s->yaw, 2.0 * M_PI);   /* shortest arc, as a quaternion slerp turns */
  if (dyaw > M_PI) dyaw -= 2.0 * M_PI;
  if (dyaw < -M_PI) dyaw += 2.0 * M_PI;
  const double ranges_norm = n > 1 ? 1.0 / ((double)n - 1.0) : 0.0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float range = r[sel ? sel[i] : i];
    if (!(range < range_cutoff && range >= s->range_min)) continue;
    const double a = s->angle_min + (double)i * inc;
    const float px = (float)(range * cos(a)), py = (float)(range * sin(a));
    const double ratio = (double)i * ranges_norm, keep = 1.0 - ratio;
    const double yaw_i = s->yaw + ratio * dyaw;
    const double cy = cos(yaw_i), sy = sin(yaw_i);
    const double tx = keep * s->x + ratio * s->x_end, ty = keep * s->y + ratio * s->y_end;
    const float gx = (float)(cy * (double)px - sy * (double)py + tx);
    const float gy = (float)(sy * (double)px + cy * (double)py + ty);
    const float orig = i < s->n_ranges ? r[i] : r[s->n_ranges - 1];   /* msg->ranges[index] of the ORIGINAL scan */
    if (m < cap) {
      out[m].sx = s->x; out[m].sy = s->y;
      out[m].ex = gx; out[m].ey = gy;
      out[m].clear_end = (isinf(orig) || orig == s->range_max) ? 1 : 0;
      out[m]._pad = 0;
    }
    ++m;
  }
  return m;
}

/* RangeMapUpdater::bufferIncomingMsg (range_map_updater.cpp:38-76): in.point = (0,0,0) -> start, then
 * in.point.x = msg->range -> end, both through tf::transformPoint (double; planar pose restated as R(yaw) p + t as
 * above); ifClearEnd = !(msg->range < msg->max_range) (:52-56).  PARITY UNPINNED like the laser path. */
void og_range_to_ray(float range, float max_range, double x, double y, double yaw, og_ray* out) {
  double c, s;
  sincos(yaw, &s, &c);   /* what gcc makes of cos(yaw), sin(yaw); glibc's sincos and cos differ in the last bit now and then */
  double px = 0.0, py = 0.0;
  out->sx = c * px - s * py + x;
  out->sy = s * px + c * py + y;
  px = range;
  out->ex = c * px - s * py + x;
  out->ey = s * px + c * py + y;
  if (range < max_range) out->clear_end = 0;
  else out->clear_end = 1;
  out->_pad = 0;
}
