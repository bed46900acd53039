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
 * og_scan_to_rays_tf takes the two transforms as tf reports them (translation + quaternion) and restates slerp, the
 * quaternion's matrix and the transform of the point from tf's LinearMath headers: any mount, not only planar ones.
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

/* tf::Quaternion::dot / length2 (tf/LinearMath/Quaternion.h) */
static double q_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

/* tf::Quaternion::slerp(q, t) with angleShortestPath (tf/LinearMath/Quaternion.h), as laser_geometry calls it through
 * tf::slerp(quat_start, quat_end, ratio):
 *   theta = angleShortestPath(q) / 2, angleShortestPath = acos(+-dot / sqrt(length2 * q.length2)) * 2 (the sign that makes
 *   the dot product non-negative); theta == 0 returns *this; else d = 1 / sin(theta), s0 = sin((1 - t) theta),
 *   s1 = sin(t theta) and every component (a * s0 + +-b * s1) * d. */
static void q_slerp(const double* a, const double* b, double t, double* out) {
  const double sl = sqrt(q_dot(a, a) * q_dot(b, b));
  const double dt = q_dot(a, b);
  /* tfAcos (tf/LinearMath/Scalar.h): the argument clamped to [-1, 1] -- rounding puts it at 1 + ulp for nearly equal
   * orientations, where acos() itself returns NaN */
  double ca = (dt < 0 ? -dt : dt) / sl;
  if (ca < -1.0) ca = -1.0;
  if (ca > 1.0) ca = 1.0;
  const double theta = (acos(ca) * 2.0) / 2.0;
  if (theta != 0.0) {
    const double d = 1.0 / sin(theta);
    const double s0 = sin((1.0 - t) * theta);
    const double s1 = sin(t * theta);
    if (dt < 0) for (int k = 0; k < 4; ++k) out[k] = (a[k] * s0 + -b[k] * s1) * d;
    else for (int k = 0; k < 4; ++k) out[k] = (a[k] * s0 + b[k] * s1) * d;
  } else {
    for (int k = 0; k < 4; ++k) out[k] = a[k];
  }
}

/* tf::Matrix3x3::setRotation(q) (tf/LinearMath/Matrix3x3.h), rows 0 and 1, then tf::Transform::operator*(Vector3):
 * row.dot(p) + origin, x and y only (LaserMapUpdater reads the cloud's x and y fields, laser_map_updater.cpp:53-60) */
static void q_apply_xy(const double* q, double tx, double ty, double x, double y, double z, double* ox, double* oy) {
  const double d = q_dot(q, q);
  const double s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  *ox = ((1.0 - (yy + zz)) * x + (xy - wz) * y + (xz + wy) * z) + tx;
  *oy = ((xy + wz) * x + (1.0 - (xx + zz)) * y + (yz - wx) * z) + ty;
}

/* both pose forms share everything but the per-beam transform: exactly one of `pl` (planar) and `tf` (full) is given */
static int scan_to_rays_core(const og_scan* pl, const og_scan_tf* tf, const float* ranges, og_ray* out, int cap) {
  const float angle_min = pl ? pl->angle_min : tf->angle_min, angle_increment = pl ? pl->angle_increment : tf->angle_increment;
  const float range_min = pl ? pl->range_min : tf->range_min, range_max = pl ? pl->range_max : tf->range_max;
  const int n_ranges = pl ? pl->n_ranges : tf->n_ranges;
  const float* r = ranges + (pl ? pl->ranges_offset : tf->ranges_offset);
  int n = n_ranges;
  float inc = angle_increment;
  static int sel_buf[1 << 16];
  int* sel = NULL;
  if (angle_increment < 0.017) {   /* laser_map_updater.cpp:82 */
    n = og_simplify_scan(n_ranges, angle_increment, sel_buf, 1 << 16, &inc);
    if (n > (1 << 16)) n = 1 << 16;
    sel = sel_buf;
  }
  const double range_cutoff = range_max;
  double dyaw = 0.0;
  if (pl) {
    dyaw = fmod(pl->yaw_end - pl->yaw, 2.0 * M_PI);   /* shortest arc, as a quaternion slerp turns */
    if (dyaw > M_PI) dyaw -= 2.0 * M_PI;
    if (dyaw < -M_PI) dyaw += 2.0 * M_PI;
  }
  const double ranges_norm = n > 1 ? 1.0 / ((double)n - 1.0) : 0.0;
  const double sx = pl ? pl->x : tf->t[0], sy = pl ? pl->y : tf->t[1];   /* tf::transformPoint of (0, 0, 0), :101-116 */
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float range = r[sel ? sel[i] : i];
    if (!(range < range_cutoff && range >= range_min)) continue;
    const double a = angle_min + (double)i * inc;
    const float px = (float)(range * cos(a)), py = (float)(range * sin(a));
    const double ratio = (double)i * ranges_norm, keep = 1.0 - ratio;
    double mx, my;
    if (pl) {
      const double yaw_i = pl->yaw + ratio * dyaw;
      const double cy = cos(yaw_i), sy_ = sin(yaw_i);
      const double tx = keep * pl->x + ratio * pl->x_end, ty = keep * pl->y + ratio * pl->y_end;
      mx = cy * (double)px - sy_ * (double)py + tx;
      my = sy_ * (double)px + cy * (double)py + ty;
    } else {
      double q[4];
      q_slerp(tf->q, tf->q_end, ratio, q);
      /* tf::Vector3::setInterpolate3(v0, v1, rt): s = 1 - rt; s * v0 + rt * v1 */
      const double tx = keep * tf->t[0] + ratio * tf->t_end[0], ty = keep * tf->t[1] + ratio * tf->t_end[1];
      q_apply_xy(q, tx, ty, (double)px, (double)py, 0.0, &mx, &my);
    }
    const float gx = (float)mx, gy = (float)my;
    const float orig = i < n_ranges ? r[i] : r[n_ranges - 1];   /* msg->ranges[index] of the ORIGINAL scan */
    if (m < cap) {
      out[m].sx = sx; out[m].sy = sy;
      out[m].ex = gx; out[m].ey = gy;
      out[m].clear_end = (isinf(orig) || orig == range_max) ? 1 : 0;
      out[m]._pad = 0;
    }
    ++m;
  }
  return m;
}

int og_scan_to_rays(const og_scan* s, const float* ranges, og_ray* out, int cap) { return scan_to_rays_core(s, NULL, ranges, out, cap); }
/* the sensor's full pose as tf::TransformListener::lookupTransform reports it (translation, quaternion x y z w) at the scan's
 * start and end time: a tilted, rolled or raised mount.  tf's LinearMath (ROS Indigo tf 1.11.x, not under /root/reference)
 * restated above -- PARITY UNPINNED like the planar path. */
int og_scan_to_rays_tf(const og_scan_tf* s, const float* ranges, og_ray* out, int cap) { return scan_to_rays_core(NULL, s, ranges, out, cap); }

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

/* the same for a full sensor pose: start = T * (0, 0, 0), end = T * (range, 0, 0), x and y kept */
void og_range_to_ray_tf(float range, float max_range, const double t[3], const double q[4], og_ray* out) {
  q_apply_xy(q, t[0], t[1], 0.0, 0.0, 0.0, &out->sx, &out->sy);
  q_apply_xy(q, t[0], t[1], (double)range, 0.0, 0.0, &out->ex, &out->ey);
  out->clear_end = (range < max_range) ? 0 : 1;
  out->_pad = 0;
}
