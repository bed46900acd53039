/*
 * himm.c -- ORACLE (test infrastructure): the simplified HIMM cell update of move_control and
 * Steerer::getRangesFromSubmap, restated in plain C.  "parity unpinned": the reference has no
 * test for these and they cannot be built here (ROS/tf/Eigen); reviewed line by line against
 * mc/include/move_control/map_updater.h:38-71, mc/src/laser_map_updater.cpp:7-21,
 * mc/src/map_provider.cpp:216-223 and mc/src/steerer.cpp:147-191.
 */
#include "rna_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* map_updater.h:61-71 */
float og_himm_clear(float v) {
  if (v <= 0 || isnan(v)) v = 0.0f;
  else v = v - 10.0f;
  if (v < 0.0f) v = 0.0f;
  return v;
}

/* map_updater.h:52-59 */
float og_himm_mark(float v) {
  if (v <= 0 || isnan(v)) return 30.0f;
  if (v <= 150.0f) return v + 30.0f;
  return v;
}

/* A ray with a non-finite coordinate, or longer than 2^20 cells, makes the reference's clipping march
 * (LineIterator.cpp:92-104) spin forever / for minutes.  Defined (here and in csrc/himm.hip): dropped whole. */
int og_ray_well_formed(const og_geom* g, const og_ray* r) {
  if (!(isfinite(r->sx) && isfinite(r->sy) && isfinite(r->ex) && isfinite(r->ey))) return 0;
  double vx = r->ex - r->sx, vy = r->ey - r->sy;
  return sqrt(vx * vx + vy * vy) <= 1048576.0 * g->res;
}

/* map_updater.h:38-50 (lineOnMap) for each buffered sample in order (laser_map_updater.cpp:14-20) */
void og_himm_update(const og_geom* g, float* layer, const og_ray* rays, int n, double bbox[4]) {
  int cap = 2 * (g->size[0] + g->size[1]) + 8;
  int* cells = (int*)malloc(sizeof(int) * 2 * (size_t)cap);
  size_t rows = (size_t)g->size[0];
  for (int r = 0; r < n; ++r) {
    double s[2] = { rays[r].sx, rays[r].sy }, e[2] = { rays[r].ex, rays[r].ey };
    if (!og_ray_well_formed(g, &rays[r])) continue;
    int nc = og_line_cells(g, s, e, cells, cap);
    for (int k = 0; k < nc; ++k) {
      float* c = &layer[(size_t)cells[2 * k + 1] * rows + (size_t)cells[2 * k]];
      *c = og_himm_clear(*c);
    }
    if (!rays[r].clear_end) {
      int ei[2];
      if (og_index_from_position(g, e, ei)) {
        float* c = &layer[(size_t)ei[1] * rows + (size_t)ei[0]];
        *c = og_himm_mark(*c);
      }
    }
    if (bbox) { /* touch(), map_updater.h:73-78 */
      if (s[0] < bbox[0]) bbox[0] = s[0];
      if (s[1] < bbox[1]) bbox[1] = s[1];
      if (s[0] > bbox[2]) bbox[2] = s[0];
      if (s[1] > bbox[3]) bbox[3] = s[1];
      if (e[0] < bbox[0]) bbox[0] = e[0];
      if (e[1] < bbox[1]) bbox[1] = e[1];
      if (e[0] > bbox[2]) bbox[2] = e[0];
      if (e[1] > bbox[3]) bbox[3] = e[1];
    }
  }
  free(cells);
}

/* map_provider.cpp:221 : map_["master"] = map_["laser"] */
void og_compose_master(float* master, const float* laser, size_t n_cells) {
  memcpy(master, laser, n_cells * sizeof(float));
}

/* angles::normalize_angle_positive (external `angles` package): fmod(fmod(a, 2pi) + 2pi, 2pi) */
static double normalize_angle_positive(double a) {
  return fmod(fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI);
}

/* steerer.cpp:147-191.  The submap is taken with GridMap::getSubmap(robotPos, Length(1.5,1.5))
 * (map_provider.cpp:93-100) and walked with GridMapIterator (column-major linear order). */
int og_ranges_from_submap(const og_geom* g, const float* master, const double robot_pos[2], double yaw,
                          double ranges[361][2]) {
  for (unsigned i = 0; i < 361; i++) ranges[i][0] = 5000.0;

  const double len[2] = { 1.5, 1.5 };
  og_geom sg;
  /* (the copy of the window: sized by the resolution.  A fixed 64 x 64 buffer stood here until round 6 and made every window of
   * a map finer than 2.4 cm a "getSubmap failure" of the oracle's own, which the reference does not have) */
  const int side = (int)(1.5 / g->res) + 3;
  const int cap = side * side;
  float* sub = (float*)malloc((size_t)cap * sizeof(float));
  if (!sub) return 0;
  if (!og_get_submap(g, master, robot_pos, len, &sg, sub, cap)) { free(sub); return 0; }

  int n = sg.size[0] * sg.size[1];
  for (int lin = 0; lin < n; ++lin) {
    int idx[2];
    og_index_from_linear((size_t)lin, sg.size, 0, idx);
    float value = sub[(size_t)idx[1] * sg.size[0] + idx[0]];
    if (isnan(value)) continue;
    if (value <= 3) continue;
    double pos[2];
    og_position_from_index(&sg, idx, pos);
    double angle = atan2(pos[1] - robot_pos[1], pos[0] - robot_pos[0]);
    double deg = normalize_angle_positive(angle - yaw + 3.14 / 2) * 180.0 / M_PI;
    if (deg > 180) continue;
    int fl = (int)floor(deg);
    int ce = (int)ceil(deg);
    double dx = robot_pos[0] - pos[0], dy = robot_pos[1] - pos[1];
    double distance = sqrt(dx * dx + dy * dy) * 1000.0;
    if (ranges[fl * 2][0] > distance) ranges[fl * 2][0] = distance;
    if (ranges[ce * 2][0] > distance) ranges[ce * 2][0] = distance;
  }
  free(sub);
  return 1;
}
