// hostiter.hip -- host-side grid_map_core walks over a geometry alone (no engine, no GPU), for the C++ mirror's
// LineIterator / CircleIterator / SubmapIterator / GridMapIterator classes and for callers that only need index math.
// Same gridmath.hpp as the kernels, so a host walk visits exactly the cells the device walks visit:
//   rna_line_cells    = what himm_prep / himm_raster rasterise for a ray  (gmc/src/iterators/LineIterator.cpp:16-150)
//   rna_circle_cells  = what rrt_kernel's blocked-disc test scans          (gmc/src/iterators/CircleIterator.cpp:16-93)
//   rna_submap_cells  = the order of SubmapIterator                        (gmc/src/iterators/SubmapIterator.cpp:28-83)
// Cells are written as (i, j) buffer-index pairs in the reference's visiting order; the return value is the length
// of the walk (only the first `cap` cells are written).
#include "engine.hpp"

using namespace rna;

namespace {

Geom to_geom(const rna_geometry* g) {
  Geom o;
  o.len[0] = g->length[0]; o.len[1] = g->length[1];
  o.pos[0] = g->position[0]; o.pos[1] = g->position[1];
  o.res = g->resolution;
  o.size[0] = g->size[0]; o.size[1] = g->size[1];
  o.start[0] = g->start_index[0]; o.start[1] = g->start_index[1];
  return o;
}

bool geom_ok(const rna_geometry* g) {
  return g && g->size[0] > 0 && g->size[1] > 0 && g->resolution > 0.0 && g->start_index[0] >= 0 && g->start_index[1] >= 0 &&
         g->start_index[0] < g->size[0] && g->start_index[1] < g->size[1];
}

// LineIterator::getIndexLimitedToMapRange (LineIterator.cpp:92-104), as himm.hip's index_limited_to_map
bool limited_to_map(const Geom& g, double sx, double sy, double ex, double ey, int idx[2]) {
  double px = sx, py = sy;
  const double vx = ex - sx, vy = ey - sy;
  const double nrm = sqrt(vx * vx + vy * vy);
  const double dx = vx / nrm, dy = vy / nrm;
  const double step = g.res - DBL_EPSILON;
  while (!index_from_position(g, px, py, idx)) {
    if (!(nrm > 0.0)) return false;
    px += step * dx;
    py += step * dy;
    const double rx = ex - px, ry = ey - py;
    if (sqrt(rx * rx + ry * ry) < step) return false;
  }
  return true;
}

// SubmapIterator's increment (GridMapMath.cpp:436-465): second index fastest
bool next_in_submap(const Geom& g, const int tl[2], const int size[2], int sub[2], int idx[2]) {
  int t[2] = {sub[0], sub[1]};
  if (t[1] + 1 < size[1]) t[1]++;
  else { t[0]++; t[1] = 0; }
  if (t[0] < 0 || t[1] < 0 || t[0] >= size[0] || t[1] >= size[1]) return false;
  int tl_u[2];
  unwrap_index(g, tl, tl_u);
  const int s[2] = {tl_u[0] + t[0], tl_u[1] + t[1]};
  buffer_index(g, s, idx);
  sub[0] = t[0]; sub[1] = t[1];
  return true;
}

}  // namespace

extern "C" int rna_geometry_index(const rna_geometry* g, double x, double y, int32_t index[2]) {
  if (!geom_ok(g) || !index) return RNA_EINVAL;
  int idx[2];
  if (!index_from_position(to_geom(g), x, y, idx)) return 0;
  index[0] = idx[0]; index[1] = idx[1];
  return 1;
}

extern "C" int rna_geometry_position(const rna_geometry* g, int32_t i, int32_t j, double position[2]) {
  if (!geom_ok(g) || !position) return RNA_EINVAL;
  if (i < 0 || j < 0 || i >= g->size[0] || j >= g->size[1]) return 0;   // GridMap::getPosition: checkIfIndexWithinRange
  const int idx[2] = {i, j};
  position_from_index(to_geom(g), idx, position);
  return 1;
}

extern "C" int rna_line_cells(const rna_geometry* gg, double sx, double sy, double ex, double ey, int32_t* cells, int cap) {
  if (!geom_ok(gg) || cap < 0 || (cap > 0 && !cells)) return RNA_EINVAL;
  const Geom g = to_geom(gg);
  // malformed rays are dropped as the HIMM kernels drop them (himm.hip ray_well_formed)
  if (!(std::isfinite(sx) && std::isfinite(sy) && std::isfinite(ex) && std::isfinite(ey))) return 0;
  {
    const double vx = ex - sx, vy = ey - sy;
    if (!(sqrt(vx * vx + vy * vy) <= 1048576.0 * g.res)) return 0;
  }
  int s[2], e[2];
  if (!limited_to_map(g, sx, sy, ex, ey, s) || !limited_to_map(g, ex, ey, sx, sy, e)) return 0;
  // integer Bresenham over buffer indices (LineIterator.cpp:60-70,133-149)
  const int dx = abs(e[0] - s[0]), dy = abs(e[1] - s[1]);
  int inc1[2], inc2[2];
  inc1[0] = inc2[0] = e[0] >= s[0] ? 1 : -1;
  inc1[1] = inc2[1] = e[1] >= s[1] ? 1 : -1;
  int den, num, add, n;
  if (dx >= dy) { inc1[0] = 0; inc2[1] = 0; den = dx; num = dx / 2; add = dy; n = dx + 1; }
  else { inc2[0] = 0; inc1[1] = 0; den = dy; num = dy / 2; add = dx; n = dy + 1; }
  int idx[2] = {s[0], s[1]};
  for (int k = 0; k < n; ++k) {
    if (k < cap) { cells[2 * k] = idx[0]; cells[2 * k + 1] = idx[1]; }
    num += add;
    if (num >= den) { num -= den; idx[0] += inc1[0]; idx[1] += inc1[1]; }
    idx[0] += inc2[0];
    idx[1] += inc2[1];
  }
  return n;
}

extern "C" int rna_submap_cells(const rna_geometry* gg, const int32_t top_left[2], const int32_t size[2], int32_t* cells, int cap) {
  if (!geom_ok(gg) || !top_left || !size || cap < 0 || (cap > 0 && !cells)) return RNA_EINVAL;
  const Geom g = to_geom(gg);
  const int tl[2] = {top_left[0], top_left[1]}, sz[2] = {size[0], size[1]};
  int idx[2] = {tl[0], tl[1]}, sub[2] = {0, 0};
  int n = 0;
  for (;;) {   // the iterator starts not-past-end: the first cell is visited whatever the size
    if (n < cap) { cells[2 * n] = idx[0]; cells[2 * n + 1] = idx[1]; }
    ++n;
    if (!next_in_submap(g, tl, sz, sub, idx)) break;
  }
  return n;
}

extern "C" int rna_circle_cells(const rna_geometry* gg, double cx, double cy, double radius, int32_t* cells, int cap) {
  if (!geom_ok(gg) || cap < 0 || (cap > 0 && !cells)) return RNA_EINVAL;
  const Geom g = to_geom(gg);
  const double r2 = radius * radius;
  double tl[2] = {cx + radius, cy + radius}, br[2] = {cx - radius, cy - radius};
  limit_position_to_range(g, tl);
  limit_position_to_range(g, br);
  // a corner lookup that fails leaves index (0, 0) (CircleIterator.cpp:89-91 reads an uninitialised Index there;
  // defined as in rrt_kernel and DESIGN.md 4)
  int s[2] = {0, 0}, e[2] = {0, 0}, su[2], eu[2];
  (void)index_from_position(g, tl[0], tl[1], s);
  (void)index_from_position(g, br[0], br[1], e);
  unwrap_index(g, s, su);
  unwrap_index(g, e, eu);
  const int size[2] = {eu[0] - su[0] + 1, eu[1] - su[1] + 1};
  int idx[2] = {s[0], s[1]}, sub[2] = {0, 0};
  int n = 0;
  for (;;) {
    double p[2];
    position_from_index(g, idx, p);
    const double ddx = p[0] - cx, ddy = p[1] - cy;
    if (ddx * ddx + ddy * ddy <= r2) {
      if (n < cap) { cells[2 * n] = idx[0]; cells[2 * n + 1] = idx[1]; }
      ++n;
    }
    if (!next_in_submap(g, s, size, sub, idx)) break;
  }
  return n;
}
