// gridmath.hpp -- grid_map index/position math shared by host and device code of librna.
// Independent implementation of the behaviour of gmc/src/GridMapMath.cpp:30-296 and
// gmc/src/GridMap.cpp:51-70 (gmc/ = grid_map-master/grid_map_core in the reference).  All of it is
// IEEE double arithmetic evaluated operation by operation: the library is built with
// -ffp-contract=off so neither hipcc's host nor its gfx950 code fuses a*b+c.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

namespace rna {

struct Geom {
  double len[2];
  double pos[2];
  double res;
  int size[2];
  int start[2];
};

#define RNA_HD __host__ __device__ __forceinline__

// hypot() as glibc 2.35 computes it (sysdeps/ieee754/dbl-64/e_hypot.c, the non-FMA kernel: Borges' corrected
// square root).  The reference's findNearNode / findClosedVertex compare hypot() values with a strict `<`, and on
// coarse grids lattice-symmetric nodes tie exactly in glibc's rounding: the device libm's hypot (also < 1 ulp, but a
// different last bit) would break such ties the other way.  Only IEEE +,-,*,/ and sqrt, evaluated without
// contraction (-ffp-contract=off), so the value is the host's bit for bit; the huge / tiny scaling branches of the
// original are not needed for map coordinates (|x| < 2^511, exact zeros handled by the first return).
RNA_HD double glibc_hypot(double x, double y) {
  x = fabs(x); y = fabs(y);
  const double ax = x < y ? y : x, ay = x < y ? x : y;
  if (ax >= ay / 0x1p-54) return ax + ay;
  double h = sqrt(ax * ax + ay * ay);
  double t1, t2;
  if (h <= 2.0 * ay) {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}

// gmc/src/GridMapMath.cpp:216-220
RNA_HD int wrap_index(int idx, int size) {
  if (idx < 0) idx += ((-idx / size) + 1) * size;
  return idx % size;
}

// gmc/src/GridMapMath.cpp:467-476 : buffer index -> index relative to the circular-buffer origin
RNA_HD void unwrap_index(const Geom& g, const int b[2], int out[2]) {
  if (g.start[0] == 0 && g.start[1] == 0) { out[0] = b[0]; out[1] = b[1]; return; }
  out[0] = wrap_index(b[0] - g.start[0], g.size[0]);
  out[1] = wrap_index(b[1] - g.start[1], g.size[1]);
}

// gmc/src/GridMapMath.cpp:70-81
RNA_HD void buffer_index(const Geom& g, const int u[2], int out[2]) {
  if (g.start[0] == 0 && g.start[1] == 0) { out[0] = u[0]; out[1] = u[1]; return; }
  out[0] = wrap_index(u[0] + g.start[0], g.size[0]);
  out[1] = wrap_index(u[1] + g.start[1], g.size[1]);
}

// gmc/src/GridMapMath.cpp:146-159
RNA_HD bool position_within_map(const Geom& g, double x, double y) {
  const double tx = -((x - g.pos[0]) - 0.5 * g.len[0]);
  const double ty = -((y - g.pos[1]) - 0.5 * g.len[1]);
  return tx >= 0.0 && ty >= 0.0 && tx < g.len[0] && ty < g.len[1];
}

// gmc/src/GridMapMath.cpp:130-144
RNA_HD bool index_from_position(const Geom& g, double x, double y, int idx[2]) {
  if (!position_within_map(g, x, y)) return false;
  int u[2];
  u[0] = -(int)(((x - 0.5 * g.len[0]) - g.pos[0]) / g.res);
  u[1] = -(int)(((y - 0.5 * g.len[1]) - g.pos[1]) / g.res);
  // A position within rounding of the far edge passes the strict `<` above and still divides to index == size.
  // getBufferIndexFromIndex wraps it to 0 on a moved buffer (kept: defined reference behaviour) and leaves it out of
  // range on an unmoved one, where the reference then indexes its matrices out of bounds.  Defined here and in
  // oracle/gridmath.c: such a position is outside the map.
  if (g.start[0] == 0 && g.start[1] == 0 &&
      ((unsigned)u[0] >= (unsigned)g.size[0] || (unsigned)u[1] >= (unsigned)g.size[1])) return false;
  buffer_index(g, u, idx);
  return true;
}

// gmc/src/GridMapMath.cpp:115-128 (no range check: callers pass valid indices)
RNA_HD void position_from_index(const Geom& g, const int idx[2], double p[2]) {
  int u[2];
  unwrap_index(g, idx, u);
  p[0] = (g.pos[0] + (0.5 * g.len[0] - 0.5 * g.res)) + g.res * (double)(-u[0]);
  p[1] = (g.pos[1] + (0.5 * g.len[1] - 0.5 * g.res)) + g.res * (double)(-u[1]);
}

// gmc/src/GridMapMath.cpp:216-239
RNA_HD void limit_position_to_range(const Geom& g, double p[2]) {
  for (int a = 0; a < 2; ++a) {
    const double vto = 0.5 * g.len[a];
    double shifted = (p[a] - g.pos[a]) + vto;
    double eps = 10.0 * DBL_EPSILON;
    if (fabs(p[a]) > 1.0) eps *= fabs(p[a]);
    if (shifted <= 0) shifted = eps;
    else if (shifted >= g.len[a]) shifted = g.len[a] - eps;
    p[a] = (shifted + g.pos[a]) - vto;
  }
}

struct SubmapInfo {
  int top_left[2];   // buffer index
  int size[2];
  double pos[2];
  double len[2];
};

// gmc/src/GridMapMath.cpp:246-296 followed by GridMap::setGeometry(SubmapGeometry)
// (gmc/src/GridMap.cpp:51-75), which re-derives the size from the length.
RNA_HD bool submap_information(const Geom& g, const double req_pos[2], const double req_len[2], SubmapInfo& o) {
  double tl[2], br[2];
  for (int a = 0; a < 2; ++a) {
    tl[a] = req_pos[a] - (-0.5 * req_len[a]);
    br[a] = req_pos[a] + (-0.5 * req_len[a]);
  }
  limit_position_to_range(g, tl);
  if (!index_from_position(g, tl[0], tl[1], o.top_left)) return false;
  int tl_u[2], br_b[2], br_u[2];
  unwrap_index(g, o.top_left, tl_u);
  limit_position_to_range(g, br);
  if (!index_from_position(g, br[0], br[1], br_b)) return false;
  unwrap_index(g, br_b, br_u);
  double corner[2];
  position_from_index(g, o.top_left, corner);
  for (int a = 0; a < 2; ++a) {
    corner[a] = corner[a] - (-(0.5 * g.res));
    const int sz = br_u[a] - tl_u[a] + 1;
    const double len = (double)sz * g.res;
    o.pos[a] = corner[a] - 0.5 * len;
    // requested-index lookup of the reference can only fail for degenerate geometry; then
    // setGeometry: size = round(len/res), len = size*res
    o.size[a] = (int)round(len / g.res);
    o.len[a] = (double)o.size[a] * g.res;
  }
  // getSubmapInformation also requires the requested position to lie inside the submap
  const double tx = -((req_pos[0] - o.pos[0]) - 0.5 * ((double)(br_u[0] - tl_u[0] + 1) * g.res));
  const double ty = -((req_pos[1] - o.pos[1]) - 0.5 * ((double)(br_u[1] - tl_u[1] + 1) * g.res));
  if (!(tx >= 0.0 && ty >= 0.0 && tx < (double)(br_u[0] - tl_u[0] + 1) * g.res &&
        ty < (double)(br_u[1] - tl_u[1] + 1) * g.res))
    return false;
  return true;
}

// GridMap::setGeometry (gmc/src/GridMap.cpp:51-70)
inline void set_geometry(Geom& g, double len_x, double len_y, double res, double px, double py) {
  g.size[0] = (int)round(len_x / res);
  g.size[1] = (int)round(len_y / res);
  g.res = res;
  g.len[0] = (double)g.size[0] * res;
  g.len[1] = (double)g.size[1] * res;
  g.pos[0] = px;
  g.pos[1] = py;
  g.start[0] = g.start[1] = 0;
}

}  // namespace rna
