/*
 * gridmath.c -- ORACLE (test infrastructure): grid_map_core index/position math, submaps,
 * circular buffer and iterators, restated in plain C.  Compile with -ffp-contract=off so that the
 * double arithmetic is evaluated operation by operation as the reference (x86-64, no FMA) does.
 * gmc/ = /root/reference/grid_map-master/grid_map_core
 */
#include "rna_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* gmc/src/GridMap.cpp:51-70 (setGeometry): size = round(length/res), length = size*res */
void og_set_geometry(og_geom* g, double len_x, double len_y, double res, double pos_x, double pos_y) {
  g->size[0] = (int)round(len_x / res);
  g->size[1] = (int)round(len_y / res);
  g->res = res;
  g->len[0] = (double)g->size[0] * res;
  g->len[1] = (double)g->size[1] * res;
  g->pos[0] = pos_x;
  g->pos[1] = pos_y;
  g->start[0] = g->start[1] = 0;
}

/* gmc/src/GridMapMath.cpp:216-220 (mapIndexWithinRange, scalar) */
int og_wrap_index(int idx, int size) {
  if (idx < 0) idx += ((-idx / size) + 1) * size;
  return idx % size;
}

/* gmc/src/GridMapMath.cpp:193-200 */
int og_index_within_range(const int idx[2], const int size[2]) {
  return idx[0] >= 0 && idx[1] >= 0 && idx[0] < size[0] && idx[1] < size[1];
}

/* gmc/src/GridMapMath.cpp:467-476 (getIndexFromBufferIndex): buffer index -> unwrapped index */
void og_unwrap_index(const int bidx[2], const int size[2], const int start[2], int out[2]) {
  if (start[0] == 0 && start[1] == 0) { out[0] = bidx[0]; out[1] = bidx[1]; return; }
  out[0] = og_wrap_index(bidx[0] - start[0], size[0]);
  out[1] = og_wrap_index(bidx[1] - start[1], size[1]);
}

/* gmc/src/GridMapMath.cpp:70-81 (getBufferIndexFromIndex): unwrapped index -> buffer index */
void og_buffer_index(const int idx[2], const int size[2], const int start[2], int out[2]) {
  if (start[0] == 0 && start[1] == 0) { out[0] = idx[0]; out[1] = idx[1]; return; }
  out[0] = og_wrap_index(idx[0] + start[0], size[0]);
  out[1] = og_wrap_index(idx[1] + start[1], size[1]);
}

/* gmc/src/GridMapMath.cpp:146-159 : 0 <= -(p - c - L/2) < L on both axes */
int og_position_within_map(const double pos[2], const double len[2], const double mpos[2]) {
  double tx = -((pos[0] - mpos[0]) - 0.5 * len[0]);
  double ty = -((pos[1] - mpos[1]) - 0.5 * len[1]);
  return tx >= 0.0 && ty >= 0.0 && tx < len[0] && ty < len[1];
}

/* gmc/src/GridMapMath.cpp:115-128 : c + (L/2 - res/2) + res * (-unwrapped) */
int og_position_from_index(const og_geom* g, const int idx[2], double pos[2]) {
  if (!og_index_within_range(idx, g->size)) return 0;
  int u[2];
  og_unwrap_index(idx, g->size, g->start, u);
  for (int a = 0; a < 2; ++a) {
    double off = 0.5 * g->len[a] - 0.5 * g->res;
    double iv = (double)(-u[a]);
    pos[a] = (g->pos[a] + off) + g->res * iv;
  }
  return 1;
}

/* gmc/src/GridMapMath.cpp:130-144 + :93-100 : idx = -(int)((p - L/2 - c)/res) (+ start, wrapped) */
int og_index_from_position(const og_geom* g, const double pos[2], int idx[2]) {
  if (!og_position_within_map(pos, g->len, g->pos)) return 0;
  int u[2];
  for (int a = 0; a < 2; ++a) {
    double iv = ((pos[a] - 0.5 * g->len[a]) - g->pos[a]) / g->res;
    u[a] = -(int)iv; /* cast<int>() truncates toward zero, then the -I transform */
  }
  /* A position within rounding of the far edge passes the strict `<` of checkIfPositionWithinMap and still divides
   * to index == size.  getBufferIndexFromIndex wraps that to 0 on a moved buffer (kept) and returns it unchanged on
   * an unmoved one, where the reference goes on to index its matrices out of bounds (undefined).  Defined here and
   * in the HIP gridmath.hpp: such a position is outside the map. */
  if (g->start[0] == 0 && g->start[1] == 0 &&
      (u[0] < 0 || u[0] >= g->size[0] || u[1] < 0 || u[1] >= g->size[1])) return 0;
  og_buffer_index(u, g->size, g->start, idx);
  return 1;
}

/* gmc/src/GridMapMath.cpp:170-183 */
void og_index_shift_from_position_shift(const double shift[2], double res, int out[2]) {
  for (int a = 0; a < 2; ++a) {
    double t = shift[a] / res;
    int iv = (int)(t + 0.5 * (t > 0 ? 1 : -1));
    out[a] = -iv;
  }
}

/* gmc/src/GridMapMath.cpp:185-191 */
void og_position_shift_from_index_shift(const int shift[2], double res, double out[2]) {
  out[0] = (double)(-shift[0]) * res;
  out[1] = (double)(-shift[1]) * res;
}

/* gmc/src/GridMapMath.cpp:216-239 */
void og_limit_position_to_range(double pos[2], const double len[2], const double mpos[2]) {
  for (int a = 0; a < 2; ++a) {
    double vto = 0.5 * len[a];
    double shifted = (pos[a] - mpos[a]) + vto;
    double eps = 10.0 * DBL_EPSILON;
    if (fabs(pos[a]) > 1.0) eps *= fabs(pos[a]);
    if (shifted <= 0) shifted = eps;
    else if (shifted >= len[a]) shifted = len[a] - eps;
    pos[a] = (shifted + mpos[a]) - vto;
  }
}

/* gmc/src/GridMapMath.cpp:246-296 */
int og_submap_information(const og_geom* g, const double req_pos[2], const double req_len[2],
                          og_submap_info* o) {
  double tl_pos[2], br_pos[2];
  int tl_u[2], br_b[2], br_u[2];
  for (int a = 0; a < 2; ++a) {
    /* req - T*0.5*len with T = -I  ==  req + 0.5*len ; and req + T*0.5*len == req - 0.5*len */
    tl_pos[a] = req_pos[a] - (-0.5 * req_len[a]);
    br_pos[a] = req_pos[a] + (-0.5 * req_len[a]);
  }
  og_limit_position_to_range(tl_pos, g->len, g->pos);
  if (!og_index_from_position(g, tl_pos, o->top_left)) return 0;
  og_unwrap_index(o->top_left, g->size, g->start, tl_u);

  og_limit_position_to_range(br_pos, g->len, g->pos);
  if (!og_index_from_position(g, br_pos, br_b)) return 0;
  og_unwrap_index(br_b, g->size, g->start, br_u);

  double corner[2];
  if (!og_position_from_index(g, o->top_left, corner)) return 0;
  for (int a = 0; a < 2; ++a) {
    corner[a] = corner[a] - (-(0.5 * g->res));
    o->size[a] = br_u[a] - tl_u[a] + 1;
    o->len[a] = (double)o->size[a] * g->res;
    o->pos[a] = corner[a] - 0.5 * o->len[a];
  }
  og_geom sub;
  sub.len[0] = o->len[0]; sub.len[1] = o->len[1];
  sub.pos[0] = o->pos[0]; sub.pos[1] = o->pos[1];
  sub.res = g->res;
  sub.size[0] = o->size[0]; sub.size[1] = o->size[1];
  sub.start[0] = sub.start[1] = 0;
  if (!og_index_from_position(&sub, req_pos, o->requested_index)) return 0;
  return 1;
}

/* gmc/src/GridMapMath.cpp:102-109 (getQuadrant) */
static int quadrant_of(const int idx[2], const int start[2]) {
  if (idx[0] >= start[0] && idx[1] >= start[1]) return 1; /* TopLeft */
  if (idx[0] >= start[0] && idx[1] < start[1]) return 2;  /* TopRight */
  if (idx[0] < start[0] && idx[1] >= start[1]) return 3;  /* BottomLeft */
  if (idx[0] < start[0] && idx[1] < start[1]) return 4;   /* BottomRight */
  return 0;
}

static void put_region(og_region* r, int i0, int i1, int s0, int s1, int q) {
  r->index[0] = i0; r->index[1] = i1; r->size[0] = s0; r->size[1] = s1; r->quadrant = q;
}

/* gmc/src/GridMapMath.cpp:306-412 */
int og_buffer_regions_for_submap(const int si[2], const int ss[2], const int bs[2],
                                 const int start[2], og_region out[4]) {
  int u[2];
  og_unwrap_index(si, bs, start, u);
  if (u[0] + ss[0] > bs[0] || u[1] + ss[1] > bs[1]) return -1;

  int br[2] = { og_wrap_index(si[0] + ss[0] - 1, bs[0]), og_wrap_index(si[1] + ss[1] - 1, bs[1]) };
  int qtl = quadrant_of(si, start);
  int qbr = quadrant_of(br, start);

  if (qtl == 1) {
    if (qbr == 1) { put_region(&out[0], si[0], si[1], ss[0], ss[1], 1); return 1; }
    if (qbr == 2) {
      int tls1 = bs[1] - si[1];
      put_region(&out[0], si[0], si[1], ss[0], tls1, 1);
      put_region(&out[1], si[0], 0, ss[0], ss[1] - tls1, 2);
      return 2;
    }
    if (qbr == 3) {
      int tls0 = bs[0] - si[0];
      put_region(&out[0], si[0], si[1], tls0, ss[1], 1);
      put_region(&out[1], 0, si[1], ss[0] - tls0, ss[1], 3);
      return 2;
    }
    if (qbr == 4) {
      int tls0 = bs[0] - si[0], tls1 = bs[1] - si[1];
      put_region(&out[0], si[0], si[1], tls0, tls1, 1);
      put_region(&out[1], si[0], 0, tls0, ss[1] - tls1, 2);
      put_region(&out[2], 0, si[1], ss[0] - tls0, tls1, 3);
      put_region(&out[3], 0, 0, ss[0] - tls0, ss[1] - tls1, 4);
      return 4;
    }
  } else if (qtl == 2) {
    if (qbr == 2) { put_region(&out[0], si[0], si[1], ss[0], ss[1], 2); return 1; }
    if (qbr == 4) {
      int trs0 = bs[0] - si[0];
      put_region(&out[0], si[0], si[1], trs0, ss[1], 2);
      put_region(&out[1], 0, si[1], ss[0] - trs0, ss[1], 4);
      return 2;
    }
  } else if (qtl == 3) {
    if (qbr == 3) { put_region(&out[0], si[0], si[1], ss[0], ss[1], 3); return 1; }
    if (qbr == 4) {
      int bls1 = bs[1] - si[1];
      put_region(&out[0], si[0], si[1], ss[0], bls1, 3);
      put_region(&out[1], si[0], 0, ss[0], ss[1] - bls1, 4);
      return 2;
    }
  } else if (qtl == 4) {
    if (qbr == 4) { put_region(&out[0], si[0], si[1], ss[0], ss[1], 4); return 1; }
  }
  return -1;
}

/* gmc/src/GridMapMath.cpp:414-434 */
int og_increment_index(int idx[2], const int size[2], const int start[2]) {
  int u[2];
  og_unwrap_index(idx, size, start, u);
  if (u[1] + 1 < size[1]) u[1]++;
  else { u[0]++; u[1] = 0; }
  if (!og_index_within_range(u, size)) return 0;
  og_buffer_index(u, size, start, idx);
  return 1;
}

/* gmc/src/GridMapMath.cpp:436-465 */
int og_increment_index_for_submap(int sub_idx[2], int idx[2], const int sub_tl[2],
                                  const int sub_size[2], const int size[2], const int start[2]) {
  int t[2] = { sub_idx[0], sub_idx[1] };
  if (t[1] + 1 < sub_size[1]) t[1]++;
  else { t[0]++; t[1] = 0; }
  if (!og_index_within_range(t, sub_size)) return 0;
  int tl_u[2], s[2];
  og_unwrap_index(sub_tl, size, start, tl_u);
  s[0] = tl_u[0] + t[0]; s[1] = tl_u[1] + t[1];
  og_buffer_index(s, size, start, idx);
  sub_idx[0] = t[0]; sub_idx[1] = t[1];
  return 1;
}

/* gmc/src/GridMapMath.cpp:478-488 */
size_t og_linear_from_index(const int idx[2], const int size[2], int row_major) {
  if (!row_major) return (size_t)idx[1] * (size_t)size[0] + (size_t)idx[0];
  return (size_t)idx[0] * (size_t)size[1] + (size_t)idx[1];
}
void og_index_from_linear(size_t lin, const int size[2], int row_major, int out[2]) {
  if (!row_major) { out[0] = (int)lin % size[0]; out[1] = (int)lin / size[0]; }
  else { out[0] = (int)lin / size[1]; out[1] = (int)lin % size[1]; }
}

/* gmc/src/GridMap.cpp:287-339 : the <=4 quadrant block copies, one layer */
int og_get_submap(const og_geom* g, const float* layer, const double pos[2], const double len[2],
                  og_geom* sg, float* out, int cap) {
  og_submap_info inf;
  if (!og_submap_information(g, pos, len, &inf)) return 0;
  /* submap.setGeometry(SubmapGeometry): re-derives size from length/res (GridMap.cpp:51-70,72-75) */
  og_set_geometry(sg, inf.len[0], inf.len[1], g->res, inf.pos[0], inf.pos[1]);
  int sr = sg->size[0], sc = sg->size[1];
  if (sr * sc > cap) return 0;
  og_region reg[4];
  int n = og_buffer_regions_for_submap(inf.top_left, sg->size, g->size, g->start, reg);
  if (n < 0) return 0;
  for (int i = 0; i < sr * sc; ++i) out[i] = NAN;
  int rows = g->size[0];
  for (int k = 0; k < n; ++k) {
    int r0, c0; /* destination corner in the submap */
    switch (reg[k].quadrant) {
      case 1: r0 = 0; c0 = 0; break;                                           /* topLeftCorner */
      case 2: r0 = 0; c0 = sc - reg[k].size[1]; break;                         /* topRightCorner */
      case 3: r0 = sr - reg[k].size[0]; c0 = 0; break;                         /* bottomLeftCorner */
      default: r0 = sr - reg[k].size[0]; c0 = sc - reg[k].size[1]; break;      /* bottomRightCorner */
    }
    for (int c = 0; c < reg[k].size[1]; ++c)
      for (int r = 0; r < reg[k].size[0]; ++r)
        out[(size_t)(c0 + c) * sr + (r0 + r)] =
            layer[(size_t)(reg[k].index[1] + c) * rows + (reg[k].index[0] + r)];
  }
  return 1;
}

static void clear_rows(const og_geom* g, float** layers, int nl, int index, int n) {
  for (int l = 0; l < nl; ++l)
    for (int c = 0; c < g->size[1]; ++c)
      for (int r = index; r < index + n; ++r) layers[l][(size_t)c * g->size[0] + r] = NAN;
}
static void clear_cols(const og_geom* g, float** layers, int nl, int index, int n) {
  for (int l = 0; l < nl; ++l)
    for (int c = index; c < index + n; ++c)
      for (int r = 0; r < g->size[0]; ++r) layers[l][(size_t)c * g->size[0] + r] = NAN;
}

/* gmc/src/GridMap.cpp:346-412 (GridMap::move) */
int og_move(og_geom* g, float** layers, int nl, const double new_pos[2], og_region* regs, int* moved) {
  int nreg = 0;
  double pshift[2] = { new_pos[0] - g->pos[0], new_pos[1] - g->pos[1] };
  int ishift[2];
  double aligned[2];
  og_index_shift_from_position_shift(pshift, g->res, ishift);
  og_position_shift_from_index_shift(ishift, g->res, aligned);

  for (int i = 0; i < 2; ++i) {
    if (ishift[i] == 0) continue;
    int asz = g->size[i];
    int ash = ishift[i] < 0 ? -ishift[i] : ishift[i];
    if (ash >= asz) {
      clear_rows(g, layers, nl, 0, g->size[0]);
      if (nreg < 4) put_region(&regs[nreg], 0, 0, g->size[0], g->size[1], 0);
      nreg++;
    } else {
      int sign = ishift[i] > 0 ? 1 : -1;
      int start_index = g->start[i] - (sign < 0 ? 1 : 0);
      int end_index = start_index - sign + ishift[i];
      int n_cells = ash;
      int index = og_wrap_index(sign > 0 ? start_index : end_index, asz);
      if (index + n_cells <= asz) {
        if (i == 0) { clear_rows(g, layers, nl, index, n_cells);
          if (nreg < 4) put_region(&regs[nreg], index, 0, n_cells, g->size[1], 0); }
        else { clear_cols(g, layers, nl, index, n_cells);
          if (nreg < 4) put_region(&regs[nreg], 0, index, g->size[0], n_cells, 0); }
        nreg++;
      } else {
        int first_n = asz - index;
        int second_n = n_cells - first_n;
        if (i == 0) {
          clear_rows(g, layers, nl, index, first_n);
          if (nreg < 4) put_region(&regs[nreg], index, 0, first_n, g->size[1], 0);
          nreg++;
          clear_rows(g, layers, nl, 0, second_n);
          if (nreg < 4) put_region(&regs[nreg], 0, 0, second_n, g->size[1], 0);
          nreg++;
        } else {
          clear_cols(g, layers, nl, index, first_n);
          if (nreg < 4) put_region(&regs[nreg], 0, index, g->size[0], first_n, 0);
          nreg++;
          clear_cols(g, layers, nl, 0, second_n);
          if (nreg < 4) put_region(&regs[nreg], 0, 0, g->size[0], second_n, 0);
          nreg++;
        }
      }
    }
  }
  g->start[0] = og_wrap_index(g->start[0] + ishift[0], g->size[0]);
  g->start[1] = og_wrap_index(g->start[1] + ishift[1], g->size[1]);
  g->pos[0] += aligned[0];
  g->pos[1] += aligned[1];
  *moved = (ishift[0] != 0 || ishift[1] != 0);
  return nreg;
}

/* gmc/src/iterators/LineIterator.cpp:92-104 (getIndexLimitedToMapRange).
 * direction = (end-start).normalized() = v / sqrt(vx*vx + vy*vy).  A zero-length ray whose start is
 * outside the map makes the reference divide by zero and spin forever; the oracle defines that
 * case as "no cells" (documented departure). */
static int index_limited_to_map(const og_geom* g, const double start[2], const double end[2], int idx[2]) {
  double p[2] = { start[0], start[1] };
  double vx = end[0] - start[0], vy = end[1] - start[1];
  double nrm = sqrt(vx * vx + vy * vy);
  double dx = vx / nrm, dy = vy / nrm;
  double step = g->res - DBL_EPSILON;
  while (!og_index_from_position(g, p, idx)) {
    if (!(nrm > 0.0)) return 0;
    p[0] += step * dx;
    p[1] += step * dy;
    double rx = end[0] - p[0], ry = end[1] - p[1];
    if (sqrt(rx * rx + ry * ry) < step) return 0;
  }
  return 1;
}

/* gmc/src/iterators/LineIterator.cpp:60-70,106-150 : integer Bresenham */
int og_line_cells_index(const int s[2], const int e[2], int* cells, int cap) {
  int dx = e[0] - s[0]; if (dx < 0) dx = -dx;
  int dy = e[1] - s[1]; if (dy < 0) dy = -dy;
  int inc1[2], inc2[2];
  inc1[0] = inc2[0] = (e[0] >= s[0]) ? 1 : -1;
  inc1[1] = inc2[1] = (e[1] >= s[1]) ? 1 : -1;
  int den, num, add, n;
  if (dx >= dy) { inc1[0] = 0; inc2[1] = 0; den = dx; num = dx / 2; add = dy; n = dx + 1; }
  else          { inc2[0] = 0; inc1[1] = 0; den = dy; num = dy / 2; add = dx; n = dy + 1; }
  int idx[2] = { s[0], s[1] };
  for (int k = 0; k < n; ++k) {
    if (k < cap) { cells[2 * k] = idx[0]; cells[2 * k + 1] = idx[1]; }
    num += add;
    if (num >= den) { num -= den; idx[0] += inc1[0]; idx[1] += inc1[1]; }
    idx[0] += inc2[0]; idx[1] += inc2[1];
  }
  return n;
}

/* gmc/src/iterators/LineIterator.cpp:16-23 */
int og_line_cells(const og_geom* g, const double start[2], const double end[2], int* cells, int cap) {
  int s[2], e[2];
  if (!index_limited_to_map(g, start, end, s)) return 0;
  if (!index_limited_to_map(g, end, start, e)) return 0;
  return og_line_cells_index(s, e, cells, cap);
}

/* gmc/src/iterators/SubmapIterator.cpp:28-83 */
int og_submap_cells(const og_geom* g, const int tl[2], const int size[2], int* out, int cap) {
  int idx[2] = { tl[0], tl[1] }, sub[2] = { 0, 0 };
  int n = 0;
  for (;;) {
    if (n < cap) { out[4 * n] = idx[0]; out[4 * n + 1] = idx[1]; out[4 * n + 2] = sub[0]; out[4 * n + 3] = sub[1]; }
    n++;
    if (!og_increment_index_for_submap(sub, idx, tl, size, g->size, g->start)) break;
  }
  return n;
}

/* gmc/src/iterators/CircleIterator.cpp:16-93 */
int og_circle_cells(const og_geom* g, const double center[2], double radius, int* cells, int cap) {
  double r2 = pow(radius, 2);
  double tl[2] = { center[0] + radius, center[1] + radius };
  double br[2] = { center[0] - radius, center[1] - radius };
  og_limit_position_to_range(tl, g->len, g->pos);
  og_limit_position_to_range(br, g->len, g->pos);
  /* CircleIterator.cpp:89-91 leaves startIndex / endIndex untouched when getIndexFromPosition fails, and endIndex is
   * an uninitialised Index there (a bottom-right corner exactly on the map's far edge fails the strict `<` of
   * checkIfPositionWithinMap even after limitPositionToRange when length = size * resolution rounds up).  The
   * reference's result is undefined in that case; this restatement and the HIP kernel define the index as (0, 0).
   * Whatever the size, the iterator visits its first cell (SubmapIterator starts not-past-end). */
  int s[2] = { 0, 0 }, e[2] = { 0, 0 }, su[2], eu[2], size[2];
  og_index_from_position(g, tl, s);
  og_index_from_position(g, br, e);
  og_unwrap_index(s, g->size, g->start, su);
  og_unwrap_index(e, g->size, g->start, eu);
  size[0] = eu[0] - su[0] + 1;
  size[1] = eu[1] - su[1] + 1;
  int idx[2] = { s[0], s[1] }, sub[2] = { 0, 0 };
  int n = 0;
  for (;;) {
    double p[2];
    og_position_from_index(g, idx, p);
    double ddx = p[0] - center[0], ddy = p[1] - center[1];
    if (ddx * ddx + ddy * ddy <= r2) {
      if (n < cap) { cells[2 * n] = idx[0]; cells[2 * n + 1] = idx[1]; }
      n++;
    }
    if (!og_increment_index_for_submap(sub, idx, s, size, g->size, g->start)) break;
  }
  return n;
}
