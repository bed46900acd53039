/*
 * astar.c -- ORACLE (test infrastructure).
 *
 * (1) Grid A*: the reference has NO grid A* (its AStarPlanner searches a hard-coded 9-vertex
 *     waypoint graph, mc/src/astar_planner.cpp:98-127), so this file DEFINES the contract that the
 *     MI355X engine must reproduce bit-exactly (DESIGN.md "Grid A* contract"):
 *       - 8-connected grid over buffer indices, linear index = i + j*rows (Eigen column-major,
 *         gmc/src/GridMapMath.cpp:478-482);
 *       - a cell is blocked iff its master value is finite and > 0 (GlobalPlanner::ifBlocked
 *         predicate, mc/include/move_control/map_global_planner.h:47-50);
 *       - a diagonal step needs the target AND both orthogonally adjacent cells free;
 *       - integer costs 1000 (straight) / 1414 (diagonal), octile heuristic (consistent);
 *       - the search settles every cell with f = g + h <= f* = g(goal);  E = that count;
 *       - the path is the canonical backtrace over the exact g field: from the goal, step to the
 *         neighbour n with g[n] + w(n,c) == g[c], lowest linear index first.
 *     The result therefore depends only on the exact distance field, not on expansion order.
 *
 * (2) Waypoint-graph A*: restates boost::astar_search (Boost.Graph 1.54, the version ROS Indigo
 *     ships; NOT vendored under /root/reference) as called at mc/src/astar_planner.cpp:72-77:
 *     4-ary indirect heap keyed on f = d + h, BFS colouring, relax() with the undirected branch,
 *     visitor throwing at examine_vertex(goal).  "parity unpinned" (no reference test, Boost absent).
 */
#include "rna_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int NB_DI[8] = { -1, 0, 1, -1, 1, -1, 0, 1 };
static const int NB_DJ[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
static const int NB_W[8] = { OG_ASTAR_COST_DIAG, OG_ASTAR_COST_STRAIGHT, OG_ASTAR_COST_DIAG,
                             OG_ASTAR_COST_STRAIGHT, OG_ASTAR_COST_STRAIGHT,
                             OG_ASTAR_COST_DIAG, OG_ASTAR_COST_STRAIGHT, OG_ASTAR_COST_DIAG };

void og_astar_blocked_mask(const float* master, size_t n, uint8_t* blocked) {
  for (size_t i = 0; i < n; ++i) {
    float v = master[i];
    blocked[i] = (!isnan(v) && v > 0.0f) ? 1 : 0;
  }
}

void og_astar_nbr_mask(const uint8_t* blocked, int rows, int cols, uint8_t* nbr) {
  for (int j = 0; j < cols; ++j) {
    for (int i = 0; i < rows; ++i) {
      size_t c = (size_t)j * rows + i;
      uint8_t m = 0;
      if (!blocked[c]) {
        for (int k = 0; k < 8; ++k) {
          int ni = i + NB_DI[k], nj = j + NB_DJ[k];
          if (ni < 0 || nj < 0 || ni >= rows || nj >= cols) continue;
          if (blocked[(size_t)nj * rows + ni]) continue;
          if (NB_DI[k] != 0 && NB_DJ[k] != 0) {
            if (blocked[(size_t)j * rows + ni]) continue;
            if (blocked[(size_t)nj * rows + i]) continue;
          }
          m |= (uint8_t)(1u << k);
        }
      }
      nbr[c] = m;
    }
  }
}

static inline int octile(int i, int j, int gi, int gj) {
  int dx = i > gi ? i - gi : gi - i;
  int dy = j > gj ? j - gj : gj - j;
  int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return OG_ASTAR_COST_STRAIGHT * mx + (OG_ASTAR_COST_DIAG - OG_ASTAR_COST_STRAIGHT) * mn;
}

typedef struct { int32_t f, g, cell; } hnode;

static void heap_push(hnode** h, size_t* n, size_t* cap, hnode v) {
  if (*n == *cap) { *cap = *cap ? *cap * 2 : 1024; *h = (hnode*)realloc(*h, *cap * sizeof(hnode)); }
  size_t i = (*n)++;
  hnode* a = *h;
  while (i > 0) {
    size_t p = (i - 1) >> 1;
    if (a[p].f < v.f || (a[p].f == v.f && a[p].g >= v.g)) break; /* ties: larger g first (deeper) */
    a[i] = a[p];
    i = p;
  }
  a[i] = v;
}

static hnode heap_pop(hnode* a, size_t* n) {
  hnode top = a[0];
  hnode v = a[--(*n)];
  size_t i = 0, sz = *n;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m;
    if (l >= sz) break;
    m = (r < sz && (a[r].f < a[l].f || (a[r].f == a[l].f && a[r].g > a[l].g))) ? r : l;
    if (v.f < a[m].f || (v.f == a[m].f && v.g >= a[m].g)) break;
    a[i] = a[m];
    i = m;
  }
  if (sz) a[i] = v;
  return top;
}

/* measurement aid (bench.py's cpu_baseline leg): how many cells the calling thread's last og_astar_query had closed when
 * the goal came off the heap -- the count of an A* that stops there (SURVEY.md 7's wording) instead of settling the whole
 * f == f* plateau as the contract does (the plateau is what makes the canonical path independent of the heap's tie order) */
static __thread int32_t og_last_settled_at_goal;
int32_t og_astar_last_settled_at_goal(void) { return og_last_settled_at_goal; }

void og_astar_query(const uint8_t* nbr, int rows, int cols, int start_lin, int goal_lin,
                    int32_t* g, int32_t* path, int path_cap, og_astar_result* res) {
  og_last_settled_at_goal = 0;
  size_t ncell = (size_t)rows * cols;
  res->status = 2; res->path_len = 0; res->cost = OG_ASTAR_INF; res->settled = 0;
  if (start_lin < 0 || goal_lin < 0 || (size_t)start_lin >= ncell || (size_t)goal_lin >= ncell) return;
  for (size_t i = 0; i < ncell; ++i) g[i] = OG_ASTAR_INF;
  /* a blocked cell has an empty mask AND no neighbour lists it; a free isolated cell also has
   * mask 0, so "blocked start/goal" is reported through the same status as "no path" unless
   * start == goal.  Callers that need the distinction test the blocked mask themselves. */
  const int gi = goal_lin % rows, gj = goal_lin / rows;
  hnode* heap = NULL; size_t hn = 0, hcap = 0;
  g[start_lin] = 0;
  hnode s = { octile(start_lin % rows, start_lin / rows, gi, gj), 0, start_lin };
  heap_push(&heap, &hn, &hcap, s);
  int32_t fstar = OG_ASTAR_INF;
  int32_t settled = 0;
  while (hn) {
    if (heap[0].f > fstar) break;
    hnode u = heap_pop(heap, &hn);
    if (u.g != g[u.cell]) continue; /* stale */
    settled++;
    if (u.cell == goal_lin && fstar == OG_ASTAR_INF) { fstar = u.g; og_last_settled_at_goal = settled; } /* keep settling ties f == f* */
    const int ui = u.cell % rows, uj = u.cell / rows;
    const uint8_t m = nbr[u.cell];
    for (int k = 0; k < 8; ++k) {
      if (!(m & (1u << k))) continue;
      const int ni = ui + NB_DI[k], nj = uj + NB_DJ[k];
      const int nc = nj * rows + ni;
      const int32_t ng = u.g + NB_W[k];
      if (ng < g[nc]) {
        g[nc] = ng;
        hnode v = { ng + octile(ni, nj, gi, gj), ng, nc };
        if (v.f <= fstar) heap_push(&heap, &hn, &hcap, v);
      }
    }
  }
  free(heap);
  res->settled = settled;
  if (fstar == OG_ASTAR_INF) { res->status = 1; return; }
  res->cost = g[goal_lin];
  /* canonical backtrace */
  int len = 0, c = goal_lin;
  int32_t* rev = (int32_t*)malloc(sizeof(int32_t) * 1024);
  size_t rcap = 1024;
  for (;;) {
    if ((size_t)len == rcap) { rcap *= 2; rev = (int32_t*)realloc(rev, rcap * sizeof(int32_t)); }
    rev[len++] = c;
    if (c == start_lin) break;
    const int ci = c % rows, cj = c / rows;
    const uint8_t m = nbr[c];
    int next = -1;
    for (int k = 0; k < 8; ++k) {
      if (!(m & (1u << k))) continue;
      const int nc = (cj + NB_DJ[k]) * rows + (ci + NB_DI[k]);
      if (g[nc] != OG_ASTAR_INF && g[nc] + NB_W[k] == g[c]) { next = nc; break; }
    }
    if (next < 0) { free(rev); res->status = 1; return; } /* cannot happen for a consistent field */
    c = next;
  }
  res->path_len = len;
  if (len > path_cap) { res->status = 3; free(rev); return; }
  for (int k = 0; k < len; ++k) path[k] = rev[len - 1 - k];
  free(rev);
  res->status = 0;
}

/* Grid A* on a map whose circular buffer has been moved (GridMap::move, gmc/src/GridMap.cpp:346-412):
 * the contract is the one above applied in MAP space -- adjacency, heuristic and the canonical
 * backtrace ("lowest linear index") use unwrapped indices (gmc/src/GridMapMath.cpp:467-476), the
 * map does not wrap around at its edges; start, goal and the returned path are buffer linear indices
 * (what GridMap::getIndex hands out).  With start index (0,0) this is og_astar_query. */
void og_astar_query_on_map(const og_geom* g, const float* master, int start_lin, int goal_lin, int32_t* g_work,
                           int32_t* path, int path_cap, og_astar_result* res) {
  const int rows = g->size[0], cols = g->size[1];
  const size_t n = (size_t)rows * cols;
  res->status = 2; res->path_len = 0; res->cost = OG_ASTAR_INF; res->settled = 0;
  if (start_lin < 0 || goal_lin < 0 || (size_t)start_lin >= n || (size_t)goal_lin >= n) return;
  float* um = (float*)malloc(n * sizeof(float));
  uint8_t* blocked = (uint8_t*)malloc(n);
  uint8_t* nbr = (uint8_t*)malloc(n);
  for (int uj = 0; uj < cols; ++uj)
    for (int ui = 0; ui < rows; ++ui) {
      const int u[2] = {ui, uj};
      int b[2];
      og_buffer_index(u, g->size, g->start, b);
      um[(size_t)uj * rows + ui] = master[(size_t)b[1] * rows + b[0]];
    }
  og_astar_blocked_mask(um, n, blocked);
  og_astar_nbr_mask(blocked, rows, cols, nbr);
  const int sb[2] = {start_lin % rows, start_lin / rows}, gb[2] = {goal_lin % rows, goal_lin / rows};
  int su[2], gu[2];
  og_unwrap_index(sb, g->size, g->start, su);
  og_unwrap_index(gb, g->size, g->start, gu);
  og_astar_query(nbr, rows, cols, su[0] + su[1] * rows, gu[0] + gu[1] * rows, g_work, path, path_cap, res);
  if (res->status == 0) {
    for (int k = 0; k < res->path_len && k < path_cap; ++k) {
      const int u[2] = {path[k] % rows, path[k] / rows};
      int b[2];
      og_buffer_index(u, g->size, g->start, b);
      path[k] = b[0] + b[1] * rows;
    }
  }
  free(um); free(blocked); free(nbr);
}

/* ------------------------------------------------------------------------------------------ */
/* Waypoint-graph A* with Boost.Graph semantics                                               */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  int* data; int n; int* index_in_heap; const float* key;
} dheap;

static void dheap_up(dheap* h, int index) {
  if (index == 0) return;
  int moving = h->data[index];
  float dist = h->key[moving];
  int levels = 0, idx = index;
  while (idx != 0) {
    int parent = (idx - 1) / 4;
    if (dist < h->key[h->data[parent]]) { ++levels; idx = parent; } else break;
  }
  idx = index;
  for (int i = 0; i < levels; ++i) {
    int parent = (idx - 1) / 4;
    h->data[idx] = h->data[parent];
    h->index_in_heap[h->data[idx]] = idx;
    idx = parent;
  }
  h->data[idx] = moving;
  h->index_in_heap[moving] = idx;
}

static void dheap_down(dheap* h) {
  if (h->n == 0) return;
  int index = 0;
  int moving = h->data[0];
  float dist = h->key[moving];
  for (;;) {
    int first = index * 4 + 1;
    if (first >= h->n) break;
    int nchild = (first + 4 <= h->n) ? 4 : h->n - first;
    int best = 0;
    float best_d = h->key[h->data[first]];
    for (int i = 1; i < nchild; ++i) {
      float d = h->key[h->data[first + i]];
      if (d < best_d) { best = i; best_d = d; }
    }
    if (best_d < dist) {
      int ci = first + best;
      int tmp = h->data[ci]; h->data[ci] = h->data[index]; h->data[index] = tmp;
      h->index_in_heap[h->data[ci]] = ci;
      h->index_in_heap[h->data[index]] = index;
      index = ci;
    } else break;
  }
}

static void dheap_push(dheap* h, int v) {
  int idx = h->n++;
  h->data[idx] = v;
  h->index_in_heap[v] = idx;
  dheap_up(h, idx);
}

static int dheap_pop(dheap* h) {
  int top = h->data[0];
  h->index_in_heap[top] = -1;
  if (h->n != 1) {
    h->data[0] = h->data[h->n - 1];
    h->index_in_heap[h->data[0]] = 0;
    h->n--;
    dheap_down(h);
  } else h->n--;
  return top;
}

/* distance_heuristic, astar_planner.cpp:16-32 : CostType = float, ::sqrt(float) */
static float graph_heuristic(const double* loc, int goal, int u) {
  float dx = (float)(loc[2 * goal] - loc[2 * u]);
  float dy = (float)(loc[2 * goal + 1] - loc[2 * u + 1]);
  return sqrtf(dx * dx + dy * dy);
}

int og_graph_closest_vertex(int nv, const double* loc, const double pos[2]) {
  /* astar_planner.cpp:129-145 : float closedDistance = 999, float distance = hypot(double,double) */
  float closed = 999.0f;
  int best = 0; /* uninitialised in the reference when nothing is closer than 999 m */
  for (int v = 0; v < nv; ++v) {
    float d = (float)hypot(pos[0] - loc[2 * v], pos[1] - loc[2 * v + 1]);
    if (d < closed) { closed = d; best = v; }
  }
  return best;
}

int og_graph_astar(int nv, const double* loc, int ne, const int* euv, const float* ew, int start,
                   int goal, int* path, int cap) {
  /* out-edge lists in add_edge order (adjacency_list<listS, vecS, undirectedS>) */
  int* deg = (int*)calloc((size_t)nv + 1, sizeof(int));
  for (int e = 0; e < ne; ++e) { deg[euv[2 * e]]++; deg[euv[2 * e + 1]]++; }
  int* off = (int*)calloc((size_t)nv + 1, sizeof(int));
  for (int v = 0; v < nv; ++v) off[v + 1] = off[v] + deg[v];
  int* fill = (int*)calloc((size_t)nv, sizeof(int));
  int* adj_v = (int*)malloc(sizeof(int) * (size_t)(2 * ne + 1));
  float* adj_w = (float*)malloc(sizeof(float) * (size_t)(2 * ne + 1));
  for (int e = 0; e < ne; ++e) {
    int u = euv[2 * e], v = euv[2 * e + 1];
    float w = ew ? ew[e] : 0.0f; /* add_edge without a property -> value-initialised weight 0 */
    adj_v[off[u] + fill[u]] = v; adj_w[off[u] + fill[u]++] = w;
    adj_v[off[v] + fill[v]] = u; adj_w[off[v] + fill[v]++] = w;
  }
  float* d = (float*)malloc(sizeof(float) * (size_t)nv);
  float* f = (float*)malloc(sizeof(float) * (size_t)nv);
  int* p = (int*)malloc(sizeof(int) * (size_t)nv);
  int* color = (int*)calloc((size_t)nv, sizeof(int)); /* 0 white 1 gray 2 black */
  dheap h;
  h.data = (int*)malloc(sizeof(int) * (size_t)(nv + 1));
  h.index_in_heap = (int*)malloc(sizeof(int) * (size_t)nv);
  h.n = 0; h.key = f;
  for (int v = 0; v < nv; ++v) { d[v] = FLT_MAX; f[v] = FLT_MAX; p[v] = v; h.index_in_heap[v] = -1; }
#define HEUR(u) graph_heuristic(loc, goal, (u))
  d[start] = 0; f[start] = HEUR(start);
  color[start] = 1;
  dheap_push(&h, start);
  int found = 0;
  while (h.n) {
    int u = h.data[0];
    dheap_pop(&h);
    if (u == goal) { found = 1; break; } /* examine_vertex -> throw found_goal */
    for (int k = off[u]; k < off[u + 1]; ++k) {
      int v = adj_v[k];
      float w = adj_w[k];
      /* relax(), boost/graph/relax.hpp (closed_plus, std::less) incl. the undirected branch */
      int decreased = 0;
      float d_u = d[u], d_v = d[v];
      float cand = (d_u == FLT_MAX || w == FLT_MAX) ? FLT_MAX : d_u + w;
      if (cand < d_v) {
        d[v] = cand;
        if (d[v] < d_v) { p[v] = u; decreased = 1; }
      } else {
        float cand2 = (d_v == FLT_MAX || w == FLT_MAX) ? FLT_MAX : d_v + w;
        if (cand2 < d_u) { d[u] = cand2; if (d[u] < d_u) { p[u] = v; decreased = 1; } }
      }
      if (color[v] == 0) {
        if (decreased) f[v] = (d[v] == FLT_MAX) ? FLT_MAX : d[v] + HEUR(v);
        color[v] = 1;
        dheap_push(&h, v);
      } else if (color[v] == 1) {
        if (decreased) { f[v] = (d[v] == FLT_MAX) ? FLT_MAX : d[v] + HEUR(v); dheap_up(&h, h.index_in_heap[v]); }
      } else {
        if (decreased) { f[v] = (d[v] == FLT_MAX) ? FLT_MAX : d[v] + HEUR(v); dheap_push(&h, v); color[v] = 1; }
      }
    }
    color[u] = 2;
  }
#undef HEUR
  int len = 0;
  if (found) {
    /* astar_planner.cpp:80-86 : walk p[] back from the goal until p[v] == v */
    int tmp[4096];
    for (int v = goal;; v = p[v]) { if (len < 4096) tmp[len] = v; len++; if (p[v] == v) break; if (len > nv + 1) break; }
    for (int k = 0; k < len && k < cap; ++k) path[k] = tmp[len - 1 - k];
  }
  free(deg); free(off); free(fill); free(adj_v); free(adj_w); free(d); free(f); free(p); free(color);
  free(h.data); free(h.index_in_heap);
  return len;
}

/* astar_planner.cpp:98-127 */
int og_reference_graph(double* loc, int* euv) {
  const int m = 8, n = 5;
  const double L[9][2] = { { 0.5 * m, 0 * n }, { 1.5 * m, 0 * n }, { 2.5 * m, 0 * n }, { 2.5 * m, 1 * n },
                           { 1.5 * m, 1 * n }, { 0.5 * m, 1 * n }, { 0.5 * m, 2 * n }, { 1.5 * m, 2 * n },
                           { 2.5 * m, 2 * n } };
  const int E[10][2] = { { 0, 1 }, { 1, 2 }, { 2, 3 }, { 3, 4 }, { 4, 5 }, { 5, 6 }, { 6, 7 }, { 7, 8 }, { 0, 5 }, { 3, 8 } };
  for (int i = 0; i < 9; ++i) { loc[2 * i] = L[i][0]; loc[2 * i + 1] = L[i][1]; }
  for (int e = 0; e < 10; ++e) { euv[2 * e] = E[e][0]; euv[2 * e + 1] = E[e][1]; }
  return 9;
}

/* astar_planner.cpp:63-96 : path = start, vertex locations..., target (appended) */
int og_graph_make_plan(const double start[2], const double target[2], double* path_xy, int cap) {
  double loc[18]; int euv[20];
  og_reference_graph(loc, euv);
  int s = og_graph_closest_vertex(9, loc, start);
  int t = og_graph_closest_vertex(9, loc, target);
  int verts[16];
  int n = og_graph_astar(9, loc, 10, euv, NULL, s, t, verts, 16);
  if (n == 0) return 0;
  int len = 0;
  if (len < cap) { path_xy[0] = start[0]; path_xy[1] = start[1]; } len++;
  for (int k = 0; k < n; ++k) { if (len < cap) { path_xy[2 * len] = loc[2 * verts[k]]; path_xy[2 * len + 1] = loc[2 * verts[k] + 1]; } len++; }
  if (len < cap) { path_xy[2 * len] = target[0]; path_xy[2 * len + 1] = target[1]; } len++;
  return len;
}
