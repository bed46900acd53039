/*
 * rrt.c -- ORACLE (test infrastructure): goal-biased RRT of move_control restated in plain C.
 * Follows mc/src/rrt_planner.cpp:4-104, mc/include/move_control/rrt_planner.h:17-36 and
 * mc/include/move_control/map_global_planner.h:10-86.  "parity unpinned" except for og_rand(),
 * which replicates glibc's TYPE_3 random()/rand() and is pinned against this libc in the tests.
 *
 * Departures (documented in DESIGN.md):
 *  - rand() is global and unseeded in the reference; every query here owns a generator seeded
 *    with srand(seed) semantics (seed 1 == the reference's very first plan after process start).
 *  - extendTree's while(true) (rrt_planner.cpp:28) is bounded by max_samples.
 */
#include "rna_oracle.h"

#include <math.h>
#include <stdlib.h>

/* glibc stdlib/random_r.c : __srandom_r for TYPE_3 (deg 31, sep 3) */
void og_srand(og_rand_state* s, unsigned seed) {
  if (seed == 0) seed = 1;
  int32_t word = (int32_t)seed;
  s->r[0] = word;
  for (int i = 1; i < 31; ++i) {
    long hi = word / 127773;
    long lo = word % 127773;
    word = (int32_t)(16807 * lo - 2836 * hi);
    if (word < 0) word += 2147483647;
    s->r[i] = word;
  }
  s->f = 3;
  s->b = 0;
  for (int i = 0; i < 310; ++i) (void)og_rand(s);
}

/* glibc stdlib/random_r.c : __random_r for TYPE_3 */
int og_rand(og_rand_state* s) {
  uint32_t val = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
  s->r[s->f] = (int32_t)val;
  int result = (int)(val >> 1);
  s->f++;
  if (s->f >= 31) { s->f = 0; s->b++; }
  else { s->b++; if (s->b >= 31) s->b = 0; }
  return result;
}

/* map_global_planner.h:39-54 : CircleIterator r = 0.3, blocked iff any finite cell > 0 */
int og_if_blocked(const og_geom* g, const float* master, const double p[2]) {
  int cells[2 * 1024];
  int n = og_circle_cells(g, p, 0.3, cells, 1024);
  if (n > 1024) n = 1024;
  for (int k = 0; k < n; ++k) {
    /* A corner within rounding of the map's far edge gives index == size (getIndexFromPosition does not wrap on an
     * unmoved map); the reference then reads the layer out of bounds.  Defined here and in the HIP kernel: skipped. */
    if (cells[2 * k] < 0 || cells[2 * k] >= g->size[0] || cells[2 * k + 1] < 0 || cells[2 * k + 1] >= g->size[1]) continue;
    float v = master[(size_t)cells[2 * k + 1] * g->size[0] + cells[2 * k]];
    if (isnan(v)) continue;
    if (v > 0.0f) return 1;
  }
  return 0;
}

typedef struct { double pos[2]; int parent; } rrt_node;

/* steer: how the new node is placed one stride from the nearest one towards the sample (rrt_planner.cpp:44-49)
 *   0  as the reference writes it: a = atan2(dy, dx); near + stride * (cos(a), sin(a))         [this libc's libm]
 *   1  the same point without the three libm calls: near + stride * (dx, dy) / sqrt(dx^2 + dy^2)  [IEEE operations only]
 * The two agree to the last bit or two of the offset; the sum with the node's coordinate then rounds differently for a few
 * per cent of the nodes, by one ulp.  The HIP kernel uses 1 (no device libm reproduces glibc's atan2 / cos / sin bit for bit
 * anyway); the GPU parity tests compare it with steer = 1 BIT FOR BIT, and tests/test_oracle_misc.py measures how rarely
 * the one-ulp difference between 0 and 1 changes a tree (a later nearest-node or blocked-disc decision at a last-bit tie). */
void og_rrt_plan_steer(const og_geom* g, const float* master, const double start[2], const double target[2],
                       double close_tol, unsigned seed, int max_samples, int steer, double* path_xy, int path_cap,
                       og_rrt_result* res) {
  const int iteratorNum = 2000;             /* rrt_planner.cpp:6 */
  const double strideStep = 0.4;            /* rrt_planner.h:23 */
  const int targetTendency = (int)0.5;      /* rrt_planner.h:24,32 : int <- 0.5 == 0 */
  const int target_inside = og_position_within_map(target, g->len, g->pos);
  rrt_node* tree = (rrt_node*)malloc(sizeof(rrt_node) * (size_t)iteratorNum);
  int n_tree = 0, samples = 0, finished = 0, aborted = 0;
  og_rand_state rs;
  og_srand(&rs, seed);

  rrt_node node;
  node.parent = -1; node.pos[0] = start[0]; node.pos[1] = start[1];

  for (int it = 0; it < iteratorNum && !aborted; ++it) {
    tree[n_tree++] = node;
    /* ifFinishPlan, map_global_planner.h:32-37,61-86 */
    int fin;
    if (target_inside) {
      fin = hypot(node.pos[0] - target[0], node.pos[1] - target[1]) < close_tol;
    } else {
      double pbx = g->pos[0] + g->len[0] / 2, pby = g->pos[1] + g->len[1] / 2;
      double mbx = g->pos[0] - g->len[0] / 2, mby = g->pos[1] - g->len[1] / 2;
      fin = ((pbx - node.pos[0]) < close_tol) || ((pby - node.pos[1]) < close_tol) ||
            ((node.pos[0] - mbx) < close_tol) || ((node.pos[1] - mby) < close_tol);
    }
    if (fin) { finished = 1; break; }

    /* extendTree, rrt_planner.cpp:26-59 */
    for (;;) {
      if (samples >= max_samples) { aborted = 1; break; }
      samples++;
      double rnd[2];
      if (og_rand(&rs) % 10 > 3) {
        int ridx[2];
        ridx[0] = og_rand(&rs) % g->size[0];
        ridx[1] = og_rand(&rs) % g->size[1];
        og_position_from_index(g, ridx, rnd);
      } else {
        rnd[0] = target[0]; rnd[1] = target[1];
      }
      /* findNearNode, rrt_planner.cpp:70-89 */
      int near = 0;
      double shortest = 9999.0;
      for (int i = 0; i < n_tree; ++i) {
        double d = hypot(rnd[0] - tree[i].pos[0], rnd[1] - tree[i].pos[1]);
        d = d + targetTendency * hypot(tree[i].pos[0] - target[0], tree[i].pos[1] - target[1]);
        if (d < shortest) { shortest = d; near = i; }
      }
      const double* np = tree[near].pos;
      double nw[2];
      if (hypot(np[0] - rnd[0], np[1] - rnd[1]) < strideStep) {
        nw[0] = rnd[0]; nw[1] = rnd[1];
      } else {
        if (steer == 0) {
          double a = atan2(rnd[1] - np[1], rnd[0] - np[0]);
          nw[0] = np[0] + strideStep * cos(a);
          nw[1] = np[1] + strideStep * sin(a);
        } else {
          const double sdx = rnd[0] - np[0], sdy = rnd[1] - np[1];
          const double sh = sqrt(sdx * sdx + sdy * sdy);
          nw[0] = np[0] + strideStep * (sdx / sh);
          nw[1] = np[1] + strideStep * (sdy / sh);
        }
      }
      if (!og_if_blocked(g, master, nw)) {
        node.pos[0] = nw[0]; node.pos[1] = nw[1]; node.parent = near;
        break;
      }
    }
  }

  res->tree_size = n_tree;
  res->samples = samples;
  res->status = aborted ? -1 : (finished ? 1 : 0);
  /* backtraceTree, rrt_planner.cpp:91-104 : from rrtTree_.back() to the root (goal -> start) */
  int len = 0;
  if (!aborted && n_tree > 0) {
    int i = n_tree - 1;
    for (;;) {
      if (len < path_cap) { path_xy[2 * len] = tree[i].pos[0]; path_xy[2 * len + 1] = tree[i].pos[1]; }
      len++;
      if (tree[i].parent == -1) break;
      i = tree[i].parent;
    }
  }
  res->path_len = len;
  free(tree);
}

/* the reference's formulation */
void og_rrt_plan(const og_geom* g, const float* master, const double start[2], const double target[2],
                 double close_tol, unsigned seed, int max_samples, double* path_xy, int path_cap,
                 og_rrt_result* res) {
  og_rrt_plan_steer(g, master, start, target, close_tol, seed, max_samples, 0, path_xy, path_cap, res);
}
