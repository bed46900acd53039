// planners.hip -- the reference's two global planners, batched: waypoint-graph A* and goal-biased RRT.
//
//   AStarPlanner::makePlan / findClosedVertex   mc/src/astar_planner.cpp:63-145 (boost::astar_search,
//                                               Boost.Graph 1.54 semantics restated, see oracle/astar.c)
//   RrtPlanner::makePlan/extendTree/sample/findNearNode/backtraceTree   mc/src/rrt_planner.cpp:4-104
//   GlobalPlanner::ifBlocked/ifFinishPlan + CircleIterator  mc/include/move_control/map_global_planner.h:32-86,
//                                               gmc/src/iterators/CircleIterator.cpp:16-93
//
// RRT layout: one query per workgroup, RRT_SPEC wavefronts that each evaluate one of the next RRT_SPEC samples
// against the current tree (see rrt_kernel).  The tree positions (<= 2000 nodes) live in LDS, the parent
// links in HBM; the nearest-node scan strides the tree across the 64 lanes and finishes with a DPP arg-min
// that keeps the lowest index on ties (the reference's strict `<` scan); the 0.3 m footprint test spreads the
// <= 13 x 13 CircleIterator window over the lanes and reduces with a ballot.  Each wavefront carries a
// glibc-compatible rand() (TYPE_3 additive feedback) in registers so sample sequences equal srand(seed); rand().
#include "engine.hpp"

#include <vector>

using namespace rna;

namespace {

// ------------------------------------------------------------------------------------------------
// waypoint-graph A*: one lane per query; graphs are tiny (the reference's has 9 vertices)
// ------------------------------------------------------------------------------------------------
constexpr int GRAPH_MAX_V = 256;

struct GraphK {
  int nv, ne;
  const double* loc;     // [nv][2]
  const int* adj_off;    // [nv+1] out-edge lists in add_edge order (listS out-edge container)
  const int* adj_v;
  const float* adj_w;
};

__device__ float graph_heuristic(const GraphK& G, int goal, int u) {  // astar_planner.cpp:16-32
  const float dx = (float)(G.loc[2 * goal] - G.loc[2 * u]);
  const float dy = (float)(G.loc[2 * goal + 1] - G.loc[2 * u + 1]);
  return sqrtf(dx * dx + dy * dy);
}

__device__ int graph_closest_vertex(const GraphK& G, double x, double y) {  // astar_planner.cpp:129-145
  float closed = 999.0f;
  int best = 0;
  for (int v = 0; v < G.nv; ++v) {
    const float d = (float)glibc_hypot(x - G.loc[2 * v], y - G.loc[2 * v + 1]);
    if (d < closed) { closed = d; best = v; }
  }
  return best;
}

// d_ary_heap_indirect<Vertex, 4> keyed on f (boost/graph/detail/d_ary_heap.hpp)
struct DHeap { int* data; int* pos; const float* key; int n; };

__device__ void dheap_up(DHeap& h, int index) {
  if (index == 0) return;
  const int moving = h.data[index];
  const float dist = h.key[moving];
  int levels = 0, idx = index;
  while (idx != 0) {
    const int parent = (idx - 1) / 4;
    if (dist < h.key[h.data[parent]]) { ++levels; idx = parent; } else break;
  }
  idx = index;
  for (int i = 0; i < levels; ++i) {
    const int parent = (idx - 1) / 4;
    h.data[idx] = h.data[parent];
    h.pos[h.data[idx]] = idx;
    idx = parent;
  }
  h.data[idx] = moving;
  h.pos[moving] = idx;
}

__device__ void dheap_down(DHeap& h) {
  if (h.n == 0) return;
  int index = 0;
  const int moving = h.data[0];
  const float dist = h.key[moving];
  for (;;) {
    const int first = index * 4 + 1;
    if (first >= h.n) break;
    const int nchild = (first + 4 <= h.n) ? 4 : h.n - first;
    int best = 0;
    float best_d = h.key[h.data[first]];
    for (int i = 1; i < nchild; ++i) {
      const float d = h.key[h.data[first + i]];
      if (d < best_d) { best = i; best_d = d; }
    }
    if (best_d < dist) {
      const int ci = first + best;
      const int tmp = h.data[ci]; h.data[ci] = h.data[index]; h.data[index] = tmp;
      h.pos[h.data[ci]] = ci;
      h.pos[h.data[index]] = index;
      index = ci;
    } else break;
  }
}

__global__ void graph_astar_kernel(GraphK G, const double* __restrict__ start_target, int n,
                                   float* __restrict__ scratch_f, int* __restrict__ scratch_i,
                                   double* __restrict__ paths, int max_len, int* __restrict__ path_len) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int nv = G.nv;
  float* d = scratch_f + (size_t)q * 2 * nv;
  float* f = d + nv;
  int* p = scratch_i + (size_t)q * 4 * nv;
  int* color = p + nv;
  DHeap h{color + nv, color + 2 * nv, f, 0};
  const double sx = start_target[4 * q], sy = start_target[4 * q + 1];
  const double tx = start_target[4 * q + 2], ty = start_target[4 * q + 3];
  const int start = graph_closest_vertex(G, sx, sy);
  const int goal = graph_closest_vertex(G, tx, ty);
  const float FINF = 3.402823466e+38f;  // numeric_limits<float>::max() = astar_search's default "inf"
  for (int v = 0; v < nv; ++v) { d[v] = FINF; f[v] = FINF; p[v] = v; color[v] = 0; h.pos[v] = -1; }
  d[start] = 0;
  f[start] = graph_heuristic(G, goal, start);
  color[start] = 1;
  h.data[0] = start; h.pos[start] = 0; h.n = 1;
  bool found = false;
  while (h.n) {
    const int u = h.data[0];
    h.pos[u] = -1;
    if (h.n != 1) { h.data[0] = h.data[h.n - 1]; h.pos[h.data[0]] = 0; h.n--; dheap_down(h); }
    else h.n--;
    if (u == goal) { found = true; break; }  // astar_goal_visitor::examine_vertex throws found_goal
    for (int k = G.adj_off[u]; k < G.adj_off[u + 1]; ++k) {
      const int v = G.adj_v[k];
      const float w = G.adj_w[k];
      bool decreased = false;
      const float d_u = d[u], d_v = d[v];
      const float cand = (d_u == FINF || w == FINF) ? FINF : d_u + w;   // closed_plus
      if (cand < d_v) {
        d[v] = cand;
        if (d[v] < d_v) { p[v] = u; decreased = true; }
      } else {
        const float cand2 = (d_v == FINF || w == FINF) ? FINF : d_v + w;  // undirected branch of relax()
        if (cand2 < d_u) { d[u] = cand2; if (d[u] < d_u) { p[u] = v; decreased = true; } }
      }
      const float fv = (d[v] == FINF) ? FINF : d[v] + graph_heuristic(G, goal, v);
      if (color[v] == 0) {
        if (decreased) f[v] = fv;
        color[v] = 1;
        h.data[h.n] = v; h.pos[v] = h.n; h.n++; dheap_up(h, h.n - 1);
      } else if (color[v] == 1) {
        if (decreased) { f[v] = fv; dheap_up(h, h.pos[v]); }
      } else if (decreased) {
        f[v] = fv;
        h.data[h.n] = v; h.pos[v] = h.n; h.n++; dheap_up(h, h.n - 1);
        color[v] = 1;
      }
    }
    color[u] = 2;
  }
  double* out = paths + (size_t)q * max_len * 2;
  int len = 0;
  if (found) {
    // astar_planner.cpp:80-94 : start, vertex locations (front-inserted walk of p[]), target
    int nvert = 0;
    for (int v = goal;; v = p[v]) { ++nvert; if (p[v] == v || nvert > nv) break; }
    len = nvert + 2;
    if (len <= max_len) {
      out[0] = sx; out[1] = sy;
      int v = goal;
      for (int k = nvert; k >= 1; --k) { out[2 * k] = G.loc[2 * v]; out[2 * k + 1] = G.loc[2 * v + 1]; v = p[v]; }
      out[2 * (nvert + 1)] = tx; out[2 * (nvert + 1) + 1] = ty;
    }
  }
  path_len[q] = len;
}

// ------------------------------------------------------------------------------------------------
// RRT
// ------------------------------------------------------------------------------------------------
constexpr int RRT_ITER = 2000;          // rrt_planner.cpp:6
#ifndef RNA_RRT_WIDE
#define RNA_RRT_WIDE 0   // 1: every wavefront carries TWO registers of look-ahead draws (124 instead of 62) and twelve wavefronts evaluate a
                         // round's samples -- the lever rounds 4-5 named for config 4's share.  Built in round 6, exact (the three GPU RRT
                         // tests: config 4's 512 queries bit for bit), and SLOWER: 57.5 ms against 41.5 (profiles/r06_rrt_wide_window.txt).
                         // A round covers 10.8 samples instead of 8.7 (+25 %: the rounds that accept a sample waste the rest either way),
                         // but it takes 10.0 us instead of 5.7: the sample bookkeeping over pairs of 62-bit limbs 0.78 -> 1.84 us on every
                         // wavefront, and wave 0 waits 4.4 us instead of 1.4 for the slowest of twelve wavefronts that share the SIMDs of
                         // one CU three deep.  With eight wavefronts and the wide window 53.6 ms, with sixteen 76.4.  Off.
#endif
#ifndef RNA_RRT_SPEC
#define RNA_RRT_SPEC (RNA_RRT_WIDE ? 12 : 8)
#endif
constexpr int RRT_SPEC = RNA_RRT_SPEC;  // wavefronts per query = samples evaluated speculatively per round
static_assert(RRT_SPEC >= 1 && RRT_SPEC <= 16, "one workgroup per query");   // (a round covers the samples whose draws lie inside the 62-draw window: at least 10 random ones, fewer wavefronts than samples just wait)

// glibc rand() (random_r, TYPE_3) is the additive lagged recurrence o[n] = o[n-31] + o[n-3] mod 2^32 with
// result o[n] >> 1.  A wavefront keeps the stream in registers: `d` holds 62 consecutive raw outputs, one per
// lane (two blocks of 31), `last` the newest block (lanes 0..30), `pos` the next unread lane of `d`.  A block
// of 31 new outputs is an inclusive scan along the three stride-3 chains -- four shuffles, no LDS state, no
// sequential draws -- so every wavefront of a query carries its own identical copy of the stream.
//
// sample() (rrt_planner.cpp:61-68) spends one draw on "goal or random" and two more on a random sample, so where
// the samples start inside the window is a chain through the draws: `three` marks the lanes whose draw says
// "random", `chain` the lanes at which a sample starts.  Both are wave masks, recomputed only when the window moves.
struct WaveRng { unsigned d, last; int pos; unsigned long long three, chain; };
constexpr unsigned long long RNG_LIVE = (1ull << 62) - 1ull;   // lanes of `d` that hold draws

__device__ __forceinline__ void rng_chain(WaveRng& r, unsigned long long from) {
  r.three = __ballot((int)(r.d >> 1) % 10 > 3) & RNG_LIVE;
  unsigned long long c = from, prev;
  do {
    prev = c;
    const unsigned long long src = c & RNG_LIVE;
    c |= ((src & ~r.three) << 1) | ((src & r.three) << 3);
  } while (c != prev);
  r.chain = c;
}

__device__ __forceinline__ unsigned rng_block(unsigned x, int lane) {
  const unsigned wrap = (unsigned)__shfl((int)x, lane < 3 ? 28 + lane : lane);
  unsigned y = x + (lane < 3 ? wrap : 0u);
#pragma unroll
  for (int s = 3; s < 31; s <<= 1) {
    const unsigned u = (unsigned)__shfl_up((int)y, s);
    if (lane >= s) y += u;
  }
  return y;
}

__device__ void rng_seed(WaveRng& r, unsigned seed, int lane) {  // glibc srandom_r, TYPE_3
  if (seed == 0) seed = 1;
  int word = (int)seed;
  for (int i = 1; i < 31; ++i) {        // lane l keeps r[l] of the seeding LCG
    if (i <= lane) {
      const long long hi = word / 127773, lo = word % 127773;
      word = (int)(16807 * lo - 2836 * hi);
      if (word < 0) word += 2147483647;
    }
  }
  // front pointer starts at r[3], rear at r[0]: the "previous 31 outputs" are r[3..30], r[0..2]
  unsigned x = (unsigned)__shfl(word, lane < 31 ? (lane + 3) % 31 : lane);
  for (int i = 0; i < 10; ++i) x = rng_block(x, lane);   // srandom_r discards 310 draws
  const unsigned y0 = rng_block(x, lane), y1 = rng_block(y0, lane);
  const unsigned up = (unsigned)__shfl((int)y1, lane >= 31 ? lane - 31 : lane);
  r.d = lane < 31 ? y0 : up;
  r.last = y1;
  r.pos = 0;
  rng_chain(r, 1ull);
}

__device__ __forceinline__ void rng_refill(WaveRng& r, int lane) {  // keep >= 31 unread draws behind pos
  while (r.pos >= 31) {
    const unsigned y = rng_block(r.last, lane);
    const unsigned lo = (unsigned)__shfl((int)r.d, lane < 31 ? lane + 31 : lane);
    const unsigned hi = (unsigned)__shfl((int)y, lane >= 31 ? lane - 31 : lane);
    r.d = lane < 31 ? lo : hi;
    r.last = y;
    r.pos -= 31;
    rng_chain(r, r.chain >> 31);
  }
}

#if RNA_RRT_WIDE
// ---- the same stream with a 124-draw window (round 6) ----
// d0 holds draws 0..61 of the window (one per lane), d1 draws 62..123; `last` the newest block of 31.  Masks over the window are
// pairs of 62-bit limbs: bit i < 62 of `lo` = draw i, bit i of `hi` = draw 62 + i (hi may carry two bits more: where the chain of
// sample starts leaves the window).
struct M2 { unsigned long long lo, hi; };
constexpr unsigned long long W_LIVE = (1ull << 62) - 1ull;
__device__ __forceinline__ M2 m2(unsigned long long lo, unsigned long long hi) { M2 r; r.lo = lo; r.hi = hi; return r; }
__device__ __forceinline__ M2 m2_and(M2 a, M2 b) { return m2(a.lo & b.lo, a.hi & b.hi); }
__device__ __forceinline__ M2 m2_andn(M2 a, M2 b) { return m2(a.lo & ~b.lo, a.hi & ~b.hi); }
__device__ __forceinline__ M2 m2_or(M2 a, M2 b) { return m2(a.lo | b.lo, a.hi | b.hi); }
__device__ __forceinline__ bool m2_any(M2 a) { return (a.lo | a.hi) != 0ull; }
__device__ __forceinline__ bool m2_eq(M2 a, M2 b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ __forceinline__ int m2_pop(M2 a) { return __popcll(a.lo) + __popcll(a.hi); }
__device__ __forceinline__ M2 m2_shl(M2 a, int k) {   // k = 1 or 3; the sources are inside the window (62-bit limbs)
  return m2((a.lo << k) & W_LIVE, (a.hi << k) | (a.lo >> (62 - k)));
}
__device__ __forceinline__ M2 m2_shr31(M2 a) { return m2(((a.lo >> 31) | (a.hi << 31)) & W_LIVE, a.hi >> 31); }
__device__ __forceinline__ M2 m2_below(int i) {   // bits of the draws before draw i (0 <= i <= 126)
  return i <= 62 ? m2(i == 62 ? W_LIVE : (1ull << i) - 1ull, 0ull) : m2(W_LIVE, (1ull << (i - 62)) - 1ull);
}
__device__ __forceinline__ bool m2_bit(M2 a, int i) { return ((i < 62 ? a.lo >> i : a.hi >> (i - 62)) & 1ull) != 0ull; }
__device__ __forceinline__ M2 m2_lowest(M2 a) { return a.lo ? m2(a.lo & (0ull - a.lo), 0ull) : m2(0ull, a.hi & (0ull - a.hi)); }
__device__ __forceinline__ M2 m2_drop_lowest(M2 a) { return a.lo ? m2(a.lo & (a.lo - 1ull), a.hi) : m2(0ull, a.hi & (a.hi - 1ull)); }
__device__ __forceinline__ int m2_first(M2 a) { return a.lo ? __ffsll((long long)a.lo) - 1 : 62 + __ffsll((long long)a.hi) - 1; }   // (a != 0)
__device__ __forceinline__ int m2_last(M2 a) { return a.hi ? 62 + 63 - __clzll((long long)a.hi) : 63 - __clzll((long long)a.lo); }   // (a != 0)

struct WaveRngW { unsigned d0, d1, last; int pos; M2 three, chain; };

__device__ __forceinline__ void rngw_chain(WaveRngW& r, M2 from) {
  r.three = m2(__ballot((int)(r.d0 >> 1) % 10 > 3) & W_LIVE, __ballot((int)(r.d1 >> 1) % 10 > 3) & W_LIVE);
  M2 c = from, prev;
  do {
    prev = c;
    const M2 src = m2(c.lo & W_LIVE, c.hi & W_LIVE);
    c = m2_or(c, m2_or(m2_shl(m2_andn(src, r.three), 1), m2_shl(m2_and(src, r.three), 3)));
  } while (!m2_eq(c, prev));
  r.chain = c;
}
__device__ void rngw_seed(WaveRngW& r, unsigned seed, int lane) {
  WaveRng n;
  rng_seed(n, seed, lane);   // d = blocks 0 and 1, last = block 1
  const unsigned y2 = rng_block(n.last, lane), y3 = rng_block(y2, lane);
  const unsigned up = (unsigned)__shfl((int)y3, lane >= 31 ? lane - 31 : lane);
  r.d0 = n.d;
  r.d1 = lane < 31 ? y2 : up;
  r.last = y3;
  r.pos = 0;
  rngw_chain(r, m2(1ull, 0ull));
}
__device__ __forceinline__ void rngw_refill(WaveRngW& r, int lane) {   // keep >= 93 unread draws behind pos
  while (r.pos >= 31) {
    const unsigned y = rng_block(r.last, lane);
    const int from = lane < 31 ? lane + 31 : lane - 31;
    const unsigned a_hi = (unsigned)__shfl((int)r.d0, from), b_lo = (unsigned)__shfl((int)r.d1, from), y_lo = (unsigned)__shfl((int)y, lane >= 31 ? lane - 31 : lane);
    r.d0 = lane < 31 ? a_hi : b_lo;    // draws 31..61 of d0, then draws 0..30 of d1
    r.d1 = lane < 31 ? b_lo : y_lo;    // draws 31..61 of d1, then the new block
    r.last = y;
    r.pos -= 31;
    rngw_chain(r, m2_shr31(r.chain));
  }
}
#endif

// GlobalPlanner::ifBlocked (map_global_planner.h:39-54) through CircleIterator (CircleIterator.cpp:16-93):
// wave-cooperative, returns the same value in every lane.
__device__ bool wave_if_blocked(const Geom& g, const float* __restrict__ master, double px, double py, int lane) {
  const double radius = 0.3;
  const double r2 = radius * radius;  // pow(radius, 2)
  // The four corner coordinates (top-left x, y = p + radius; bottom-right x, y = p - radius) go through
  // limitPositionToRange and getIndexFromPosition in lanes 0..3 at once: one pass of the double-precision
  // divide instead of four.
  const int a = lane & 1;
  const double c = a ? py : px;
  const double len = g.len[a], pos = g.pos[a];
  double v = (lane & 2) ? c - radius : c + radius;
  {  // limit_position_to_range, one coordinate
    const double vto = 0.5 * len;
    double shifted = (v - pos) + vto;
    double eps = 10.0 * DBL_EPSILON;
    if (fabs(v) > 1.0) eps *= fabs(v);
    if (shifted <= 0) shifted = eps;
    else if (shifted >= len) shifted = len - eps;
    v = (shifted + pos) - vto;
  }
  const double t = -((v - pos) - 0.5 * len);
  int uc = -(int)(((v - 0.5 * len) - pos) / g.res);
  // index_from_position (gridmath.hpp), one coordinate per lane: a corner within rounding of the far edge yields
  // index == size, which wraps to 0 on a moved buffer and counts as outside the map on an unmoved one
  const bool moved = g.start[0] != 0 || g.start[1] != 0;
  if (moved) uc = wrap_index(uc, g.size[a]);
  const unsigned long long inside = __ballot(t >= 0.0 && t < len && (unsigned)uc < (unsigned)g.size[a]);
  // index_from_position leaves {0,0} (a buffer index) when the position is outside: unwrapped, that is -start
  const bool tl_ok = (inside & 3ull) == 3ull, br_ok = (inside & 12ull) == 12ull;
  const int su0 = tl_ok ? __builtin_amdgcn_readlane(uc, 0) : wrap_index(-g.start[0], g.size[0]);
  const int su1 = tl_ok ? __builtin_amdgcn_readlane(uc, 1) : wrap_index(-g.start[1], g.size[1]);
  const int tu0 = br_ok ? __builtin_amdgcn_readlane(uc, 2) : wrap_index(-g.start[0], g.size[0]);
  const int tu1 = br_ok ? __builtin_amdgcn_readlane(uc, 3) : wrap_index(-g.start[1], g.size[1]);
  // a non-positive size (a corner index that failed, see oracle/gridmath.c og_circle_cells) still visits the
  // first cell: SubmapIterator starts not-past-end and CircleIterator tests that cell before incrementing
  const bool whole = (tu0 - su0 + 1) > 0 && (tu1 - su1 + 1) > 0;
  const int ni = whole ? tu0 - su0 + 1 : 1, nj = whole ? tu1 - su1 + 1 : 1;
  bool hit = false;
  const int total = ni * nj;
  const float rcp_ni = 1.0f / (float)ni;
  const double ox = g.pos[0] + (0.5 * g.len[0] - 0.5 * g.res), oy = g.pos[1] + (0.5 * g.len[1] - 0.5 * g.res);
  // three cells per lane and trip (the 0.3 m disc at 0.05 m resolution spans <= 169 cells): the map reads of one
  // trip are issued together, so the test costs one memory round trip instead of three
  for (int k0 = lane; k0 < total; k0 += 192) {
    // (round 6: a cell's map read needs its index alone -- it leaves BEFORE the cell's centre is tested against the disc, and the
    // double-precision arithmetic of that test runs while the read is on its way instead of in front of it; a cell of the
    // bounding box outside the disc is read and not looked at)
    size_t at[3];
    bool ok[3];
    int w0s[3], w1s[3];
#pragma unroll
    for (int u3 = 0; u3 < 3; ++u3) {   // index arithmetic first ...
      const int k = k0 + 64 * u3;
      int row = (int)((float)k * rcp_ni);      // k / ni without the integer divide (k < 2^23), corrected below
      int col = k - row * ni;
      {  // (selects)
        const int fix = col < 0 ? -1 : (col >= ni ? 1 : 0);
        row += fix; col -= fix * ni;
      }
      const int u[2] = {su0 + col, su1 + row};
      int bi[2];
      buffer_index(g, u, bi);
      // position_from_index(bi) unwraps bi again: that is u itself, wrapped into the map
      w0s[u3] = (unsigned)u[0] < (unsigned)g.size[0] ? u[0] : wrap_index(u[0], g.size[0]);
      w1s[u3] = (unsigned)u[1] < (unsigned)g.size[1] ? u[1] : wrap_index(u[1], g.size[1]);
      // an index past the map (unmoved map, corner on the far edge) is read out of bounds by the reference: skipped here
      ok[u3] = (k < total) & ((unsigned)bi[0] < (unsigned)g.size[0]) & ((unsigned)bi[1] < (unsigned)g.size[1]);
      at[u3] = ok[u3] ? (size_t)bi[1] * g.size[0] + bi[0] : 0;
    }
    float val[3];
#pragma unroll
    for (int u3 = 0; u3 < 3; ++u3) val[u3] = master[at[u3]];   // ... then the three reads together (cell 0 for lanes without one)
#pragma unroll
    for (int u3 = 0; u3 < 3; ++u3) {   // ... and the disc test while they travel
      const double x = ox + g.res * (double)(-w0s[u3]);
      const double y = oy + g.res * (double)(-w1s[u3]);
      const double dx = x - px, dy = y - py;
      ok[u3] = ok[u3] & (dx * dx + dy * dy <= r2);
    }
#pragma unroll
    for (int u3 = 0; u3 < 3; ++u3)
      if (ok[u3] && !(val[u3] != val[u3]) && val[u3] > 0.0f) hit = true;
  }
  return __ballot(hit) != 0ULL;
}

// Wave-wide minimum of (non-negative double, index) pairs -- smallest value, lowest index among equals --
// with DPP row shifts / row broadcasts instead of six rounds of ds_bpermute shuffles (which were two
// thirds of the nearest-node time).  All 64 lanes must be active; every lane receives the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_min_step(unsigned& lo, unsigned& hi, int& idx) {
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xF, false);
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xF, false);
  const int oidx = __builtin_amdgcn_update_dpp(idx, idx, CTRL, ROW_MASK, 0xF, false);
  const unsigned long long a = ((unsigned long long)ohi << 32) | olo, b = ((unsigned long long)hi << 32) | lo;
  const bool take = (a < b) | ((a == b) & (oidx < idx));   // (selects: as an `if` each of the six steps kept an exec-masked branch)
  lo = take ? olo : lo; hi = take ? ohi : hi; idx = take ? oidx : idx;
}
__device__ __forceinline__ void wave_min_pair(double& v, int& idx) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);   // v >= 0: bit order == value order
  unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
  dpp_min_step<0x111, 0xF>(lo, hi, idx);   // row_shr:1
  dpp_min_step<0x112, 0xF>(lo, hi, idx);   // row_shr:2
  dpp_min_step<0x114, 0xF>(lo, hi, idx);   // row_shr:4
  dpp_min_step<0x118, 0xF>(lo, hi, idx);   // row_shr:8  -> lane 15 of every row holds the row minimum
  dpp_min_step<0x142, 0xA>(lo, hi, idx);   // row_bcast:15 into rows 1 and 3
  dpp_min_step<0x143, 0xC>(lo, hi, idx);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
  lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
  hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
  idx = __builtin_amdgcn_readlane(idx, 63);
  v = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#ifdef RNA_RRT_STATS
__device__ unsigned long long g_rrt_stat[8];
#define RRT_T(v) const unsigned long long v = wall_clock64()
#define RRT_ACC(slot, a, b) rrt_acc[slot] += (b) - (a)
#else
#define RRT_T(v)
#define RRT_ACC(slot, a, b)
#endif

// One sample against the current tree: findNearNode, the steering step of extendTree, ifBlocked at the new point.  Wave-cooperative
// (all 64 lanes), the same values in every lane.
struct RrtEval { double wx, wy; int near; bool blocked; };
#ifdef RNA_RRT_STATS
#define RRT_STAT_PARAM , unsigned long long* rrt_acc, unsigned long long t1
#define RRT_STAT_ARG , rrt_acc, t1
#else
#define RRT_STAT_PARAM
#define RRT_STAT_ARG
#endif
__device__ __forceinline__ RrtEval rrt_evaluate(const Geom& g, const float* __restrict__ master, const double2* tn, int n_tree,
                                                double rx, double ry, int lane RRT_STAT_PARAM) {
  const double strideStep = 0.4;   // rrt_planner.h:23
  // findNearNode, rrt_planner.cpp:70-89 : strict < on hypot() keeps the lowest index among equals.
  // hypot costs ~10x a squared distance, so the scan runs on squared distances first; hypot (glibc's, error
  // < 1 ulp, i.e. 2^-52 relative) can only reorder nodes whose squared distances agree to within
  // 2^-48, and exactly those are re-examined with glibc_hypot in the reference's order.
  // One pass: every lane keeps its nearest node by squared distance (lowest index among equals) and
  // its second-smallest squared distance.  After the wave reduction, hypot is evaluated for at most one
  // node per lane; only if some lane holds TWO nodes inside the 2^-46 band (practically never) the
  // band is rescanned in full.
  double l1 = 1.0e300, l2 = 1.0e300;
  int i1 = 0x7fffffff;
  for (int i0 = lane; i0 < n_tree; i0 += 256) {   // four nodes per lane and trip: the LDS reads go out together
    double2 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = tn[min(i0 + 64 * u, RRT_ITER - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 64 * u;
      const double dx = rx - t[u].x, dy = ry - t[u].y;
      const double d2 = i < n_tree ? dx * dx + dy * dy : 1.0e300;
      // (selects, not an if / else-if: four exec-masked branch pairs per trip of the scan otherwise)
      const bool nearer = d2 < l1, second = d2 < l2;
      l2 = nearer ? l1 : (second ? d2 : l2);
      i1 = nearer ? i : i1;
      l1 = nearer ? d2 : l1;
    }
  }
  double m2 = l1;
  int imin = i1;
  wave_min_pair(m2, imin);                       // nearest by squared distance, lowest index among equals
  const double band = m2 + m2 * 0x1p-46 + 1.0e-300;
  double best = 9999.0;
  int best_i = 0x7fffffff;
  if (__ballot(l2 <= band || (l1 <= band && i1 != imin)) == 0ull) {
    // the usual case: no other node within the band, the answer is imin (if it beats the reference's 9999 start)
    // hypot < 9999 needs no hypot when the squared distance is far below 9999^2
    if (imin != 0x7fffffff) {
      const double2 t = tn[imin];
      if (m2 < 9.0e7 || glibc_hypot(rx - t.x, ry - t.y) < best) best_i = imin;
    }
  } else {
    for (int i = lane; i < n_tree; i += 64) {
      const double2 t = tn[i];
      const double dx = rx - t.x, dy = ry - t.y;
      if (dx * dx + dy * dy <= band) {
        const double d = glibc_hypot(dx, dy);
        if (d < best) { best = d; best_i = i; }
      }
    }
    wave_min_pair(best, best_i);
  }
  const int near = (best_i == 0x7fffffff) ? 0 : best_i;
  RRT_T(t2);
  RRT_ACC(1, t1, t2);
  const double npx = tn[near].x, npy = tn[near].y;
  double wx, wy;
  // hypot < 0.4 is decided by the squared distance except within rounding of 0.16
  const double ddx = npx - rx, ddy = npy - ry, dd2 = ddx * ddx + ddy * ddy;
  const bool within = dd2 < 0.1599 ? true : (dd2 > 0.1601 ? false : glibc_hypot(ddx, ddy) < strideStep);
  if (within) { wx = rx; wy = ry; }
  else {
#ifdef RNA_RRT_LIBM_STEER
    const double a = atan2(ry - npy, rx - npx);   // as written in the reference (rrt_planner.cpp:45-47), with the device's libm
    wx = npx + strideStep * cos(a);
    wy = npy + strideStep * sin(a);
#else
    // cos(atan2(dy, dx)) = dx / hypot(dx, dy): one square root and two divisions, all IEEE-rounded, instead of three
    // double-precision libm calls -- which are what held the kernel at 127 VGPRs (78 without them: twelve
    // wavefronts per query instead of eight fit the chip).  Against glibc's atan2 / cos / sin the offset differs in
    // its last bits either way (the device libm's does too, DESIGN.md 4 "What stays open"): the sum with the node's
    // coordinate then rounds differently for a few per cent of the nodes, by one ulp, which only matters at the
    // last-bit ties scripts/fuzz_rrt.py classifies.
    const double sdx = rx - npx, sdy = ry - npy;
    const double sh = sqrt(sdx * sdx + sdy * sdy);
    wx = npx + strideStep * (sdx / sh);
    wy = npy + strideStep * (sdy / sh);
#endif
  }
  RRT_T(t3);
  RRT_ACC(2, t2, t3);
  const bool blocked = wave_if_blocked(g, master, wx, wy, lane);
  RRT_T(t4);
  RRT_ACC(3, t3, t4);
  return RrtEval{wx, wy, near, blocked};
}

// One query per workgroup, RRT_SPEC wavefronts.  extendTree (rrt_planner.cpp:26-59) draws samples until one is
// not blocked; the draws do not depend on the outcomes and the tree only changes when a sample is accepted, so
// the next RRT_SPEC samples are evaluated against the current tree at once, one per wavefront, and the first
// unblocked one IN DRAW ORDER is accepted -- the later ones are discarded and their draws re-used.  Every
// wavefront carries the rand() stream and takes the same decisions, so one barrier per round is all the
// synchronisation there is.  (On the bench maps 94 % of the samples are blocked; with the goal samples answered
// from the last evaluation, see below, a round consumes 8.7 samples on average and 13 on a stuck tree.)
// Round 6 also ran up to four rounds back to back before ONE barrier on trees that had stopped accepting (the later rounds' samples
// are known once the earlier ones are assumed blocked; on an accept the stream is put back).  Exact -- the GPU tests and 8 704
// fuzzed queries -- and slower, 46.7 ms against 41.4 for config 4's share: what the timers call "barrier wait" is not the cost of
// the barrier but the spread between the wavefronts of a round, and it adds up over the rounds of a group all the same
// (0.82 us a round, 3.24 us a group of four: profiles/r06_rrt_rounds_ahead.txt).  Not kept.
struct RrtSlot { double wx, wy; int near, blocked; };

// 4 wavefronts per SIMD (<= 128 VGPRs): two 8-wavefront workgroups per CU, 512 queries resident on the chip
#ifndef RNA_RRT_WAVES_PER_EU
#define RNA_RRT_WAVES_PER_EU (RNA_RRT_WIDE ? 6 : 4)   // (wide: two 12-wavefront workgroups per CU need six wavefronts per SIMD, i.e. <= 80 VGPRs)
#endif
__global__ void __launch_bounds__(64 * RRT_SPEC) __attribute__((amdgpu_waves_per_eu(RNA_RRT_WAVES_PER_EU, RNA_RRT_WAVES_PER_EU)))
rrt_kernel(Geom g, const float* __restrict__ master, const rna_rrt_query* __restrict__ queries, int n,
           int* __restrict__ tree_parent,
           double* __restrict__ paths, int max_path_len, rna_rrt_result* __restrict__ results) {
  __shared__ double2 tn[RRT_ITER];          // node positions: 32 KB, one 16-byte LDS read per node
  __shared__ RrtSlot slot[2][RRT_SPEC];     // per-round results, double-buffered by round parity
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int q = blockIdx.x;
  const rna_rrt_query qu = queries[q];
  int* tp = tree_parent + (size_t)q * RRT_ITER;
#if RNA_RRT_WIDE
  WaveRngW rs;
  rngw_seed(rs, qu.seed, lane);
#else
  WaveRng rs;
  rng_seed(rs, qu.seed, lane);
#endif

  // (strideStep 0.4, rrt_planner.h:23, lives in rrt_evaluate; targetTendency_ is int(0.5) == 0, rrt_planner.h:24,32)
  const bool target_inside = position_within_map(g, qu.target[0], qu.target[1]);
  const double pbx = g.pos[0] + g.len[0] / 2, pby = g.pos[1] + g.len[1] / 2;
  const double mbx = g.pos[0] - g.len[0] / 2, mby = g.pos[1] - g.len[1] / 2;

#ifdef RNA_RRT_STATS
  unsigned long long rrt_acc[8] = {};
#endif
  double nx = qu.start[0], ny = qu.start[1];
  int nparent = -1;
  int n_tree = 0, samples = 0, goal_blocked_at = -1;
  unsigned round = 0;
  bool finished = false, aborted = false;

  for (int it = 0; it < RRT_ITER && !aborted; ++it) {
    // every wavefront appends the node itself (same values): its own scan sees it without a barrier
    if (lane == 0) { tn[n_tree] = make_double2(nx, ny); if (wave == 0) tp[n_tree] = nparent; }
    n_tree++;
    __builtin_amdgcn_wave_barrier();
    bool fin;  // ifFinishPlan, map_global_planner.h:32-37,56-86
    if (target_inside) fin = glibc_hypot(nx - qu.target[0], ny - qu.target[1]) < qu.close_tolerance;
    else fin = ((pbx - nx) < qu.close_tolerance) || ((pby - ny) < qu.close_tolerance) ||
               ((nx - mbx) < qu.close_tolerance) || ((ny - mby) < qu.close_tolerance);
    if (fin) { finished = true; break; }

    for (;;) {  // extendTree, rrt_planner.cpp:26-59, RRT_SPEC samples per round
      if (samples >= qu.max_samples) { aborted = true; break; }
      const int valid = qu.max_samples - samples;
      RRT_T(t0);
#if RNA_RRT_WIDE
      rngw_refill(rs, lane);
#else
      rng_refill(rs, lane);
#endif
      // sample(), rrt_planner.cpp:61-68: one draw decides goal or random, a random sample takes two more.
      // A goal sample's outcome depends on the tree alone, and an unblocked one is accepted on the spot, so
      // once one has been found blocked every later goal sample is known blocked until the tree grows: those
      // take no wavefront.  The round therefore covers samples in draw order until it holds RRT_SPEC that need
      // evaluating (or the 62-draw window ends) -- on the bench maps 40 % of the samples are goal samples.
#if RNA_RRT_WIDE
      // (the same bookkeeping over the 124-draw window: masks are pairs of 62-bit limbs, see M2)
      const int res0 = (int)(rs.d0 >> 1), res1 = (int)(rs.d1 >> 1);
      const M2 three = rs.three;
      const bool goal_known = goal_blocked_at == n_tree;
      // sample starts from pos on whose draws lie inside the window (a random sample needs three draws: it starts at 121 at the latest)
      M2 startmask = m2_and(m2_andn(m2(rs.chain.lo & W_LIVE, rs.chain.hi & W_LIVE), m2_below(rs.pos)), m2(W_LIVE, (W_LIVE & ~three.hi) | (three.hi & (W_LIVE >> 2))));
      const M2 goals = m2_andn(startmask, three);
      M2 slotmask = m2_and(startmask, three);
      if (!goal_known) slotmask = m2_or(slotmask, m2_lowest(goals));
      M2 over = slotmask;                          // samples past the RRT_SPEC-th one that needs a wavefront wait
#pragma unroll
      for (int k = 0; k < RRT_SPEC; ++k) over = m2_drop_lowest(over);
      if (m2_any(over)) startmask = m2_and(startmask, m2_below(m2_first(over)));
      if (m2_pop(startmask) > valid) {             // ... and so do samples past the budget
        M2 m = startmask;
        for (int k = 0; k < valid; ++k) m = m2_drop_lowest(m);
        startmask = m2_and(startmask, m2_below(m2_first(m)));
      }
      slotmask = m2_and(slotmask, startmask);
      const int goal_lane = (!goal_known && m2_any(m2_and(goals, startmask))) ? m2_first(goals) : -1;
      const int nslot = m2_pop(slotmask);
      const int last_start = m2_last(startmask);
      const int p_end = last_start + (m2_bit(three, last_start) ? 3 : 1);
#define RRT_IS_RANDOM(i) m2_bit(three, (i))
#define RRT_DRAW(i) ((i) < 62 ? __builtin_amdgcn_readlane(res0, (i)) : __builtin_amdgcn_readlane(res1, (i) - 62))
#else
      const int res = (int)(rs.d >> 1);
      const unsigned long long three = rs.three;
      const bool goal_known = goal_blocked_at == n_tree;
      // sample starts from pos on whose draws lie inside the window (a random sample needs three lanes)
      unsigned long long startmask = rs.chain & ~((1ull << rs.pos) - 1ull) & ((RNG_LIVE & ~three) | (three & (RNG_LIVE >> 2)));
      const unsigned long long goals = startmask & ~three;
      unsigned long long slotmask = (startmask & three) | (goal_known ? 0ull : (goals & (0ull - goals)));
      unsigned long long over = slotmask;          // samples past the RRT_SPEC-th one that needs a wavefront wait
#pragma unroll
      for (int k = 0; k < RRT_SPEC; ++k) over &= over - 1ull;
      if (over) startmask &= (over & (0ull - over)) - 1ull;
      if (__popcll(startmask) > valid) {           // ... and so do samples past the budget
        unsigned long long m = startmask;
        for (int k = 0; k < valid; ++k) m &= m - 1ull;
        startmask &= (m & (0ull - m)) - 1ull;
      }
      slotmask &= startmask;
      const int goal_lane = (!goal_known && (goals & startmask)) ? __ffsll((long long)goals) - 1 : -1;
      const int nslot = __popcll(slotmask);
      const int last_start = 63 - __clzll((long long)startmask);
      const int p_end = last_start + (((three >> last_start) & 1ull) ? 3 : 1);
#define RRT_IS_RANDOM(i) (((three >> (i)) & 1ull) != 0ull)
#define RRT_DRAW(i) __builtin_amdgcn_readlane(res, (i))
#endif
      RrtSlot* const out = slot[round & 1u];
      if (wave < nslot) {
#if RNA_RRT_WIDE
        M2 m = slotmask;
        for (int k = 0; k < wave; ++k) m = m2_drop_lowest(m);
        const int mine = __builtin_amdgcn_readfirstlane(m2_first(m));
#else
        unsigned long long m = slotmask;
        for (int k = 0; k < wave; ++k) m &= m - 1;
        const int mine = __ffsll((long long)m) - 1;
#endif
        double rx, ry;
        if (RRT_IS_RANDOM(mine)) {
          const int r1 = RRT_DRAW(mine + 1), r2 = RRT_DRAW(mine + 2);
          const int ridx[2] = {r1 % g.size[0], r2 % g.size[1]};
          double pp[2];
          position_from_index(g, ridx, pp);
          rx = pp[0]; ry = pp[1];
        } else {
          rx = qu.target[0]; ry = qu.target[1];
        }
        RRT_T(t1);
        RRT_ACC(0, t0, t1);
        const RrtEval ev = rrt_evaluate(g, master, tn, n_tree, rx, ry, lane RRT_STAT_ARG);
        const double wx = ev.wx, wy = ev.wy;
        const int near = ev.near;
        const bool blocked = ev.blocked;
        if (lane == 0) out[wave] = RrtSlot{wx, wy, near, blocked ? 1 : 0};
#if defined(RNA_RRT_DEBUG_SAMPLE) && !RNA_RRT_WIDE
        {
          const int sno = samples + __popcll(startmask & ((2ull << mine) - 1ull));
          if (lane == 0 && sno == RNA_RRT_DEBUG_SAMPLE)
            printf("[rrt dbg] q %d sample %d rnd (%.17g, %.17g) near %d n_tree %d new (%.17g, %.17g) blocked %d\n",
                   q, sno, rx, ry, near, n_tree, wx, wy, (int)blocked);
        }
#endif
      }
      RRT_T(t5);
      __syncthreads();
      RRT_T(t6);
      RRT_ACC(5, t5, t6);
      ++round;
      // outcome per sample, one sample start per lane: a random sample reads its slot, a goal sample the slot of
      // the round's evaluated goal sample (or is known blocked); the first open one in draw order is accepted
#if RNA_RRT_WIDE
      // outcome per sample: lane l speaks for the starts at draw l and at draw 62 + l
      const int goal_slot = goal_lane >= 0 ? m2_pop(m2_and(slotmask, m2_below(goal_lane))) : 0;
      bool open_a = false, open_b = false;
      {
        const int ia = lane, ib = 62 + lane;
        const bool st_a = lane < 62 && m2_bit(startmask, ia), st_b = lane < 62 && m2_bit(startmask, ib);
        const bool rn_a = m2_bit(three, ia), rn_b = lane < 62 && m2_bit(three, ib);
        if (st_a && (rn_a || goal_lane >= 0)) open_a = out[rn_a ? m2_pop(m2_and(slotmask, m2_below(ia))) : goal_slot].blocked == 0;
        if (st_b && (rn_b || goal_lane >= 0)) open_b = out[rn_b ? m2_pop(m2_and(slotmask, m2_below(ib))) : goal_slot].blocked == 0;
      }
      const M2 open = m2(__ballot(open_a), __ballot(open_b));
      if (m2_any(open)) {
        const int wl = m2_first(open);
        const bool wr = m2_bit(three, wl);
        samples += m2_pop(m2_and(startmask, m2_below(wl + 1)));
        rs.pos = wl + (wr ? 3 : 1);
        const RrtSlot w = out[wr ? m2_pop(m2_and(slotmask, m2_below(wl))) : goal_slot];
        nx = w.wx; ny = w.wy; nparent = w.near;
        break;
      }
      samples += m2_pop(startmask);
      rs.pos = p_end;
      if (goal_lane >= 0) goal_blocked_at = n_tree;
#else
      const unsigned long long below = (1ull << lane) - 1ull;
      const int goal_slot = goal_lane >= 0 ? __popcll(slotmask & ((1ull << goal_lane) - 1ull)) : 0;
      const bool is_start = (startmask >> lane) & 1ull, is_random = (three >> lane) & 1ull;
      bool is_open = false;
      if (is_start && (is_random || goal_lane >= 0)) is_open = out[is_random ? __popcll(slotmask & below) : goal_slot].blocked == 0;
      const unsigned long long open = __ballot(is_open);
      if (open) {
        const int wl = __ffsll((long long)open) - 1;
        const bool wr = (three >> wl) & 1ull;
        samples += __popcll(startmask & ((2ull << wl) - 1ull));
        rs.pos = wl + (wr ? 3 : 1);
        const RrtSlot w = out[wr ? __popcll(slotmask & ((1ull << wl) - 1ull)) : goal_slot];
        nx = w.wx; ny = w.wy; nparent = w.near;
#ifdef RNA_RRT_DEBUG_SAMPLE
        if (wave == 0 && lane == 0) printf("[rrt acc] q %d sample %d node %d near %d new (%.17g, %.17g)\n", q, samples, n_tree, nparent, nx, ny);
#endif
        break;
      }
      samples += __popcll(startmask);
      rs.pos = p_end;
      if (goal_lane >= 0) goal_blocked_at = n_tree;
#endif
#undef RRT_IS_RANDOM
#undef RRT_DRAW
    }
  }

  // backtraceTree, rrt_planner.cpp:91-104 : goal -> start
  int len = 0;
  if (wave == 0) {
    __threadfence_block();
    if (!aborted && n_tree > 0) {
      double* out = paths + (size_t)q * max_path_len * 2;
      int i = n_tree - 1;
      for (;;) {
        if (lane == 0 && len < max_path_len) { out[2 * len] = tn[i].x; out[2 * len + 1] = tn[i].y; }
        len++;
        const int par = tp[i];
        if (par == -1) break;
        i = par;
      }
    }
    if (lane == 0) results[q] = rna_rrt_result{aborted ? -1 : (finished ? 1 : 0), len, n_tree, samples};
#ifdef RNA_RRT_STATS
    if (lane == 0) { for (int k = 0; k < 6; ++k) atomicAdd(&g_rrt_stat[k], rrt_acc[k]); atomicAdd(&g_rrt_stat[4], (unsigned long long)samples); atomicAdd(&g_rrt_stat[6], (unsigned long long)round); }
#endif
  }
}

}  // namespace

extern "C" int rna_graph_astar_batch(rna_engine* e, int nv, const double* vertex_xy, int ne, const int32_t* edge_uv,
                                     const float* edge_weight, const double* start_target_xy, int n, double* paths_xy,
                                     int max_len, int32_t* path_len) {
  if (!e || nv <= 0 || nv > GRAPH_MAX_V || ne < 0 || !vertex_xy || (ne > 0 && !edge_uv) || n < 0 || max_len < 3 ||
      (n > 0 && (!start_target_xy || !paths_xy || !path_len)))
    return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  for (int k = 0; k < 2 * ne; ++k)
    if (edge_uv[k] < 0 || edge_uv[k] >= nv) return RNA_EINVAL;
  RNA_ENTER(e);
  // out-edge lists in add_edge order: add_edge(u,v) appends to u's list and to v's list
  std::vector<int> off(nv + 1, 0), fill(nv, 0), adj_v(2 * ne + 1);
  std::vector<float> adj_w(2 * ne + 1);
  for (int k = 0; k < ne; ++k) { off[edge_uv[2 * k] + 1]++; off[edge_uv[2 * k + 1] + 1]++; }
  for (int v = 0; v < nv; ++v) off[v + 1] += off[v];
  for (int k = 0; k < ne; ++k) {
    const int u = edge_uv[2 * k], v = edge_uv[2 * k + 1];
    const float w = edge_weight ? edge_weight[k] : 0.0f;  // add_edge without a property: weight 0 (astar_planner.cpp:116-125)
    adj_v[off[u] + fill[u]] = v; adj_w[off[u] + fill[u]++] = w;
    adj_v[off[v] + fill[v]] = u; adj_w[off[v] + fill[v]++] = w;
  }
  double *d_loc = nullptr, *d_st = nullptr, *d_paths = nullptr;
  int *d_off = nullptr, *d_adjv = nullptr, *d_si = nullptr, *d_len = nullptr;
  float *d_adjw = nullptr, *d_sf = nullptr;
  int rc = RNA_OK;
  auto cleanup = [&]() {
    dev_free(&d_loc); dev_free(&d_st); dev_free(&d_paths); dev_free(&d_off); dev_free(&d_adjv); dev_free(&d_si);
    dev_free(&d_len); dev_free(&d_adjw); dev_free(&d_sf);
  };
#define GA_TRY(x) do { rc = (x); if (rc != RNA_OK) { cleanup(); return rc; } } while (0)
#define GA_HIP(x) do { if ((x) != hipSuccess) { cleanup(); return fail(e, RNA_EHIP, #x); } } while (0)
  GA_TRY(dev_alloc(e, &d_loc, (size_t)2 * nv));
  GA_TRY(dev_alloc(e, &d_st, (size_t)4 * n));
  GA_TRY(dev_alloc(e, &d_paths, (size_t)2 * max_len * n));
  GA_TRY(dev_alloc(e, &d_off, (size_t)nv + 1));
  GA_TRY(dev_alloc(e, &d_adjv, adj_v.size()));
  GA_TRY(dev_alloc(e, &d_adjw, adj_w.size()));
  GA_TRY(dev_alloc(e, &d_sf, (size_t)2 * nv * n));
  GA_TRY(dev_alloc(e, &d_si, (size_t)4 * nv * n));
  GA_TRY(dev_alloc(e, &d_len, (size_t)n));
  GA_HIP(hipMemcpyAsync(d_loc, vertex_xy, sizeof(double) * 2 * nv, hipMemcpyHostToDevice, e->stream));
  GA_HIP(hipMemcpyAsync(d_st, start_target_xy, sizeof(double) * 4 * n, hipMemcpyHostToDevice, e->stream));
  GA_HIP(hipMemcpyAsync(d_off, off.data(), sizeof(int) * (nv + 1), hipMemcpyHostToDevice, e->stream));
  GA_HIP(hipMemcpyAsync(d_adjv, adj_v.data(), sizeof(int) * adj_v.size(), hipMemcpyHostToDevice, e->stream));
  GA_HIP(hipMemcpyAsync(d_adjw, adj_w.data(), sizeof(float) * adj_w.size(), hipMemcpyHostToDevice, e->stream));
  GraphK G{nv, ne, d_loc, d_off, d_adjv, d_adjw};
  hipLaunchKernelGGL(graph_astar_kernel, dim3((n + 63) / 64), dim3(64), 0, e->stream, G, d_st, n, d_sf, d_si,
                     d_paths, max_len, d_len);
  GA_HIP(hipGetLastError());
  GA_HIP(hipMemcpyAsync(paths_xy, d_paths, sizeof(double) * 2 * max_len * n, hipMemcpyDeviceToHost, e->stream));
  GA_HIP(hipMemcpyAsync(path_len, d_len, sizeof(int) * n, hipMemcpyDeviceToHost, e->stream));
  GA_HIP(hipStreamSynchronize(e->stream));
#undef GA_TRY
#undef GA_HIP
  cleanup();
  return RNA_OK;
}

static int rrt_launch(rna_engine* e, const rna_rrt_query* q_dev, int n, double* paths_dev, int max_len,
                      rna_rrt_result* res_dev, double** tx, double** ty, int** tp) {
  int rc;
  (void)tx; (void)ty;   // node positions live in LDS; only the parent links go through HBM
  if ((rc = dev_alloc(e, tp, (size_t)n * RRT_ITER)) != RNA_OK) return rc;
  KernelTimer kt(e, RNA_K_RRT);
  hipLaunchKernelGGL(rrt_kernel, dim3(n), dim3(64 * RRT_SPEC), 0, e->stream, e->geom,
                     e->layer[RNA_LAYER_MASTER], q_dev, n, *tp, paths_dev, max_len, res_dev);
  RNA_HIP(e, hipGetLastError());
#ifdef RNA_RRT_STATS
  {
    RNA_HIP(e, hipStreamSynchronize(e->stream));
    unsigned long long st[8];
    RNA_HIP(e, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_rrt_stat), sizeof(st)));
    static const unsigned long long zero[8] = {};
    RNA_HIP(e, hipMemcpyToSymbol(HIP_SYMBOL(g_rrt_stat), zero, sizeof(zero)));
    const double ns = (double)st[4];
    const double nr = (double)st[6];
    fprintf(stderr, "[rrt stats] samples %.0f rounds %.0f (%.2f samples per round) | wave 0, us per round: rng+sample %.2f  nearest %.2f  "
            "steer %.2f  ifBlocked %.2f  barrier wait %.2f\n", ns, nr, ns / nr,
            st[0] * 0.01 / nr, st[1] * 0.01 / nr, st[2] * 0.01 / nr, st[3] * 0.01 / nr, st[5] * 0.01 / nr);
  }
#endif
  return RNA_OK;
}

extern "C" int rna_rrt_batch_device(rna_engine* e, const rna_rrt_query* queries, int n, double* paths_xy,
                                    int max_path_len, rna_rrt_result* results) {
  if (!e || n < 0 || max_path_len <= 0 || (n > 0 && (!queries || !paths_xy || !results))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_ENTER(e);
  double *tx = nullptr, *ty = nullptr;
  int* tp = nullptr;
  int rc = rrt_launch(e, queries, n, paths_xy, max_path_len, results, &tx, &ty, &tp);
  if (rc == RNA_OK && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail(e, RNA_EHIP, "rrt sync");
  dev_free(&tx); dev_free(&ty); dev_free(&tp);
  return rc;
}

extern "C" int rna_rrt_batch(rna_engine* e, const rna_rrt_query* queries_host, int n, double* paths_xy_host,
                             int max_path_len, rna_rrt_result* results_host) {
  if (!e || n < 0 || max_path_len <= 0 || (n > 0 && (!queries_host || !paths_xy_host || !results_host)))
    return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_ENTER(e);
  rna_rrt_query* dq = nullptr;
  rna_rrt_result* dr = nullptr;
  double *dp = nullptr, *tx = nullptr, *ty = nullptr;
  int* tp = nullptr;
  int rc = RNA_OK;
  auto cleanup = [&]() { dev_free(&dq); dev_free(&dr); dev_free(&dp); dev_free(&tx); dev_free(&ty); dev_free(&tp); };
  if ((rc = dev_alloc(e, &dq, (size_t)n)) != RNA_OK || (rc = dev_alloc(e, &dr, (size_t)n)) != RNA_OK ||
      (rc = dev_alloc(e, &dp, (size_t)n * max_path_len * 2)) != RNA_OK) { cleanup(); return rc; }
  if (hipMemcpyAsync(dq, queries_host, sizeof(rna_rrt_query) * n, hipMemcpyHostToDevice, e->stream) != hipSuccess) {
    cleanup(); return fail(e, RNA_EHIP, "rrt H2D");
  }
  rc = rrt_launch(e, dq, n, dp, max_path_len, dr, &tx, &ty, &tp);
  if (rc == RNA_OK) {
    if (hipMemcpyAsync(results_host, dr, sizeof(rna_rrt_result) * n, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipMemcpyAsync(paths_xy_host, dp, sizeof(double) * 2 * max_path_len * n, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
      rc = fail(e, RNA_EHIP, "rrt D2H");
  }
  cleanup();
  return rc;
}
