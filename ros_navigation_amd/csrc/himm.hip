// himm.hip -- order-faithful HIMM ray batches on gfx950.
//
// Replaces MapUpdater::lineOnMap / clearCell / markCell applied to every buffered RangeSample in
// arrival order (mc/include/move_control/map_updater.h:38-71, mc/src/laser_map_updater.cpp:7-21,
// mc/src/range_map_updater.cpp:7-21) with LineIterator's clipping walk and integer Bresenham
// (gmc/src/iterators/LineIterator.cpp:60-70,92-150).
//
// clearCell and markCell do not commute, so a plain atomic scatter is not parity-safe.  Design:
//   * clears commute with each other (they are the same function applied k times), so every cell
//     that receives NO mark in this batch takes its clears as order-free compare-and-swap updates;
//   * cells that hold >= 1 mark (ray end points; at most n of them) are found through a 1-bit-per-
//     cell bitmap + an open-addressing hash; their clears are only COUNTED, per interval between
//     consecutive marks of that cell (marks sorted by ray sequence number), and one thread per
//     marked cell then replays  clear^k0 mark clear^k1 mark ... clear^kn  exactly.
// The result equals the sequential reference for any float contents of the layer.
//
// Kernels: himm_prep (1 thread/ray: double-precision clipping, mark registration),
//          himm_collect (1 thread/hash slot: gather + sort that cell's marks),
//          himm_bin_count / himm_bin_scan / himm_bin_fill (rays -> the 64 x 64 tiles they cross),
//          himm_tile_raster (one workgroup per touched tile: its rays' clears are COUNTED per cell in LDS --
//              k clears are clear^k, whatever their order -- then every touched line of the layer is read once,
//              coalesced, and written once; clears of marked cells go to the interval counters instead),
//          himm_apply (1 thread/hash slot: ordered replay on marked cells).
// Round 5 ran the whole chain (and the compose) as the phases of ONE persistent kernel with device-wide barriers: exact, and
// slower in the replan loop (95-130 k against 146 k cycles/s, profiles/r05_fused_map_update_*.txt) -- next to the searches every
// phase took as long as the kernel it replaced (their time there is memory latency under load, not launch overhead), the compose
// phase serialised the update with the side work, and a barrier whose workgroups cannot all be resident can only trap.  Removed;
// the kernels' bodies stayed the device functions it was built from.
// Round 1 rasterised ray by ray with a 4-byte compare-and-swap per cell straight on HBM: 295 MB of traffic for the
// 61 MB the batch needs (rays from one origin re-touch the same lines hundreds of times).  Per tile it is at most
// one read and one write of 16 KB.
#include "engine.hpp"

#include <algorithm>

using namespace rna;

namespace {


__device__ __forceinline__ float himm_clear(float v) {  // map_updater.h:61-71
  if (v <= 0.0f || v != v) v = 0.0f;
  else v = v - 10.0f;
  if (v < 0.0f) v = 0.0f;
  return v;
}

__device__ __forceinline__ float himm_mark(float v) {  // map_updater.h:52-59
  if (v <= 0.0f || v != v) return 30.0f;
  if (v <= 150.0f) return v + 30.0f;
  return v;
}

__device__ __forceinline__ float himm_clear_n(float v, unsigned k) {
  for (unsigned i = 0; i < k; ++i) {
    v = himm_clear(v);
    if (__float_as_int(v) == 0) break;  // +0.0 is a fixed point of clearCell
  }
  return v;
}

__device__ __forceinline__ unsigned hash_cell(unsigned c) {
  c ^= c >> 16; c *= 0x7feb352dU; c ^= c >> 15; c *= 0x846ca68bU; c ^= c >> 16;
  return c;
}

// A ray the reference could not digest either: a non-finite coordinate makes its clipping march
// (LineIterator.cpp:92-104) spin forever, a start millions of cells away makes it spin for minutes.  Defined here
// and in oracle/himm.c: such a ray is dropped whole (no clears, no mark).  HIMM_MAX_RAY_CELLS cells is 52 km at
// the reference's 0.05 m resolution; the march below needs at most length / step + 1 iterations.
constexpr double HIMM_MAX_RAY_CELLS = 1048576.0;
__device__ __forceinline__ bool ray_well_formed(const Geom& g, const rna_ray& r) {
  if (!(isfinite(r.sx) && isfinite(r.sy) && isfinite(r.ex) && isfinite(r.ey))) return false;
  const double vx = r.ex - r.sx, vy = r.ey - r.sy;
  return sqrt(vx * vx + vy * vy) <= HIMM_MAX_RAY_CELLS * g.res;
}

// LineIterator::getIndexLimitedToMapRange (LineIterator.cpp:92-104): march `start` towards `end`
// in (res - eps) steps until it is inside the map.
__device__ bool index_limited_to_map(const Geom& g, double sx, double sy, double ex, double ey, int idx[2]) {
  double px = sx, py = sy;
  const double vx = ex - sx, vy = ey - sy;
  const double nrm = sqrt(vx * vx + vy * vy);
  const double dx = vx / nrm, dy = vy / nrm;
  const double step = g.res - DBL_EPSILON;
  while (!index_from_position(g, px, py, idx)) {
    if (!(nrm > 0.0)) return false;  // zero-length ray outside the map (reference: undefined)
    px += step * dx;
    py += step * dy;
    const double rx = ex - px, ry = ey - py;
    if (sqrt(rx * rx + ry * ry) < step) return false;
  }
  return true;
}

__global__ void himm_init_slots_kernel(HimmSlot* __restrict__ slots, int n_slots, int* __restrict__ total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_slots) slots[i] = HimmSlot{-1, -1, 0, 0};
  if (i == 0) *total = 0;
}

// one ray: clipping, cell count, mark registration; returns the cell count and leaves the clipped ends in `d`
__device__ __forceinline__ int himm_prep_ray(const Geom& g, const rna_ray* __restrict__ rays, int r, int4* __restrict__ desc,
                                             int* __restrict__ ncells, int* __restrict__ next, HimmSlot* __restrict__ slots,
                                             int slot_mask, unsigned* __restrict__ mark_bitmap, int4 win, int4& d) {
  const rna_ray ray = rays[r];
  int s[2], t[2];
  int nc = 0;
  d = make_int4(0, 0, 0, 0);
  if (!ray_well_formed(g, ray)) {
    ncells[r] = 0;
    next[r] = -1;
    return 0;
  }
  if (index_limited_to_map(g, ray.sx, ray.sy, ray.ex, ray.ey, s) &&
      index_limited_to_map(g, ray.ex, ray.ey, ray.sx, ray.sy, t)) {
    const int dx = abs(t[0] - s[0]), dy = abs(t[1] - s[1]);
    nc = (dx >= dy ? dx : dy) + 1;
    d = make_int4(s[0], s[1], t[0], t[1]);
    desc[r] = d;
  }
  ncells[r] = nc;
  next[r] = -1;
  if (!ray.clear_end) {  // map_updater.h:44-49
    int ei[2];
    if (index_from_position(g, ray.ex, ray.ey, ei) && ei[0] >= win.x && ei[0] < win.y && ei[1] >= win.z &&
        ei[1] < win.w) {  // marks outside the owner window belong to another tile's GPU
      const int cell = ei[1] * g.size[0] + ei[0];
      unsigned h = hash_cell((unsigned)cell) & (unsigned)slot_mask;
      for (;;) {
        const int prev = atomicCAS(&slots[h].cell, -1, cell);
        if (prev == -1 || prev == cell) break;
        h = (h + 1) & (unsigned)slot_mask;
      }
      next[r] = atomicExch(&slots[h].head, r);
      atomicAdd(&slots[h].len, 1);
      atomicOr(&mark_bitmap[cell >> 5], 1u << (cell & 31));
    }
  }
  return nc;
}

__global__ void himm_prep_kernel(Geom g, const rna_ray* __restrict__ rays, int n, int4* __restrict__ desc,
                                 int* __restrict__ ncells, int* __restrict__ next, HimmSlot* __restrict__ slots,
                                 int slot_mask, unsigned* __restrict__ mark_bitmap, int4 win) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int4 d;
  (void)himm_prep_ray(g, rays, r, desc, ncells, next, slots, slot_mask, mark_bitmap, win, d);
}

__device__ void sift_down(int* a, int start, int end) {
  int root = start;
  for (;;) {
    int child = 2 * root + 1;
    if (child > end) break;
    if (child + 1 <= end && a[child] < a[child + 1]) child++;
    if (a[root] < a[child]) { const int t = a[root]; a[root] = a[child]; a[child] = t; root = child; }
    else break;
  }
}

// one hash slot per thread, a workgroup's worth of slots at a time (all threads of the workgroup call: three barriers inside);
// s_wave: one int per wavefront of the workgroup, s_base: one int
__device__ __forceinline__ void himm_collect_slot(int h, HimmSlot* __restrict__ slots, int n_slots, const int* __restrict__ next,
                                                  int* __restrict__ seqs, unsigned* __restrict__ before,
                                                  unsigned* __restrict__ after, int* __restrict__ total, int* s_wave, int* s_base) {
  HimmSlot sl = h < n_slots ? slots[h] : HimmSlot{-1, -1, 0, 0};
  const bool live = sl.cell >= 0;
  // One atomic per WORKGROUP on the shared running total: returning atomics on one L2 address take ~12 ns each, and the
  // compiler's one-per-wavefront aggregation still left 4096 of them -- 49 of the kernel's 51 us.
  int off = 0;
  {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
    int incl = live ? sl.len : 0;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    __syncthreads();   // (the previous chunk's offsets have been read)
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int w = 0; w < nwv; ++w) { const int t = s_wave[w]; s_wave[w] = run; run += t; }
      *s_base = run > 0 ? atomicAdd(total, run) : 0;
    }
    __syncthreads();
    off = *s_base + s_wave[wv] + incl - (live ? sl.len : 0);
  }
  if (!live) return;
  slots[h].offset = off;

  int* a = seqs + off;
  int k = 0;
  for (int r = sl.head; r >= 0; r = next[r]) a[k++] = r;
  const int len = k;

  if (len <= 24) {  // insertion sort
    for (int i = 1; i < len; ++i) {
      const int v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
      a[j + 1] = v;
    }
  } else {  // heap sort: many hits on one cell (e.g. a wall seen at close range)
    for (int st = (len - 2) / 2; st >= 0; --st) sift_down(a, st, len - 1);
    for (int end = len - 1; end > 0; --end) {
      const int t = a[end]; a[end] = a[0]; a[0] = t;
      sift_down(a, 0, end - 1);
    }
  }
  for (int i = 0; i < len; ++i) { before[off + i] = 0; after[off + i] = 0; }
}

__global__ void __launch_bounds__(1024) himm_collect_kernel(HimmSlot* __restrict__ slots, int n_slots, const int* __restrict__ next,
                                    int* __restrict__ seqs, unsigned* __restrict__ before,
                                    unsigned* __restrict__ after, int* __restrict__ total) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  himm_collect_slot(blockIdx.x * blockDim.x + threadIdx.x, slots, n_slots, next, seqs, before, after, total, s_wave, &s_base);
}

// ---- rays -> tiles ------------------------------------------------------------------------------------------------
// A ray is walked over its major axis in 64-cell tile columns; inside one it changes the minor tile at most once
// (slope <= 1), so it is registered with <= 2 tiles per column -- exactly the tiles that hold one of its cells.
struct RayWalk {
  int M0, m0, s, sm, den, add, num0, nc;
  bool xmajor;
  __device__ __forceinline__ RayWalk(const int4 d, int ncells) {
    const int adx = abs(d.z - d.x), ady = abs(d.w - d.y);
    xmajor = adx >= ady;                      // LineIterator.cpp:133-149
    den = xmajor ? adx : ady;
    add = xmajor ? ady : adx;
    num0 = den / 2;
    nc = ncells;
    const int sx = d.z >= d.x ? 1 : -1, sy = d.w >= d.y ? 1 : -1;
    M0 = xmajor ? d.x : d.y; m0 = xmajor ? d.y : d.x;
    s = xmajor ? sx : sy; sm = xmajor ? sy : sx;
  }
  // cell k of the Bresenham walk in closed form: the minor axis has stepped floor((num0 + k*add) / den) times
  // after k increments (LineIterator.cpp:60-70)
  __device__ __forceinline__ int minor(int k) const {
    // k, add, den < 46 341 (rows * cols < 2^31): the numerator fits 32 unsigned bits; a 32-bit division is a
    // handful of instructions, the 64-bit one a subroutine of ~200
    return m0 + (den > 0 ? (int)(((unsigned)num0 + (unsigned)k * (unsigned)add) / (unsigned)den) : 0) * sm;
  }
  __device__ __forceinline__ int major(int k) const { return M0 + k * s; }
  // Lock-step form for a whole wavefront (lanes without a ray pass nc = 0): f(active, tile) is called by ALL lanes the
  // same number of times, so that f can combine the lanes that name the same tile (wave_tile_add below).
  template <class F>
  __device__ __forceinline__ void for_each_tile_lockstep(int tiles_i, F f) const {
    int k = 0;
    while (__builtin_amdgcn_ballot_w64(k < nc)) {
      const bool act = k < nc;
      const int M = major(act ? k : 0), tM = M >> 6;
      int kend = s > 0 ? k + ((tM << 6) + 63 - M) : k + (M - (tM << 6));
      if (kend > nc - 1) kend = nc - 1;
      const int ta = minor(act ? k : 0) >> 6, tb = minor(act ? kend : 0) >> 6;
      f(act, xmajor ? ta * tiles_i + tM : tM * tiles_i + ta);
      f(act && tb != ta, xmajor ? tb * tiles_i + tM : tM * tiles_i + tb);
      if (act) k = kend + 1;
    }
  }
};

// counters[t] += 1 for every active lane, one atomic per DISTINCT tile of the wavefront; returns the lane's own slot
// (old value + its rank among the lanes of the same tile).  Rays of one scan are consecutive and start in the same
// tile: without this the 1563 rays of an origin queue up on one L2 atomic unit per tile they share.
__device__ __forceinline__ int wave_tile_add(int* __restrict__ counters, bool active, int t) {
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
  int slot = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lt = __shfl(t, leader);
    const unsigned long long same = __builtin_amdgcn_ballot_w64(active && t == lt);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counters[lt], __popcll(same));
    base = __shfl(base, leader);
    if (active && t == lt) slot = base + __popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return slot;
}

__global__ void himm_bin_count_kernel(const int4* __restrict__ desc, const int* __restrict__ ncells, int n, int tiles_i,
                                      int* __restrict__ tile_count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = r < n ? ncells[r] : 0;
  const RayWalk w(nc ? desc[r] : make_int4(0, 0, 0, 0), nc);
  w.for_each_tile_lockstep(tiles_i, [&](bool act, int t) { (void)wave_tile_add(tile_count, act, t); });
}

// exclusive prefix sum of the tile counts (one workgroup; ntile <= 65536 at 4096 x 4096 cells per 64 x 64 tile), and the
// list of the tiles that hold at least one ray: the rasteriser's workgroups take (tile, half) jobs from that list with a
// ticket instead of one workgroup per half tile of the map -- of the 8192 half tiles of a 4096^2 map a 100 k-ray batch
// touches ~1500, and next to the search workgroups that fill every CU each EMPTY workgroup still had to wait for a slot.
template <int NT>   // threads of the (one) workgroup that runs it; LDS: three arrays of NT ints
__device__ __forceinline__ void himm_bin_scan(int* __restrict__ tile_count, int ntile, int* __restrict__ tile_off,
                                              int* __restrict__ tile_cursor, int4* __restrict__ active,
                                              int* __restrict__ n_active, int* __restrict__ ticket, int* s_part, int* s_act, int* s_act2) {
  // s_act: tiles with many rays -- their jobs go first (the longest jobs must not be the last ones started); s_act2: the other tiles with rays
  constexpr int HEAVY = 512;     // rays: more than one round of a rasteriser workgroup
  const int per = (ntile + NT - 1) / NT;
  const int lo = min(ntile, (int)threadIdx.x * per), hi = min(ntile, lo + per);
  int sum = 0, nact = 0, nact2 = 0;
  for (int t = lo; t < hi; ++t) { const int c = tile_count[t]; sum += c; nact += c >= HEAVY; nact2 += c > 0 && c < HEAVY; }
  s_part[threadIdx.x] = sum;
  s_act[threadIdx.x] = nact;
  s_act2[threadIdx.x] = nact2;
  __syncthreads();
  for (int o = 1; o < NT; o <<= 1) {
    const bool in = (int)threadIdx.x >= o;
    const int v = in ? s_part[threadIdx.x - o] : 0, a = in ? s_act[threadIdx.x - o] : 0, a2 = in ? s_act2[threadIdx.x - o] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    s_act[threadIdx.x] += a;
    s_act2[threadIdx.x] += a2;
    __syncthreads();
  }
  const int n_heavy = s_act[NT - 1];
  int run = s_part[threadIdx.x] - sum, arun = s_act[threadIdx.x] - nact, arun2 = n_heavy + s_act2[threadIdx.x] - nact2;
  for (int t = lo; t < hi; ++t) {
    const int c = tile_count[t];
    tile_off[t] = run; tile_cursor[t] = 0; run += c;
    // the rasteriser's job record: tile, first pair, pairs -- one load
    if (c >= HEAVY) active[arun++] = make_int4(t, run - c, c, 0);
    else if (c > 0) active[arun2++] = make_int4(t, run - c, c, 0);
    tile_count[t] = 0;   // (counts: zero again for the next batch)
  }
  if (threadIdx.x == NT - 1) { tile_off[ntile] = s_part[NT - 1]; *n_active = s_act[NT - 1] + s_act2[NT - 1]; *ticket = 0; }
}

__global__ void __launch_bounds__(1024) himm_bin_scan_kernel(int* __restrict__ tile_count, int ntile, int* __restrict__ tile_off,
                                                             int* __restrict__ tile_cursor, int4* __restrict__ active,
                                                             int* __restrict__ n_active, int* __restrict__ ticket) {
  __shared__ int s_part[1024];
  __shared__ int s_act[1024];
  __shared__ int s_act2[1024];
  himm_bin_scan<1024>(tile_count, ntile, tile_off, tile_cursor, active, n_active, ticket, s_part, s_act, s_act2);
}

__global__ void himm_bin_fill_kernel(const int4* __restrict__ desc, const int* __restrict__ ncells, int n, int tiles_i,
                                     const int* __restrict__ tile_off, int* __restrict__ tile_cursor, int* __restrict__ pairs) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = r < n ? ncells[r] : 0;
  const RayWalk w(nc ? desc[r] : make_int4(0, 0, 0, 0), nc);
  w.for_each_tile_lockstep(tiles_i, [&](bool act, int t) {
    const int slot = wave_tile_add(tile_cursor, act, t);
    if (act) pairs[tile_off[t] + slot] = r;
  });
}

// Two jobs per 64 x 64 tile that at least one ray crosses, one per half (32 columns j): 8 KB of counters each -- a
// workgroup of 8 wavefronts and 12.4 KB of LDS is what fits a CU that has lost one of its four search workgroups
// (tests/test_kernel_budgets.py; whole-tile jobs with 16 KB were measured no faster).
// One LANE per ray: the lane loads its ray (256 independent loads per round instead of a dependent chain per ray),
// steps along its cells inside the tile with the integer Bresenham recurrence and counts clears in LDS.  A clear that
// lands on a MARKED cell (ray end points) belongs to an interval between that cell's marks.  The half tile's marked
// cells are looked up once (hash probe per cell, all threads in parallel) into an LDS table; a cell with a single
// mark -- almost all of them -- then takes its clears as two more LDS counters (before / after that mark), flushed
// with one global add each at the end.  Only cells with several marks, or beyond the table, walk the interval
// counters in HBM (hash probe + binary search: a chain of dependent loads).
constexpr int HIMM_MTAB = 256;   // marked cells per half tile with an LDS entry

#ifdef RNA_HIMM_STATS
// developer build: where a rasteriser job's time goes (100 MHz ticks of thread 0, summed over all jobs), printed by himm_release
__device__ unsigned long long g_himm_stat[16];
#define HST(var) const unsigned long long var = wall_clock64()
#define HACC(slot, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_himm_stat[slot], (unsigned long long)((b) - (a))); } while (0)
#define HCNT(slot, v) do { if (threadIdx.x == 0) atomicAdd(&g_himm_stat[slot], (unsigned long long)(v)); } while (0)
#else
#define HST(var)
#define HACC(slot, a, b)
#define HCNT(slot, v)
#endif

// a workgroup barrier that orders LDS only: global loads that are in flight stay in flight (himm_tile_raster_kernel)
__device__ __forceinline__ void himm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void himm_count_marked_clear(int cell, int r, const HimmSlot* __restrict__ slots, int slot_mask,
                                                        const int* __restrict__ seqs, unsigned* __restrict__ before,
                                                        unsigned* __restrict__ after) {
  // count this clear in the interval before the first mark with seq >= r
  unsigned h = hash_cell((unsigned)cell) & (unsigned)slot_mask;
  while (slots[h].cell != cell) h = (h + 1) & (unsigned)slot_mask;
  const int off = slots[h].offset, len = slots[h].len;
  int lo = 0, hi = len;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seqs[off + mid] < r) lo = mid + 1; else hi = mid;
  }
  if (lo < len) atomicAdd(&before[off + lo], 1u);
  else atomicAdd(&after[off + len - 1], 1u);
}

#ifndef RNA_HIMM_TR_ROWS
#define RNA_HIMM_TR_ROWS 32   // rows (j) of a 64 x 64 tile per job: 32 = half-tile jobs (8 KB of counters), 64 = whole tiles (16 KB)
#endif
constexpr int HIMM_TR_ROWS = RNA_HIMM_TR_ROWS;
#ifndef RNA_HIMM_TR_THREADS
#define RNA_HIMM_TR_THREADS 512
#endif
constexpr int HIMM_TR_THREADS = RNA_HIMM_TR_THREADS;   // (a power of two: the lane -> ray permutation of the rasteriser relies on it) // 8 wavefronts: two per SIMD next to the four of a resident search workgroup
// LDS of a rasteriser workgroup
struct RasterLds {
  unsigned cnt[HIMM_TR_ROWS * 64];   // clears per cell of this job's rows, index (j - j0) * 64 + (i & 63)
  unsigned mark[HIMM_TR_ROWS * 2];   // mark bits of those cells: word lc >> 5, bit lc & 31
  unsigned short mrank[HIMM_TR_ROWS * 2];   // marked cells in the words before this one
  int moff[HIMM_MTAB];     // per marked cell (by rank): offset of its marks in seqs / before / after, -1: not a single-mark cell
  int mseq[HIMM_MTAB];     // ... the ray sequence number of its one mark
  unsigned mcnt[2 * HIMM_MTAB];   // ... clears before / after that mark
  int touched, job;
  int4 rec;
};

// the (tile, half) jobs of a batch, taken by the calling workgroup (HIMM_TR_THREADS threads): `first_job` first, then by
// ticket (`grid` = workgroups that take part)
__device__ __forceinline__ void himm_raster_jobs(RasterLds& L, const int first_job, const int grid, int rows, int cols, int tiles_i, const int4* __restrict__ desc,
                                                               const int* __restrict__ ncells,
                                                               const int* __restrict__ tile_off, const int* __restrict__ pairs,
                                                               const int4* __restrict__ active, const int* __restrict__ n_active,
                                                               int* __restrict__ ticket,
                                                               float* __restrict__ layer, const unsigned* __restrict__ mark_bitmap,
                                                               const HimmSlot* __restrict__ slots, int slot_mask,
                                                               const int* __restrict__ seqs, unsigned* __restrict__ before,
                                                               unsigned* __restrict__ after, unsigned* __restrict__ dirty_tiles, int4 win) {

  constexpr int JPT = 64 / HIMM_TR_ROWS, NW = HIMM_TR_ROWS * 2;   // jobs per 64 x 64 tile; words of mark bits per job
  const int njobs = JPT * *n_active;
  // Which ray of a round a lane takes (-1: none).  A round is up to HIMM_TR_THREADS rays; a shorter one is dealt to the
  // first `span` lanes only (span = the next power of two, at least a wavefront): half the tiles of a batch hold fewer
  // than a hundred rays, and spread over all eight wavefronts each of them walked its 64 steps for a dozen lanes.
  // Inside the span the lanes take every 33rd ray (x -> 33 x mod span is a bijection), see the walk below.
  auto ray_of_lane = [](int rem) -> int {
    int span = HIMM_TR_THREADS;
    if (rem < HIMM_TR_THREADS) { span = 64; while (span < rem) span <<= 1; }
    if ((int)threadIdx.x >= span) return -1;
    const int p = (int)((threadIdx.x * 33u) & (unsigned)(span - 1));
    return p < rem ? p : -1;
  };
  // (tile, part) jobs by ticket; the first one is the workgroup's own index (a thousand workgroups asking for a ticket at
  // once queue up on one L2 address for ~12 ns each).
  // A job is a chain of dependent memory round trips (ticket, job record, mark bits, the marked cells' slots, the rays,
  // the cells to rewrite) with little arithmetic in between, and the 32 CUs the search streams leave to this kernel hold
  // few workgroups -- 17 us per job on an idle CU, of which the arithmetic is 3.  So: the ticket and the record of the
  // NEXT job are fetched by thread 0 while this job runs; the barriers only order LDS (s_waitcnt lgkmcnt(0) + s_barrier:
  // __syncthreads() would drain the global loads in flight at every barrier), so each step's loads are issued before the
  // previous step's LDS work; and the cells to rewrite are loaded all at once.
  int job = first_job;
  int4 rec = job < njobs ? active[job / JPT] : make_int4(0, 0, 0, 0);   // tile, its first (tile, ray) pair, its pairs (> 0)
  for (;;) {
  if (job >= njobs) break;
  HST(t_0);
  int next_ticket = 0;
  if (threadIdx.x == 0) next_ticket = atomicAdd(ticket, 1);   // (in flight until the walk is over)
  const int t = rec.x, part = job % JPT, np = rec.z;
  const int* mine = pairs + rec.y;
  const int ti = t % tiles_i, tj = t / tiles_i;
  const int i0 = ti << 6, j0 = (tj << 6) + part * HIMM_TR_ROWS;
  // round trip 1: this lane's first ray and the half tile's mark bits
  const int p0 = ray_of_lane(np);
  const int r0 = p0 >= 0 ? mine[p0] : -1;
  unsigned bits = 0u;
  if (threadIdx.x < NW) {
    // the tile's 64 cells of column j are two words of the bitmap when rows is a multiple of 32; gathered bit by
    // bit otherwise
    const int jl = threadIdx.x >> 1, wi = threadIdx.x & 1;
    const int j = j0 + jl, ib = i0 + (wi << 5);
    if (j < cols && ib < rows) {
      const size_t cell = (size_t)j * rows + ib;
      if ((cell & 31) == 0 && ib + 32 <= rows) bits = mark_bitmap[cell >> 5];
      else
        for (int b = 0; b < 32 && ib + b < rows; ++b) bits |= ((mark_bitmap[(cell + b) >> 5] >> ((cell + b) & 31)) & 1u) << b;
    }
  }
  for (int c = threadIdx.x; c < HIMM_TR_ROWS * 64; c += HIMM_TR_THREADS) L.cnt[c] = 0u;
  for (int c = threadIdx.x; c < 2 * HIMM_MTAB; c += HIMM_TR_THREADS) L.mcnt[c] = 0u;
  for (int c = threadIdx.x; c < HIMM_MTAB; c += HIMM_TR_THREADS) L.moff[c] = -1;
  if (threadIdx.x == 0) L.touched = 0;
  // round trip 2: the ray itself (used after the marked cells have been looked up)
  int4 d0 = make_int4(0, 0, 0, 0);
  int nc0 = 0;
  if (r0 >= 0) { d0 = desc[r0]; nc0 = ncells[r0]; }
  if (threadIdx.x < NW) L.mark[threadIdx.x] = bits;
  himm_lds_barrier();
  HST(t_1);
  if (threadIdx.x < 64) {   // marked cells in the words before this one: a prefix sum across the first wavefront (NW / 64 words per lane)
    constexpr int PER = NW / 64;
    int mine_n = 0;
    for (int k = 0; k < PER; ++k) mine_n += __popc(L.mark[threadIdx.x * PER + k]);
    int incl = mine_n;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += v; }
    int run = incl - mine_n;
    for (int k = 0; k < PER; ++k) { L.mrank[threadIdx.x * PER + k] = (unsigned short)run; run += __popc(L.mark[threadIdx.x * PER + k]); }
  }
  himm_lds_barrier();
  // one hash probe per marked cell of the half tile (also round trip 2): the slot is one 16-byte load, and a cell with
  // a single mark has that mark's ray in `head`
  for (int w = threadIdx.x >> 5, b = threadIdx.x & 31; w < NW; w += HIMM_TR_THREADS / 32) {
    if (!((L.mark[w] >> b) & 1u)) continue;
    const int lc = (w << 5) + b;
    L.cnt[lc] = 0x80000000u;   // "marked": seen by the walk in the value its add returns (zeroed before the first barrier)
    const int rank = L.mrank[w] + __popc(L.mark[w] & ((1u << b) - 1u));
    if (rank >= HIMM_MTAB) continue;
    const int cell = (j0 + (lc >> 6)) * rows + i0 + (lc & 63);
    unsigned h = hash_cell((unsigned)cell) & (unsigned)slot_mask;
    int4 sl = *reinterpret_cast<const int4*>(&slots[h]);   // cell, head, len, offset
    while (sl.x != cell) { h = (h + 1) & (unsigned)slot_mask; sl = *reinterpret_cast<const int4*>(&slots[h]); }
    if (sl.z == 1) { L.moff[rank] = sl.w; L.mseq[rank] = sl.y; }
  }
  HST(t_2);
  himm_lds_barrier();
  HST(t_3);
  bool touched = false;
  // this job's cells: the half tile cut to the owner window (tiled single map, SURVEY 8e mode 2: the line is rasterised
  // on the GLOBAL geometry and only the cells of this GPU's window are written)
  const int ilo = max(i0, win.x), ihi = min(i0 + 64, win.y), jlo = max(j0, win.z), jhi = min(j0 + HIMM_TR_ROWS, win.w);
  if (ilo < ihi && jlo < jhi)
  for (int base = 0; base < np; base += HIMM_TR_THREADS) {
    // The rays of a scan are consecutive in the list and leave their origin a quarter of a degree apart: taken lane by
    // lane, the 64 lanes of a wavefront would walk the same few cells for dozens of steps and their LDS adds would
    // serialise on the address.  The lanes take every 33rd ray instead (x -> 33 x mod 512 is a bijection): neighbouring
    // lanes are 7.6 degrees apart and part after a few cells.
    const int pl = base == 0 ? p0 : ray_of_lane(np - base);
    if (pl < 0) continue;
    const int r = base == 0 ? r0 : mine[base + pl];
    const RayWalk w(base == 0 ? d0 : desc[r], base == 0 ? nc0 : ncells[r]);
    // The ray's cells inside this job's rectangle are ONE run of k: the major coordinate moves by one per cell, the minor
    // one is monotone (it has stepped q(k) = floor((num0 + k * add) / den) times after k cells).  Both bounds in closed
    // form -- the walk below then has no test per cell.  (Round 3 walked every cell of the ray's 64-cell tile column in
    // both half-tile jobs and tested each against tile, half and window: twice the cells, three times the instructions;
    // on the 32 CUs the search streams leave to this kernel it took 0.67 ms per batch.)
    const int Mlo = w.xmajor ? ilo : jlo, Mhi = w.xmajor ? ihi : jhi, mlo = w.xmajor ? jlo : ilo, mhi = w.xmajor ? jhi : ihi;
    int k_lo = w.s > 0 ? Mlo - w.M0 : w.M0 - (Mhi - 1);
    int k_hi = w.s > 0 ? Mhi - 1 - w.M0 : w.M0 - Mlo;
    if (k_lo < 0) k_lo = 0;
    if (k_hi > w.nc - 1) k_hi = w.nc - 1;
    // minor steps allowed: q in [qa, qb]
    int qa = w.sm > 0 ? mlo - w.m0 : w.m0 - (mhi - 1);
    const int qb = w.sm > 0 ? mhi - 1 - w.m0 : w.m0 - mlo;
    if (qb < 0) continue;
    if (qa < 0) qa = 0;
    if (w.add > 0) {
      // q(k) >= qa  <=>  k >= ceil((qa * den - num0) / add);  q(k) <= qb  <=>  k <= floor(((qb + 1) * den - num0 - 1) / add)
      // (den, add, q < 46 341: the products fit 32 unsigned bits, see RayWalk::minor)
      const unsigned need = (unsigned)qa * (unsigned)w.den;
      if (need > (unsigned)w.num0) {
        const int ka = (int)((need - (unsigned)w.num0 + (unsigned)w.add - 1u) / (unsigned)w.add);
        if (ka > k_lo) k_lo = ka;
      }
      const unsigned lim = ((unsigned)qb + 1u) * (unsigned)w.den - (unsigned)w.num0 - 1u;
      const int kb = (int)(lim / (unsigned)w.add);
      if (kb < k_hi) k_hi = kb;
    } else if (qa > 0) {
      continue;   // the minor coordinate never moves and lies outside
    }
    if (k_lo > k_hi) continue;
    touched = true;
    // integer Bresenham from cell k_lo on (LineIterator.cpp:133-149): minor steps when the numerator passes den
    const unsigned acc = (unsigned)w.num0 + (unsigned)k_lo * (unsigned)w.add;   // < 2^32, see RayWalk::minor
    const unsigned q = w.den > 0 ? acc / (unsigned)w.den : 0u;
    int m = w.m0 + (int)q * w.sm;
    int num = w.den > 0 ? (int)(acc - q * (unsigned)w.den) : 0;
    int M = w.major(k_lo);
    // LDS index of a cell: (j & 31) * 64 + (i & 63); one step along the major / minor axis moves it by a constant
    const int dM = w.xmajor ? w.s : w.s * 64, dm = w.xmajor ? w.sm * 64 : w.sm;
    int lc = w.xmajor ? ((m - j0) << 6) + (M - i0) : ((M - j0) << 6) + (m - i0);
    // One LDS operation per cell: the counter of a MARKED cell carries bit 31 (set by the prologue), and the add returns
    // the old value -- which is looked at one step later, when the next cell's add is already on its way (with a read of
    // the mark bits in front of every add a step was two dependent LDS round trips, ~300 cycles; 8.7 of a job's 14.5 us).
    // A marked cell's clear belongs to an interval between that cell's marks, not to L.cnt (whose value for such a cell
    // is never used).
    auto marked_clear = [&](int c) {
      const unsigned mw = L.mark[c >> 5];
      const int rank = L.mrank[c >> 5] + __popc(mw & ((1u << (c & 31)) - 1u));
      const int off = rank < HIMM_MTAB ? L.moff[rank] : -1;
      // the clear belongs to the interval before the first mark with seq >= r
      if (off >= 0) atomicAdd(&L.mcnt[2 * rank + (L.mseq[rank] < r ? 1 : 0)], 1u);
      else himm_count_marked_clear((j0 + (c >> 6)) * rows + i0 + (c & 63), r, slots, slot_mask, seqs, before, after);
    };
    unsigned old_prev = 0u;
    int lc_prev = 0;
    for (int k = k_lo; k <= k_hi; ++k) {
      const unsigned old = atomicAdd(&L.cnt[lc], 1u);
      if (old_prev >> 31) marked_clear(lc_prev);
      old_prev = old;
      lc_prev = lc;
      num += w.add;
      lc += dM;
      if (num >= w.den && w.den > 0) { num -= w.den; lc += dm; }
    }
    if (old_prev >> 31) marked_clear(lc_prev);
  }
  HST(t_4);
  if (touched) L.touched = 1;
  // the next job: its ticket has long arrived; its record is loaded now, next to the cells this job rewrites
  int4 rec_next = make_int4(0, 0, 0, 0);
  int job_next = 0;
  if (threadIdx.x == 0) {
    job_next = grid + next_ticket;
    if (job_next < njobs) rec_next = active[job_next / JPT];
  }
  himm_lds_barrier();
  HST(t_5);
  if (L.touched) {
    // flags exist for the laser layer only ("laser differs from master here")
    if (threadIdx.x == 0 && dirty_tiles) reinterpret_cast<volatile unsigned char*>(dirty_tiles)[t] = 1;
    for (int c = threadIdx.x; c < HIMM_MTAB; c += HIMM_TR_THREADS)
      if (L.moff[c] >= 0) {
        const unsigned kb = L.mcnt[2 * c], ka = L.mcnt[2 * c + 1];
        if (kb) atomicAdd(&before[L.moff[c]], kb);
        if (ka) atomicAdd(&after[L.moff[c]], ka);
      }
    // apply: lanes run along i (contiguous in the column-major layer); only cells that were counted are touched.  All of
    // a thread's cells are loaded before the first is used (one memory round trip, not one per cell).
    constexpr int PER_T = HIMM_TR_ROWS * 64 / HIMM_TR_THREADS;
    unsigned kk[PER_T];
    float vv[PER_T];
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      kk[u] = L.cnt[threadIdx.x + u * HIMM_TR_THREADS];
      if (kk[u] >> 31) kk[u] = 0u;   // a marked cell: rewritten by himm_apply from its interval counters
    }
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int c = threadIdx.x + u * HIMM_TR_THREADS;
      vv[u] = kk[u] ? layer[(size_t)(j0 + (c >> 6)) * rows + i0 + (c & 63)] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      if (kk[u] == 0u) continue;
      const int c = threadIdx.x + u * HIMM_TR_THREADS;
      const float nv = himm_clear_n(vv[u], kk[u]);
      if (__float_as_int(nv) != __float_as_int(vv[u])) layer[(size_t)(j0 + (c >> 6)) * rows + i0 + (c & 63)] = nv;
    }
  }
  HST(t_6);
  if (threadIdx.x == 0) { L.job = job_next; L.rec = rec_next; }
  himm_lds_barrier();   // this job's LDS has been read; the next job is known
  HST(t_7);
  HACC(0, t_0, t_1); HACC(1, t_1, t_2); HACC(2, t_2, t_3); HACC(3, t_3, t_4); HACC(4, t_4, t_5); HACC(5, t_5, t_6); HACC(6, t_6, t_7);
  HCNT(7, 1); HCNT(8, np);
  job = L.job;
  rec = L.rec;
  }   // next job
}

__global__ void __launch_bounds__(HIMM_TR_THREADS) himm_tile_raster_kernel(int rows, int cols, int tiles_i, const int4* __restrict__ desc,
                                                               const int* __restrict__ ncells,
                                                               const int* __restrict__ tile_off, const int* __restrict__ pairs,
                                                               const int4* __restrict__ active, const int* __restrict__ n_active,
                                                               int* __restrict__ ticket,
                                                               float* __restrict__ layer, const unsigned* __restrict__ mark_bitmap,
                                                               const HimmSlot* __restrict__ slots, int slot_mask,
                                                               const int* __restrict__ seqs, unsigned* __restrict__ before,
                                                               unsigned* __restrict__ after, unsigned* __restrict__ dirty_tiles, int4 win) {
  __shared__ RasterLds L;
  himm_raster_jobs(L, (int)blockIdx.x, (int)gridDim.x, rows, cols, tiles_i, desc, ncells, tile_off, pairs, active, n_active, ticket, layer, mark_bitmap,
                   slots, slot_mask, seqs, before, after, dirty_tiles, win);
}

// one hash slot: the ordered replay on its marked cell; leaves bitmap and table empty for the next batch
__device__ __forceinline__ void himm_apply_slot(int h, int rows, HimmSlot* __restrict__ slots, int n_slots, int* __restrict__ total,
                                                const unsigned* __restrict__ before, const unsigned* __restrict__ after,
                                                float* __restrict__ layer, unsigned* __restrict__ mark_bitmap,
                                                unsigned* __restrict__ dirty_tiles, int tiles_i) {
  if (h == 0) *total = 0;   // (himm_collect, the only user, has finished)
  if (h >= n_slots) return;
  const HimmSlot sl = slots[h];
  if (sl.cell < 0) return;
  float v = layer[sl.cell];
  for (int k = 0; k < sl.len; ++k) {
    v = himm_clear_n(v, before[sl.offset + k]);
    v = himm_mark(v);
  }
  v = himm_clear_n(v, after[sl.offset + sl.len - 1]);
  layer[sl.cell] = v;
  atomicAnd(&mark_bitmap[sl.cell >> 5], ~(1u << (sl.cell & 31)));  // leave the bitmap all-zero ...
  slots[h] = HimmSlot{-1, -1, 0, 0};                               // ... and the hash table empty for the next batch
  const int i = sl.cell % rows, j = sl.cell / rows;
  const int tile = (j >> 6) * tiles_i + (i >> 6);
  // (most of the ~50 marked cells of a tile find the flag set already: a cached read instead of a store each)
  if (dirty_tiles && reinterpret_cast<volatile unsigned char*>(dirty_tiles)[tile] != 1) reinterpret_cast<volatile unsigned char*>(dirty_tiles)[tile] = 1;
}

__global__ void himm_apply_kernel(int rows, HimmSlot* __restrict__ slots, int n_slots, int* __restrict__ total,
                                  const unsigned* __restrict__ before, const unsigned* __restrict__ after,
                                  float* __restrict__ layer, unsigned* __restrict__ mark_bitmap,
                                  unsigned* __restrict__ dirty_tiles, int tiles_i) {
  himm_apply_slot(blockIdx.x * blockDim.x + threadIdx.x, rows, slots, n_slots, total, before, after, layer, mark_bitmap, dirty_tiles, tiles_i);
}


// (tile, ray) pairs of a batch.  Worst case: a ray is registered with at most two tiles per 64-cell tile column it crosses,
// i.e. 2 * ceil(max(rows, cols) / 64) + 2 pairs -- 130 ints per ray on a 4096^2 map although the bench's rays make 2.3.
// While that is small (<= 256 MB) it is simply allocated: nothing to check, nothing to wait for.  Beyond (a million
// rays on a large map: 0.5 GB and up, out of the HBM the A* page pools are sized from) the buffer starts at 16 pairs per
// ray, every batch's exact count -- the binning prefix sum's total -- is read back (4 bytes) before the pairs are written,
// and the buffer grows when a batch needs more.
static size_t himm_pairs_worst(const rna_engine* e, size_t cap_rays) {
  return cap_rays * (size_t)(2 * ((std::max(e->geom.size[0], e->geom.size[1]) + 63) / 64) + 2);
}
static size_t himm_pairs_initial(const rna_engine* e, size_t cap_rays, bool* checked) {
  const size_t worst = himm_pairs_worst(e, cap_rays);
  size_t limit = (size_t)64 << 20;   // ints
  if (const char* v = getenv("RNA_HIMM_PAIRS_LIMIT")) limit = (size_t)atoll(v);   // developer knob (tests): ints allocated without checking
  *checked = worst > limit;
  return *checked ? std::min(worst, std::max(cap_rays * 16, limit / 4)) : worst;
}

// tile_bins: count [ntile] | offset [ntile + 1] | cursor [ntile] | number of tiles with rays | ticket | (16-byte aligned)
// job records of the tiles with rays, int4 [ntile]
static size_t himm_bins_active_at(size_t ntile) { return (3 * ntile + 3 + 3) & ~(size_t)3; }

int ensure_scratch(rna_engine* e, int n) {
  HimmScratch& s = e->himm;
  int rc;
  if (!s.mark_bitmap) {
    const size_t words = (e->ncell + 31) / 32;
    if ((rc = dev_alloc(e, &s.mark_bitmap, words)) != RNA_OK) return rc;
    RNA_HIP(e, hipMemsetAsync(s.mark_bitmap, 0, words * sizeof(unsigned), e->stream));
    if ((rc = dev_alloc(e, &s.total, 1)) != RNA_OK) return rc;
    // per 64 x 64 tile: count | offset (ntile + 1) | cursor | list of the tiles with rays, then their number and the
    // rasteriser's ticket; the counts are all zero between batches (the scan leaves them so)
    const size_t bins = himm_bins_active_at((size_t)e->tiles_i * e->tiles_j) + (size_t)4 * e->tiles_i * e->tiles_j;
    if ((rc = dev_alloc(e, &s.tile_bins, bins)) != RNA_OK) return rc;
    RNA_HIP(e, hipMemsetAsync(s.tile_bins, 0, bins * sizeof(int), e->stream));
  }
  if (n <= s.cap_rays) return RNA_OK;
  int cap = 1024;
  while (cap < n) cap <<= 1;
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  // a failed regrow leaves no half-sized scratch behind: the next call starts from nothing again
  s.cap_rays = 0;
  s.n_slots = 0;
  if ((rc = dev_alloc(e, &s.rays_dev, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.desc, (size_t)cap)) != RNA_OK ||
      (rc = dev_alloc(e, &s.ncells, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.next, (size_t)cap)) != RNA_OK ||
      (rc = dev_alloc(e, &s.seqs, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.before, (size_t)cap)) != RNA_OK ||
      (rc = dev_alloc(e, &s.after, (size_t)cap)) != RNA_OK || (rc = dev_alloc(e, &s.slots, (size_t)cap * 2)) != RNA_OK ||
      (rc = dev_alloc(e, &s.pairs, himm_pairs_initial(e, (size_t)cap, &s.pairs_checked))) != RNA_OK) {
    const std::string why = e->err;
    (void)himm_release(e);
    e->err = why;
    return rc;
  }
  s.pairs_cap = himm_pairs_initial(e, (size_t)cap, &s.pairs_checked);
  if (s.pairs_checked && !s.pairs_total_host)
    RNA_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&s.pairs_total_host), sizeof(int), hipHostMallocDefault));
  s.n_slots = cap * 2;
  s.cap_rays = cap;
  hipLaunchKernelGGL(himm_init_slots_kernel, dim3((s.n_slots + 255) / 256), dim3(256), 0, e->stream, s.slots, s.n_slots, s.total);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

int himm_launch(rna_engine* e, int layer, const rna_ray* rays_dev, int n) {
  HimmScratch& s = e->himm;
  const Geom g = e->geom;
  // hash sized to the batch: a power of two >= 2n keeps probe sequences short
  int n_slots = 1024;
  while (n_slots < 2 * n) n_slots <<= 1;
  if (n_slots > s.n_slots) n_slots = s.n_slots;
  const int4 win = s.win[1] > 0 ? make_int4(s.win[0], s.win[1], s.win[2], s.win[3])
                                : make_int4(0, g.size[0], 0, g.size[1]);
  {
    KernelTimer kt(e, RNA_K_HIMM_PREP);
    // (the hash table is empty and *total is 0: himm_apply leaves them so, ensure_scratch starts them so)
    hipLaunchKernelGGL(himm_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, g, rays_dev, n, s.desc,
                       s.ncells, s.next, s.slots, n_slots - 1, s.mark_bitmap, win);
    hipLaunchKernelGGL(himm_collect_kernel, dim3((n_slots + 511) / 512), dim3(512), 0, e->stream, s.slots, n_slots,
                       s.next, s.seqs, s.before, s.after, s.total);
    RNA_HIP(e, hipGetLastError());
  }
  {
    KernelTimer kt(e, RNA_K_HIMM_RASTER);
    const int ntile = e->tiles_i * e->tiles_j;
    int* count = s.tile_bins;
    int* off = s.tile_bins + ntile;
    int* cursor = s.tile_bins + 2 * ntile + 1;
    int* n_active = s.tile_bins + 3 * ntile + 1;
    int* ticket = s.tile_bins + 3 * ntile + 2;
    int4* active = reinterpret_cast<int4*>(s.tile_bins + himm_bins_active_at((size_t)ntile));
    hipLaunchKernelGGL(himm_bin_count_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, s.desc, s.ncells, n, e->tiles_i, count);
    hipLaunchKernelGGL(himm_bin_scan_kernel, dim3(1), dim3(1024), 0, e->stream, count, ntile, off, cursor, active, n_active, ticket);
    if (s.pairs_checked) {   // the buffer is smaller than the worst case: this batch's exact pair count, before anything is written
      RNA_HIP(e, hipMemcpyAsync(s.pairs_total_host, off + ntile, sizeof(int), hipMemcpyDeviceToHost, e->stream));
      RNA_HIP(e, hipStreamSynchronize(e->stream));
      const size_t need = (size_t)*s.pairs_total_host;
      if (need > s.pairs_cap) {
        const size_t grown = std::min(himm_pairs_worst(e, (size_t)s.cap_rays), need + need / 4);
        int rc = dev_alloc(e, &s.pairs, grown);
        if (rc != RNA_OK) { s.pairs_cap = 0; s.cap_rays = 0; return rc; }   // (the next call starts from nothing)
        s.pairs_cap = grown;
      }
    }
    hipLaunchKernelGGL(himm_bin_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, s.desc, s.ncells, n, e->tiles_i, off,
                       cursor, s.pairs);
    // (tile, half) jobs are taken by ticket: as many workgroups as can be resident and useful, not one per half tile
    // (measured in the loop, next to the searches: 256 / 512 / 1024 workgroups -> 0.59 / 0.54 / 0.60 ms per batch)
    int raster_wgs = std::min((64 / HIMM_TR_ROWS) * ntile, std::max(256, 2 * e->cu_count));
    if (const char* w = getenv("RNA_HIMM_RASTER_WGS")) raster_wgs = std::max(1, std::min((64 / HIMM_TR_ROWS) * ntile, atoi(w)));   // developer knob
    hipLaunchKernelGGL(himm_tile_raster_kernel, dim3(raster_wgs), dim3(HIMM_TR_THREADS), 0, e->stream, g.size[0], g.size[1], e->tiles_i, s.desc,
                       s.ncells, off, s.pairs, active, n_active, ticket, e->layer[layer], s.mark_bitmap, s.slots, n_slots - 1, s.seqs, s.before,
                       s.after, layer == RNA_LAYER_LASER ? e->dirty_tiles : nullptr, win);
    RNA_HIP(e, hipGetLastError());
  }
  {
    KernelTimer kt(e, RNA_K_HIMM_APPLY);
    hipLaunchKernelGGL(himm_apply_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, e->stream, g.size[0], s.slots,
                       n_slots, s.total, s.before, s.after, e->layer[layer], s.mark_bitmap, layer == RNA_LAYER_LASER ? e->dirty_tiles : nullptr, e->tiles_i);
    RNA_HIP(e, hipGetLastError());
  }
  if (layer == RNA_LAYER_MASTER) { e->nbr_all_dirty = true; e->master_diverged = true; }
  return RNA_OK;
}

}  // namespace

namespace rna {
int himm_release(rna_engine* e) {
  HimmScratch& s = e->himm;
#ifdef RNA_HIMM_STATS
  {
    unsigned long long st[16];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_himm_stat), sizeof(st)) == hipSuccess && st[7] > 0) {
      const double j = (double)st[7];
      fprintf(stderr, "[himm raster stats] jobs %.0f, rays per job %.1f | thread 0, us per job: loads + zeroing -> barrier 1 %.2f, rank + probes %.2f, "
              "barrier 3 %.2f, walk %.2f, barrier 4 %.2f, flush + apply %.2f, last barrier %.2f\n", j, st[8] / j, st[0] * 0.01 / j,
              st[1] * 0.01 / j, st[2] * 0.01 / j, st[3] * 0.01 / j, st[4] * 0.01 / j, st[5] * 0.01 / j, st[6] * 0.01 / j);
    }
  }
#endif
  dev_free(&s.rays_dev); dev_free(&s.desc); dev_free(&s.ncells); dev_free(&s.next); dev_free(&s.slots);
  dev_free(&s.seqs); dev_free(&s.before); dev_free(&s.after); dev_free(&s.total); dev_free(&s.mark_bitmap);
  dev_free(&s.pairs); dev_free(&s.tile_bins);
  if (s.pairs_total_host) { (void)hipHostFree(s.pairs_total_host); s.pairs_total_host = nullptr; }
  s.pairs_cap = 0;
  s.cap_rays = 0;
  s.n_slots = 0;
  return RNA_OK;
}
}  // namespace rna

extern "C" int rna_himm_set_window(rna_engine* e, int i0, int j0, int ni, int nj) {
  if (!e) return RNA_EINVAL;
  if (ni <= 0 || nj <= 0) {  // back to the whole map
    e->himm.win[0] = e->himm.win[1] = e->himm.win[2] = e->himm.win[3] = 0;
    return RNA_OK;
  }
  if (i0 < 0 || j0 < 0 || i0 + ni > e->geom.size[0] || j0 + nj > e->geom.size[1])
    return fail(e, RNA_EINVAL, "rna_himm_set_window: window outside the map");
  e->himm.win[0] = i0; e->himm.win[1] = i0 + ni; e->himm.win[2] = j0; e->himm.win[3] = j0 + nj;
  return RNA_OK;
}

extern "C" int rna_himm_update_device(rna_engine* e, int layer, const rna_ray* rays_device, int n) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS || n < 0 || (n > 0 && !rays_device)) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  // the ray batch of the laser (or range) layer touches nothing the side work reads (VFH+: master, snapshots: the masks)
  if (layer == RNA_LAYER_MASTER) RNA_ENTER(e);
  else RNA_ENTER_NOJOIN(e);
  int rc = ensure_scratch(e, n);
  if (rc != RNA_OK) return rc;
  return himm_launch(e, layer, rays_device, n);
}

extern "C" int rna_himm_update(rna_engine* e, int layer, const rna_ray* rays_host, int n) {
  if (!e || layer < 0 || layer >= RNA_NUM_LAYERS || n < 0 || (n > 0 && !rays_host)) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_ENTER(e);
  int rc = ensure_scratch(e, n);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(e->himm.rays_dev, rays_host, (size_t)n * sizeof(rna_ray), hipMemcpyHostToDevice,
                            e->stream));
  rc = himm_launch(e, layer, e->himm.rays_dev, n);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_update_map_device(rna_engine* e, const rna_ray* rays_device, int n, int compose_mode) {
  int rc = rna_himm_update_device(e, RNA_LAYER_LASER, rays_device, n);
  if (rc != RNA_OK) return rc;
  return rna_compose_master(e, compose_mode);
}

extern "C" int rna_update_map(rna_engine* e, const rna_ray* rays_host, int n, int compose_mode) {
  int rc = rna_himm_update(e, RNA_LAYER_LASER, rays_host, n);
  if (rc != RNA_OK) return rc;
  rc = rna_compose_master(e, compose_mode);
  if (rc != RNA_OK) return rc;
  return rna_synchronize(e);
}
