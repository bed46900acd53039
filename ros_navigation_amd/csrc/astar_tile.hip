// astar_tile.hip -- tile-synchronous grid A* for gfx950 ("TSA"): the same contract and the same
// label-correcting argument as astar.hip, but the relaxation runs inside LDS.  This file holds the
// default search kernel of the engine: one workgroup (16 wavefronts) per query, rounds of tile jobs.
//
// Why tiles: the frontier kernel of astar.hip pays two HBM/L2 round trips plus two workgroup barriers
// per search GENERATION (~5 us), and a long query has >10^4 generations.  Here the search field lives
// in 32 x 32-cell PAGES (4 KiB, word = g << 8) and a wavefront owns one tile at a time:
//   1. grab-and-clear the tile's pending bits (cells improved since its last visit),
//   2. load the tile + a one-cell halo (and the tile's neighbour masks) into LDS with coalesced loads,
//   3. relax to the tile-local fixed point of the current f-bucket entirely in LDS: a wave-synchronous
//      queue, one lane per popped cell, plain LDS reads and writes (no LDS atomics), ~1 us per generation,
//   4. write the tile back (inner cells: coalesced stores; edge ring and improved halo cells:
//      atomicMin, because neighbouring tiles may be in flight on other wavefronts), hand improved
//      halo cells to their tiles as pending bits and activate those tiles.
// Pages are handed out on first touch: every query owns a contiguous run of `cap` pages and a tile ->
// page table (tmap), so the 2-4 % of the map a search visits sits in a few MiB of HBM instead of being
// scattered over a 64 MiB field (TLB reach, L2 hit rate), and the next launch on the same pipeline stage
// resets exactly the pages that were handed out.  Page 0 is shared, never written and always
// "unreached": reads of tiles without a page go there.  Two pending bitmaps per page (current bucket /
// next bucket) replace the frontier queues, so nothing can overflow except the page pool itself
// (status 5; the pool covers the whole map per query whenever HBM allows, see ensure_config).
// Exactness: every update is a min over lengths of real paths and the schedule runs every bucket to its
// fixed point, so at termination g is exact for f <= f*, which is all the canonical backtrace reads
// (DESIGN.md "Grid A* contract").  Measurements and the experiments that were dropped: DESIGN.md 5.
#include "engine.hpp"
#include <algorithm>
#include <vector>

using namespace rna;

namespace rna {

constexpr int TS = 32;                 // tile edge (cells)
constexpr int TW = TS + 2;             // LDS row pitch incl. halo
constexpr int TILE_WORDS = TS * TS;    // 1024
#ifndef RNA_TSA_WAVES
#define RNA_TSA_WAVES 16
#endif
#ifndef RNA_TSA_POLL_SLEEP
#define RNA_TSA_POLL_SLEEP 32
#endif
#ifndef RNA_TSA_UNR
#define RNA_TSA_UNR 1
#endif
constexpr int TSA_WAVES = RNA_TSA_WAVES;
constexpr int TSA_THREADS = TSA_WAVES * 64;
constexpr int TSA_MAX_TILE_WORDS = 2048;   // active-tile bitset words -> up to 65536 tiles
#ifndef RNA_TSA_JOBS
#define RNA_TSA_JOBS 2048
#endif
constexpr int TSA_JOBS = RNA_TSA_JOBS;             // tile jobs per round (more stay flagged for the next round)
#ifndef RNA_TSA_LQ
#define RNA_TSA_LQ 1024
#endif
constexpr int LQ = RNA_TSA_LQ;   // per-wave local queue (u16 LDS positions, power of two).  Live entries are distinct interior cells
                                 // (in-queue flag), so 1024 can never overflow; a smaller queue falls back to a rescan of the flags
constexpr int COST_S = 1000, COST_D = 1414;
constexpr int INF = 0x7fffffff;
constexpr unsigned G_INF = 0xFFFFFFu;

__device__ __forceinline__ int tsa_octile(int i, int j, int gi, int gj) {
  const int dx = abs(i - gi), dy = abs(j - gj);
  const int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return COST_S * mx + (COST_D - COST_S) * mn;
}
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// number of set bits of a wave mask below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ unsigned tsa_rank(unsigned long long m, unsigned base = 0u) {   // base + rank: v_mbcnt adds for free
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, base));
}
// buffer linear index <-> map-space (unwrapped) linear index (gmc/src/GridMapMath.cpp:467-476, 70-81)
__device__ __forceinline__ int tsa_unwrap_lin(int lin, int rows, int cols, int s0, int s1) {
  int i = lin % rows - s0, j = lin / rows - s1;
  if (i < 0) i += rows;
  if (j < 0) j += cols;
  return j * rows + i;
}
__device__ __forceinline__ int tsa_buffer_lin(int lin, int rows, int cols, int s0, int s1) {
  int i = lin % rows + s0, j = lin / rows + s1;
  if (i >= rows) i -= rows;
  if (j >= cols) j -= cols;
  return j * rows + i;
}
// wave-wide OR with DPP row shifts / row broadcasts; every lane receives the result (all 64 lanes active)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned tsa_dpp_or_step(unsigned v) {
  return v | (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned tsa_wave_or(unsigned v) {
  v = tsa_dpp_or_step<0x111, 0xF>(v);   // row_shr:1
  v = tsa_dpp_or_step<0x112, 0xF>(v);   // row_shr:2
  v = tsa_dpp_or_step<0x114, 0xF>(v);   // row_shr:4
  v = tsa_dpp_or_step<0x118, 0xF>(v);   // row_shr:8
  v = tsa_dpp_or_step<0x142, 0xA>(v);   // row_bcast:15
  v = tsa_dpp_or_step<0x143, 0xC>(v);   // row_bcast:31 -> lane 63 holds the OR of the wave
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// word index of cell (i, j) in a tile-major array (the neighbour-mask snapshot); tile and in-page offset of a cell
__device__ __forceinline__ size_t tm_index(int i, int j, int tiles_i) {
  return ((size_t)((j >> 5) * tiles_i + (i >> 5)) << 10) + ((j & 31) << 5) + (i & 31);
}
__device__ __forceinline__ int tile_of(int i, int j, int tiles_i) { return (j >> 5) * tiles_i + (i >> 5); }
__device__ __forceinline__ int in_page(int i, int j) { return ((j & 31) << 5) + (i & 31); }

// Device view of one pipeline stage.  Global page index of query q's local page p >= 1 is q*cap + p; page 0 is
// the shared "unreached" page.  Invariant between launches: every page word is (G_INF << 8), every pending word
// and every tmap entry is 0, EXCEPT what belongs to the local pages 1..nalloc[q] of each query; the next launch
// on the stage resets exactly those (tsa_reset_kernel) instead of rewriting 64 MiB per query.
constexpr unsigned TSA_BUSY = 0xFFFFFFFFu;   // tmap entry while its page is being handed out
constexpr int PEND_WORDS = 2 * TS;           // per page: two pending bitmaps (current / next bucket) of 32 column words
struct TsaStage {
  unsigned* pages;      // [1 + max_queries*cap][1024]
  unsigned* ppend;      // [1 + max_queries*cap][64]
  unsigned* tmap;       // [max_queries][ntile] tile -> local page, 0 = none
  unsigned* owner;      // [max_queries][cap + 1] local page -> tile
  int* nalloc;          // [max_queries] local pages handed out by the last search
  uint8_t* nbr_tm;      // [ntile][32][32] neighbour masks of this launch (snapshot), 0 outside the map
  int* perm;            // [max_queries] launch order of this batch: the k-th workgroup to START serves query perm[k]
  int* ticket;          // next position of perm to hand out (reset by every launch's init)
  int cap;              // pages per query
};
__host__ __device__ inline size_t tsa_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// snapshot of the neighbour masks in tile-major MAP-space order (4 cells per thread), taken at launch so that a
// later map update cannot disturb a search in flight
__global__ void tsa_snapshot_kernel(const uint8_t* __restrict__ nbr, int rows, int cols, int tiles_i, int tiles_j,
                                    uint8_t* __restrict__ nbr_tm, int s0, int s1) {
  const size_t nw = (size_t)tiles_i * tiles_j * TILE_WORDS;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t w4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w4 < nw / 4; w4 += step) {
    const size_t w = w4 * 4;
    const int t = (int)(w >> 10), l = (int)(w & 1023);
    const int i = (t % tiles_i) * TS + (l & 31), j = (t / tiles_i) * TS + (l >> 5);
    unsigned v = 0u;   // (i, j) is a MAP-space (unwrapped) index; nbr is stored at buffer indices
    if (j < cols) {
      const int bj = j + s1 >= cols ? j + s1 - cols : j + s1;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < rows) {
          const int bi = i + k + s0 >= rows ? i + k + s0 - rows : i + k + s0;
          v |= (unsigned)nbr[(size_t)bj * rows + bi] << (8 * k);
        }
    }
    reinterpret_cast<unsigned*>(nbr_tm)[w4] = v;
  }
}
// back to the invariant: the pages the previous launch on this stage handed out.  Block (x, q) takes the local
// pages 1 + x, 1 + x + gridDim.x, ... of query q (256 threads: one uint4 of the page each).
__global__ void __launch_bounds__(256) tsa_reset_kernel(TsaStage S, int ntile) {
  const int q = blockIdx.y;
  const int used = S.nalloc[q] < S.cap ? S.nalloc[q] : S.cap;
  for (int p = 1 + (int)blockIdx.x; p <= used; p += (int)gridDim.x) {
    const size_t gp = (size_t)q * S.cap + p;
    reinterpret_cast<uint4*>(S.pages + (gp << 10))[threadIdx.x] = make_uint4(0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u);
    if (threadIdx.x < PEND_WORDS) S.ppend[gp * PEND_WORDS + threadIdx.x] = 0u;
    if (threadIdx.x == 0) S.tmap[(size_t)q * ntile + S.owner[(size_t)q * (S.cap + 1) + p]] = 0u;
  }
}
__global__ void tsa_reset_done_kernel(TsaStage S, int max_queries) {
  for (int q = threadIdx.x; q < max_queries; q += blockDim.x) S.nalloc[q] = 0;
  if (threadIdx.x == 0) *S.ticket = 0;
}
// fresh allocation: every page "unreached" (the zero parts are a hipMemsetAsync)
__global__ void tsa_fill_pages_kernel(uint4* __restrict__ p, size_t n4) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < n4; w += step)
    p[w] = make_uint4(0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u);
}

// Launch order of a batch: longest expected search first (key = Chebyshev distance start -> goal,
// ties by index).  Workgroups are dispatched in index order and land on the XCDs round-robin, so this
// both starts the long queries early and deals them evenly over the eight XCDs; with the caller's
// (arbitrary) order one XCD regularly ended up with most of the long searches (+11 % throughput).
__global__ void __launch_bounds__(256) tsa_order_kernel(const rna_astar_query* __restrict__ queries, int n, int rows, int cols,
                                                          int* __restrict__ perm) {
  extern __shared__ int s_key[];
  const int ncell = rows * cols;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const rna_astar_query q = queries[i];
    int key = -1;   // invalid queries go last
    if (q.start >= 0 && q.goal >= 0 && q.start < ncell && q.goal < ncell) {
      // the circular-buffer offset cancels in the differences except across the seam; the key is only a heuristic
      const int di = abs(q.start % rows - q.goal % rows), dj = abs(q.start / rows - q.goal / rows);
      key = di > dj ? di : dj;
    }
    s_key[i] = key;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int ki = s_key[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const int kj = s_key[j];
      rank += (kj > ki || (kj == ki && j < i)) ? 1 : 0;
    }
    perm[rank] = i;
  }
}
__global__ void tsa_identity_order_kernel(int n, int* __restrict__ perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = i;
}

#ifdef RNA_TSA_STATS
// developer build: phase timers (100 MHz wall clock ticks summed over all jobs), printed by the host
__device__ unsigned long long g_tsa_stat[32];
#define TSA_T(var) const unsigned long long var = wall_clock64()
#define TSA_ACC(slot, t0, t1) tsa_acc[slot] += (unsigned long long)((t1) - (t0))
#define TSA_CNT(slot, v) tsa_acc[slot] += (unsigned long long)(v)
#define TSA_ACC_PARAM , unsigned long long* tsa_acc
#define TSA_ACC_ARG , tsa_acc
#else
#define TSA_ACC_PARAM
#define TSA_ACC_ARG
#define TSA_T(var)
#define TSA_ACC(slot, t0, t1)
#define TSA_CNT(slot, v)
#endif

constexpr int TSA_SCRATCH = 2 * TW + 4;   // words behind the tile that idle lanes of a relaxation step read and rewrite
struct alignas(16) TsaWave {
  unsigned tile[TW * TW + TSA_SCRATCH];   // (g << 8) | job flags, halo included; index (jl+1)*TW + (il+1).  Flags: bit0 in
                                 // the local queue, bit1 halo cell improved by this job, bit2 interior cell
                                 // improved beyond the current bucket
  unsigned short lq[LQ];         // local queue of LDS positions
  unsigned char mask[TILE_WORDS];    // neighbour masks of the 32 x 32 interior cells, index jl*32 + il
  unsigned nbpg[8];              // local pages of the eight neighbouring tiles as seen at load time (kept out of the VGPRs)
#ifdef RNA_TSA_STATS_REEXP
  unsigned char seen[TW * TW + TSA_SCRATCH];   // developer build: cell already expanded in this job
#endif
};

// Per-query context of a tile job (wave-uniform).
struct TsaCtx {
  int rows, cols, tiles_i, tiles_j;
  unsigned* pages;            // stage-wide page array
  unsigned* ppend;            // stage-wide pending bitmaps
  unsigned* tmap;             // this query's tile -> local page table
  unsigned* owner;            // this query's local page -> tile list
  size_t page_base;           // q * cap: global page = page_base + local page (local >= 1)
  int cap;
  int* nalloc;                // LDS: local pages handed out so far
  const uint8_t* nbr_tm;
  int gi, gj;
  __device__ __forceinline__ size_t gpage(unsigned local) const { return local ? page_base + local : 0; }
};

// Make sure tile `nt` of this query has a page; called by ONE lane per tile of a wave (lanes of a wave never ask
// for the same tile, so a lane only ever waits for another wavefront).  Returns the local page, 0 = pool exhausted.
__device__ __forceinline__ unsigned tsa_page_get(const TsaCtx& C, int nt, unsigned seen) {
  unsigned v = seen;
  if (v == 0u) {
    const unsigned old = atomicCAS(&C.tmap[nt], 0u, TSA_BUSY);
    if (old == 0u) {
      const int p = atomicAdd(C.nalloc, 1) + 1;
      if (p > C.cap) {
        __hip_atomic_store(&C.tmap[nt], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0u;
      }
      C.owner[p] = (unsigned)nt;
      __hip_atomic_store(&C.tmap[nt], (unsigned)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return (unsigned)p;
    }
    v = old;
  }
  while (v == TSA_BUSY) {
    __builtin_amdgcn_s_sleep(1);
    v = ld_l2(&C.tmap[nt]);
  }
  return v;
}

// One tile job, executed by one wavefront (lane = this wave's lane id).  `sch` supplies the
// scheduler-specific pieces: best() / improve_best(g) (upper bound on f*), act_cur(tile) /
// act_far(tile) (a tile received pending cells for the current / the next bucket), overflow(), pool_exhausted().
// `pg` is the local page of tile t (a tile only becomes a job after its page exists); `role` selects which of the
// page's two pending bitmaps is the current bucket's.  Returns the number of cell expansions.
template <class Sched>
__device__ __forceinline__ int tsa_job(Sched& sch, TsaWave& W, const int lane, const int t, const unsigned pg, const TsaCtx& C,
                                       const int role, const long long bucket_end TSA_ACC_PARAM) {
  int expanded = 0;
  const int rows = C.rows, cols = C.cols, tiles_i = C.tiles_i, tiles_j = C.tiles_j, gi = C.gi, gj = C.gj;
  const int ti = t % tiles_i, tj = t / tiles_i;
  const int i0 = ti * TS, j0 = tj * TS;
  unsigned* ftile = C.pages + (C.gpage(pg) << 10);
  unsigned* ptile = C.ppend + C.gpage(pg) * PEND_WORDS;
  unsigned* pend_cur = ptile + role * TS;
  unsigned* pend_far = ptile + (role ^ 1) * TS;

  TSA_T(t_a);
  // 1. grab-and-clear the pending bits of this tile (lane = column jl); in the same round trip lane k < 8 looks up
  //    the page of the neighbouring tile in direction k (0 = none yet, or outside the map)
  unsigned seed = 0u;
  if (lane < TS) seed = atomicExch(&pend_cur[lane], 0u);
  unsigned nb_pg = 0u;      // lanes 0..7: local page of neighbour tile k as seen now (TSA_BUSY: being handed out)
  int nb_t = -1;            // lanes 0..7: that tile, -1 outside the map
  if (lane < 8) {
    const int kdi = (lane == 0 || lane == 3 || lane == 5) ? -1 : ((lane == 2 || lane == 4 || lane == 7) ? 1 : 0);
    const int kdj = lane < 3 ? -1 : (lane > 4 ? 1 : 0);
    const int nti = ti + kdi, ntj = tj + kdj;
    if (nti >= 0 && ntj >= 0 && nti < tiles_i && ntj < tiles_j) { nb_t = ntj * tiles_i + nti; nb_pg = ld_l2(&C.tmap[nb_t]); }
  }
  // 2. tile + halo -> LDS (after the pending bits: every grabbed bit's value is already in L2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    // Everything a job reads from HBM is issued before the first wait: the three halo words of a lane (neighbour
    // column, neighbour row, corner), its 16 tile words (cell r*64 + lane sits 2r rows below cell `lane`: one base
    // address + compile-time offsets) and its 16 mask bytes -- ONE memory round trip after the pending bits.
    // A neighbour whose page is still being handed out has nothing in it yet: read the shared "unreached" page.
    const unsigned nb_rd = nb_pg == TSA_BUSY ? 0u : nb_pg;
    const int h = lane & 31;
    const bool second = lane >= 32;
    // halo columns: tile (ti, tj-1) column 31 -> LDS jl=-1 ; tile (ti, tj+1) column 0 -> LDS jl=32
    // halo rows:    tile (ti-1, tj) row 31 -> LDS il=-1 ; tile (ti+1, tj) row 0 -> LDS il=32 ; then the 4 corners
    const unsigned* pcol = C.pages + (C.gpage((unsigned)__shfl((int)nb_rd, second ? 6 : 1)) << 10);
    const unsigned* prow = C.pages + (C.gpage((unsigned)__shfl((int)nb_rd, second ? 4 : 3)) << 10);
    const unsigned hv_col = ld_l2(&pcol[((second ? 0 : 31) << 5) + h]);
    const unsigned hv_row = ld_l2(&prow[(h << 5) + (second ? 0 : 31)]);
    const int cdi = (lane & 1) ? 1 : -1, cdj = (lane & 2) ? 1 : -1;
    // corner of lane c < 4: direction (cdi, cdj) = neighbour index {0, 2, 5, 7}[c]
    const unsigned* pcor = C.pages + (C.gpage((unsigned)__shfl((int)nb_rd, (lane & 2) ? ((lane & 1) ? 7 : 5) : ((lane & 1) ? 2 : 0))) << 10);
    unsigned hv_cor = 0xFFFFFF00u;
    if (lane < 4) hv_cor = ld_l2(&pcor[((cdj > 0 ? 0 : 31) << 5) + (cdi > 0 ? 0 : 31)]);
    unsigned* tp = &W.tile[((lane >> 5) + 1) * TW + (lane & 31) + 1];
    unsigned tv[TILE_WORDS / 64];
#pragma unroll
    for (int r = 0; r < TILE_WORDS / 64; ++r) tv[r] = ld_l2(&ftile[r * 64 + lane]);
    const uint4 mv = *reinterpret_cast<const uint4*>(C.nbr_tm + ((size_t)t << 10) + lane * 16);
    W.tile[(second ? TS + 1 : 0) * TW + h + 1] = hv_col;
    W.tile[(h + 1) * TW + (second ? TS + 1 : 0)] = hv_row;
    if (lane < 4) W.tile[(cdj > 0 ? TS + 1 : 0) * TW + (cdi > 0 ? TS + 1 : 0)] = hv_cor;
#pragma unroll
    for (int r = 0; r < TILE_WORDS / 64; ++r) tp[r * 2 * TW] = tv[r];
    *reinterpret_cast<uint4*>(&W.mask[lane * 16]) = mv;
    if (lane < 8) W.nbpg[lane] = nb_pg;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  TSA_T(t_b);
  TSA_ACC(0, t_a, t_b);
  if (lane < TSA_SCRATCH) W.tile[TW * TW + lane] = 0xFFFFFF00u;   // scratch words behind the tile (see the relaxation loop)
#ifdef RNA_TSA_STATS_REEXP
  for (int w = lane; w < TW * TW + TSA_SCRATCH; w += 64) W.seen[w] = 0;
#endif
  // 3. seed the local queue from the pending bits
  int head = 0, tail = 0;   // wave-uniform
  bool lq_full = false;     // a push did not fit: those cells keep their in-queue flag and are found by a rescan
  {
    unsigned bits = seed;   // lane jl holds the bits (il) of its column
    for (;;) {
      const bool has = bits != 0u;
      const unsigned long long m = __ballot(has);
      if (!m) break;
      const int cnt = __popcll(m);
      const bool fits = LQ >= TILE_WORDS || tail + cnt <= LQ;
      if (has) {
        const int il = __ffs(bits) - 1;
        bits &= bits - 1;
        const int p = (lane + 1) * TW + il + 1;
        if (fits) W.lq[(tail + (int)tsa_rank(m)) & (LQ - 1)] = (unsigned short)p;
        W.tile[p] |= 1u;
      }
      if (fits) tail += cnt; else lq_full = true;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // 4. relax to the tile-local fixed point of the current bucket.  One lane per popped cell; the
  //    eight directions are visited one after the other.  Within one direction all lanes target
  //    DIFFERENT cells (target = own cell + the same offset), so the min-update of a neighbour and
  //    the test-and-set of its in-queue flag are one plain LDS read and one plain LDS write of the
  //    same word -- no LDS atomics (ds_min_rtn on 16 waves was the bottleneck of an earlier version)
  //    and no divergent branches.  All f-tests (prune against the upper bound, defer to the next
  //    bucket) happen once per POPPED cell, so the direction body is only compare / select / store.
  //    (Tried and dropped: reading all eight neighbours at once and resolving write conflicts with
  //    non-returning ds_min_u32 plus a returning ds_and at the pop -- four dependent LDS round trips
  //    instead of ten, yet 27 % slower: LDS atomics cost more than the round trips they save.)
  const int best_in = sch.best();
  const int bend = bucket_end > (long long)INF ? INF : (int)bucket_end;
  const int goal_p = (gi >= i0 && gi < i0 + TS && gj >= j0 && gj < j0 + TS) ? (gj - j0 + 1) * TW + (gi - i0 + 1) : -1;
  bool ovf = false;
  unsigned dirty_lane = 0u;   // rows (jl) of the cells this lane popped: only those rows can have changed
  for (;;) {
    if (tail == head) {
      if (LQ >= TILE_WORDS || !lq_full) break;
      // rare: the queue overflowed earlier; it is empty now, so every flagged cell is un-queued: queue them again
      lq_full = false;
#pragma unroll 1
      for (int r = 0; r < TILE_WORDS / 64; ++r) {
        const int pos = ((lane >> 5) + 1 + 2 * r) * TW + (lane & 31) + 1;
        const unsigned long long fq = __builtin_amdgcn_ballot_w64((W.tile[pos] & 1u) != 0u);
        const int cnt = __popcll(fq);
        if (tail - head + cnt <= LQ) {
          if (__builtin_amdgcn_inverse_ballot_w64(fq)) W.lq[(tail + (int)tsa_rank(fq)) & (LQ - 1)] = (unsigned short)pos;
          tail += cnt;
        } else if (cnt) {
          lq_full = true;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (tail == head) break;
    }
    const int n = tail - head;
    const int take = n < 64 ? n : 64;
    const bool act = lane < take;
    // lanes without a cell point at the scratch words behind the tile: every lane can then run the same
    // unpredicated read-select-write per direction without ever touching a cell another lane updates
    const int p = act ? (int)W.lq[(head + lane) & (LQ - 1)] : (TW * TW + TW + 1);
    head += take;
    TSA_CNT(8, 1);
    TSA_CNT(9, take);
    const unsigned cw = W.tile[p];
    const int g = (int)(cw >> 8);
    const int pil = p % TW - 1, pjl = p / TW - 1;
    const unsigned mk = W.mask[act ? pjl * TS + pil : 0];
    dirty_lane |= act ? 1u << pjl : 0u;
    const unsigned ax = (unsigned)abs(i0 + pil - gi), ay = (unsigned)abs(j0 + pjl - gj);
    const int fc = g + (int)(__umul24(ax > ay ? ax : ay, COST_S) + __umul24(ax > ay ? ay : ax, COST_D - COST_S));
    const int sb = sch.best();
    const int best_now = best_in < sb ? best_in : sb;
    const bool live = act && fc <= best_now;      // else pruned: f > upper bound on f*
    const bool later = live && fc >= bend;        // belongs to a later bucket: flag it, do not expand
    if (act) W.tile[p] = (cw & ~1u) | (later ? 4u : 0u);   // popped: may be queued again
    const bool ex = live && !later;               // this lane expands its cell
#ifdef RNA_TSA_STATS_REEXP
    {
      const bool again = ex && W.seen[p] != 0;
      TSA_CNT(14, __popcll(__builtin_amdgcn_ballot_w64(again)));
      TSA_CNT(15, __popcll(__builtin_amdgcn_ballot_w64(ex)));
      if (ex) W.seen[p] = 1;
    }
#endif
    expanded += ex ? 1 : 0;
    // rare events, kept out of the straight-line path: the popped cell is the goal (tighten the bound, do not
    // expand it), or its g is about to leave the 24-bit range
    const bool at_goal = ex && p == goal_p;
    const bool too_far = ex && g >= (int)G_INF - 3 * COST_D;
    if (__builtin_amdgcn_ballot_w64(at_goal || too_far)) {
      if (at_goal) sch.improve_best(g);
      ovf |= too_far;
    }
    const unsigned m = (ex && !at_goal && !too_far) ? mk : 0u;
    // candidate words of a straight / diagonal step: new g in the high bits; "| 0xff" for the test
    // g + w < g(neighbour) on whole words; flag byte of the stored word = in-queue (interior) or
    // halo-dirty.  Old flags need not be kept: a re-queued cell is re-tested when it is popped.
    // All predicates are kept as 64-bit wave masks in SGPRs (ballot / inverse ballot), so the
    // boolean algebra runs on the scalar unit and the vector unit only compares, selects and stores.
    const unsigned gs = ((unsigned)(g + COST_S) << 8), gd = ((unsigned)(g + COST_D) << 8);
    unsigned c_st = gs | 0xffu, c_dt = gd | 0xffu, c_si = gs | 1u, c_sh = gs | 2u, c_di = gd | 1u, c_dh = gd | 2u;
    int pb = p - TW - 1;   // lowest neighbour: all eight offsets are non-negative immediates
    // keep these in registers: recomputing them from g in every direction costs more than it saves
    asm volatile("" : "+v"(pb), "+v"(c_st), "+v"(c_dt), "+v"(c_si), "+v"(c_sh), "+v"(c_di), "+v"(c_dh));
    unsigned* const nb = &W.tile[pb];
    const unsigned nm = ~m;
    unsigned tail_v = (unsigned)tail;   // the queue tail as a (uniform) vector register: v_mbcnt adds it to the rank for free
    asm volatile("" : "+v"(tail_v));
    const unsigned long long il_lo = __builtin_amdgcn_ballot_w64(pil == 0), il_hi = __builtin_amdgcn_ballot_w64(pil == TS - 1);
    const unsigned long long jl_lo = __builtin_amdgcn_ballot_w64(pjl == 0), jl_hi = __builtin_amdgcn_ballot_w64(pjl == TS - 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int di = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0);
      const int dj = k < 3 ? -1 : (k > 4 ? 1 : 0);
      const int off = (di + 1) + (dj + 1) * TW;
      const bool straight = (k == 1 || k == 3 || k == 4 || k == 6);
      const unsigned nwv = nb[off];
      // candidate word, all ones where direction k is not allowed (bit k of ~m set): "candidate < neighbour" is
      // then the whole improvement test g + w < g(neighbour)
      const unsigned cand = (straight ? c_st : c_dt) | (unsigned)__builtin_amdgcn_sbfe((int)nm, k, 1);
      const unsigned long long improve = __builtin_amdgcn_ballot_w64(cand < nwv);
      // flag byte: 0 idle, 1 in the queue, 2 halo-dirty, 4 deferred to the next bucket (never combined: a popped
      // cell drops to 0 / 4, an improved one is rewritten to exactly 1 / 2)
      const unsigned long long inq = __builtin_amdgcn_ballot_w64((nwv & 0xffu) == 1u);
      const unsigned long long halo = (di < 0 ? il_lo : (di > 0 ? il_hi : 0ull)) | (dj < 0 ? jl_lo : (dj > 0 ? jl_hi : 0ull));
      const unsigned word = __builtin_amdgcn_inverse_ballot_w64(halo) ? (straight ? c_sh : c_dh) : (straight ? c_si : c_di);
      nb[off] = __builtin_amdgcn_inverse_ballot_w64(improve) ? word : nwv;   // unpredicated: rewrites the old word otherwise
      const unsigned long long push = improve & ~halo & ~inq;
      const int cnt = __popcll(push);
      if (LQ >= TILE_WORDS || (int)__builtin_amdgcn_readfirstlane(tail_v) - head + cnt <= LQ) {
        if (__builtin_amdgcn_inverse_ballot_w64(push))
          W.lq[tsa_rank(push, tail_v) & (LQ - 1)] = (unsigned short)(pb + off);
        tail_v += (unsigned)cnt;
      } else {
        lq_full = true;   // the stored words carry the in-queue flag: the rescan picks these cells up
      }
      // Direction k + 1 must read what direction k stored: another lane's neighbour may be the same cell.  Within
      // one lane the eight addresses are distinct, so without this the compiler may hoist the later reads above the
      // store (seen once the code between them became straight-line); the hardware itself executes a wavefront's
      // LDS instructions in order.
      asm volatile("" ::: "memory");
    }
    tail = (int)__builtin_amdgcn_readfirstlane(tail_v);
    __builtin_amdgcn_wave_barrier();
  }
  if (ovf) sch.overflow();
#ifdef RNA_TSA_STATS
  {  // histogram of job sizes (expansions per job): slots 10..15 = 0, 1-15, 16-63, 64-255, 256-1023, 1024+
    int ex = expanded;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ex += __shfl_xor(ex, o);
    const int b = ex == 0 ? 0 : (ex < 16 ? 1 : (ex < 64 ? 2 : (ex < 256 ? 3 : (ex < 1024 ? 4 : 5))));
    tsa_acc[10 + b] += 1;
  }
#endif

  TSA_T(t_c);
  TSA_ACC(1, t_b, t_c);
  // 5. write back.  Inner 30 x 30 cells are private to this tile: coalesced stores.  Edge ring:
  //    atomicMin (a neighbouring tile's job may have improved them in HBM meanwhile).  Only rows that hold a
  //    popped cell can differ from what was loaded (every improved interior cell is queued, hence popped).
  const unsigned dirty = tsa_wave_or(dirty_lane);
  // Everything after the relaxation loop derives its lane arithmetic from an opaque copy of the lane id, so that no
  // address or index computed for the tile load stays alive across the loop: 104 -> 86 VGPRs (91 with the one-batch
  // load above).  The register budget of this kernel decides how many wavefronts of the map-update and VFH kernels
  // fit next to four searches per SIMD.
  int lane_b = lane;
  asm volatile("" : "+v"(lane_b));
  {
    const unsigned* tp = &W.tile[((lane_b >> 5) + 1) * TW + (lane_b & 31) + 1];
    const int il = lane_b & 31;
    const bool edge_col = il == 0 || il == TS - 1;
#pragma unroll 2
    for (int r = 0; r < TILE_WORDS / 64; ++r) {
      if (((dirty >> (2 * r)) & 3u) == 0u) continue;            // neither row of this pair changed
      if (!((dirty >> (2 * r + (lane_b >> 5))) & 1u)) continue;   // this lane's row did not
      const unsigned v = tp[r * 2 * TW] & 0xFFFFFF00u;
      const bool edge = edge_col || (r == 0 && lane_b < 32) || (r == TILE_WORDS / 64 - 1 && lane_b >= 32);
      if (edge) {
        if ((v >> 8) != G_INF) (void)atomicMin(&ftile[r * 64 + lane_b], v);
      } else {
        ftile[r * 64 + lane_b] = v;
      }
    }
  }
  //    far-bucket cells of this tile: column jl -> one pending word (bit il); deferred cells were popped too
  {
    bool anyfar = false;
#pragma unroll 4
    for (int r = 0; r < TS / 2; ++r) {
      if (((dirty >> (2 * r)) & 3u) == 0u) continue;
      const int jl = 2 * r + (lane_b >> 5), il = lane_b & 31;
      const bool f = (W.tile[(jl + 1) * TW + il + 1] & 4u) != 0u;
      const unsigned long long bm = __ballot(f);
      const unsigned word = (unsigned)(bm >> (lane_b & 32));
      if ((lane_b & 31) == 0 && word) { atomicOr(&pend_far[jl], word); }
      anyfar |= bm != 0ull;
    }
    if (anyfar && lane_b == 0) sch.act_far(t);
  }
  TSA_T(t_d);
  TSA_ACC(2, t_c, t_d);
  //    improved halo cells -> their tiles: value first, then the pending bit, then the activation.  The 136 ring
  //    positions are three cells per lane; each stage is issued for all three before its single wait, so the
  //    hand-over costs two memory round trips, not two per cell (plus one when a neighbour needs its first page).
  {
    constexpr int HK = (4 * TW + 63) / 64;
    unsigned hv[HK];
    unsigned hof[HK];   // (neighbour direction << 10) | word inside its page
    int hfn[HK];
    bool hdo[HK];
    unsigned need = 0u;   // directions this lane hands cells to
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      const int hh = lane_b + 64 * k;
      hdo[k] = false; hv[k] = 0u; hof[k] = 0u; hfn[k] = 0;
      if (hh < 4 * TW) {
        // ring positions: hh in [0,TW): jl=-1 row; [TW,2TW): jl=32 row; [2TW,3TW): il=-1 col; [3TW,4TW): il=32 col
        const int side = hh / TW, u = hh % TW;
        int pil, pjl;
        if (side == 0) { pjl = -1; pil = u - 1; }
        else if (side == 1) { pjl = TS; pil = u - 1; }
        else if (side == 2) { pil = -1; pjl = u - 1; }
        else { pil = TS; pjl = u - 1; }
        const bool corner_dup = side >= 2 && (pjl < 0 || pjl >= TS);   // corners are covered by the row sides
        const unsigned tw_ = W.tile[(pjl + 1) * TW + pil + 1];
        const int ni = i0 + pil, nj = j0 + pjl;
        if (!corner_dup && (tw_ & 2u) && ni >= 0 && nj >= 0 && ni < rows && nj < cols) {
          const unsigned v = tw_ & 0xFFFFFF00u;
          const int fn = (int)(v >> 8) + tsa_octile(ni, nj, gi, gj);
          if (fn <= sch.best()) {                                // else pruned: f > upper bound on f*
            if (ni == gi && nj == gj) sch.improve_best((int)(v >> 8));
            const int idx = ((pjl < 0 ? 0 : (pjl >= TS ? 2 : 1)) * 3) + (pil < 0 ? 0 : (pil >= TS ? 2 : 1));   // 0..8 without 4
            const unsigned kd = (unsigned)(idx < 4 ? idx : idx - 1);
            hdo[k] = true; hv[k] = v; hof[k] = (kd << 10) | (unsigned)in_page(ni, nj); hfn[k] = fn;
            need |= 1u << kd;
          }
        }
      }
    }
    // first touch of a neighbouring tile: lane k < 8 hands out the page of direction k (other wavefronts of this
    // workgroup may be doing the same for the same tile: tsa_page_get settles that)
    need = tsa_wave_or(need);
    unsigned nb_pg = 0u;
    if (need) {
      if (lane_b < 8) nb_pg = W.nbpg[lane_b];
      if (lane_b < 8 && ((need >> lane_b) & 1u) && (nb_pg == 0u || nb_pg == TSA_BUSY)) {
        const int kdi = (lane_b == 0 || lane_b == 3 || lane_b == 5) ? -1 : ((lane_b == 2 || lane_b == 4 || lane_b == 7) ? 1 : 0);
        const int kdj = lane_b < 3 ? -1 : (lane_b > 4 ? 1 : 0);
        nb_pg = tsa_page_get(C, (tj + kdj) * tiles_i + ti + kdi, nb_pg);
        if (nb_pg == 0u) sch.pool_exhausted();
      }
    }
    unsigned hold[HK];
    unsigned* hpage[HK];
    unsigned* hpend[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      const unsigned lp = (unsigned)__shfl((int)nb_pg, (int)(hof[k] >> 10));
      hdo[k] = hdo[k] && lp != 0u;   // no page left: the search is being abandoned (status 5)
      const size_t gp = C.gpage(lp);
      hpage[k] = C.pages + (gp << 10);
      hpend[k] = C.ppend + gp * PEND_WORDS;
    }
#pragma unroll
    for (int k = 0; k < HK; ++k) hold[k] = hdo[k] ? atomicMin(&hpage[k][hof[k] & 1023u], hv[k]) : 0u;
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      hdo[k] = hdo[k] && hv[k] < hold[k];
      if (hdo[k]) {
        const unsigned l = hof[k] & 1023u;
        atomicOr(&hpend[k][((hfn[k] >= bucket_end) ? (role ^ 1) : role) * TS + (l >> 5)], 1u << (l & 31));
      }
    }
    // the values and their pending bits must be performed at L2 before the tiles can be scheduled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < HK; ++k)
      if (hdo[k]) {
        const unsigned kd = hof[k] >> 10;
        const int kdi = (kd == 0 || kd == 3 || kd == 5) ? -1 : ((kd == 2 || kd == 4 || kd == 7) ? 1 : 0);
        const int kdj = kd < 3 ? -1 : (kd > 4 ? 1 : 0);
        const int nt = (tj + kdj) * tiles_i + ti + kdi;
        if (hfn[k] >= bucket_end) sch.act_far(nt); else sch.act_cur(nt);
      }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_wave_barrier();
  TSA_T(t_e);
  TSA_ACC(3, t_d, t_e);
  return expanded;
}

// scheduler state of the one-workgroup-per-query kernel lives in LDS
struct TsaLocalSched {
  int* best_;
  int* state_;
  unsigned* act_cur_;
  unsigned* act_far_;
  __device__ __forceinline__ int best() const { return __hip_atomic_load(best_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ void improve_best(int g) { atomicMin(best_, g); }
  __device__ __forceinline__ void overflow() { *state_ = 4; }
  __device__ __forceinline__ void pool_exhausted() { *state_ = 5; }
  __device__ __forceinline__ void act_cur(int t) { atomicOr(&act_cur_[t >> 5], 1u << (t & 31)); }
  __device__ __forceinline__ void act_far(int t) { atomicOr(&act_far_[t >> 5], 1u << (t & 31)); }
};

__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Kernel arguments that are the same for every query of a launch.
struct TsaLaunch {
  int rows, cols, tiles_i, tiles_j, s0, s1;
  const rna_astar_query* queries;
  TsaStage S;
  int bucket_width;
  int32_t* paths;
  int max_path_len;
  int32_t* rev_all;
  int rev_cap;
  rna_astar_result* results;
};
constexpr int TSA_FOUND = -1000;        // provisional status between the search and the backtrace kernel

// One workgroup of 16 wavefronts per query.  The kernel ends with the exact distance field in HBM and a provisional
// result; the path is traced by tsa_backtrace_kernel (one wavefront per query, next kernel on the stream), which
// needs none of this kernel's LDS and therefore overlaps with the searches of the other batches in flight.
// (Several queries per workgroup -- the union of their active tiles as one job list per round -- was measured again
// on the paged fields: wavefronts spend 83 % instead of 53 % of their life inside jobs, but four relaxing wavefronts
// per SIMD already fill its VALU issue slots, every job gets slower by the same factor and the kernel needs 26 more
// VGPRs, which starves the map-update kernels: 20.9 k / 18.0 k / 13.6 k cycles/s for 1 / 2 / 4 queries per workgroup.)
__global__ void __launch_bounds__(TSA_THREADS) tsa_search_kernel(const TsaLaunch A) {
  __shared__ TsaWave s_w[TSA_WAVES];
  extern __shared__ unsigned s_dyn[];   // tile bitsets, sized by the launch: 2 x ((ntile + 31) / 32) words
  __shared__ unsigned short s_jobs[TSA_JOBS];
  __shared__ unsigned s_jobpg[TSA_JOBS];   // local page of each job's tile (looked up once per round by the list builders)
  __shared__ int s_njobs, s_first_fail, s_job_next, s_best, s_state, s_bucket, s_bucket0, s_rounds, s_role, s_expanded, s_nalloc;

  // Workgroups take their query when they START (a ticket), not by blockIdx: the hardware deals workgroup
  // indices round-robin to the XCDs, so a fixed mapping lets one XCD with several long searches hold back
  // its share of the batch while the other XCDs idle.  With tickets a free CU anywhere takes the next
  // (longest remaining) query.
  __shared__ int s_q;
  if (threadIdx.x == 0) s_q = A.S.perm[atomicAdd(A.S.ticket, 1)];
  __syncthreads();
  const int q = s_q;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rows = A.rows, cols = A.cols, tiles_i = A.tiles_i, tiles_j = A.tiles_j;
  rna_astar_query qu = A.queries[q];   // buffer linear indices; the search itself runs in map space
  const int ncell = rows * cols;
  const int ntile = tiles_i * tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  unsigned* const s_act[2] = {s_dyn, s_dyn + nt_words};   // [0] current bucket (next round), [1] next bucket
  rna_astar_result* const results = A.results;

  const bool valid = qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell;
  if (!valid) {
    if (tid == 0) results[q] = rna_astar_result{2, 0, INF, 0, 0, 0};
    return;
  }
  qu.start = tsa_unwrap_lin(qu.start, rows, cols, A.s0, A.s1);
  qu.goal = tsa_unwrap_lin(qu.goal, rows, cols, A.s0, A.s1);
  const int si = qu.start % rows, sj = qu.start / rows;
  const int gi = qu.goal % rows, gj = qu.goal / rows;

  TsaCtx C;
  C.rows = rows; C.cols = cols; C.tiles_i = tiles_i; C.tiles_j = tiles_j;
  C.pages = A.S.pages; C.ppend = A.S.ppend;
  C.tmap = A.S.tmap + (size_t)q * ntile;
  C.owner = A.S.owner + (size_t)q * (A.S.cap + 1);
  C.page_base = (size_t)q * A.S.cap;
  C.cap = A.S.cap;
  C.nalloc = &s_nalloc;
  C.nbr_tm = A.S.nbr_tm;
  C.gi = gi; C.gj = gj;

  for (int w = tid; w < nt_words; w += TSA_THREADS) { s_act[0][w] = 0u; s_act[1][w] = 0u; }
  // a goal without a single traversable neighbour cannot be reached (blocked or walled in); nothing
  // has been written yet, so no page is in use
  if (qu.goal != qu.start && C.nbr_tm[tm_index(gi, gj, tiles_i)] == 0) {
    if (tid == 0) results[q] = rna_astar_result{1, 0, INF, 0, 0, 0};
    return;
  }
  if (tid == 0) {
    s_best = INF; s_state = 0; s_rounds = 0; s_role = 0; s_expanded = 0;
    s_bucket = tsa_octile(si, sj, gi, gj) / A.bucket_width;
    s_bucket0 = s_bucket;
    // the start tile takes local page 1: g(start) = 0, pending in the current bucket
    const int ts = tile_of(si, sj, tiles_i);
    s_nalloc = 1;
    C.owner[1] = (unsigned)ts;
    __hip_atomic_store(&C.tmap[ts], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&C.pages[(C.gpage(1u) << 10) + in_page(si, sj)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicOr(&C.ppend[C.gpage(1u) * PEND_WORDS + (sj & 31)], 1u << (si & 31));
  }
  __syncthreads();
  if (tid == 0) {
    const int ts = tile_of(si, sj, tiles_i);
    s_act[0][ts >> 5] = 1u << (ts & 31);
  }
  __syncthreads();

  TsaWave& W = s_w[wv];
  int my_expanded = 0;
#ifdef RNA_TSA_STATS
  unsigned long long tsa_acc[16] = {};
  const unsigned long long t_life0 = wall_clock64();
#endif
  TsaLocalSched sch{&s_best, &s_state, s_act[0], s_act[1]};

  for (;;) {
    // ---- build this round's job list from the active-tile bitset ----
    TSA_T(t_r0);
    if (tid == 0) { s_njobs = 0; s_job_next = 0; s_first_fail = TSA_JOBS; }
    __syncthreads();
    for (int w = tid; w < nt_words; w += TSA_THREADS) {
      unsigned bits = s_act[0][w];
      if (!bits) continue;
      const int cnt = __popc(bits);
      const int base = atomicAdd(&s_njobs, cnt);
      if (base + cnt <= TSA_JOBS) {
        s_act[0][w] = 0u;
        int k = base;
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          const int t = (w << 5) + b;
          s_jobs[k] = (unsigned short)t;
          s_jobpg[k] = ld_l2(&C.tmap[t]);   // final: a tile is activated only after its page was published
          ++k;
        }
      } else {
        atomicMin(&s_first_fail, base);   // job list full: these tiles stay flagged for the next round
      }
    }
    __syncthreads();
    const int njobs = s_njobs < s_first_fail ? s_njobs : s_first_fail;
#ifdef RNA_TSA_STATS
    if (tid == 0 && njobs > 0) {   // rounds by size: slots 16..21 = 1-4, 5-8, 9-16, 17-32, 33-64, 65+ jobs; 22 = rounds
      const int b = njobs <= 4 ? 0 : (njobs <= 8 ? 1 : (njobs <= 16 ? 2 : (njobs <= 32 ? 3 : (njobs <= 64 ? 4 : 5))));
      atomicAdd(&g_tsa_stat[16 + b], 1ull);
      atomicAdd(&g_tsa_stat[24 + b], (unsigned long long)njobs);
    }
#endif
    if (njobs == 0) {
      // bucket k is at its fixed point: every cell with f < (k+1)*B has its exact g
      __syncthreads();
      if (tid == 0) {
        const long long done_below = ((long long)s_bucket + 1) * A.bucket_width;
        if (s_best != INF && (long long)s_best < done_below) s_state = 1;
        else s_state = -1;  // try the next bucket
      }
      __syncthreads();
      if (s_state == 1) break;
      // advance: tiles with next-bucket pending cells become the active set
      int any = 0;
      for (int w = tid; w < nt_words; w += TSA_THREADS) {
        const unsigned b = s_act[1][w];
        s_act[0][w] = b;
        s_act[1][w] = 0u;
        any |= (b != 0u);
      }
      any = __syncthreads_or(any);
      if (tid == 0) {
        if (!any) s_state = (s_best != INF) ? 1 : 2;   // nothing left anywhere
        else { s_state = 0; s_bucket += 1; s_role ^= 1; }
      }
      __syncthreads();
      if (s_state != 0) break;
      continue;
    }

    const int role = s_role;
    const long long bucket_end = ((long long)s_bucket + 1) * A.bucket_width;

    // ---- tile jobs: one wavefront per job ----
    TSA_T(t_r1);
    TSA_ACC(4, t_r0, t_r1);   // list build + barriers of this round
    TSA_CNT(6, 1);            // rounds (per wave)
    for (;;) {
      int job = 0;
      if (lane == 0) job = atomicAdd(&s_job_next, 1);
      job = __builtin_amdgcn_readfirstlane(job);   // wave-uniform: the loop is a scalar branch
      if (job >= njobs) break;
      const int t = s_jobs[job];
      const unsigned pg = s_jobpg[job];
      TSA_CNT(7, 1);
      my_expanded += tsa_job(sch, W, lane, t, pg, C, role, bucket_end TSA_ACC_ARG);
    }
    // all stores / atomics of this round are performed before any wave loads tiles in the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int stop = __syncthreads_or(s_state >= 4);
    if (tid == 0) s_rounds += 1;
    if (stop) break;
  }
  atomicAdd(&s_expanded, my_expanded);
  __syncthreads();
#ifdef RNA_TSA_STATS
  tsa_acc[5] = wall_clock64() - t_life0;   // wave lifetime inside the search loop
  if (lane == 0) for (int k = 0; k < 16; ++k) atomicAdd(&g_tsa_stat[k], tsa_acc[k]);
#endif
  // what the next launch on this stage has to reset
  if (tid == 0) A.S.nalloc[q] = s_nalloc < C.cap ? s_nalloc : C.cap;

  // provisional result: the backtrace kernel (next on the stream) finishes the found ones
  if (tid == 0) {
    const int state = s_state;
    results[q] = rna_astar_result{state == 1 ? TSA_FOUND : (state >= 4 ? state : 1), 0, state == 1 ? s_best : INF, s_expanded, s_rounds,
                                  s_bucket - s_bucket0 + 1};
  }
}

// Canonical backtrace, one wavefront per query: walk from the goal to the neighbour n with g[n] + w(n, c) == g[c],
// lowest linear index first (lane k probes neighbour k).  The walk runs in LDS: the 32 x 32 tile of the current
// cell plus its halo ring and the tile's neighbour masks are loaded once, then every step is one LDS round trip
// until the path leaves the tile's interior -- a path of 2 000 cells is ~70 tile loads instead of 2 000 dependent
// HBM round trips (which held a whole CU for 2-4 ms per query while the search kernel still did this itself).
__global__ void __launch_bounds__(64) tsa_backtrace_kernel(const TsaLaunch A) {
  __shared__ unsigned s_tile[TW * TW];
  __shared__ unsigned char s_mask[TILE_WORDS];
  const int q = blockIdx.x, lane = threadIdx.x;
  const rna_astar_result r = A.results[q];
  if (r.status != TSA_FOUND) return;
  const int rows = A.rows, cols = A.cols, tiles_i = A.tiles_i, tiles_j = A.tiles_j;
  const int ncell = rows * cols, ntile = tiles_i * tiles_j;
  const unsigned* tmap = A.S.tmap + (size_t)q * ntile;
  const size_t page_base = (size_t)q * A.S.cap;
  const int start = tsa_unwrap_lin(A.queries[q].start, rows, cols, A.s0, A.s1);
  const int goal = tsa_unwrap_lin(A.queries[q].goal, rows, cols, A.s0, A.s1);
  const int si = start % rows, sj = start / rows;
  int ci = goal % rows, cj = goal / rows;
  int* rev = A.rev_all + (size_t)q * A.rev_cap;
  const int k = lane & 7;
  const int wk = (k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D;
  const int di = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0);
  const int dj = k < 3 ? -1 : (k > 4 ? 1 : 0);
  const int off = di + dj * TW;
  int len = 0;
  bool ok = true, done = false;
  while (ok && !done) {
    // ---- tile of the current cell + halo ring + masks -> LDS ----
    const int ti = ci >> 5, tj = cj >> 5;
    const int t = tj * tiles_i + ti;
    unsigned pgl = 0u;   // lane k < 8: page of the neighbouring tile in direction k; lane 8: page of this tile
    {
      const int nti = lane < 8 ? ti + di : ti, ntj = lane < 8 ? tj + dj : tj;
      if (lane < 9 && nti >= 0 && ntj >= 0 && nti < tiles_i && ntj < tiles_j) pgl = ld_l2(&tmap[ntj * tiles_i + nti]);
      if (pgl == TSA_BUSY) pgl = 0u;
    }
    auto page = [&](int src) -> const unsigned* {
      const unsigned lp = (unsigned)__shfl((int)pgl, src);
      return A.S.pages + ((lp ? page_base + lp : (size_t)0) << 10);
    };
    {
      const int h = lane & 31;
      const bool second = lane >= 32;
      const unsigned* pcol = page(second ? 6 : 1);
      const unsigned* prow = page(second ? 4 : 3);
      const unsigned* pcor = page((lane & 2) ? ((lane & 1) ? 7 : 5) : ((lane & 1) ? 2 : 0));
      const unsigned* pown = page(8);
      const int cdi = (lane & 1) ? 1 : -1, cdj = (lane & 2) ? 1 : -1;
      const unsigned hv_col = ld_l2(&pcol[((second ? 0 : 31) << 5) + h]);
      const unsigned hv_row = ld_l2(&prow[(h << 5) + (second ? 0 : 31)]);
      unsigned hv_cor = 0xFFFFFF00u;
      if (lane < 4) hv_cor = ld_l2(&pcor[((cdj > 0 ? 0 : 31) << 5) + (cdi > 0 ? 0 : 31)]);
      unsigned tv[TILE_WORDS / 64];
#pragma unroll
      for (int rr = 0; rr < TILE_WORDS / 64; ++rr) tv[rr] = ld_l2(&pown[rr * 64 + lane]);
      const uint4 mv = *reinterpret_cast<const uint4*>(A.S.nbr_tm + ((size_t)t << 10) + lane * 16);
      __builtin_amdgcn_wave_barrier();   // the previous tile's walk has finished reading the LDS image
      s_tile[(second ? TS + 1 : 0) * TW + h + 1] = hv_col;
      s_tile[(h + 1) * TW + (second ? TS + 1 : 0)] = hv_row;
      if (lane < 4) s_tile[(cdj > 0 ? TS + 1 : 0) * TW + (cdi > 0 ? TS + 1 : 0)] = hv_cor;
      unsigned* tp = &s_tile[((lane >> 5) + 1) * TW + (lane & 31) + 1];
#pragma unroll
      for (int rr = 0; rr < TILE_WORDS / 64; ++rr) tp[rr * 2 * TW] = tv[rr];
      *reinterpret_cast<uint4*>(&s_mask[lane * 16]) = mv;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- walk inside the tile ----
    int il = ci & 31, jl = cj & 31;
    unsigned gc = s_tile[(jl + 1) * TW + il + 1] >> 8;
    for (;;) {
      if (lane == 0 && len < A.rev_cap) rev[len] = cj * rows + ci;
      ++len;
      if (ci == si && cj == sj) { done = true; break; }
      if (len > ncell || gc == G_INF) { ok = false; break; }
      const unsigned mc = s_mask[jl * TS + il];
      const unsigned gn = s_tile[(jl + 1) * TW + il + 1 + off] >> 8;
      const bool hit = lane < 8 && ((mc >> k) & 1u) && gn != G_INF && gn + (unsigned)wk == gc;
      const unsigned long long m = __ballot(hit);
      if (!m) { ok = false; break; }
      const int src = __ffsll((long long)m) - 1;
      gc = (unsigned)__shfl((int)gn, src);
      const int sdi = (src == 0 || src == 3 || src == 5) ? -1 : ((src == 2 || src == 4 || src == 7) ? 1 : 0);
      const int sdj = src < 3 ? -1 : (src > 4 ? 1 : 0);
      ci += sdi; cj += sdj; il += sdi; jl += sdj;
      if (il < 0 || jl < 0 || il >= TS || jl >= TS) break;   // left the interior: load that tile
    }
  }
  if (!ok) {
    if (lane == 0) A.results[q] = rna_astar_result{1, 0, INF, r.expanded, r.rounds, r.buckets};
    return;
  }
  if (len > A.max_path_len || len > A.rev_cap) {
    if (lane == 0) A.results[q] = rna_astar_result{3, len, r.cost, r.expanded, r.rounds, r.buckets};
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  int32_t* path = A.paths + (size_t)q * A.max_path_len;
  for (int i = lane; i < len; i += 64) path[i] = tsa_buffer_lin(rev[len - 1 - i], rows, cols, A.s0, A.s1);
  if (lane == 0) A.results[q] = rna_astar_result{0, len, r.cost, r.expanded, r.rounds, r.buckets};
}

// |{n : g(n) + h(n) <= f*}| per query from the pages still resident in HBM (measurement utility)
__global__ void tsa_settled_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries,
                                   const rna_astar_result* __restrict__ results, TsaStage S, int32_t* __restrict__ counts, int s0, int s1) {
  __shared__ int s_cnt;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const rna_astar_result r = results[q];
  int cnt = 0;
  if (r.status == 0 || r.status == 3) {
    const int goal = tsa_unwrap_lin(queries[q].goal, rows, cols, s0, s1);
    const int gi = goal % rows, gj = goal / rows;
    const int used = S.nalloc[q];
    for (size_t w = threadIdx.x; w < ((size_t)used << 10); w += blockDim.x) {
      const int p = 1 + (int)(w >> 10), l = (int)(w & 1023);
      const int t = (int)S.owner[(size_t)q * (S.cap + 1) + p];
      const int i = (t % tiles_i) * TS + (l & 31), j = (t / tiles_i) * TS + (l >> 5);
      const unsigned gv = S.pages[(((size_t)q * S.cap + p) << 10) + l] >> 8;
      if (gv != G_INF && i < rows && j < cols && (int)gv + tsa_octile(i, j, gi, gj) <= r.cost) ++cnt;
    }
  }
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[q] = s_cnt;
}

// ---- host entry points used by astar.hip ----
static inline int tsa_ntile(const rna_engine* e) {
  return ((e->geom.size[0] + TS - 1) / TS) * ((e->geom.size[1] + TS - 1) / TS);
}
bool tsa_supported(const rna_engine* e) { return (size_t)tsa_ntile(e) <= (size_t)TSA_MAX_TILE_WORDS * 32; }
int tsa_tiles(const rna_engine* e) { return tsa_ntile(e); }
// HBM of one pipeline stage: pages and pending bitmaps (cap per query + the shared page 0) ...
size_t tsa_pool_bytes(int max_queries, int cap) { return ((size_t)max_queries * cap + 1) * (TILE_WORDS + PEND_WORDS) * sizeof(unsigned); }
// ... and one allocation that must start zeroed: ticket | nalloc | perm | mask snapshot | tmap | owner
size_t tsa_aux_bytes(const rna_engine* e, int max_queries, int cap) {
  const size_t ntile = (size_t)tsa_ntile(e);
  return 256 + 2 * tsa_align256((size_t)max_queries * sizeof(int)) + tsa_align256(ntile * TILE_WORDS) +
         tsa_align256((size_t)max_queries * ntile * sizeof(unsigned)) + tsa_align256((size_t)max_queries * ((size_t)cap + 1) * sizeof(unsigned));
}
static TsaStage tsa_stage_view(const rna_engine* e, int slot) {
  const AstarDevice& a = e->astar;
  const size_t ntile = (size_t)tsa_ntile(e);
  char* base = static_cast<char*>(a.tsa_aux[slot]);
  TsaStage S;
  S.cap = a.page_cap;
  S.pages = reinterpret_cast<unsigned*>(a.g[slot]);
  S.ppend = S.pages + (((size_t)a.max_queries * S.cap + 1) << 10);
  S.ticket = reinterpret_cast<int*>(base);
  base += 256;
  S.nalloc = reinterpret_cast<int*>(base);
  base += tsa_align256((size_t)a.max_queries * sizeof(int));
  S.perm = reinterpret_cast<int*>(base);
  base += tsa_align256((size_t)a.max_queries * sizeof(int));
  S.nbr_tm = reinterpret_cast<uint8_t*>(base);
  base += tsa_align256(ntile * TILE_WORDS);
  S.tmap = reinterpret_cast<unsigned*>(base);
  base += tsa_align256((size_t)a.max_queries * ntile * sizeof(unsigned));
  S.owner = reinterpret_cast<unsigned*>(base);
  return S;
}
// fresh stage: pages "unreached" (everything else was zeroed by the caller's hipMemsetAsync)
int tsa_stage_prepare(rna_engine* e, int slot) {
  const TsaStage S = tsa_stage_view(e, slot);
  const size_t pages = (size_t)e->astar.max_queries * S.cap + 1;
  RNA_HIP(e, hipMemsetAsync(S.ppend, 0, pages * PEND_WORDS * sizeof(unsigned), e->stream));
  hipLaunchKernelGGL(tsa_fill_pages_kernel, dim3(8192), dim3(256), 0, e->stream, reinterpret_cast<uint4*>(S.pages), pages * TILE_WORDS / 4);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

int tsa_launch(rna_engine* e, int slot, hipStream_t init_stream, hipStream_t search_stream, hipEvent_t ev_init,
               const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len, rna_astar_result* res_dev) {
  AstarDevice& a = e->astar;
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  const TsaStage S = tsa_stage_view(e, slot);
  {
    // snapshot the neighbour masks and bring the pages the last search on this stage used back to "unreached"
    KernelTimer kt(e, RNA_K_ASTAR_INIT, init_stream);
    hipLaunchKernelGGL(tsa_snapshot_kernel, dim3(2048), dim3(256), 0, init_stream, e->nbr, rows, cols, ti, tj, S.nbr_tm,
                       e->geom.start[0], e->geom.start[1]);
    hipLaunchKernelGGL(tsa_reset_kernel, dim3(128, a.max_queries), dim3(256), 0, init_stream, S, ti * tj);   // ~5 pages per block at 4096^2
    hipLaunchKernelGGL(tsa_reset_done_kernel, dim3(1), dim3(256), 0, init_stream, S, a.max_queries);
    if (n <= 2048)   // the ranking is O(n^2 / 256) per thread: beyond this the caller order is kept
      // one small workgroup: it has to fit next to the search workgroups that fill every CU when batches are pipelined
      hipLaunchKernelGGL(tsa_order_kernel, dim3(1), dim3(256), (size_t)n * sizeof(int), init_stream, q_dev, n, rows, cols, S.perm);
    else
      hipLaunchKernelGGL(tsa_identity_order_kernel, dim3((n + 255) / 256), dim3(256), 0, init_stream, n, S.perm);
    RNA_HIP(e, hipGetLastError());
  }
  if (ev_init) {
    RNA_HIP(e, hipEventRecord(ev_init, init_stream));
    RNA_HIP(e, hipStreamWaitEvent(search_stream, ev_init, 0));
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_SEARCH, search_stream);
    const size_t nt_bytes = (size_t)((ti * tj + 31) / 32) * sizeof(unsigned);
    TsaLaunch A;
    A.rows = rows; A.cols = cols; A.tiles_i = ti; A.tiles_j = tj; A.s0 = e->geom.start[0]; A.s1 = e->geom.start[1];
    A.queries = q_dev; A.S = S; A.bucket_width = a.bucket_width; A.paths = paths_dev; A.max_path_len = max_len;
    A.rev_all = a.rev[slot]; A.rev_cap = a.rev_cap; A.results = res_dev;
    hipLaunchKernelGGL(tsa_search_kernel, dim3(n), dim3(TSA_THREADS), 2 * nt_bytes, search_stream, A);
    hipLaunchKernelGGL(tsa_backtrace_kernel, dim3(n), dim3(64), 0, search_stream, A);
    RNA_HIP(e, hipGetLastError());
  }
  return RNA_OK;
}

#ifdef RNA_TSA_STATS
void tsa_stats_dump() {
  unsigned long long st[32];
  if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tsa_stat), sizeof(st)) != hipSuccess) return;
  {
    double rounds = 0, jj = 0;
    for (int b = 0; b < 6; ++b) { rounds += (double)st[16 + b]; jj += (double)st[24 + b]; }
    if (rounds > 0)
      fprintf(stderr, "[tsa stats] rounds by jobs 1-4: %.1f%% (%.1f%% of jobs)  5-8: %.1f%% (%.1f%%)  9-16: %.1f%% (%.1f%%)  17-32: %.1f%% (%.1f%%)  33-64: %.1f%% (%.1f%%)  65+: %.1f%% (%.1f%%)\n",
              100 * st[16] / rounds, 100 * st[24] / jj, 100 * st[17] / rounds, 100 * st[25] / jj, 100 * st[18] / rounds, 100 * st[26] / jj,
              100 * st[19] / rounds, 100 * st[27] / jj, 100 * st[20] / rounds, 100 * st[28] / jj, 100 * st[21] / rounds, 100 * st[29] / jj);
  }
  const double jobs = (double)st[7];
  if (jobs <= 0) return;
  const double busy = (double)(st[0] + st[1] + st[2] + st[3]);
  fprintf(stderr, "[tsa stats] expansions %.0f, of which repeated inside the same job %.0f (%.1f%%)\n", (double)st[15], (double)st[14], 100.0 * (double)st[14] / (double)std::max<unsigned long long>(1, st[15]));
  fprintf(stderr, "[tsa stats, tile kernel, all launches] jobs %.0f | per job us: load %.2f relax %.2f wb %.2f handover %.2f | wave lifetime %.1f wave-ms, in jobs %.1f wave-ms (%.1f%%), in round set-up %.1f wave-ms (%.1f%%, %.2f us per round) | relax iters/job %.1f cells/iter %.1f\n",
          jobs, st[0] * 0.01 / jobs, st[1] * 0.01 / jobs, st[2] * 0.01 / jobs, st[3] * 0.01 / jobs, st[5] * 1e-5, busy * 1e-5,
          100.0 * busy / (double)st[5], st[4] * 1e-5, 100.0 * (double)st[4] / (double)st[5], st[4] * 0.01 / (double)std::max<unsigned long long>(1, st[6]),
          st[8] / jobs, st[9] / (double)std::max<unsigned long long>(1, st[8]));
}
#endif

int tsa_settled(rna_engine* e, int slot, const rna_astar_query* q, const rna_astar_result* r, int n, int32_t* d_counts) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  hipLaunchKernelGGL(tsa_settled_kernel, dim3(n), dim3(1024), 0, e->stream, rows, cols, ti, tj, q, r, tsa_stage_view(e, slot),
                     d_counts, e->geom.start[0], e->geom.start[1]);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

}  // namespace rna
