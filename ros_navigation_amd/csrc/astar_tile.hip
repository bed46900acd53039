// astar_tile.hip -- tile-synchronous grid A* for gfx950 ("TSA"): the same contract and the same
// label-correcting argument as astar.hip, but the relaxation runs inside LDS.  This file holds the
// default search kernel of the engine and its variants:
//   tsa_search_kernel<false>  one workgroup (16 wavefronts) per query, rounds of tile jobs   (default)
//   tsa_search_kernel<true>   the same without round barriers (RNA_ASTAR_KERNEL=async, experiment)
//   tsa_multi_kernel<QB>      QB queries share a workgroup (RNA_TSA_QUERIES_PER_BLOCK, experiment)
//   tsa_persist_kernel        every wavefront of the GPU pulls (query, tile) jobs from per-XCD rings
//                             (RNA_ASTAR_KERNEL=persist: lowest latency of a single batch)
// All of them run the same tile job (tsa_job) and pass the same parity tests.
//
// Why tiles: the frontier kernel of astar.hip pays two HBM/L2 round trips plus two workgroup barriers
// per search GENERATION (~5 us), and a long query has >10^4 generations.  Here the search field is
// stored TILE-MAJOR (32 x 32 cells = 4 KiB per tile, word = g << 8) and a wavefront owns one tile at a
// time:
//   1. grab-and-clear the tile's pending bits (cells improved since its last visit),
//   2. load the tile + a one-cell halo (and the tile's neighbour masks) into LDS with coalesced loads,
//   3. relax to the tile-local fixed point of the current f-bucket entirely in LDS: a wave-synchronous
//      queue, one lane per popped cell, plain LDS reads and writes (no LDS atomics), ~1 us per generation,
//   4. write the tile back (inner cells: coalesced stores; edge ring and improved halo cells:
//      atomicMin, because neighbouring tiles may be in flight on other wavefronts), hand improved
//      halo cells to their tiles as pending bits and activate those tiles.
// Two pending bitmaps per query (current bucket / next bucket) replace the frontier queues, so nothing
// can overflow.  Fields are never initialised wholesale: each search records the tiles it wrote and
// the next launch on that pipeline stage resets exactly those (tsa_init_kernel).  Exactness: every
// update is a min over lengths of real paths and the schedule runs every bucket to its fixed point, so
// at termination g is exact for f <= f*, which is all the canonical backtrace reads (DESIGN.md "Grid A*
// contract").  Measurements, the ceiling of the design and the experiments that were dropped: DESIGN.md 5.
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
#define RNA_TSA_JOBS 4096
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
// word index of cell (i, j) in a tile-major field
__device__ __forceinline__ size_t tm_index(int i, int j, int tiles_i) {
  return ((size_t)((j >> 5) * tiles_i + (i >> 5)) << 10) + ((j & 31) << 5) + (i & 31);
}

// Per pipeline stage, next to the fields: a tile-major snapshot of the neighbour masks (taken at
// launch, so a later map update cannot disturb a search in flight), one touched-tile bitset per query,
// and the `clean` flag.
//
// Invariant between launches: every field word is (G_INF << 8) and every pending word is 0, EXCEPT in
// the tiles flagged in touched[q].  A search flags every tile it wrote (its jobs and the tiles it
// handed cells to); the next launch on the same stage resets exactly those tiles -- typically a few
// per cent of the map -- instead of rewriting 64 MiB per query.  clean == 0 (fresh allocation, or a
// scheduler abort) makes the next launch rewrite everything.
struct TsaAux {
  int* clean;
  uint8_t* nbr_tm;      // [ntile][32][32] neighbour masks, 0 outside the map
  unsigned* touched;    // [max_queries][nt_words]
  int* perm;            // [max_queries] launch order of this batch: the k-th workgroup to START serves query perm[k]
  int* ticket;          // next position of perm to hand out (reset by every launch's init)
};
__host__ __device__ inline size_t tsa_align256(size_t x) { return (x + 255) & ~(size_t)255; }

__global__ void tsa_init_kernel(const uint8_t* __restrict__ nbr, int rows, int cols, int tiles_i, int tiles_j,
                                unsigned* __restrict__ field, size_t field_stride, unsigned* __restrict__ pend,
                                size_t pend_stride, int max_queries, TsaAux aux, int s0, int s1) {
  const int ntile = tiles_i * tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  const size_t nw = (size_t)ntile * TILE_WORDS;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (a) snapshot of the neighbour masks in tile-major MAP-space order (4 cells per thread)
  for (size_t w4 = gid; w4 < nw / 4; w4 += step) {
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
    reinterpret_cast<unsigned*>(aux.nbr_tm)[w4] = v;
  }
  if (*aux.clean == 0) {
    // (b1) everything: fields to "unreached", pending bitmaps and touched bitsets to zero
    for (int q = 0; q < max_queries; ++q) {
      uint4* f4 = reinterpret_cast<uint4*>(field + (size_t)q * field_stride);
      for (size_t w4 = gid; w4 < nw / 4; w4 += step) f4[w4] = make_uint4(0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u);
    }
    for (size_t w = gid; w < (size_t)max_queries * pend_stride; w += step) pend[w] = 0u;
    for (size_t w = gid; w < (size_t)max_queries * nt_words; w += step) aux.touched[w] = 0u;
    return;
  }
  // (b2) only the tiles the previous launch on this stage wrote: one block per (query, bitset word) pair
  const size_t pend_words = (size_t)ntile * TS;
  for (size_t item = blockIdx.x; item < (size_t)max_queries * nt_words; item += gridDim.x) {
    unsigned bits = aux.touched[item];
    if (!bits) continue;
    const int q = (int)(item / nt_words), w = (int)(item % nt_words);
    unsigned* fq = field + (size_t)q * field_stride;
    unsigned* pq = pend + (size_t)q * pend_stride;
    while (bits) {
      const int t = (w << 5) + (__ffs(bits) - 1);
      bits &= bits - 1;
      reinterpret_cast<uint4*>(fq + ((size_t)t << 10))[threadIdx.x] = make_uint4(0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u, 0xFFFFFF00u);
      if (threadIdx.x < 2 * TS) pq[(size_t)(threadIdx.x >> 5) * pend_words + (size_t)t * TS + (threadIdx.x & 31)] = 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) aux.touched[item] = 0u;
  }
}
__global__ void tsa_mark_clean_kernel(int* clean, int* ticket) { *clean = 1; *ticket = 0; }

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
#ifdef RNA_TSA_STATS_REEXP
  unsigned char seen[TW * TW + TSA_SCRATCH];   // developer build: cell already expanded in this job
#endif
};

// One tile job, executed by one wavefront (lane = this wave's lane id).  `sch` supplies the
// scheduler-specific pieces: best() / improve_best(g) (upper bound on f*), act_cur(tile) /
// act_far(tile) (a tile received pending cells for the current / the next bucket), overflow().
// Returns the number of cell expansions.
template <class Sched>
__device__ __forceinline__ int tsa_job(Sched& sch, TsaWave& W, const int lane, const int t, const int rows, const int cols,
                                       const int tiles_i, const int tiles_j, unsigned* __restrict__ field,
                                       const uint8_t* __restrict__ nbr_tm, unsigned* __restrict__ pend_cur, unsigned* __restrict__ pend_far,
                                       const long long bucket_end, const int gi, const int gj TSA_ACC_PARAM) {
  int expanded = 0;
  const int ti = t % tiles_i, tj = t / tiles_i;
  const int i0 = ti * TS, j0 = tj * TS;
  unsigned* ftile = field + ((size_t)t << 10);

  TSA_T(t_a);
  // 1. grab-and-clear the pending bits of this tile (lane = column jl)
  unsigned seed = 0u;
  if (lane < TS) seed = atomicExch(&pend_cur[(size_t)t * TS + lane], 0u);
  // 2. tile + halo -> LDS (after the pending bits: every grabbed bit's value is already in L2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    // Everything a job reads from HBM is issued before the first wait: the three halo words of a lane (neighbour
    // column, neighbour row, corner), its 16 tile words (cell r*64 + lane sits 2r rows below cell `lane`: one base
    // address + compile-time offsets) and its 16 mask bytes -- ONE memory round trip after the pending bits.  (Until
    // the lane arithmetic of the later phases was cut loose from this one, see lane_b below, the kernel had no
    // registers for that and loaded in two batches of eight plus the halo: three round trips.)
    const int h = lane & 31;
    const bool second = lane >= 32;
    // halo columns: tile (ti, tj-1) column 31 -> LDS jl=-1 ; tile (ti, tj+1) column 0 -> LDS jl=32
    // halo rows:    tile (ti-1, tj) row 31 -> LDS il=-1 ; tile (ti+1, tj) row 0 -> LDS il=32 ; then the 4 corners
    const int ntj = second ? tj + 1 : tj - 1;
    const int nti = second ? ti + 1 : ti - 1;
    unsigned hv_col = 0xFFFFFF00u, hv_row = 0xFFFFFF00u, hv_cor = 0xFFFFFF00u;
    if (ntj >= 0 && ntj < tiles_j) hv_col = ld_l2(&field[((size_t)(ntj * tiles_i + ti) << 10) + ((second ? 0 : 31) << 5) + h]);
    if (nti >= 0 && nti < tiles_i) hv_row = ld_l2(&field[((size_t)(tj * tiles_i + nti) << 10) + (h << 5) + (second ? 0 : 31)]);
    const int cdi = (lane & 1) ? 1 : -1, cdj = (lane & 2) ? 1 : -1;
    if (lane < 4) {
      const int cti = ti + cdi, ctj = tj + cdj;
      if (cti >= 0 && cti < tiles_i && ctj >= 0 && ctj < tiles_j)
        hv_cor = ld_l2(&field[((size_t)(ctj * tiles_i + cti) << 10) + ((cdj > 0 ? 0 : 31) << 5) + (cdi > 0 ? 0 : 31)]);
    }
    unsigned* tp = &W.tile[((lane >> 5) + 1) * TW + (lane & 31) + 1];
    unsigned tv[TILE_WORDS / 64];
#pragma unroll
    for (int r = 0; r < TILE_WORDS / 64; ++r) tv[r] = ld_l2(&ftile[r * 64 + lane]);
    const uint4 mv = *reinterpret_cast<const uint4*>(nbr_tm + ((size_t)t << 10) + lane * 16);
    W.tile[(second ? TS + 1 : 0) * TW + h + 1] = hv_col;
    W.tile[(h + 1) * TW + (second ? TS + 1 : 0)] = hv_row;
    if (lane < 4) W.tile[(cdj > 0 ? TS + 1 : 0) * TW + (cdi > 0 ? TS + 1 : 0)] = hv_cor;
#pragma unroll
    for (int r = 0; r < TILE_WORDS / 64; ++r) tp[r * 2 * TW] = tv[r];
    *reinterpret_cast<uint4*>(&W.mask[lane * 16]) = mv;
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
      if ((lane_b & 31) == 0 && word) { atomicOr(&pend_far[(size_t)t * TS + jl], word); }
      anyfar |= bm != 0ull;
    }
    if (anyfar && lane_b == 0) sch.act_far(t);
  }
  TSA_T(t_d);
  TSA_ACC(2, t_c, t_d);
  //    improved halo cells -> their tiles: value first, then the pending bit, then the activation.  The 136 ring
  //    positions are three cells per lane; each stage is issued for all three before its single wait, so the
  //    hand-over costs two memory round trips, not two per cell.
  {
    constexpr int HK = (4 * TW + 63) / 64;
    unsigned hv[HK];
    unsigned hix[HK];
    int hfn[HK];
    bool hdo[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      const int hh = lane_b + 64 * k;
      hdo[k] = false; hv[k] = 0u; hix[k] = 0u; hfn[k] = 0;
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
            hdo[k] = true; hv[k] = v; hix[k] = (unsigned)tm_index(ni, nj, tiles_i); hfn[k] = fn;
          }
        }
      }
    }
    unsigned hold[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) hold[k] = hdo[k] ? atomicMin(&field[hix[k]], hv[k]) : 0u;
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      hdo[k] = hdo[k] && hv[k] < hold[k];
      if (hdo[k]) {
        const unsigned nt = hix[k] >> 10, l = hix[k] & 1023u;
        atomicOr(&(hfn[k] >= bucket_end ? pend_far : pend_cur)[(size_t)nt * TS + (l >> 5)], 1u << (l & 31));
      }
    }
    // the values and their pending bits must be performed at L2 before the tiles can be scheduled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < HK; ++k)
      if (hdo[k]) {
        const int nt = (int)(hix[k] >> 10);
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
  __device__ __forceinline__ void act_cur(int t) { atomicOr(&act_cur_[t >> 5], 1u << (t & 31)); }
  __device__ __forceinline__ void act_far(int t) { atomicOr(&act_far_[t >> 5], 1u << (t & 31)); }
};

// Barrier-free variant of the same kernel: the active-tile set is a bitset in LDS that every
// wavefront scans and claims from on its own; `outstanding` = active bits + running jobs, and the
// wave that brings it to zero owns the query alone and opens the next bucket (or ends the search).
struct TsaAsyncSched {
  int* best_;
  int* state_;
  unsigned* act_;
  unsigned* far_;
  int* outstanding_;
  __device__ __forceinline__ int best() const { return __hip_atomic_load(best_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ void improve_best(int g) { atomicMin(best_, g); }
  __device__ __forceinline__ void overflow() { *state_ = 4; }
  __device__ __forceinline__ void act_cur(int t) {
    const unsigned bit = 1u << (t & 31);
    if (__hip_atomic_load(&act_[t >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) return;   // already active
    // count the tile BEFORE it can be claimed: a claimer that finishes first must not see the sum at 0
    atomicAdd(outstanding_, 1);
    if (atomicOr(&act_[t >> 5], bit) & bit) atomicSub(outstanding_, 1);   // lost the race to another activator
  }
  __device__ __forceinline__ void act_far(int t) { atomicOr(&far_[t >> 5], 1u << (t & 31)); }
};
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <bool ASYNC>
__global__ void __launch_bounds__(TSA_THREADS)
tsa_search_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries,
                  unsigned* __restrict__ field_all, size_t field_stride, unsigned* __restrict__ pend_all,
                  size_t pend_stride, const uint8_t* __restrict__ nbr_tm, unsigned* __restrict__ touched_all,
                  int bucket_width, int32_t* __restrict__ paths, int max_path_len,
                  int32_t* __restrict__ rev_all, int rev_cap, rna_astar_result* __restrict__ results, int s0, int s1,
                  const int* __restrict__ perm, int* __restrict__ ticket) {
  __shared__ TsaWave s_w[TSA_WAVES];
  extern __shared__ unsigned s_dyn[];   // tile bitsets, sized by the launch: (ASYNC ? 4 : 3) x ((ntile + 31) / 32) words
  __shared__ unsigned short s_jobs[ASYNC ? 2 : TSA_JOBS];
  __shared__ int s_njobs, s_first_fail, s_job_next, s_best, s_state, s_bucket, s_bucket0, s_rounds, s_role, s_expanded, s_len;
  __shared__ int s_outstanding;

  // Workgroups take their query when they START (a ticket), not by blockIdx: the hardware deals workgroup
  // indices round-robin to the XCDs, so a fixed mapping lets one XCD with several long searches hold back
  // its share of the batch while the other XCDs idle.  With tickets a free CU anywhere takes the next
  // (longest remaining) query.
  __shared__ int s_q;
  if (threadIdx.x == 0) s_q = perm[atomicAdd(ticket, 1)];
  __syncthreads();
  const int q = s_q;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  rna_astar_query qu = queries[q];   // buffer linear indices; the search itself runs in map space
  const int ncell = rows * cols;
  const int ntile = tiles_i * tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  unsigned* const s_act[2] = {s_dyn, s_dyn + nt_words};   // [0] current bucket (next round), [1] next bucket
  unsigned* const s_touched = s_dyn + 2 * nt_words;        // every tile that ever became a job
  unsigned* const s_running = s_dyn + 3 * nt_words;        // ASYNC only: a wavefront is inside a job of this tile
  unsigned* field = field_all + (size_t)q * field_stride;
  unsigned* pend0 = pend_all + (size_t)q * pend_stride;   // two bitmaps of ntile*32 words each
  const size_t pend_words = (size_t)ntile * TS;

  const bool valid = qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell;
  if (!valid) {
    if (tid == 0) results[q] = rna_astar_result{2, 0, INF, 0, 0, 0};
    return;
  }
  qu.start = tsa_unwrap_lin(qu.start, rows, cols, s0, s1);
  qu.goal = tsa_unwrap_lin(qu.goal, rows, cols, s0, s1);
  const int si = qu.start % rows, sj = qu.start / rows;
  const int gi = qu.goal % rows, gj = qu.goal / rows;

  for (int w = tid; w < nt_words; w += TSA_THREADS) {
    s_act[0][w] = 0u; s_act[1][w] = 0u; s_touched[w] = 0u;
    if (ASYNC) s_running[w] = 0u;
  }
  // a goal without a single traversable neighbour cannot be reached (blocked or walled in); nothing
  // has been written yet, so the field stays clean
  if (qu.goal != qu.start && nbr_tm[tm_index(gi, gj, tiles_i)] == 0) {
    if (tid == 0) results[q] = rna_astar_result{1, 0, INF, 0, 0, 0};
    return;
  }
  if (tid == 0) {
    s_best = INF; s_state = 0; s_rounds = 0; s_role = 0; s_expanded = 0; s_outstanding = 1;
    s_bucket = tsa_octile(si, sj, gi, gj) / bucket_width;
    s_bucket0 = s_bucket;
    const size_t ws = tm_index(si, sj, tiles_i);
    __hip_atomic_store(&field[ws], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // g(start) = 0
    const int ts = (sj >> 5) * tiles_i + (si >> 5);
    atomicOr(&pend0[(size_t)ts * TS + (sj & 31)], 1u << (si & 31));
  }
  __syncthreads();
  if (tid == 0) {
    const int ts = (sj >> 5) * tiles_i + (si >> 5);
    s_act[0][ts >> 5] = 1u << (ts & 31);
  }
  __syncthreads();

  TsaWave& W = s_w[wv];
  int my_expanded = 0;
#ifdef RNA_TSA_STATS
  unsigned long long tsa_acc[16] = {};
  const unsigned long long t_life0 = wall_clock64();
#endif
  if constexpr (ASYNC) {
    TsaAsyncSched sch{&s_best, &s_state, s_act[0], s_act[1], &s_outstanding};
    int cursor = (int)(((long long)wv * nt_words) / TSA_WAVES);   // the waves start their sweeps at different words
    for (;;) {
      if (lds_ld(&s_state) != 0) break;
      // ---- find and claim an active tile: sweep the bitset from `cursor`, 64 words per step ----
      int t = -1;
      for (int base = 0; base < nt_words && t < 0; base += 64) {
        int w = cursor + base + lane;
        if (w >= nt_words) w -= nt_words;
        const unsigned bits = (base + lane < nt_words) ? lds_ld(&s_act[0][w]) : 0u;
        unsigned long long m = __ballot(bits != 0u);
        while (m && t < 0) {
          const int src = __ffsll((long long)m) - 1;
          m &= m - 1;
          const unsigned cb = __shfl(bits, src);
          const int cw = __shfl(w, src);
          int got = -1;
          if (lane == 0) {
            unsigned rem = cb;
            while (rem) {
              const int b = __ffs(rem) - 1;
              rem &= rem - 1;
              const unsigned bit = 1u << b;
              if (atomicOr(&s_running[cw], bit) & bit) continue;        // a job of this tile is in flight: it stays active
              if (atomicAnd(&s_act[0][cw], ~bit) & bit) { got = (cw << 5) + b; break; }   // claimed: active -> running
              atomicAnd(&s_running[cw], ~bit);                           // somebody else took it meanwhile
            }
          }
          got = __shfl(got, 0);
          if (got >= 0) t = got;
        }
      }
      if (t < 0) { __builtin_amdgcn_s_sleep(8); continue; }
      cursor = (t >> 5) + 1;
      if (cursor >= nt_words) cursor = 0;
      const int role = lds_ld(&s_role), bucket = lds_ld(&s_bucket);
      unsigned* pend_cur = pend0 + (size_t)role * pend_words;
      unsigned* pend_far = pend0 + (size_t)(role ^ 1) * pend_words;
      const long long bucket_end = ((long long)bucket + 1) * bucket_width;
      if (lane == 0) atomicOr(&s_touched[t >> 5], 1u << (t & 31));
      TSA_CNT(7, 1);
      my_expanded += tsa_job(sch, W, lane, t, rows, cols, tiles_i, tiles_j, field, nbr_tm, pend_cur, pend_far, bucket_end, gi, gj TSA_ACC_ARG);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // everything this job wrote is at L2 before the tile is released
      int left = 1;
      if (lane == 0) {
        atomicAnd(&s_running[t >> 5], ~(1u << (t & 31)));
        atomicAdd(&s_rounds, 1);
        left = atomicSub(&s_outstanding, 1) - 1;
      }
      left = __shfl(left, 0);
      if (left != 0) continue;
      // ---- this wave emptied the bucket: no tile is active or running, it owns the query alone ----
      if (lds_ld(&s_state) != 0) break;   // overflow was flagged
      const int best = lds_ld(&s_best);
      const long long done_below = ((long long)bucket + 1) * bucket_width;
      if (best != INF && (long long)best < done_below) { if (lane == 0) s_state = 1; break; }   // goal settled, ties included
      int cnt = 0;
      for (int w = lane; w < nt_words; w += 64) cnt += __popc(lds_ld(&s_act[1][w]));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
      if (cnt == 0) { if (lane == 0) s_state = (best != INF) ? 1 : 2; break; }   // nothing left anywhere
      if (lane == 0) { s_bucket = bucket + 1; s_role = role ^ 1; s_outstanding = cnt; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      for (int w = lane; w < nt_words; w += 64) {   // publish the next bucket's tiles (count and role are in place)
        const unsigned b = s_act[1][w];
        if (b) { s_act[1][w] = 0u; atomicOr(&s_act[0][w], b); }
      }
    }
  } else {
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
        s_touched[w] |= bits;
        int k = base;
        while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; s_jobs[k++] = (unsigned short)((w << 5) + b); }
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
        const long long done_below = ((long long)s_bucket + 1) * bucket_width;
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
    unsigned* pend_cur = pend0 + (size_t)role * pend_words;
    unsigned* pend_far = pend0 + (size_t)(role ^ 1) * pend_words;
    const long long bucket_end = ((long long)s_bucket + 1) * bucket_width;

    // ---- tile jobs: one wavefront per job ----
    TSA_T(t_r1);
    TSA_ACC(4, t_r0, t_r1);   // list build + barriers of this round
    TSA_CNT(6, 1);            // rounds (per wave)
    for (;;) {
      int job = 0;
      if (lane == 0) job = atomicAdd(&s_job_next, 1);
      job = __shfl(job, 0);
      if (job >= njobs) break;
      const int t = s_jobs[job];
      TSA_CNT(7, 1);
      my_expanded += tsa_job(sch, W, lane, t, rows, cols, tiles_i, tiles_j, field, nbr_tm, pend_cur, pend_far, bucket_end, gi, gj TSA_ACC_ARG);
    }
    // all stores / atomics of this round are performed before any wave loads tiles in the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int stop = __syncthreads_or(s_state == 4);
    if (tid == 0) s_rounds += 1;
    if (stop) break;
  }
  }   // !ASYNC
  atomicAdd(&s_expanded, my_expanded);
  __syncthreads();
#ifdef RNA_TSA_STATS
  tsa_acc[5] = wall_clock64() - t_life0;   // wave lifetime inside the search loop
  if (lane == 0) for (int k = 0; k < 16; ++k) atomicAdd(&g_tsa_stat[k], tsa_acc[k]);
#endif
  // every tile this search wrote: its jobs plus the tiles that were handed cells but never ran
  {
    unsigned* touched = touched_all + (size_t)q * nt_words;
    for (int w = tid; w < nt_words; w += TSA_THREADS) touched[w] = s_touched[w] | s_act[0][w] | s_act[1][w];
  }

  const int state = s_state;
  const int n_buckets = s_bucket - s_bucket0 + 1;
  if (state != 1) {
    if (tid == 0) results[q] = rna_astar_result{state == 4 ? 4 : 1, 0, INF, s_expanded, s_rounds, n_buckets};
    return;
  }

  // ---- canonical backtrace by the first wavefront (lane k probes neighbour k) ----
  int* rev = rev_all + (size_t)q * rev_cap;
  if (tid < 64) {
    int ci = gi, cj = gj;
    int len = 0;
    bool ok = true;
    const int k = tid & 7;
    const int w = (k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D;
    const int di = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0);
    const int dj = k < 3 ? -1 : (k > 4 ? 1 : 0);
    for (;;) {
      if (tid == 0 && len < rev_cap) rev[len] = cj * rows + ci;
      ++len;
      if (ci == si && cj == sj) break;
      if (len > ncell) { ok = false; break; }
      const int ni = ci + di, nj = cj + dj;
      const bool inb = ni >= 0 && nj >= 0 && ni < rows && nj < cols;
      const unsigned wc = ld_l2(&field[tm_index(ci, cj, tiles_i)]);
      const unsigned mc = nbr_tm[tm_index(ci, cj, tiles_i)];
      const unsigned wn = ld_l2(&field[inb ? tm_index(ni, nj, tiles_i) : tm_index(ci, cj, tiles_i)]);
      const bool hit = tid < 8 && inb && ((mc >> k) & 1u) && ((wn >> 8) != G_INF) && ((wn >> 8) + (unsigned)w == (wc >> 8));
      const unsigned long long mask = __ballot(hit);
      if (!mask) { ok = false; break; }
      const int src = __ffsll((long long)mask) - 1;
      ci = __shfl(ni, src);
      cj = __shfl(nj, src);
    }
    if (tid == 0) s_len = ok ? len : -1;
  }
  __syncthreads();
  const int len = s_len;
  if (len < 0) {
    if (tid == 0) results[q] = rna_astar_result{1, 0, INF, s_expanded, s_rounds, n_buckets};
    return;
  }
  if (len > max_path_len || len > rev_cap) {
    if (tid == 0) results[q] = rna_astar_result{3, len, s_best, s_expanded, s_rounds, n_buckets};
    return;
  }
  int32_t* path = paths + (size_t)q * max_path_len;
  for (int i = tid; i < len; i += TSA_THREADS) path[i] = tsa_buffer_lin(rev[len - 1 - i], rows, cols, s0, s1);
  if (tid == 0) results[q] = rna_astar_result{0, len, s_best, s_expanded, s_rounds, n_buckets};
}

// -------------------------------------------------------------------------------------------------
// Several queries per workgroup.  Profiling the kernel above shows 61 % of its rounds holding <= 16 tile
// jobs for 16 wavefronts (they carry 16 % of the jobs but 30 % of the time): a single query rarely has
// enough active tiles to keep a CU's wavefronts busy.  Here QB queries share the 16 wavefronts of one
// workgroup: every round's job list is the union of their active tiles, so thin rounds of one query are
// filled by the others.  Each query keeps its own bucket, role, bound and bitsets; a query whose current
// bucket ran dry opens its next bucket at the start of the round, independently of its neighbours.
// -------------------------------------------------------------------------------------------------
constexpr int TSA_MQ_JOBS = 2048;   // job list entries per round: (local query << 16) | tile

template <int QB>
__global__ void __launch_bounds__(TSA_THREADS)
tsa_multi_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries, int n_queries,
                 unsigned* __restrict__ field_all, size_t field_stride, unsigned* __restrict__ pend_all, size_t pend_stride,
                 const uint8_t* __restrict__ nbr_tm, unsigned* __restrict__ touched_all, int bucket_width,
                 int32_t* __restrict__ paths, int max_path_len, int32_t* __restrict__ rev_all, int rev_cap,
                 rna_astar_result* __restrict__ results, int s0, int s1, const int* __restrict__ perm) {
  __shared__ TsaWave s_w[TSA_WAVES];
  extern __shared__ unsigned s_dyn[];   // [QB][3][nt_words]: active (current bucket), active (next bucket), touched
  __shared__ unsigned s_jobs[TSA_MQ_JOBS];
  __shared__ int s_njobs, s_first_fail, s_job_next, s_running;
  // per-query state: 0 searching, 1 goal settled, 2 no path, 4 cost overflow, 5 never started (invalid / walled-in goal)
  __shared__ int s_state[QB], s_best[QB], s_bucket[QB], s_bucket0[QB], s_role[QB], s_rounds[QB], s_expanded[QB], s_len[QB];
  __shared__ int s_any[QB], s_adv[QB], s_si[QB], s_sj[QB], s_gi[QB], s_gj[QB], s_q[QB];   // s_q: global query index, -1 = none

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ncell = rows * cols;
  const int ntile = tiles_i * tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  const size_t pend_words = (size_t)ntile * TS;
  auto act = [&](int ql, int which) -> unsigned* { return s_dyn + (size_t)(ql * 3 + which) * nt_words; };

  for (int w = tid; w < QB * 3 * nt_words; w += TSA_THREADS) s_dyn[w] = 0u;
  if (tid < QB) {
    const int ql = tid;
    // queries are dealt round-robin in launch order, so each workgroup gets a mix of long and short searches
    const int slot = ql * (int)gridDim.x + (int)blockIdx.x;
    const int q = slot < n_queries ? perm[slot] : -1;
    s_q[ql] = q;
    s_best[ql] = INF; s_rounds[ql] = 0; s_role[ql] = 0; s_expanded[ql] = 0; s_bucket[ql] = 0; s_bucket0[ql] = 0; s_len[ql] = 0;
    s_state[ql] = 5;
    if (q >= 0) {
      rna_astar_query qu = queries[q];
      if (!(qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell)) {
        results[q] = rna_astar_result{2, 0, INF, 0, 0, 0};
      } else {
        qu.start = tsa_unwrap_lin(qu.start, rows, cols, s0, s1);
        qu.goal = tsa_unwrap_lin(qu.goal, rows, cols, s0, s1);
        const int si = qu.start % rows, sj = qu.start / rows, gi = qu.goal % rows, gj = qu.goal / rows;
        s_si[ql] = si; s_sj[ql] = sj; s_gi[ql] = gi; s_gj[ql] = gj;
        if (qu.goal != qu.start && nbr_tm[tm_index(gi, gj, tiles_i)] == 0) {
          results[q] = rna_astar_result{1, 0, INF, 0, 0, 0};   // walled-in goal: nothing written
        } else {
          s_state[ql] = 0;
          s_bucket[ql] = s_bucket0[ql] = tsa_octile(si, sj, gi, gj) / bucket_width;
          unsigned* field = field_all + (size_t)q * field_stride;
          unsigned* pend0 = pend_all + (size_t)q * pend_stride;
          __hip_atomic_store(&field[tm_index(si, sj, tiles_i)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // g(start) = 0
          const int ts = (sj >> 5) * tiles_i + (si >> 5);
          atomicOr(&pend0[(size_t)ts * TS + (sj & 31)], 1u << (si & 31));
        }
      }
    }
  }
  __syncthreads();
  if (tid < QB && s_state[tid] == 0) {
    const int ts = (s_sj[tid] >> 5) * tiles_i + (s_si[tid] >> 5);
    act(tid, 0)[ts >> 5] = 1u << (ts & 31);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  TsaWave& W = s_w[wv];
  int my_expanded[QB];
#pragma unroll
  for (int k = 0; k < QB; ++k) my_expanded[k] = 0;
#ifdef RNA_TSA_STATS
  unsigned long long tsa_acc[16] = {};
  const unsigned long long t_life0 = wall_clock64();
#endif

  for (;;) {
    // ---- 1. which searching queries still have tiles in their current bucket? ----
    if (tid < QB) { s_any[tid] = 0; s_adv[tid] = 0; }
    if (tid == 0) { s_njobs = 0; s_job_next = 0; s_first_fail = TSA_MQ_JOBS; s_running = 0; }
    __syncthreads();
#pragma unroll
    for (int ql = 0; ql < QB; ++ql) {
      if (s_state[ql] != 0) continue;
      const unsigned* a = act(ql, 0);
      unsigned any = 0u;
      for (int w = tid; w < nt_words; w += TSA_THREADS) any |= a[w];
      if (__ballot(any != 0u) && lane == 0) s_any[ql] = 1;
    }
    __syncthreads();
    // ---- 2. a query whose bucket ran dry is at that bucket's fixed point: finish it or open the next bucket ----
    if (tid < QB && s_state[tid] == 0 && !s_any[tid]) {
      const int ql = tid;
      const long long done_below = ((long long)s_bucket[ql] + 1) * bucket_width;
      if (s_best[ql] != INF && (long long)s_best[ql] < done_below) s_state[ql] = 1;   // goal settled, ties included
      else s_adv[ql] = 1;
    }
    __syncthreads();
#pragma unroll
    for (int ql = 0; ql < QB; ++ql) {
      if (!s_adv[ql]) continue;
      unsigned* a0 = act(ql, 0);
      unsigned* a1 = act(ql, 1);
      unsigned any = 0u;
      for (int w = tid; w < nt_words; w += TSA_THREADS) {
        const unsigned b = a1[w];
        a0[w] = b;
        a1[w] = 0u;
        any |= b;
      }
      if (__ballot(any != 0u) && lane == 0) s_any[ql] = 1;
    }
    __syncthreads();
    if (tid < QB && s_adv[tid]) {
      const int ql = tid;
      if (!s_any[ql]) s_state[ql] = (s_best[ql] != INF) ? 1 : 2;   // nothing left anywhere
      else { s_bucket[ql] += 1; s_role[ql] ^= 1; }
    }
    __syncthreads();
    // ---- 3. this round's job list: the active tiles of every searching query ----
#pragma unroll
    for (int ql = 0; ql < QB; ++ql) {
      if (s_state[ql] != 0) continue;
      if (tid == 0) s_running = 1;
      unsigned* a0 = act(ql, 0);
      unsigned* tch = act(ql, 2);
      for (int w = tid; w < nt_words; w += TSA_THREADS) {
        unsigned bits = a0[w];
        if (!bits) continue;
        const int cnt = __popc(bits);
        const int base = atomicAdd(&s_njobs, cnt);
        if (base + cnt <= TSA_MQ_JOBS) {
          a0[w] = 0u;
          tch[w] |= bits;
          int k = base;
          while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; s_jobs[k++] = ((unsigned)ql << 16) | (unsigned)((w << 5) + b); }
        } else {
          atomicMin(&s_first_fail, base);   // job list full: these tiles stay flagged for the next round
        }
      }
    }
    __syncthreads();
    if (!s_running) break;
    const int njobs = s_njobs < s_first_fail ? s_njobs : s_first_fail;
    if (tid < QB && s_state[tid] == 0) s_rounds[tid] += 1;

    // ---- 4. tile jobs: one wavefront per job ----
    for (;;) {
      int job = 0;
      if (lane == 0) job = atomicAdd(&s_job_next, 1);
      job = __shfl(job, 0);
      if (job >= njobs) break;
      const unsigned je = s_jobs[job];
      const int ql = (int)(je >> 16), t = (int)(je & 0xffffu);
      const int q = s_q[ql];
      unsigned* field = field_all + (size_t)q * field_stride;
      unsigned* pend0 = pend_all + (size_t)q * pend_stride;
      const int role = s_role[ql];
      const long long bucket_end = ((long long)s_bucket[ql] + 1) * bucket_width;
      TsaLocalSched sch{&s_best[ql], &s_state[ql], act(ql, 0), act(ql, 1)};
      TSA_CNT(7, 1);
      const int ex = tsa_job(sch, W, lane, t, rows, cols, tiles_i, tiles_j, field, nbr_tm, pend0 + (size_t)role * pend_words,
                             pend0 + (size_t)(role ^ 1) * pend_words, bucket_end, s_gi[ql], s_gj[ql] TSA_ACC_ARG);
#pragma unroll
      for (int k = 0; k < QB; ++k) my_expanded[k] += (k == ql) ? ex : 0;
    }
    // all stores / atomics of this round are performed before any wave loads tiles in the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < QB; ++k) atomicAdd(&s_expanded[k], my_expanded[k]);
  __syncthreads();
#ifdef RNA_TSA_STATS
  tsa_acc[5] = wall_clock64() - t_life0;
  if (lane == 0) for (int k = 0; k < 16; ++k) atomicAdd(&g_tsa_stat[k], tsa_acc[k]);
#endif
  // every tile a search wrote: its jobs plus the tiles that were handed cells but never ran
  for (int ql = 0; ql < QB; ++ql) {
    if (s_q[ql] < 0) continue;
    unsigned* touched = touched_all + (size_t)s_q[ql] * nt_words;
    const unsigned *a0 = act(ql, 0), *a1 = act(ql, 1), *tch = act(ql, 2);
    for (int w = tid; w < nt_words; w += TSA_THREADS) touched[w] = tch[w] | a0[w] | a1[w];
  }

  // ---- canonical backtrace: wavefront ql serves query ql (lane k probes neighbour k) ----
  if (wv < QB && s_q[wv] >= 0 && s_state[wv] != 5) {
    const int ql = wv, q = s_q[ql];
    const int state = s_state[ql];
    const int n_buckets = s_bucket[ql] - s_bucket0[ql] + 1;
    const unsigned* field = field_all + (size_t)q * field_stride;
    if (state != 1) {
      if (lane == 0) results[q] = rna_astar_result{state == 4 ? 4 : 1, 0, INF, s_expanded[ql], s_rounds[ql], n_buckets};
    } else {
      const int si = s_si[ql], sj = s_sj[ql];
      int* rev = rev_all + (size_t)q * rev_cap;
      int ci = s_gi[ql], cj = s_gj[ql];
      int len = 0;
      bool ok = true;
      const int k = lane & 7;
      const int w = (k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D;
      const int di = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0);
      const int dj = k < 3 ? -1 : (k > 4 ? 1 : 0);
      for (;;) {
        if (lane == 0 && len < rev_cap) rev[len] = cj * rows + ci;
        ++len;
        if (ci == si && cj == sj) break;
        if (len > ncell) { ok = false; break; }
        const int ni = ci + di, nj = cj + dj;
        const bool inb = ni >= 0 && nj >= 0 && ni < rows && nj < cols;
        const unsigned wc = ld_l2(&field[tm_index(ci, cj, tiles_i)]);
        const unsigned mc = nbr_tm[tm_index(ci, cj, tiles_i)];
        const unsigned wn = ld_l2(&field[inb ? tm_index(ni, nj, tiles_i) : tm_index(ci, cj, tiles_i)]);
        const bool hit = lane < 8 && inb && ((mc >> k) & 1u) && ((wn >> 8) != G_INF) && ((wn >> 8) + (unsigned)w == (wc >> 8));
        const unsigned long long mask = __ballot(hit);
        if (!mask) { ok = false; break; }
        const int src = __ffsll((long long)mask) - 1;
        ci = __shfl(ni, src);
        cj = __shfl(nj, src);
      }
      if (!ok) {
        if (lane == 0) results[q] = rna_astar_result{1, 0, INF, s_expanded[ql], s_rounds[ql], n_buckets};
      } else if (len > max_path_len || len > rev_cap) {
        if (lane == 0) results[q] = rna_astar_result{3, len, s_best[ql], s_expanded[ql], s_rounds[ql], n_buckets};
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int32_t* path = paths + (size_t)q * max_path_len;
        for (int i = lane; i < len; i += 64) path[i] = tsa_buffer_lin(rev[len - 1 - i], rows, cols, s0, s1);
        if (lane == 0) results[q] = rna_astar_result{0, len, s_best[ql], s_expanded[ql], s_rounds[ql], n_buckets};
      }
    }
  }
}

// |{n : g(n) + h(n) <= f*}| per query from the resident tile-major fields (measurement utility)
__global__ void tsa_settled_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries,
                                   const rna_astar_result* __restrict__ results, const unsigned* __restrict__ field_all,
                                   size_t field_stride, int32_t* __restrict__ counts, int s0, int s1) {
  __shared__ int s_cnt;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const rna_astar_result r = results[q];
  int cnt = 0;
  if (r.status == 0 || r.status == 3) {
    const int goal = tsa_unwrap_lin(queries[q].goal, rows, cols, s0, s1);
    const int gi = goal % rows, gj = goal / rows;
    const unsigned* field = field_all + (size_t)q * field_stride;
    const size_t nw = (size_t)tiles_i * tiles_j * TILE_WORDS;
    for (size_t w = threadIdx.x; w < nw; w += blockDim.x) {
      const int t = (int)(w >> 10), l = (int)(w & 1023);
      const int i = (t % tiles_i) * TS + (l & 31), j = (t / tiles_i) * TS + (l >> 5);
      const unsigned gv = field[w] >> 8;
      if (gv != G_INF && i < rows && j < cols && (int)gv + tsa_octile(i, j, gi, gj) <= r.cost) ++cnt;
    }
  }
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[q] = s_cnt;
}

// =================================================================================================
// Persistent, cross-CU scheduler ("TSA-P"): every wavefront of the whole GPU is a worker that pulls
// (query, tile) jobs from a queue.  A long query is therefore served by many CUs at once and the
// batch is load-balanced at tile granularity -- the one-workgroup-per-query kernels above finish
// only when their slowest query does (the mean query is 5x shorter than the longest one).
//
//  * Queries are bound to the XCD whose worker first picks them and all their jobs stay on that
//    XCD's queue: a query's field, pending bitmaps and tile states are then only touched through ONE
//    L2 (L2s of different XCDs are not coherent); L1 is bypassed with sc1 loads and every hand-over
//    is "s_waitcnt vmcnt(0), then a device-scope atomic".  The XCD id is read from HW_REG_XCC_ID, so
//    nothing depends on the block -> XCD placement.
//  * Tile state machine (2 bits per tile): IDLE -> QUEUED -> RUNNING -> IDLE, or RUNNING ->
//    RUNNING_DIRTY (re-activated while in flight) -> QUEUED.  A tile is never run by two waves.
//  * outstanding[q] counts queued + running jobs of the current bucket; the wave that brings it to 0
//    owns the query alone and either finishes it or opens the next bucket (tiles flagged in far_act).
//  * No worker ever waits for a particular other worker: it waits for queue entries, which any
//    resident worker of that XCD can produce/consume, so co-residency of the grid is not required.
//    Every spin is bounded; on timeout the abort flag ends the launch and the host reports it.
// =================================================================================================
constexpr int TSA_QN = 1 << 16;         // ring entries per XCD
constexpr unsigned TSA_NOJOB = 0xFFFFFFFFu;
constexpr int TSA_SPIN_LIMIT = 1 << 22;

struct TsaQ {   // per-query state (one 64-byte line)
  int best, bucket, bucket0, role, outstanding, status, xcc, expanded, jobs, overflow, start, goal, pad[4];
};
struct TsaCtl {
  int next_query, remaining, abort, n, pad[12];
  unsigned head[8][16];   // one 64-byte line per XCD
  unsigned tail[8][16];
};

__device__ __forceinline__ int ld_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int read_xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}

// Consecutive ring positions are spread 264 B apart (odd multiplier = bijection on the ring): the
// waiting workers of an XCD then poll different L2 channels instead of one cache line.
__device__ __forceinline__ unsigned tsa_slot(unsigned pos) { return (pos * 33u) & (unsigned)(TSA_QN - 1); }
// bounded MPMC ring: entry = (sequence << 32) | job; the slot of position i starts with sequence i
__device__ bool tsa_enqueue(TsaCtl* ctl, unsigned long long* ring, int x, unsigned job) {
  const unsigned pos = atomicAdd(&ctl->tail[x][0], 1u);
  unsigned long long* slot = &ring[(size_t)x * TSA_QN + tsa_slot(pos)];
  for (int spin = 0;; ++spin) {
    if ((unsigned)(ld_u64(slot) >> 32) == pos) break;          // slot is free for this lap
    if (spin > TSA_SPIN_LIMIT || ((spin & 1023) == 1023 && ld_i32(&ctl->abort))) { atomicCAS(&ctl->abort, 0, 1); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  st_u64(slot, ((unsigned long long)(pos + 1u) << 32) | job);  // one 64-bit store publishes job + sequence
  return true;
}
// Ticket dequeue: one atomicAdd takes the next position of this XCD's ring, then the worker waits
// for that slot to be published (a CAS-on-head dequeue collapsed under 512 contending wavefronts).
// Returns TSA_NOJOB only when the launch is over (no query left unfinished) or aborted.
__device__ unsigned tsa_dequeue(TsaCtl* ctl, unsigned long long* ring, int x) {
  const unsigned pos = atomicAdd(&ctl->head[x][0], 1u);
  unsigned long long* slot = &ring[(size_t)x * TSA_QN + tsa_slot(pos)];
  for (int spin = 0;; ++spin) {
    const unsigned long long v = ld_u64(slot);
    if ((unsigned)(v >> 32) == pos + 1u) {
      st_u64(slot, (unsigned long long)(pos + (unsigned)TSA_QN) << 32);   // free the slot for the next lap
      return (unsigned)v;
    }
    if ((spin & 15) == 15) {
      if (ld_i32(&ctl->remaining) <= 0 || ld_i32(&ctl->abort)) return TSA_NOJOB;
      if (spin > 400000) { atomicCAS(&ctl->abort, 0, 5); return TSA_NOJOB; }   // ~5 s without work
    }
    if (spin < 64) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(RNA_TSA_POLL_SLEEP);
  }
}

struct TsaGlobalSched {
  TsaQ* qs;
  TsaCtl* ctl;
  unsigned long long* ring;
  unsigned* tstate;     // 2 bits per tile of this query
  unsigned* far_act;    // bit per tile: has next-bucket pending cells
  int q, xcc, best_cache;
  __device__ __forceinline__ int best() const { return best_cache; }
  __device__ __forceinline__ void improve_best(int g) { atomicMin(&qs->best, g); if (g < best_cache) best_cache = g; }
  __device__ __forceinline__ void overflow() { st_i32(&qs->overflow, 1); }
  __device__ __forceinline__ void act_far(int t) { atomicOr(&far_act[t >> 5], 1u << (t & 31)); }
  __device__ void act_cur(int t) {
    unsigned* w = &tstate[t >> 4];
    const int sh = (t & 15) * 2;
    for (int spin = 0; spin < TSA_SPIN_LIMIT; ++spin) {
      const unsigned old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned st = (old >> sh) & 3u;
      if (st == 1u || st == 3u) return;                               // already scheduled to run (again)
      const unsigned nw = (st == 0u) ? (old | (1u << sh)) : (old | (3u << sh));   // IDLE->QUEUED, RUNNING->RUNNING_DIRTY
      if (atomicCAS(w, old, nw) == old) {
        if (st == 0u) {
          atomicAdd(&qs->outstanding, 1);                            // count the job BEFORE it becomes visible
          tsa_enqueue(ctl, ring, xcc, ((unsigned)q << 16) | (unsigned)t);
        }
        return;
      }
    }
    atomicCAS(&ctl->abort, 0, 2);
  }
};

struct TsaPersistArgs {
  int rows, cols, tiles_i, tiles_j, n, bucket_width;
  int s0, s1;                  // circular-buffer start index: queries/paths are buffer indices, the search is in map space
  const rna_astar_query* queries;
  unsigned* field; size_t field_stride;
  unsigned* pend; size_t pend_stride;
  TsaQ* qstate; TsaCtl* ctl; unsigned long long* ring;
  unsigned* tstate; size_t tstate_stride;
  unsigned* far_act; size_t far_stride;
  const uint8_t* nbr_tm;       // tile-major neighbour masks (snapshot of this launch)
  unsigned* touched;           // [n][nt_words] tiles written by each query
  int* clean;                  // cleared on abort: the next launch rewrites every field
  const int* perm;             // launch order (longest expected search first)
};

__global__ void tsa_persist_init_kernel(TsaPersistArgs A) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t w = gid; w < (size_t)A.n * A.tstate_stride; w += step) A.tstate[w] = 0u;
  for (size_t w = gid; w < (size_t)A.n * A.far_stride; w += step) A.far_act[w] = 0u;
  for (size_t w = gid; w < (size_t)8 * TSA_QN; w += step)
    A.ring[(w & ~(size_t)(TSA_QN - 1)) + tsa_slot((unsigned)w & (TSA_QN - 1))] = (unsigned long long)(w & (TSA_QN - 1)) << 32;
  for (size_t w = gid; w < (size_t)A.n * (sizeof(TsaQ) / 4); w += step)
    reinterpret_cast<int*>(A.qstate)[w] = ((w % (sizeof(TsaQ) / 4)) == 5) ? -2 : 0;   // status = -2: not started
  if (gid == 0) {
    A.ctl->next_query = 0; A.ctl->remaining = A.n; A.ctl->abort = 0; A.ctl->n = A.n;
    for (int x = 0; x < 8; ++x) { A.ctl->head[x][0] = 0u; A.ctl->tail[x][0] = 0u; }
  }
}

__global__ void __launch_bounds__(TSA_THREADS) tsa_persist_kernel(TsaPersistArgs A) {
  __shared__ TsaWave s_w[TSA_WAVES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  TsaWave& W = s_w[wv];
  const int xcc = read_xcc_id();
  const int ntile = A.tiles_i * A.tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  const size_t pend_words = (size_t)ntile * TS;
#ifdef RNA_TSA_STATS
  unsigned long long tsa_acc[16] = {};
#endif
  for (;;) {
    // ---- unstarted queries first: the worker that starts a query binds it to its own XCD ----
    int qn = -1;
    if (lane == 0 && ld_i32(&A.ctl->next_query) < A.n) {
      qn = atomicAdd(&A.ctl->next_query, 1);   // caller's order: longest-first (A.perm) measured slower here
      if (qn >= A.n) qn = -1;
    }
    qn = __shfl(qn, 0);
    if (qn >= 0) {
      if (lane == 0) {
        TsaQ* qs = &A.qstate[qn];
        rna_astar_query qu = A.queries[qn];
        const int ncell = A.rows * A.cols;
        unsigned* field = A.field + (size_t)qn * A.field_stride;
        int status = -1;
        if (!(qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell)) status = 2;
        else {
          qu.start = tsa_unwrap_lin(qu.start, A.rows, A.cols, A.s0, A.s1);
          qu.goal = tsa_unwrap_lin(qu.goal, A.rows, A.cols, A.s0, A.s1);
          const int si = qu.start % A.rows, sj = qu.start / A.rows, gi = qu.goal % A.rows, gj = qu.goal / A.rows;
          const int b0 = tsa_octile(si, sj, gi, gj) / A.bucket_width;
          qs->best = INF; qs->bucket = b0; qs->bucket0 = b0; qs->role = 0; qs->outstanding = 1; qs->xcc = xcc;
          qs->expanded = 0; qs->jobs = 0; qs->overflow = 0; qs->start = qu.start; qs->goal = qu.goal;
#ifdef RNA_TSA_STATS
          qs->pad[0] = (int)(wall_clock64() & 0x7fffffff);
#endif
          if (qu.goal != qu.start && A.nbr_tm[tm_index(gi, gj, A.tiles_i)] == 0) status = 1;  // walled-in goal: nothing written
          else {
            __hip_atomic_store(&field[tm_index(si, sj, A.tiles_i)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // g(start) = 0
            const int ts = (sj >> 5) * A.tiles_i + (si >> 5);
            unsigned* pend0 = A.pend + (size_t)qn * A.pend_stride;
            atomicOr(&pend0[(size_t)ts * TS + (sj & 31)], 1u << (si & 31));
            atomicOr(&(A.tstate + (size_t)qn * A.tstate_stride)[ts >> 4], 1u << ((ts & 15) * 2));     // QUEUED
            st_i32(&qs->status, -1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tsa_enqueue(A.ctl, A.ring, xcc, ((unsigned)qn << 16) | (unsigned)ts);
          }
        }
        if (status >= 0) {
          st_i32(&qs->status, status);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          atomicSub(&A.ctl->remaining, 1);
        }
      }
      continue;
    }
    // ---- next job of this XCD (blocks until one is published or the launch is over) ----
    unsigned job = TSA_NOJOB;
    TSA_T(t_w0);
    if (lane == 0) job = tsa_dequeue(A.ctl, A.ring, xcc);
    job = __shfl(job, 0);
    TSA_T(t_w1);
    TSA_ACC(4, t_w0, t_w1);
    if (job == TSA_NOJOB) break;
    TSA_CNT(7, 1);

    // ---- run one tile job ----
    const int q = (int)(job >> 16), t = (int)(job & 0xffffu);
    TsaQ* qs = &A.qstate[q];
    unsigned* tstate = A.tstate + (size_t)q * A.tstate_stride;
    unsigned* far_act = A.far_act + (size_t)q * A.far_stride;
    unsigned* tw = &tstate[t >> 4];
    const int sh = (t & 15) * 2;
    if (lane == 0) {   // QUEUED -> RUNNING
      atomicOr(&A.touched[(size_t)q * nt_words + (t >> 5)], 1u << (t & 31));   // this search writes tile t
      for (int spin = 0;; ++spin) {
        const unsigned old = __hip_atomic_load(tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((old >> sh) & 3u) == 1u && atomicCAS(tw, old, (old & ~(3u << sh)) | (2u << sh)) == old) break;
        if (spin > TSA_SPIN_LIMIT) { atomicCAS(&A.ctl->abort, 0, 3); break; }
      }
    }
    const int bucket = ld_i32(&qs->bucket), role = ld_i32(&qs->role);
    const int goal = ld_i32(&qs->goal);
    const int gi = goal % A.rows, gj = goal / A.rows;
    unsigned* field = A.field + (size_t)q * A.field_stride;
    unsigned* pend0 = A.pend + (size_t)q * A.pend_stride;
    TsaGlobalSched sch{qs, A.ctl, A.ring, tstate, far_act, q, xcc, ld_i32(&qs->best)};
    const long long bucket_end = ((long long)bucket + 1) * A.bucket_width;
    int exp = tsa_job(sch, W, lane, t, A.rows, A.cols, A.tiles_i, A.tiles_j, field, A.nbr_tm, pend0 + (size_t)role * pend_words,
                      pend0 + (size_t)(role ^ 1) * pend_words, bucket_end, gi, gj TSA_ACC_ARG);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // everything this job wrote is at L2 before the tile is released
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) exp += __shfl_xor(exp, o);
    int last = 0;
    if (lane == 0) {
      atomicAdd(&qs->expanded, exp);
      atomicAdd(&qs->jobs, 1);
      for (int spin = 0;; ++spin) {
        const unsigned old = __hip_atomic_load(tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned st = (old >> sh) & 3u;
        if (st == 2u) {            // RUNNING -> IDLE
          if (atomicCAS(tw, old, old & ~(3u << sh)) == old) { last = (atomicSub(&qs->outstanding, 1) == 1); break; }
        } else if (st == 3u) {     // re-activated while running: RUNNING_DIRTY -> QUEUED, same job count
          if (atomicCAS(tw, old, (old & ~(3u << sh)) | (1u << sh)) == old) { tsa_enqueue(A.ctl, A.ring, xcc, job); break; }
        }
        if (spin > TSA_SPIN_LIMIT) { atomicCAS(&A.ctl->abort, 0, 4); break; }
      }
    }
    last = __shfl(last, 0);
    TSA_T(t_w2);
    TSA_ACC(5, t_w1, t_w2);
    if (!last) continue;

    // ---- this wave emptied the bucket: it owns the query alone until it enqueues new jobs ----
    const int best = ld_i32(&qs->best);
    const long long done_below = ((long long)bucket + 1) * A.bucket_width;
    int status = -1;
    if (ld_i32(&qs->overflow)) status = 4;
    else if (best != INF && (long long)best < done_below) status = 0;   // goal settled, ties included
    else {
      unsigned* scratch = W.tile;   // grabbed far bits (nt_words <= TW*TW)
      int cnt = 0;
      for (int w = lane; w < nt_words; w += 64) {
        const unsigned bits = atomicExch(&far_act[w], 0u);
        scratch[w] = bits;
        cnt += __popc(bits);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
      if (cnt == 0) status = (best != INF) ? 0 : 1;   // nothing left anywhere
      else {
        if (lane == 0) { st_i32(&qs->bucket, bucket + 1); st_i32(&qs->role, role ^ 1); st_i32(&qs->outstanding, cnt); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // pass 1: every tile of the new bucket becomes QUEUED before the first job is visible (a job
        // that is already running may otherwise activate -- and enqueue -- one of them a second time)
        for (int w = lane; w < nt_words; w += 64) {
          unsigned bits = scratch[w];
          while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const int tt = (w << 5) + b;
            atomicOr(&tstate[tt >> 4], 1u << ((tt & 15) * 2));   // IDLE -> QUEUED (all tiles are idle here)
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // pass 2: publish the jobs
        for (int w = lane; w < nt_words; w += 64) {
          unsigned bits = scratch[w];
          while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            tsa_enqueue(A.ctl, A.ring, xcc, ((unsigned)q << 16) | (unsigned)((w << 5) + b));
          }
        }
      }
    }
    if (status >= 0) {   // tiles that were handed next-bucket cells but never ran were written as well
      for (int w = lane; w < nt_words; w += 64) {
        const unsigned bits = ld_l2(&far_act[w]);
        if (bits) atomicOr(&A.touched[(size_t)q * nt_words + w], bits);
      }
    }
    if (status >= 0 && lane == 0) {
#ifdef RNA_TSA_STATS
      qs->pad[1] = (int)(wall_clock64() & 0x7fffffff);
#endif
      st_i32(&qs->status, status);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      atomicSub(&A.ctl->remaining, 1);
    }
    TSA_T(t_w3);
    TSA_ACC(6, t_w2, t_w3);
  }
#ifdef RNA_TSA_STATS
  if (lane == 0) for (int k = 0; k < 16; ++k) atomicAdd(&g_tsa_stat[k], tsa_acc[k]);
#endif
}

// canonical backtrace + result records for the persistent scheduler: one wavefront per query
__global__ void tsa_backtrace_kernel(int rows, int cols, int s0, int s1, int tiles_i, const unsigned* __restrict__ field_all, size_t field_stride,
                                     const uint8_t* __restrict__ nbr_tm, int* __restrict__ clean,
                                     const TsaQ* __restrict__ qstate, const TsaCtl* __restrict__ ctl, int32_t* __restrict__ paths,
                                     int max_path_len, int32_t* __restrict__ rev_all, int rev_cap,
                                     rna_astar_result* __restrict__ results) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const TsaQ qs = qstate[q];
  const unsigned* field = field_all + (size_t)q * field_stride;
  const int n_buckets = qs.bucket - qs.bucket0 + 1;
  if (ctl->abort || qs.status < 0) {   // scheduler timeout: report it as a capacity/scheduling failure
    if (tid == 0) *clean = 0;          // queued tiles were never recorded: rewrite every field next time
    if (tid == 0) results[q] = rna_astar_result{(int)RNA_ECAPACITY, ctl->abort, qs.status, qs.expanded, qs.jobs, qs.outstanding};
    return;
  }
  if (qs.status != 0) {
    if (tid == 0) results[q] = rna_astar_result{qs.status, 0, INF, qs.expanded, qs.jobs, qs.status == 2 ? 0 : n_buckets};
    return;
  }
  const int ncell = rows * cols;
  const int si = qs.start % rows, sj = qs.start / rows, gi = qs.goal % rows, gj = qs.goal / rows;
  int* rev = rev_all + (size_t)q * rev_cap;
  int ci = gi, cj = gj, len = 0;
  bool ok = true;
  const int k = tid & 7;
  const int w = (k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D;
  const int di = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0);
  const int dj = k < 3 ? -1 : (k > 4 ? 1 : 0);
  for (;;) {
    if (tid == 0 && len < rev_cap) rev[len] = cj * rows + ci;
    ++len;
    if (ci == si && cj == sj) break;
    if (len > ncell) { ok = false; break; }
    const int ni = ci + di, nj = cj + dj;
    const bool inb = ni >= 0 && nj >= 0 && ni < rows && nj < cols;
    const unsigned wc = field[tm_index(ci, cj, tiles_i)];
    const unsigned mc = nbr_tm[tm_index(ci, cj, tiles_i)];
    const unsigned wn = field[inb ? tm_index(ni, nj, tiles_i) : tm_index(ci, cj, tiles_i)];
    const bool hit = tid < 8 && inb && ((mc >> k) & 1u) && ((wn >> 8) != G_INF) && ((wn >> 8) + (unsigned)w == (wc >> 8));
    const unsigned long long mask = __ballot(hit);
    if (!mask) { ok = false; break; }
    const int src = __ffsll((long long)mask) - 1;
    ci = __shfl(ni, src);
    cj = __shfl(nj, src);
  }
  if (!ok) { if (tid == 0) results[q] = rna_astar_result{1, 0, INF, qs.expanded, qs.jobs, n_buckets}; return; }
  if (len > max_path_len || len > rev_cap) {
    if (tid == 0) results[q] = rna_astar_result{3, len, qs.best, qs.expanded, qs.jobs, n_buckets};
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int32_t* path = paths + (size_t)q * max_path_len;
  for (int i = tid; i < len; i += 64) path[i] = tsa_buffer_lin(rev[len - 1 - i], rows, cols, s0, s1);
  if (tid == 0) results[q] = rna_astar_result{0, len, qs.best, qs.expanded, qs.jobs, n_buckets};
}

// ---- host entry points used by astar.hip ----
size_t tsa_field_words(const rna_engine* e) {
  const size_t ti = (e->geom.size[0] + TS - 1) / TS, tj = (e->geom.size[1] + TS - 1) / TS;
  return ti * tj * TILE_WORDS;
}
size_t tsa_pend_words(const rna_engine* e) {
  const size_t ti = (e->geom.size[0] + TS - 1) / TS, tj = (e->geom.size[1] + TS - 1) / TS;
  return 2 * ti * tj * TS;
}
bool tsa_supported(const rna_engine* e) {
  const size_t ti = (e->geom.size[0] + TS - 1) / TS, tj = (e->geom.size[1] + TS - 1) / TS;
  return ti * tj <= (size_t)TSA_MAX_TILE_WORDS * 32;
}

// one allocation per pipeline stage: clean flag | tile-major neighbour masks | touched bitsets.
// It must start zeroed (clean == 0: the first launch writes every field).
size_t tsa_aux_bytes(const rna_engine* e, int max_queries) {
  const size_t ti = (e->geom.size[0] + TS - 1) / TS, tj = (e->geom.size[1] + TS - 1) / TS;
  const size_t ntile = ti * tj;
  return 256 + tsa_align256(ntile * TILE_WORDS) + tsa_align256((size_t)max_queries * ((ntile + 31) / 32) * 4) +
         tsa_align256((size_t)max_queries * sizeof(int));
}
static TsaAux tsa_aux_view(const rna_engine* e, void* aux) {
  const size_t ti = (e->geom.size[0] + TS - 1) / TS, tj = (e->geom.size[1] + TS - 1) / TS;
  char* base = static_cast<char*>(aux);
  TsaAux v;
  v.clean = reinterpret_cast<int*>(base);
  v.ticket = reinterpret_cast<int*>(base + 64);
  v.nbr_tm = reinterpret_cast<uint8_t*>(base + 256);
  v.touched = reinterpret_cast<unsigned*>(base + 256 + tsa_align256(ti * tj * TILE_WORDS));
  v.perm = reinterpret_cast<int*>(base + 256 + tsa_align256(ti * tj * TILE_WORDS) +
                                  tsa_align256((size_t)e->astar.max_queries * ((ti * tj + 31) / 32) * 4));
  return v;
}
// snapshot the neighbour masks and bring every field of this stage back to "unreached"
static void tsa_launch_init(rna_engine* e, hipStream_t stream, unsigned* field, size_t field_stride, unsigned* pend,
                            size_t pend_stride, int max_queries, const TsaAux& aux, const rna_astar_query* q_dev, int n) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  hipLaunchKernelGGL(tsa_init_kernel, dim3(4096), dim3(256), 0, stream, e->nbr, rows, cols, ti, tj, field, field_stride, pend,
                     pend_stride, max_queries, aux, e->geom.start[0], e->geom.start[1]);
  hipLaunchKernelGGL(tsa_mark_clean_kernel, dim3(1), dim3(1), 0, stream, aux.clean, aux.ticket);
  if (n <= 2048)   // the ranking is O(n^2 / 256) per thread: beyond this the caller order is kept
    // one small workgroup: it has to fit next to the search workgroups that fill every CU when batches are pipelined
    hipLaunchKernelGGL(tsa_order_kernel, dim3(1), dim3(256), (size_t)n * sizeof(int), stream, q_dev, n, rows, cols, aux.perm);
  else
    hipLaunchKernelGGL(tsa_identity_order_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, aux.perm);
}

int tsa_launch(rna_engine* e, hipStream_t init_stream, hipStream_t search_stream, hipEvent_t ev_init, unsigned* field,
               size_t field_stride, unsigned* pend, size_t pend_stride, void* aux_mem, int max_queries, int32_t* rev,
               int rev_cap, const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len, rna_astar_result* res_dev) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  const TsaAux aux = tsa_aux_view(e, aux_mem);
  {
    KernelTimer kt(e, RNA_K_ASTAR_INIT, init_stream);
    tsa_launch_init(e, init_stream, field, field_stride, pend, pend_stride, max_queries, aux, q_dev, n);
    RNA_HIP(e, hipGetLastError());
  }
  if (ev_init) {
    RNA_HIP(e, hipEventRecord(ev_init, init_stream));
    RNA_HIP(e, hipStreamWaitEvent(search_stream, ev_init, 0));
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_SEARCH, search_stream);
    const size_t nt_bytes = (size_t)((ti * tj + 31) / 32) * sizeof(unsigned);
    // queries per workgroup: as many (<= 4) as the three bitsets per query leave room for in the 160 KB of LDS
    int qb = 1;
    if (const char* m = getenv("RNA_TSA_QUERIES_PER_BLOCK")) qb = atoi(m);
    if (qb > 1) {   // opt-in: several queries per workgroup (fills thin rounds, but measured no faster when pipelined)
      const size_t fixed = sizeof(TsaWave) * TSA_WAVES + sizeof(unsigned) * TSA_MQ_JOBS + 1024;
      if (qb > 4) qb = 4;
      if (qb == 3) qb = 2;
      while (qb > 1 && fixed + 3 * nt_bytes * qb > 160 * 1024) qb >>= 1;
    }
#define RNA_LAUNCH_MQ(QB)                                                                                               \
  hipLaunchKernelGGL(tsa_multi_kernel<QB>, dim3((n + QB - 1) / QB), dim3(TSA_THREADS), 3 * nt_bytes * QB, search_stream, rows,   \
                     cols, ti, tj, q_dev, n, field, field_stride, pend, pend_stride, aux.nbr_tm, aux.touched,          \
                     e->astar.bucket_width, paths_dev, max_len, rev, rev_cap, res_dev, e->geom.start[0], e->geom.start[1], aux.perm)
    if (e->astar.mode == 1 && qb == 4) RNA_LAUNCH_MQ(4);
    else if (e->astar.mode == 1 && qb == 2) RNA_LAUNCH_MQ(2);
    else if (e->astar.mode == 3)
      hipLaunchKernelGGL(tsa_search_kernel<true>, dim3(n), dim3(TSA_THREADS), 4 * nt_bytes, search_stream, rows, cols, ti, tj, q_dev,
                         field, field_stride, pend, pend_stride, aux.nbr_tm, aux.touched, e->astar.bucket_width, paths_dev, max_len,
                         rev, rev_cap, res_dev, e->geom.start[0], e->geom.start[1], aux.perm, aux.ticket);
    else
      hipLaunchKernelGGL(tsa_search_kernel<false>, dim3(n), dim3(TSA_THREADS), 3 * nt_bytes, search_stream, rows, cols, ti, tj, q_dev,
                         field, field_stride, pend, pend_stride, aux.nbr_tm, aux.touched, e->astar.bucket_width, paths_dev, max_len,
                         rev, rev_cap, res_dev, e->geom.start[0], e->geom.start[1], aux.perm, aux.ticket);
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

int tsa_settled(rna_engine* e, const unsigned* field, size_t field_stride, const rna_astar_query* q, const rna_astar_result* r,
                int n, int32_t* d_counts) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  hipLaunchKernelGGL(tsa_settled_kernel, dim3(n), dim3(1024), 0, e->stream, rows, cols, ti, tj, q, r, field, field_stride,
                     d_counts, e->geom.start[0], e->geom.start[1]);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}


size_t tsa_persist_state_bytes(const rna_engine* e, int max_queries, size_t* tstate_stride, size_t* far_stride) {
  const size_t ti = (e->geom.size[0] + TS - 1) / TS, tj = (e->geom.size[1] + TS - 1) / TS;
  const size_t ntile = ti * tj;
  *tstate_stride = ((ntile + 15) / 16 + 15) / 16 * 16;
  *far_stride = ((ntile + 31) / 32 + 15) / 16 * 16;
  return sizeof(TsaCtl) + (size_t)max_queries * sizeof(TsaQ) + (size_t)8 * TSA_QN * 8 +
         (size_t)max_queries * (*tstate_stride + *far_stride) * 4;
}

// persistent launch: init (field + scheduler state) on init_stream, then the worker grid and the
// backtrace on search_stream.  `state` is one allocation laid out as ctl | qstate | ring | tstate | far_act.
int tsa_persist_launch(rna_engine* e, hipStream_t init_stream, hipStream_t search_stream, hipEvent_t ev_init, unsigned* field,
                       size_t field_stride, unsigned* pend, size_t pend_stride, void* aux_mem, void* state, int max_queries,
                       int32_t* rev, int rev_cap, const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len,
                       rna_astar_result* res_dev) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  if (n > 32767 || (size_t)ti * tj > 65536) return fail(e, RNA_EINVAL, "persistent A*: too many queries or tiles");
  TsaPersistArgs A{};
  A.rows = rows; A.cols = cols; A.tiles_i = ti; A.tiles_j = tj; A.n = n; A.bucket_width = e->astar.bucket_width;
  A.s0 = e->geom.start[0]; A.s1 = e->geom.start[1];
  A.queries = q_dev; A.field = field; A.field_stride = field_stride; A.pend = pend; A.pend_stride = pend_stride;
  size_t ts_stride = 0, far_stride = 0;
  (void)tsa_persist_state_bytes(e, max_queries, &ts_stride, &far_stride);
  char* base = static_cast<char*>(state);
  A.ctl = reinterpret_cast<TsaCtl*>(base); base += sizeof(TsaCtl);
  A.qstate = reinterpret_cast<TsaQ*>(base); base += (size_t)max_queries * sizeof(TsaQ);
  A.ring = reinterpret_cast<unsigned long long*>(base); base += (size_t)8 * TSA_QN * 8;
  A.tstate = reinterpret_cast<unsigned*>(base); A.tstate_stride = ts_stride; base += (size_t)max_queries * ts_stride * 4;
  A.far_act = reinterpret_cast<unsigned*>(base); A.far_stride = far_stride;
  const TsaAux aux = tsa_aux_view(e, aux_mem);
  A.nbr_tm = aux.nbr_tm; A.touched = aux.touched; A.clean = aux.clean; A.perm = aux.perm;
  {
    KernelTimer kt(e, RNA_K_ASTAR_INIT, init_stream);
    tsa_launch_init(e, init_stream, field, field_stride, pend, pend_stride, max_queries, aux, q_dev, n);
    hipLaunchKernelGGL(tsa_persist_init_kernel, dim3(512), dim3(256), 0, init_stream, A);
    RNA_HIP(e, hipGetLastError());
  }
  if (ev_init) {
    RNA_HIP(e, hipEventRecord(ev_init, init_stream));
    RNA_HIP(e, hipStreamWaitEvent(search_stream, ev_init, 0));
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_SEARCH, search_stream);
    int cus = 0;
    RNA_HIP(e, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
    if (cus <= 0) cus = 256;
    if (const char* m = getenv("RNA_TSA_BLOCKS_PER_CU")) cus *= std::max(1, atoi(m));
    hipLaunchKernelGGL(tsa_persist_kernel, dim3(cus), dim3(TSA_THREADS), 0, search_stream, A);
    hipLaunchKernelGGL(tsa_backtrace_kernel, dim3(n), dim3(64), 0, search_stream, rows, cols, A.s0, A.s1, ti, field, field_stride,
                       aux.nbr_tm, aux.clean, A.qstate, A.ctl, paths_dev, max_len, rev, rev_cap, res_dev);
    RNA_HIP(e, hipGetLastError());
  }
#ifdef RNA_TSA_STATS
  {
    RNA_HIP(e, hipStreamSynchronize(search_stream));
    unsigned long long st[32];
    RNA_HIP(e, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tsa_stat), sizeof(st)));
    static const unsigned long long zero[32] = {};
    RNA_HIP(e, hipMemcpyToSymbol(HIP_SYMBOL(g_tsa_stat), zero, sizeof(zero)));
    std::vector<TsaQ> qh(n);
    RNA_HIP(e, hipMemcpy(qh.data(), A.qstate, (size_t)n * sizeof(TsaQ), hipMemcpyDeviceToHost));
    int t0 = 0x7fffffff;
    for (auto& qq : qh) if (qq.status != 2 && qq.pad[0] < t0) t0 = qq.pad[0];
    std::vector<double> fin;
    for (auto& qq : qh) if (qq.status != 2) fin.push_back((qq.pad[1] - t0) * 1e-5);
    std::sort(fin.begin(), fin.end());
    const double jobs = (double)st[7];
    fprintf(stderr, "[tsa stats] jobs %.0f | per job us: load %.2f relax %.2f wb %.2f handover %.2f finish %.2f advance %.2f | wait total %.1f ms-waves | relax iters/job %.1f cells/iter %.1f\n",
            jobs, st[0] * 0.01 / jobs, st[1] * 0.01 / jobs, st[2] * 0.01 / jobs, st[3] * 0.01 / jobs, (st[5] - st[0] - st[1] - st[2] - st[3]) * 0.01 / jobs,
            st[6] * 0.01 / jobs, st[4] * 1e-5, st[8] / jobs, st[9] / (double)std::max<unsigned long long>(1, st[8]));
    fprintf(stderr, "[tsa stats] jobs by expansions: 0: %.1f%%  1-15: %.1f%%  16-63: %.1f%%  64-255: %.1f%%  256-1023: %.1f%%  1024+: %.1f%%\n",
            100.0 * st[10] / jobs, 100.0 * st[11] / jobs, 100.0 * st[12] / jobs, 100.0 * st[13] / jobs, 100.0 * st[14] / jobs, 100.0 * st[15] / jobs);
    if (!fin.empty())
      fprintf(stderr, "[tsa stats] query finish ms: p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f | busy wave-ms %.1f\n", fin[fin.size() / 10],
              fin[fin.size() / 2], fin[fin.size() * 9 / 10], fin[fin.size() * 99 / 100], fin.back(), (st[5] + st[6]) * 1e-5);
  }
#endif
  return RNA_OK;
}

}  // namespace rna
