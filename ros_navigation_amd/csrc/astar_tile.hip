// astar_tile.hip -- tile-synchronous grid A* for gfx950 ("TSA"): the same contract and the same
// label-correcting argument as astar.hip, but the relaxation runs in REGISTERS.  This file holds the
// search kernel of the engine: one workgroup (8 or 16 wavefronts) per query, tile jobs taken from an open list in LDS.
//
// The search field lives in PAGES of one 64 x 16-cell tile each (4 KiB; word = KU - g, 0 = unreached, so that a
// fresh page is all zeros and "better" is "larger").  A wavefront owns one tile at a time: lane = column (the
// contiguous axis i), register = row (j), i.e. the whole tile is 16 VGPRs.  A tile job
//   1. loads its page, the two neighbouring rows, the two neighbouring columns (kept by their owners as a copy in
//      the page's aux words, so they are one 64-byte read) and four corners -- ONE memory round trip after the page
//      table look-up -- and applies what those halo cells can contribute once (they do not change during the job),
//   2. relaxes the tile to the fixed point of the current f-bucket with alternating down / up sweeps: a row takes its
//      three vertical candidates from the row before it (one in-lane, two through DPP wave shifts by one lane) and
//      its two horizontal ones through the same shifts; ~16 VALU instructions per 64 cells, no LDS, no atomics, no
//      divergence.  A row that changed runs on along itself at once (up to 16 short passes).  Rows whose sources did
//      not change since their last evaluation are skipped (three 16-bit flag words on the scalar unit), so a job
//      costs what its moving front costs, not 1024 cells per sweep,
//   3. stores the rows that changed (it is the only writer of its page: plain coalesced stores), and wakes a
//      neighbouring tile only if one of its own edge cells beats -- by a step that cell's mask allows -- what the
//      neighbour held when the job loaded its halo.  Neighbours PULL: nothing is ever written into another tile.
// There are no rounds: a tile that was woken sits in the query's open list in LDS with a key (the lowest f its waker
// offers it), free wavefronts take the tile with the lowest key, and an f-bucket is at its fixed point when every
// wavefront of the workgroup is idle (TsaLocalSched below; round 2 ran red-black rounds separated by workgroup barriers).
// Pages are handed out by the job that first changes a tile; every query slot owns a contiguous run of `cap` + 1 pages and
// a tile -> page table (tmap), so the 2-4 % of the map a search visits sits in a few MiB of HBM, and the workgroup that
// takes the slot next resets exactly the pages that were handed out.  Local page 0 of a slot is never written and always
// "unreached": reads of tiles without a page go there.
// Exactness: every update is a max over (KU - length) of real paths, every improvement that can matter to a neighbour
// sets its wake-up bit after the data is visible, and a bucket only ends with nothing pending -- so at termination g is
// exact for f <= f* whatever the order of the jobs, which is all the canonical backtrace reads (DESIGN.md "Grid A*
// contract").  scripts/sim_async.c is a CPU model of exactly this schedule (checked against the oracle's cost and E).
// How the kernel got here (the LDS worklist kernel of round 1, the dense sweeps and red-black rounds of round 2):
// docs/history.md; the measurements of round 3: DESIGN.md 5.
#include "engine.hpp"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace rna;

namespace rna {

constexpr int TI = 64, TJ = 16;               // tile: 64 cells along i (lanes) x 16 along j (registers)
constexpr int TILE_WORDS = TI * TJ;           // 1024
constexpr int AUX_WORDS = 64;                 // per page: [0..15] copy of column 0, [16..31] copy of column 63
constexpr int MASK_STRIDE = TILE_WORDS + 32 + 128;  // snapshot bytes per tile: lane-major masks + the two edge columns again + 16 "free" bits per lane
#ifndef RNA_TSA_WAVES
#define RNA_TSA_WAVES 8   // wavefronts per workgroup = per query; 8 wavefronts per SIMD -> four workgroups per CU (16 x 2: 60.5 k, 8 x 4: 63.5 k, 4 x 8: 42.4 k)
#endif
#ifndef RNA_TSA_SUPER
#define RNA_TSA_SUPER 2   // what a row does that is still moving after RNA_TSA_HPASS one-cell passes along itself:
                          // 0: it is looked at again in the next sweep            134.4 k cycles/s (profiles/r04_ab_row_scan.txt)
                          // 1: log steps inside the 16-lane DPP rows (TSA_SUPER)  134.7 k at best
                          // 2: its fixed point in one prefix-maximum scan          144.1 k (tsa_row_fixpoint)
#endif
#ifndef RNA_TSA_HPASS
#define RNA_TSA_HPASS (RNA_TSA_SUPER == 2 ? 4 : (RNA_TSA_SUPER ? 8 : 16))  // one-cell passes first (mode 2: 3 / 4 / 5 -> 144.1 / 144.0 / 143.6 k; mode 0: 8 / 16 -> 55.8 / 56.7 k in round 2)
#endif
#ifndef RNA_TSA_FRESH
#define RNA_TSA_FRESH 0   // 1: the wake tests at the end of a job read the neighbours' edges again instead of using the halo as loaded.
                          // Measured (profiles/r05_ab_fresh_wake_tests.txt): the jobs that find nothing fall as the model says
                          // (non-sticky ones to 7 % of all jobs), the bench from 154.3 k to 152.7 k cycles/s -- three more loads and
                          // their addresses in every working job cost more than the cheap jobs they save.  Off.
#endif
#ifndef RNA_TSA_CONFIRM
#define RNA_TSA_CONFIRM 0   // 1: a side whose wake test (against the halo as loaded) says "wake" is looked at AGAIN before the wake-up goes
                            // out: the neighbour's edge is read as it is NOW and the test repeated -- round 5's fresh wake tests, but only
                            // where a wake-up is about to be queued.  Measured (profiles/r06_ab_confirm.txt): exact (A* parity tests, fuzz
                            // seeds), jobs per search 2 709 -> 2 489, no-op share 33.6 -> 29.8 %, and the bench 162.8 -> 158.9 k cycles/s:
                            // the second look runs in 61 % of the working jobs and costs more than the 220 cheap jobs it saves.  Off.
#endif
#ifndef RNA_TSA_IDLE_SLEEP
#define RNA_TSA_IDLE_SLEEP 4   // an idle wavefront looks at the entry counter every 64 x this many clocks (1 / 4 / 12: profiles/r05_ab_idle_sleep.txt)
#endif
#ifndef RNA_TSA_FIRST_ROWS
#define RNA_TSA_FIRST_ROWS 1   // 1: a tile's first job in a bucket evaluates the rows that hold a cell the new bound releases, not all sixteen both ways
#endif
#ifndef RNA_TSA_REDBLACK
#define RNA_TSA_REDBLACK 1   // rounds alternate between the two checkerboard colours of the tiles
#endif
#ifndef RNA_TSA_WAVES_PER_EU
#define RNA_TSA_WAVES_PER_EU 8   // eight wavefronts per SIMD: the kernel must fit 64 VGPRs
#endif
constexpr int TSA_WAVES = RNA_TSA_WAVES;
constexpr int TSA_MAX_TILE_WORDS = 2048;   // active-tile bitset words -> up to 65536 tiles
constexpr int COST_S = 1000, COST_D = 1414;
constexpr int INF = 0x7fffffff;
constexpr int KU = 0x40000000;             // field word u = KU - g; 0 = unreached
constexpr int SCR_CNT = 84 + 192 + 48 + 64;     // four counters behind the scratch proper: rows written, jobs that got past the halo step, all turns
                                                // (jobs and sticky turns), sticky turns (kept in LDS: a register each across the job loop was one too many)
constexpr int SCR_WORDS = SCR_CNT + 4;          // per-wave LDS scratch: column transposition (68 + a zero tail of 16), the halo rows and
                                           // columns as loaded (3 x 64), this tile's edge columns at the end of the job (16 + 16 + 16),
                                           // the masks of the tile's edge-column cells as loaded (64)
static_assert(SCR_WORDS % 4 == 0 && 276 % 4 == 0, "a wavefront's scratch starts on 16 bytes and so do the edge columns in it (ds_write_b128)");

__device__ __forceinline__ int tsa_octile(int i, int j, int gi, int gj) {
  const int dx = abs(i - gi), dy = abs(j - gj);
  const int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return COST_S * mx + (COST_D - COST_S) * mn;
}
// Loads of the search field.  A query's pages, edge copies and page table are read and written by ONE workgroup (the
// slot is its alone), whose wavefronts share a CU and therefore a vector L1 that is coherent among them (workgroups are
// not split over CUs: no -mtgsplit); RNA_TSA_SCOPE selects the scope the loads are coherent at -- agent: every load
// goes to L2 (sc1), workgroup: it may hit the CU's L1.
#ifndef RNA_TSA_SCOPE
#define RNA_TSA_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, RNA_TSA_SCOPE);
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
// DPP wave shifts by one lane (all 64 lanes active).  from_below(v)[l] = v[l-1], lane 0 receives `edge`;
// from_above(v)[l] = v[l+1], lane 63 receives `edge`.
__device__ __forceinline__ int lane_m1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }    // wave_shr:1
__device__ __forceinline__ int lane_p1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }    // wave_shl:1
__device__ __forceinline__ int lane_m1(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int lane_p1(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xF, 0xF, false); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
// Wave-wide reductions on the VALU (DPP row shifts, then the two row broadcasts): no LDS, ~8 instructions.
// row_shr:1/2/4/8 leave the inclusive prefix of each 16-lane row in its lane 15; row_bcast:15 hands rows 0 / 2 to rows
// 1 / 3, row_bcast:31 hands rows 0+1 to rows 2 and 3: lane 63 holds the whole wave, lane 31 the lower half.
#define TSA_DPP_RED(op, v, ident)                                                                          \
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xF, 0xF, false));                                \
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xF, 0xF, false));                                \
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xF, 0xF, false));                                \
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xF, 0xF, false));                                \
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xA, 0xF, false));
__device__ __forceinline__ int tsa_imax(int a, int b) { return max(a, b); }
__device__ __forceinline__ int tsa_umin(int a, int b) { return (int)min((unsigned)a, (unsigned)b); }
// maxima of the two 32-lane halves: lane 31 / lane 63
__device__ __forceinline__ void wave_halves_max_i32(int v, int& lo, int& hi) {
  TSA_DPP_RED(tsa_imax, v, (int)0x80000000)
  lo = __builtin_amdgcn_readlane(v, 31);
  hi = __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
  TSA_DPP_RED(tsa_imax, v, (int)0x80000000)
  v = max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, 0x143, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned u) {
  int v = (int)u;
  TSA_DPP_RED(tsa_umin, v, -1)
  v = tsa_umin(v, __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xC, 0xF, false));
  return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
// all ones if bit `bit` of m is set, else 0.  Opaque to the optimiser on purpose: left to itself it hoists the 64 tests
// of a tile job (16 rows x 4 diagonal moves) out of the sweeps as 64-bit lane masks, 128 SGPRs that it then spills into
// VGPR lanes and from there into scratch.
#define TSA_OPEN(m, bit) ({ int r_; asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(r_) : "v"(m), "i"(bit)); r_; })
// octile heuristic from max / min of the two offsets with full-rate 24-bit multiplies (left to itself the compiler
// emits two quarter-rate v_mul_lo_u32 here, in the hottest recompute of the sweeps)
__device__ __forceinline__ int tsa_h24(int mx, int mn) {
  int a, r;
  asm("v_mul_u32_u24 %0, %2, %1" : "=v"(a) : "v"(mx), "s"(COST_S));
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(mn), "s"(COST_D - COST_S), "v"(a));
  return r;
}
// 29 t + 8191 lane with full-rate 24-bit multiplies (a start for the free-node probe; the compiler's version was a
// quarter-rate v_mul_lo_u32 and a 64-bit multiply-add)
__device__ __forceinline__ unsigned tsa_seed24(unsigned t, unsigned lane) {
  unsigned a, r;
  asm("v_mul_u32_u24 %0, %1, 29" : "=v"(a) : "v"(t));
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(lane), "s"(8191u), "v"(a));
  return r;
}
// the octile heuristic of a cell through tsa_h24 (offsets below 2^24: the map has at most 2^30 cells)
__device__ __forceinline__ int tsa_octile24(int i, int j, int gi, int gj) {
  const int dx = abs(i - gi), dy = abs(j - gj);
  return tsa_h24(max(dx, dy), min(dx, dy));
}
// u if the cell may pass its value on in this bucket (f < lim  <=>  u - h >= thr), else "unreached"
__device__ __forceinline__ int tsa_prop(int u, int h, int thr) { return (u - h >= thr) ? u : 0; }

__device__ __forceinline__ int tile_of(int i, int j, int tiles_i) { return (j >> 4) * tiles_i + (i >> 6); }
__device__ __forceinline__ int in_page(int i, int j) { return ((j & 15) << 6) + (i & 63); }
// neighbouring tile k (same numbering as the neighbour-mask bits): offsets along i and j
__device__ __forceinline__ int kdi_of(int k) { return (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0); }
__device__ __forceinline__ int kdj_of(int k) { return k < 3 ? -1 : (k > 4 ? 1 : 0); }

// Device view of one pipeline stage.  Global page index of query q's local page p >= 1 is q*cap + p; page 0 is
// the shared "unreached" page.  Invariant between launches: every page word, aux word and tmap entry is 0, EXCEPT
// what belongs to the local pages 1..nalloc[q] of each query; the next launch on the stage resets exactly those
// (the workgroup that takes the slot next does, before it searches) instead of rewriting 64 MiB per query.
struct TsaStage {
  unsigned* pages;      // [max_queries][cap + 1][1024]: local page 0 of a slot stays "unreached" (what a tile without a page reads)
  unsigned* paux;       // [max_queries][cap + 1][64]
  unsigned* tmap;       // [max_queries][ntile] tile -> local page, 0 = none
  unsigned* owner;      // [max_queries][cap + 1] local page -> tile
  int* nalloc;          // [max_queries] local pages handed out by the last search
  uint8_t* nbr_tm;      // [ntile][MASK_STRIDE] neighbour masks of this launch (snapshot), 0 outside the map
  int* perm;            // [max_queries] launch order of this batch: the k-th workgroup to START serves query perm[k]
  int* ticket;          // next position of perm to hand out (reset by every launch's init)
  int cap;              // pages per query
};
__host__ __device__ inline size_t tsa_align256(size_t x) { return (x + 255) & ~(size_t)255; }
// What the search kernels count about themselves (rna_astar_job_counters): sixteen 64-bit words per stage view, in the second
// half of the 256-byte block in front of the view's page counts (its first half: the ticket, and in a retry view the queries the
// retry slots served).  [0] searches, [1] tiles that got a page, [2] jobs (every turn), [3] jobs that found nothing, [4] sticky
// turns, [5] rows written, [6] buckets, [7] buckets run again because the queue ran out of nodes; [8] / [9] wavefront life / idle ticks in a -DRNA_TSA_IDLE developer build.
constexpr int TSA_COUNTER_WORDS = 16;
__host__ __device__ inline unsigned long long* tsa_counters_of(const TsaStage& S) {
  return reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(S.nalloc) - TSA_COUNTER_WORDS * sizeof(unsigned long long));
}

// snapshot of the neighbour masks, taken at launch so that a later map update cannot disturb a search in flight.
// Per tile (MAP-space, unwrapped indices): byte a*16 + b = mask of cell (a, b) -- the 16 rows of a lane are one
// 16-byte load --, then the 16 masks of column 0 and the 16 of column 63 once more for the halo step.
__global__ void __launch_bounds__(256) tsa_snapshot_kernel(const uint8_t* __restrict__ nbr, int rows, int cols, int tiles_i, int tiles_j,
                                                             uint8_t* __restrict__ nbr_tm, int s0, int s1) {
  __shared__ unsigned s_free[256];
  const int a = threadIdx.x & 63, bq = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < tiles_i * tiles_j; t += gridDim.x) {
    const int i = (t % tiles_i) * TI + a, j0 = (t / tiles_i) * TJ + bq * 4;
    unsigned v = 0u;   // (i, j) is a MAP-space (unwrapped) index; nbr is stored at buffer indices
    if (i < rows) {
      const int bi = i + s0 >= rows ? i + s0 - rows : i + s0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (j0 + k < cols) {
          const int bj = j0 + k + s1 >= cols ? j0 + k + s1 - cols : j0 + k + s1;
          v |= (unsigned)nbr[(size_t)bj * rows + bi] << (8 * k);
        }
    }
    uint8_t* out = nbr_tm + (size_t)t * MASK_STRIDE;
    reinterpret_cast<unsigned*>(out)[a * 4 + bq] = v;
    if (a == 0) reinterpret_cast<unsigned*>(out + TILE_WORDS)[bq] = v;
    if (a == TI - 1) reinterpret_cast<unsigned*>(out + TILE_WORDS + 16)[bq] = v;
    // bit b of a lane's "free" word: the cell in row b has at least one traversable neighbour, i.e. is free and inside
    // the map (a job would otherwise derive the 16 bits from the mask bytes every time: ~50 vector instructions)
    s_free[threadIdx.x] = (((v & 0xffu) != 0u) ? 1u : 0u) | (((v & 0xff00u) != 0u) ? 2u : 0u) | (((v & 0xff0000u) != 0u) ? 4u : 0u) |
                          (((v & 0xff000000u) != 0u) ? 8u : 0u);
    __syncthreads();
    if (bq == 0)
      reinterpret_cast<unsigned short*>(out + TILE_WORDS + 32)[a] =
          (unsigned short)(s_free[a] | (s_free[64 + a] << 4) | (s_free[128 + a] << 8) | (s_free[192 + a] << 12));
    __syncthreads();
  }
}
// The ticket back to zero and the launch order of the batch, see below.
// One small workgroup: it has to fit next to the search workgroups that fill every CU when batches are pipelined.
__global__ void __launch_bounds__(256) tsa_prepare_kernel(TsaStage S, const rna_astar_query* __restrict__ queries, int n, int rows, int cols,
                                                          int ranked, int* __restrict__ served) {
  extern __shared__ int s_key[];
  if (threadIdx.x == 0) *S.ticket = 0;
  if (served && threadIdx.x < TSA_RETRY) served[threadIdx.x] = -1;   // no retry slot has served a query of this batch yet
  if (served && threadIdx.x == TSA_RETRY) served[TSA_RETRY] = 0;     // ... and none has asked for one (tsa_retry_count)
  if (!ranked) {   // the ranking is O(n^2 / 256) per thread: large batches keep the caller's order
    for (int i = threadIdx.x; i < n; i += blockDim.x) S.perm[i] = i;
    return;
  }
  // Launch order of a batch: longest expected search first (key = Chebyshev distance start -> goal,
  // ties by index).  Workgroups are dispatched in index order and land on the XCDs round-robin, so this
  // both starts the long queries early and deals them evenly over the eight XCDs; with the caller's
  // (arbitrary) order one XCD regularly ended up with most of the long searches (+11 % throughput).
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
    S.perm[rank] = i;
  }
}
// fresh allocation: every page "unreached"
__global__ void tsa_fill_pages_kernel(uint4* __restrict__ p, size_t n4) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < n4; w += step) p[w] = make_uint4(0u, 0u, 0u, 0u);
}

constexpr int TSA_NCLS = 256;   // key classes of the open list (four per lane in a pop), see TsaLocalSched

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

// Per-query context of a tile job (wave-uniform).
struct TsaCtx {
  int rows, cols, tiles_i, tiles_j;
  unsigned tiles_magic;       // floor(2^32 / tiles_i) + 1 (0 when tiles_i == 1): t / tiles_i == umulhi(t, tiles_magic) for t < 2^16
  unsigned* pages;            // this query's pages: local page p at pages + (p << 10); page 0 stays "unreached"
  unsigned* paux;             // this query's edge-column copies, AUX_WORDS per page
  unsigned* tmap;             // this query's tile -> local page table
  unsigned* owner;            // this query's local page -> tile list
  int cap;
  int* nalloc;                // LDS: local pages handed out so far
  const uint8_t* nbr_tm;
  int gi, gj;
  int ts, sa, sb;             // start: tile, lane, row
  int tg, ga, gb;             // goal: tile, lane, row
  __device__ __forceinline__ unsigned page_of(int t) const { return ld_l2(&tmap[t]); }
};

// ---- a whole row to the fixed point of its horizontal steps in one go (RNA_TSA_SUPER == 2) ----
// Along a row of free cells a value decays by 1000 per cell, so the fixed point is the upper envelope of cones:
// u'(l) = max over sources s of the same free run of pp(s) - 1000 |l - s|.  With key(s) = pp(s) + 1000 s the right-moving
// half is an inclusive PREFIX MAXIMUM of the keys, restarted at every blocked cell (u'(l) = best key - 1000 l); the
// left-moving half the same on the lanes in reverse order (two ds_bpermute).  The restart is the top of the key: the
// number of blocked cells before the lane -- it never decreases along the lanes, a lane's own entry carries its own count,
// so the maximum of a prefix is the best key of the lane's own run.  The scan is the usual DPP one (row_shr 1 / 2 / 4 / 8,
// row_bcast 15 / 31), six v_max_u32 per direction; with 64-bit keys (count, value) it was 30 instructions per direction
// and +2.6 % instead of +7.1 %.  A cell takes the value only if the cell before it on the way may pass it on (f does not
// decrease along a path, so that one test covers every cell in between): value + 1000 >= h + thr of the neighbour lane.
// One application from the values as they are is the fixed point: what a cell receives this way it cannot hand back
// better than its source hands it directly.
// inclusive prefix maximum over the 64 lanes: four shifts inside the 16-lane DPP rows, then lane 15 of rows 0 and 2 to
// rows 1 and 3 and lane 31 to rows 2 and 3.  (In assembly: a lane without a source keeps its value, so each step is ONE
// v_max_u32 with a DPP operand; from __builtin_amdgcn_update_dpp the compiler made copy + v_mov_dpp + v_max of it.)
__device__ __forceinline__ unsigned tsa_prefix_max(unsigned v) {
  asm("s_nop 1\n\t"   // two wait states between a vector write and a DPP read of the same register
      "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
// The key of a cell that passes its value on: 7 bits of "blocked cells before me" above 25 bits of value.  The value
// is pp - thr + 1 + 1000 * position = (lim - cost) + 1000 * position, in [1, 2^25) as long as lim + 63000 < 2^25
// (TSA_SCAN_LIM; searches beyond it -- paths of more than 33 000 straight cells -- keep to one-cell passes).
constexpr int TSA_SCAN_SHIFT = 25;
[[maybe_unused]] constexpr unsigned TSA_SCAN_LIM = (1u << TSA_SCAN_SHIFT) - 63u * (unsigned)COST_S - 1u;
__device__ __forceinline__ unsigned long long tsa_row_fixpoint(int& g, int& pp, const int open_, const int ht_, const int thr, const int lane) {
  const int gold = g;
  const unsigned long long blk = ~__builtin_amdgcn_ballot_w64(open_ != 0);
  int best = 0;
  // A direction in which no one-cell step improves anything is at its fixed point already (the first cell of a longer
  // improving run would be improved by that step), and a front that travels along a row mostly travels one way.
#ifndef RNA_TSA_SCAN_BOTH
  if (__builtin_amdgcn_ballot_w64(((lane_m1(pp) - COST_S) & open_) > g))
#endif
  {
    // right-moving: sources at lower lanes.  A source in an earlier free run has a smaller count, so the maximum is the
    // best source of the lane's own run -- or the lane's own empty key, count << 25.
    const int off = (int)__umul24((unsigned)lane, (unsigned)COST_S) + 1 - thr;   // (full-rate 24-bit multiplies: v_mul_lo_u32 is quarter rate)
    const unsigned c = __builtin_amdgcn_mbcnt_hi((unsigned)(blk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)blk, 0u)) << TSA_SCAN_SHIFT;
    const unsigned t = tsa_prefix_max(c + (pp != 0 ? (unsigned)(pp + off) : 0u)) - c;
    const int cand = (int)t - off;
    const int ht_before = lane_m1(ht_, INF);   // (formed for ALL lanes, outside any condition: a wave shift inside a short-circuit `&&` runs with the other lanes switched off)
    best = ((t != 0u) & (cand + COST_S >= ht_before)) ? cand : 0;
  }
#ifndef RNA_TSA_SCAN_BOTH
  if (__builtin_amdgcn_ballot_w64(((lane_p1(pp) - COST_S) & open_) > g))
#endif
  {
    // left-moving: the same on the lanes in reverse order (lane p of the reversed wave is lane 63 - p)
    const unsigned long long blk_rev = __builtin_bitreverse64(blk);
    const int rl = 63 - lane;
    const int off = (int)__umul24((unsigned)rl, (unsigned)COST_S) + 1 - thr;
    const unsigned c = __builtin_amdgcn_mbcnt_hi((unsigned)(blk_rev >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)blk_rev, 0u)) << TSA_SCAN_SHIFT;
    const unsigned key = pp != 0 ? (unsigned)(pp + off) : 0u;
    const unsigned t_rev = tsa_prefix_max(c + (unsigned)__builtin_amdgcn_ds_bpermute(rl << 2, (int)key)) - c;
    const unsigned t = (unsigned)__builtin_amdgcn_ds_bpermute(rl << 2, (int)t_rev);
    const int cand = (int)t - off;
    const int ht_before = lane_p1(ht_, INF);
    best = max(best, ((t != 0u) & (cand + COST_S >= ht_before)) ? cand : 0);
  }
  g = max(g, best & open_);
  pp = g >= ht_ ? g : 0;
  return __builtin_amdgcn_ballot_w64(g != gold);
}

// One tile job, executed by one wavefront (lane = this wave's lane id = the cell's column inside the tile).
// `sch` supplies the scheduler-specific pieces: best() / improve_best(g) (upper bound on f*), act_cur(tile) /
// act_far(tile) (run the tile in the next round / when the next bucket opens), pool_exhausted().  `pg` is the tile's
// local page (0: none yet -- it is all "unreached"); `first`: the tile's first job in this bucket (cells that were
// held back by the previous bucket's bound may now pass their values on, so every row is evaluated and unchanged edge
// cells inside the new band wake the neighbours too).  Returns the number of rows it wrote.
//
// The 16 rows are 2 x 16 NAMED scalars (g0..g15: the field, pp0..pp15: what a cell may pass on in this bucket) and
// every per-row step is a macro pasted 16 times: with `int g[16]` and unrolled loops the optimiser turns the rows into
// one <16 x i32> value and copies all 16 registers at every row update (and spills them as a block).
// Code layout.  The search kernel is 74 KB of code and a tile job walks most of it; the instruction cache is 64 KB for two
// CUs, whose 64 wavefronts are at different places of the job at any time.  Blocks that are executed rarely (a row's scan: 9 % of
// the row evaluations, 32 copies of 400 bytes; the first job of a bucket; the start and the goal tile; overflow) carry a branch
// weight, so that the block placement moves them behind the hot path instead of leaving them inside the row sequence.
#ifndef RNA_TSA_NO_LAYOUT_HINTS
#define TSA_UNLIKELY(c) __builtin_expect(!!(c), 0)
#else
#define TSA_UNLIKELY(c) (c)
#endif
#define TSA_CAT_(a, b) a##b
#define TSA_CAT(a, b) TSA_CAT_(a, b)
#define TSA_R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define TSA_DEC_1 0
#define TSA_DEC_2 1
#define TSA_DEC_3 2
#define TSA_DEC_4 3
#define TSA_DEC_5 4
#define TSA_DEC_6 5
#define TSA_DEC_7 6
#define TSA_DEC_8 7
#define TSA_DEC_9 8
#define TSA_DEC_10 9
#define TSA_DEC_11 10
#define TSA_DEC_12 11
#define TSA_DEC_13 12
#define TSA_DEC_14 13
#define TSA_DEC_15 14
#define TSA_INC_0 1
#define TSA_INC_1 2
#define TSA_INC_2 3
#define TSA_INC_3 4
#define TSA_INC_4 5
#define TSA_INC_5 6
#define TSA_INC_6 7
#define TSA_INC_7 8
#define TSA_INC_8 9
#define TSA_INC_9 10
#define TSA_INC_10 11
#define TSA_INC_11 12
#define TSA_INC_12 13
#define TSA_INC_13 14
#define TSA_INC_14 15
#define TSA_MK_0 mk0
#define TSA_MK_1 mk0
#define TSA_MK_2 mk0
#define TSA_MK_3 mk0
#define TSA_MK_4 mk1
#define TSA_MK_5 mk1
#define TSA_MK_6 mk1
#define TSA_MK_7 mk1
#define TSA_MK_8 mk2
#define TSA_MK_9 mk2
#define TSA_MK_10 mk2
#define TSA_MK_11 mk2
#define TSA_MK_12 mk3
#define TSA_MK_13 mk3
#define TSA_MK_14 mk3
#define TSA_MK_15 mk3
#define TSA_G(b) TSA_CAT(g, b)
#define TSA_PP(b) TSA_CAT(pp, b)
#define TSA_MKW(b) TSA_CAT(TSA_MK_, b)   // the mask word that holds row b's byte, at bit 8 * (b & 3)
template <class Sched>
__device__ __forceinline__ void tsa_job(Sched& sch, unsigned* scr, const int lane_in, const int t_in, const TsaCtx& C,
                                       const int first_in, const unsigned bucket_end, const int key_base, const int key_shift, int* spare TSA_ACC_PARAM) {
  // STICKY TILES (round 5).  A tile that is woken while its job runs used to be queued again by its wavefront (class 0)
  // and, as a rule, taken again at once by the same wavefront: a push, a pop, a claim, the sixteen rows, the masks and the
  // sixteen pass-on values loaded and formed again -- for a third of all jobs (scripts/sim_async.c: 36 % of the jobs are
  // such re-runs, half of them find nothing).  Now the wavefront KEEPS the tile: when a wake-up is pending at the end of
  // the job (Sched::finish) it goes round the loop below with the rows and the masks still in
  // registers and pulls only the halo again (`sticky`).  The schedule is the one the queue would have produced (class 0 is
  // taken first), so jobs, results and exactness are unchanged; what goes is ~430 instructions and the queue round trip.
  // (loop-carried flags as 32-bit scalars behind an opaque copy: as `bool` they become lane-mask phis, every selection on them
  // a v_cndmask, and the scalar row sets that depend on `first` end up in VGPRs -- "illegal VGPR to SGPR copy")
  int first_w = first_in, sticky_w = 0;
  int g0, g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15;
  unsigned mk0 = 0u, mk1 = 0u, mk2 = 0u, mk3 = 0u, fbits_ld = 0u;   // byte b of mk* of this lane = neighbour mask of cell (lane, b); bit b of fbits: the cell is free
  unsigned pg = 0u;
  for (;;) {   // one turn per job of this tile
  {
  // (the tile number behind an opaque copy per turn, like the lane id below: what is derived from it -- the tile's bit masks
  // for the scheduler's words, its number as a vector value for the page list -- is otherwise hoisted out of this loop as
  // loop-invariant VGPRs and spilled to scratch, whose reloads inside the job wait for every store in flight)
  int t = t_in;
  asm volatile("" : "+s"(t));
  const int tiles_i = C.tiles_i, tiles_j = C.tiles_j, gi = C.gi, gj = C.gj;
  // tile number -> (ti, tj): a multiplication by floor(2^32 / tiles_i) + 1, exact for t < 2^16 and tiles_i <= 2^16 (the error term
  // t * e / 2^32 with e <= tiles_i stays below one tiles_i-th); the compiler's division by a run-time value was 18 scalar
  // instructions at the top of every turn
  const int tj = C.tiles_magic ? (int)__umulhi((unsigned)t, C.tiles_magic) : t;
  const int ti = t - tj * tiles_i;
  const int i0 = ti * TI, j0 = tj * TJ;
  first_w = __builtin_amdgcn_readfirstlane(first_w);
  sticky_w = __builtin_amdgcn_readfirstlane(sticky_w);
  const bool first = first_w != 0, sticky = sticky_w != 0;
  // an opaque copy of the lane id per job (and one more for the results phase): everything derived from it is then
  // recomputed here instead of being hoisted out of the job loop, kept alive across the sweeps and spilled to scratch
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  TSA_T(t_a);
  asm volatile("; TSA_MARK job_begin");
  // ---- 1. page table look-up of the eight neighbouring tiles (lane k < 8: direction k; 0 = none or outside) and of
  //         the tile itself (lane 8) ----
  unsigned nb_pg = 0u;
  int nb_t = -1;
  {
    // (arithmetic instead of kdi_of / kdj_of: their comparisons became a cascade of branches on the lane id.  The own
    // page is final while this job runs: only the tile's own job changes it, and a tile never runs twice at once.)
    const int kk = lane < 4 ? lane : (lane < 8 ? lane + 1 : 4);   // cell of the 3 x 3 block of tiles, row-major; lane 8: the centre
    // kj = kk / 3 = (11 kk) >> 5 and kk - 3 kj with full-rate 24-bit multiplies (the compiler took v_mul_lo_u32 for both)
    int kj, ki;
    asm("v_mul_u32_u24 %0, %1, 11" : "=v"(kj) : "v"(kk));
    kj >>= 5;
    asm("v_mad_i32_i24 %0, %1, -3, %2" : "=v"(ki) : "v"(kj), "v"(kk));
    const int nti = ti + ki - 1, ntj = tj + kj - 1;
    if (lane < 9 && nti >= 0 && ntj >= 0 && nti < tiles_i && ntj < tiles_j) {
      const int tt = (int)__umul24((unsigned)ntj, (unsigned)tiles_i) + nti;   // (ntj >= 0 here; at most 2^20 tiles)
      nb_pg = C.page_of(tt);   // (a sticky turn looks them up again: kept in LDS across the turns and looked up only where there was none, the bench measured the same -- profiles/r05_ab_sticky_pages.txt)
      nb_t = lane < 8 ? tt : -1;
    }
  }
  pg = (unsigned)__builtin_amdgcn_readlane((int)nb_pg, 8);
#if RNA_TSA_FRESH || RNA_TSA_CONFIRM
  if (lane < 8) scr[lane] = nb_pg;   // for the end of the job: the wake tests read the neighbours' edges again (TSA_FRESH / TSA_CONFIRM)
#endif
  // ---- 2. everything the job reads, issued before the first wait ----
  // (the tile's snapshot through a scalar base and 32-bit lane offsets of known range: as `base + lane * 16` and
  // `(ushort*)(base + 1056)[lane]` the second address was a 64-bit multiply-add per lane)
  const uint8_t* const snap = C.nbr_tm + (size_t)t * MASK_STRIDE;
  const unsigned ulane_ld = (unsigned)lane & 63u;
  if (!sticky) {   // (a sticky turn still holds the rows as it stored them, and the masks)
    const unsigned* own = C.pages + (pg << 10);
#define TSA_LOAD(b) TSA_G(b) = (int)ld_l2(&own[(b) * TI + lane]);
    TSA_R16(TSA_LOAD)
#undef TSA_LOAD
    const uint4 mv = *reinterpret_cast<const uint4*>(snap + (ulane_ld << 4));
    mk0 = mv.x; mk1 = mv.y; mk2 = mv.z; mk3 = mv.w;
    fbits_ld = *reinterpret_cast<const unsigned short*>(snap + TILE_WORDS + 32 + (ulane_ld << 1));
  }
  // X: the two halo columns and the four corners.  Lanes 0..17 hold the left one top-down (lane 0 = corner (-1,-1),
  // lanes 1..16 = rows 0..15, lane 17 = corner (-1,16)), lanes 32..49 the right one.
  const int xl = lane & 31;
  const bool xr = lane >= 32, xcell = xl >= 1 && xl <= TJ;
  int top, bot, X = 0, gcol = 0;
  unsigned eb = 0u;   // lanes 1..16 / 33..48: mask of the tile's own edge cell (0, xl-1) / (63, xl-1)
  {
    // (ds_bpermute with the byte address of the source lane, not __shfl: that adds the wavefront's base lane -- zero here, but a
    // vector value the compiler keeps across the whole kernel, i.e. one of the 64 registers, or a scratch reload at the top of a job)
    const unsigned pgN = (unsigned)__builtin_amdgcn_ds_bpermute(1 << 2, (int)nb_pg), pgS = (unsigned)__builtin_amdgcn_ds_bpermute(6 << 2, (int)nb_pg);
    top = (int)ld_l2(&C.pages[(pgN << 10) + (TJ - 1) * TI + lane]);
    bot = (int)ld_l2(&C.pages[(pgS << 10) + lane]);
    // direction of the tile a halo-column lane reads: NW W SW = 0 3 5 on the left, NE E SE = 2 4 7 on the right
    // (arithmetic: as nested selections it became three levels of exec-mask branches)
    const int xz = (xl > 0 ? 1 : 0) + (xl > TJ ? 1 : 0);
    const int xdir = 3 * xz - (xz >> 1) + (xr ? 2 - (xz & 1) : 0);
    const unsigned pgX = (unsigned)__builtin_amdgcn_ds_bpermute(xdir << 2, (int)nb_pg);
    const int xrow = xl == 0 ? TJ - 1 : (xl <= TJ ? xl - 1 : 0);
    if (xl <= TJ + 1) X = (int)ld_l2(&C.paux[pgX * AUX_WORDS + (xr ? 0 : 16) + xrow]);   // a left tile's column 63 / a right tile's column 0
    if (xcell) eb = snap[TILE_WORDS + ((ulane_ld & 31u) - 1u) + (ulane_ld & 32u ? 16u : 0u)];   // (xl - 1 + (xr ? 16 : 0))
    // the tile's own columns 0 and 63 in the same layout (the copy it keeps for its neighbours)
    if (xcell) gcol = (int)ld_l2(&C.paux[pg * AUX_WORDS + (xr ? 16 : 0) + xl - 1]);
  }
  asm volatile("; TSA_MARK loads_issued");
  // ---- 3. this bucket's bound; which cells are free ----
  // A blocked (or outside) cell keeps u = 0 for ever: every candidate is ANDed with the cell's free bit, and an
  // "unreached" cell never passes anything on.  The heuristic of a row is recomputed where it is needed (one scalar
  // |dj| and four vector instructions) instead of living in 16 registers: the kernel has to fit 64 VGPRs so that eight
  // wavefronts share a SIMD.
  const int best_in = __builtin_amdgcn_readfirstlane(sch.best());   // (wave-uniform: the sticky turn branches on the bound it gives)
  // pass on iff f < lim = min(end of the bucket, best + 1), in unsigned 32 bits: `bucket_end` arrives clamped to 2^31,
  // best + 1 <= 2^31 (as 64-bit integers these few scalar values cost vector compares and four spilled registers)
  const unsigned best1 = (unsigned)best_in + 1u;
  const unsigned lim_u = bucket_end < best1 ? bucket_end : best1;
  const int thr = __builtin_amdgcn_readfirstlane(KU - (int)(lim_u > (unsigned)INF ? (unsigned)INF : lim_u) + 1);   // (wave-uniform by construction: the sticky turn branches on it)
#if RNA_TSA_SUPER >= 2
  const bool scan_ok = lim_u < TSA_SCAN_LIM;   // the keys of tsa_row_fixpoint fit
#endif
  const int dxl = abs(i0 + lane - gi);
  const int dx414 = (int)__umul24((unsigned)dxl, (unsigned)(COST_D - COST_S)) + thr;   // 414 dx + thr: the usual addend rides in the lane's register
  // h + add of row b:  1000 max(dx, dy) + 414 min(dx, dy) = 586 max(dx, dy) + 414 dx + 414 dy -- one maximum, one 24-bit
  // multiply-add and one addition of a scalar (414 dy + add) per use, with the lane's 414 dx kept in a register (five
  // instructions when the minimum was formed as well; the second register is there since the row sets stopped spilling).
  // (all three instructions are volatile asm: otherwise the 16 row heuristics are hoisted out of the sweeps into 16 VGPRs --
  // or their scalar halves into 32 SGPRs -- again.  |jg + b| is ONE s_absdiff_i32; as abs() of an opaque copy it was a
  // copy, an addition, a negation and a maximum.  With thr inside the lane's register the scalar part of the common case,
  // h + thr, is that and one s_mulk_i32.)
  const int jg = j0 - gj;
#define TSA_HC(b, add) ({ int dy_, m_, r_; asm volatile("s_absdiff_i32 %0, %1, %2" : "=s"(dy_) : "s"(jg), "n"(-(b)) : "scc");   /* |jg + b| */ \
                          asm volatile("v_max_i32 %0, %1, %2" : "=v"(m_) : "s"(dy_), "v"(dxl));                           \
                          asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r_) : "v"(m_), "s"(2 * COST_S - COST_D), "v"(dx414)); \
                          r_ + ((COST_D - COST_S) * dy_ + ((add) - thr)); })
#define TSA_H(b) TSA_HC(b, 0)
#define TSA_HT(b) TSA_HC(b, thr)   /* a cell passes on iff u >= h + thr */
  // bit b: this lane's cell in row b is free (and inside the map) -- from the snapshot, loaded with the masks
  const unsigned fbits = fbits_ld;
  TSA_T(t_h0);
  TSA_ACC(9, t_a, t_h0);   // issuing the loads + waiting for them
  asm volatile("; TSA_MARK halo_begin");
  // ---- 4. what the halo can contribute (once: it does not change during the job) ----
  // rows to evaluate in the next down / up sweep: a row is evaluated from above when the row above it changed (or it
  // took something from the halo itself), from below when the row below it changed.  An evaluation leaves the row at
  // the fixed point of its own horizontal steps (the passes along the row run until nothing moves, see TSA_ROW), so a
  // row never has to be looked at again for its own sake.
  unsigned nd = 0u, nu = 0u;
  unsigned rowchg = 0u;                 // rows that changed at all in this job
  unsigned long long q0 = 0ull, q15 = 0ull, qany = 0ull;   // cells of row 0 / row 15 that changed and may pass their value on; lanes that changed in any row
  int pp0, pp1, pp2, pp3, pp4, pp5, pp6, pp7, pp8, pp9, pp10, pp11, pp12, pp13, pp14, pp15;
  // a row took better values in the lanes `up`: flag the rows next to it, remember which cells wake neighbours
  // DIR: 0 = the row was evaluated in a down sweep (its vertical candidates came from the row above), 1 = in an up
  // sweep; `pure`: every lane that improved took a VERTICAL candidate and no pass along the row moved anything after.
  // Such a row cannot improve the row its values came from: a candidate back into that row is the source row's own
  // value minus two steps (>= 2000), and inside the source row -- which is at the fixed point of its horizontal steps --
  // the same cell is reached from the same source within two horizontal steps (<= 2000) whenever the return is legal
  // (a diagonal step needs both orthogonal cells free, which are exactly the cells of the in-row route; a cell of that
  // route that is held back by the bucket bound holds back its successor as well: f does not decrease along it).  So
  // the row BEHIND the sweep is not flagged -- half of all row evaluations used to be such echoes that found nothing
  // (scripts/sim_async.c, SIM_VPURE: 19.4 -> 13.4 evaluations per job, cost and settled count still the oracle's).
  // (DIR == 2: a change from the halo, both neighbours are flagged; the flag behind the sweep is set by five scalar
  // instructions in inline assembly -- the compiler's version went through a vector select and a readfirstlane)
#define TSA_FLAG_BEHIND(word, bit, hsrc, left)                                                                   \
  if ((bit) != 0u) {                                                                                             \
    unsigned t_;                                                                                                 \
    asm volatile("s_cmp_lg_u64 %[h], 0\n\t"                                                                      \
                 "s_cselect_b32 %[t], %[b], 0\n\t"                                                               \
                 "s_cmp_lg_u32 %[l], %[i]\n\t"                                                                   \
                 "s_cselect_b32 %[t], %[b], %[t]\n\t"                                                            \
                 "s_or_b32 %[w], %[w], %[t]"                                                                     \
                 : [w] "+s"(word), [t] "=&s"(t_)                                                                 \
                 : [h] "s"(hsrc), [l] "s"(left), [b] "n"(bit), [i] "n"(1u << (RNA_TSA_HPASS - 1))                \
                 : "scc");                                                                                       \
  }
#define TSA_ROW_CHANGED(b, up, DIR, hsrc, left)                                                                  \
  {                                                                                                              \
    if ((DIR) == 0) { nd |= (2u << (b)) & 0xffffu; TSA_FLAG_BEHIND(nu, (1u << (b)) >> 1, hsrc, left) }           \
    else { nu |= (1u << (b)) >> 1; TSA_FLAG_BEHIND(nd, (2u << (b)) & 0xffffu, hsrc, left) }                      \
    rowchg |= 1u << (b);                                                                                         \
    qany |= (up);                                                                                                \
    if ((b) == 0) q0 |= (up) & __builtin_amdgcn_ballot_w64(TSA_PP(b) != 0);                                      \
    if ((b) == TJ - 1) q15 |= (up) & __builtin_amdgcn_ballot_w64(TSA_PP(b) != 0);                                \
  }
  {
    int cT, cB, cX;
    {
      const int pT = top >= TSA_HT(-1) ? top : 0, pB = bot >= TSA_HT(TJ) ? bot : 0;
      const int hX = tsa_octile24(xr ? i0 + TI : i0 - 1, j0 + xl - 1, gi, gj);
      const int pX = xl <= TJ + 1 ? tsa_prop(X, hX, thr) : 0;
      const int sTL = __builtin_amdgcn_readlane(pX, 0), sBL = __builtin_amdgcn_readlane(pX, TJ + 1);
      const int sTR = __builtin_amdgcn_readlane(pX, 32), sBR = __builtin_amdgcn_readlane(pX, 32 + TJ + 1);
      // rows 0 and 15 from the rows beyond them (masks: k0 k1 k2 = (-1,-1) (0,-1) (1,-1); k5 k6 k7 = (-1,1) (0,1) (1,1));
      // a straight step needs no mask test: a blocked source reads "unreached", a blocked target is gated by its free bit
      cT = max3i(pT - COST_S, (lane_m1(pT, sTL) - COST_D) & __builtin_amdgcn_sbfe((int)mk0, 0, 1),
                 (lane_p1(pT, sTR) - COST_D) & __builtin_amdgcn_sbfe((int)mk0, 2, 1)) & TSA_OPEN(fbits, 0);
      cB = max3i(pB - COST_S, (lane_m1(pB, sBL) - COST_D) & __builtin_amdgcn_sbfe((int)mk3, 24 + 5, 1),
                 (lane_p1(pB, sBR) - COST_D) & __builtin_amdgcn_sbfe((int)mk3, 24 + 7, 1)) & TSA_OPEN(fbits, TJ - 1);
      // columns 0 and 63 from the columns beyond them, computed where the halo column sits (lane xl = row + 1)
      const int kN = xr ? 2 : 0, kS = xr ? 7 : 5;
      cX = max3i(pX - COST_S, (lane_m1(pX) - COST_D) & -(int)((eb >> kN) & 1u), (lane_p1(pX) - COST_D) & -(int)((eb >> kS) & 1u));
      cX = (xcell && eb != 0u) ? cX : 0;
    }
    unsigned planted = 0u;
    if (TSA_UNLIKELY(t == C.ts)) {   // the start cell: g = 0, whatever its mask says (a blocked start still answers start == goal)
      // (the row as a one-bit set tested with a shift: compared as `b == C.sb` the sixteen conditions were hoisted out of
      // the job loop as sixteen lane masks, 32 spilled SGPRs)
      unsigned srow = 1u << C.sb;
      asm volatile("" : "+s"(srow));
#define TSA_PLANT(b)                                                                        \
  if (((srow >> (b)) & 1u) && __builtin_amdgcn_readlane(TSA_G(b), C.sa) != KU) {            \
    TSA_G(b) = lane == C.sa ? KU : TSA_G(b);                                                \
    planted = 1u << (b);                                                                    \
  }
      TSA_R16(TSA_PLANT)
#undef TSA_PLANT
    }
    // What improves?  Rows 0 and 15 across the wave; the two columns where the halo column sits, against the tile's
    // own columns in the same layout (gcol) -- one comparison for the 32 edge cells of rows 0..15.  Most wake-ups of
    // an asynchronous schedule find the tile up to date already (a third of all jobs): they end here, before the 16
    // rows' pass-on values are formed.
    const unsigned long long upT = __builtin_amdgcn_ballot_w64(cT > g0), upB = __builtin_amdgcn_ballot_w64(cB > g15);
    const unsigned long long imask = __builtin_amdgcn_ballot_w64(cX > gcol);
    if (!(upT | upB | imask) && !planted && !first) {   // the wake-up brought nothing better
      TSA_CNT(10, 1); if (sticky) TSA_CNT(23, 1); TSA_T(t_n); TSA_ACC(15, t_a, t_n);
      goto tsa_job_done;   // (counted in Sched::finish like every turn; the jobs that get past this point are counted at the end of the job)
    }
    asm volatile("; TSA_MARK noop_decided");
#if !RNA_TSA_FRESH
    scr[84 + lane] = (unsigned)top;        // kept for the end of the job: does a changed edge row beat what the
    scr[84 + 64 + lane] = (unsigned)bot;   // neighbour already has?
    scr[84 + 128 + lane] = (unsigned)X;
#endif
    scr[84 + 192 + 48 + lane] = eb;
    const unsigned cl = (unsigned)(imask >> 1) & 0xffffu, cr = (unsigned)(imask >> 33) & 0xffffu;   // rows whose cell in lane 0 / lane 63 improves
    const unsigned crow = cl | cr | planted;
    if (upT) g0 = max(g0, cT);
    if (upB) g15 = max(g15, cB);
    // a column candidate goes from its lane of cX into lane 0 / 63 of the row's register through the scalar unit: only
    // rows named in cl / cr, and most jobs have none (round 3 tested every row for it, twice, inside the per-row block
    // below: 14 scalar instructions per row; the kernel is as sensitive to a scalar instruction as to 0.6 vector ones)
    if (cl | cr) {
      // A named cell's candidate beat the tile's own column as loaded, and rows 1..14 have not changed since: it is
      // written as it is.  Rows 0 and 15 may just have taken cT / cB: they take the larger of the two; a planted start
      // cell is planted again afterwards.
#define TSA_APPLY_COL(b)                                                                                         \
  if (((cl | cr) >> (b)) & 1u) {                                                                                 \
    const bool keep_ = (b) == 0 || (b) == TJ - 1;                                                                \
    if ((cl >> (b)) & 1u) {                                                                                      \
      int c_ = __builtin_amdgcn_readlane(cX, (b) + 1);                                                           \
      if (keep_) c_ = max(c_, __builtin_amdgcn_readlane(TSA_G(b), 0));                                           \
      asm volatile("v_writelane_b32 %0, %1, 0" : "+v"(TSA_G(b)) : "s"(c_));                                      \
    }                                                                                                            \
    if ((cr >> (b)) & 1u) {                                                                                      \
      int c_ = __builtin_amdgcn_readlane(cX, 33 + (b));                                                          \
      if (keep_) c_ = max(c_, __builtin_amdgcn_readlane(TSA_G(b), TI - 1));                                      \
      asm volatile("v_writelane_b32 %0, %1, 63" : "+v"(TSA_G(b)) : "s"(c_));                                     \
    }                                                                                                            \
  }
      TSA_R16(TSA_APPLY_COL)
#undef TSA_APPLY_COL
      if (planted) {
#define TSA_REPLANT(b) if ((planted >> (b)) & 1u) TSA_G(b) = lane == C.sa ? KU : TSA_G(b);
        TSA_R16(TSA_REPLANT)
#undef TSA_REPLANT
      }
    }
    // what each cell may pass on in this bucket: sixteen rows, no branches (also on a sticky turn: kept across the turns, the
    // sixteen values were live through the page look-ups and the halo step, and the kernel spilled into scratch -- whose
    // loads wait for every store in flight; a turn that finds nothing has left before this point anyway)
#define TSA_APPLY_PP(b) TSA_PP(b) = TSA_G(b) >= TSA_HT(b) ? TSA_G(b) : 0;
    TSA_R16(TSA_APPLY_PP)
#undef TSA_APPLY_PP
    // the rows the halo changed, as whole words: each is evaluated in the first sweep (its own horizontal steps) and
    // flags the rows next to it
    {
      const unsigned chg = crow | (upT ? 1u : 0u) | (upB ? 1u << (TJ - 1) : 0u);
      nd |= (chg | (chg << 1)) & 0xffffu;
      nu |= chg >> 1;
      rowchg |= chg;
      const unsigned long long l0 = 1ull, l63 = 1ull << (TI - 1), lsa = planted ? 1ull << C.sa : 0ull;
      qany |= upT | upB | (cl ? l0 : 0ull) | (cr ? l63 : 0ull) | lsa;
      // cells of rows 0 / 15 that changed and may pass their value on (they wake the tiles beyond)
      const unsigned long long up0 = upT | ((cl & 1u) ? l0 : 0ull) | ((cr & 1u) ? l63 : 0ull) | ((planted & 1u) ? lsa : 0ull);
      const unsigned long long up15 = upB | ((cl >> (TJ - 1)) ? l0 : 0ull) | ((cr >> (TJ - 1)) ? l63 : 0ull) | ((planted >> (TJ - 1)) ? lsa : 0ull);
      if (up0) q0 |= up0 & __builtin_amdgcn_ballot_w64(pp0 != 0);
      if (up15) q15 |= up15 & __builtin_amdgcn_ballot_w64(pp15 != 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("; TSA_MARK halo_end");
#if RNA_TSA_FIRST_ROWS
  // A tile's FIRST job in a bucket used to evaluate every row both ways (33.8 evaluations for 9.7 rows that change: 4 % of the
  // jobs made 8 % of all evaluations).  What the new bound changes inside the tile is that cells it RELEASES pass their values
  // on: cells with f in [the bucket's start, the new bound) -- they pass on now (pp != 0) and u - h <= KU - (f at the bucket's
  // start) = key_base.  Everything with a smaller f was at the fixed point of the tile when its last job ended, so only the
  // rows that hold a released cell (and, through them, their neighbours) have anything new to offer; what the halo brings is
  // flagged by the halo step as in any job.  Exact on the model (scripts/sim_async.c: first jobs with released rows only, cost
  // and settled count of every query the oracle's, at 96 000 and 24 000 per bucket).  A bucket that is run again (parked
  // wake-ups) releases the same cells again: evaluated once more, harmless.
  if (TSA_UNLIKELY(first)) {
    unsigned rel = 0u;
    // (pp is the cell's value or 0, and h + key_base > 0: `pp != 0 && pp <= h + key_base` is ONE unsigned comparison -- as two
    // conditions the compiler built an exec-masked block per row, 17 instructions)
#define TSA_RELEASED(b) if (__builtin_amdgcn_ballot_w64((unsigned)(TSA_PP(b) - 1) < (unsigned)TSA_HC(b, key_base)) != 0ull) rel |= 1u << (b);
    TSA_R16(TSA_RELEASED)
#undef TSA_RELEASED
    nd |= (rel | (rel << 1)) & 0xffffu;
    nu |= rel >> 1;
  }
#else
  if (first) nd = nu = 0xffffu;
#endif
  TSA_T(t_b);
  TSA_ACC(0, t_a, t_b);
  // ---- 5. sweeps ----
#ifdef RNA_TSA_STATS
  int evals = 0, hpass = 0;
#define TSA_STAT_INC(v) v += 1
#define TSA_STAT_HP(left) hpass += (left) ? RNA_TSA_HPASS - 1 - (31 - __builtin_clz(left)) : RNA_TSA_HPASS   /* passes that moved something */
#else
#define TSA_STAT_INC(v)
#define TSA_STAT_HP(left)
#endif
  // the (negative) step costs live in VGPRs: v_add_u32 with a DPP source cannot take a literal, and only then does the
  // wave shift fold into the add (one instruction instead of v_mov_dpp + v_add)
  int nS = -COST_S, nD = -COST_D;
  asm volatile("" : "+v"(nS), "+v"(nD));
  // one row: VERT = the best of the three candidates from the row before it in sweep direction (0 for the first row),
  // then the two from the row's own neighbours; improved lanes take the candidate (a plain max: the others' is not
  // better than what they hold).  A row that changed runs on along itself until nothing moves (a front that travels
  // along the lanes would otherwise advance one cell per sweep) -- at most RNA_TSA_HPASS passes, then the row is
  // flagged for the next sweep (AGAIN).  The loop body is 8 vector and 3 scalar instructions per pass: the row's free
  // mask and its pass-on threshold h + thr are formed once per evaluation.
  /* developer builds: extra instructions per row evaluation, to measure which issue port the kernel is sensitive to */
#define TSA_STR_(x) #x
#define TSA_STR(x) TSA_STR_(x)
#if defined(RNA_TSA_PAD_VALU)
#define TSA_PAD asm volatile(".rept " TSA_STR(RNA_TSA_PAD_VALU) "\n v_nop\n .endr")
#elif defined(RNA_TSA_PAD_SALU)
#define TSA_PAD { int pad_ = 0; asm volatile(".rept " TSA_STR(RNA_TSA_PAD_SALU) "\n s_add_u32 %0, %0, 1\n .endr" : "+s"(pad_) : : "scc"); }
#else
#define TSA_PAD
#endif
#define TSA_VERT(b, src, kA, kC)                                                                                                   \
  max3i(TSA_PP(src) + nS, (lane_m1(TSA_PP(src)) + nD) & TSA_OPEN(TSA_MKW(b), 8 * ((b) & 3) + (kA)),                                \
        (lane_p1(TSA_PP(src)) + nD) & TSA_OPEN(TSA_MKW(b), 8 * ((b) & 3) + (kC)))
  // A front that travels ALONG a row advances one cell per pass: 9 % of the row evaluations ran out of their 16 passes and
  // made 68 % of all passes (scripts/sim_async.c), and crossing the tile that way takes 64 passes per row where crossing it
  // the other way takes 16 row evaluations for all 64 lanes at once.  So after RNA_TSA_HPASS one-cell passes that all moved
  // something the row goes on in LOG STEPS: a super-pass is one one-cell pass (wave shifts: it crosses the 16-lane DPP
  // rows) followed by shifts of 2, 4 and 8 lanes inside the DPP rows, each in both directions -- values travel up to 15
  // cells per super-pass.  A shift by d is a legal shortcut for d one-cell steps iff the d cells up to the target are
  // free (run masks: mR_d = AND of the free bits of lanes l-d+1..l, by doubling on the scalar unit) and every cell in
  // between may pass the value on -- f does not decrease along a path (consistent heuristic), so it is enough to test the
  // LAST cell in between: candidate + 1000 >= h + thr of the lane next to the target (htl_ / htr_).
  // Measured (profiles/r04_ab_log_step_rows.txt): a super-pass is ~70 instructions and crosses one 16-lane DPP row, so a run
  // of 20-30 cells costs about what its one-cell passes cost (12 each); with the 26 KB of cold code the 32 copies add, the
  // kernel is 0.4 % faster at best.  Mode 2 replaces the log steps by one scan over the whole wavefront.
#if RNA_TSA_SUPER == 2
#define TSA_HP_OUT_OF_PASSES
#define TSA_SUPER(b, AGVAR)                                                                                      \
  if (TSA_UNLIKELY(left_ == 0u)) {   /* the one-cell passes ran out while the row was still moving: the rest in one scan */     \
    if (scan_ok) {                                                                                               \
      TSA_CNT(14, 1);                                                                                            \
      up_ |= tsa_row_fixpoint(TSA_G(b), TSA_PP(b), open_, ht_, thr, lane);                                       \
    } else {                                                                                                     \
      AGVAR |= 1u << (b);                                                                                        \
    }                                                                                                            \
  }
#elif RNA_TSA_SUPER
#define TSA_HP_OUT_OF_PASSES
#define TSA_SUPER_STEP(b, d, mr, ml)                                                                             \
  {                                                                                                              \
    const int cr_ = __builtin_amdgcn_update_dpp(0, TSA_PP(b), 0x110 + (d), 0xF, 0xF, true) - (d) * COST_S;       \
    const int cl_ = __builtin_amdgcn_update_dpp(0, TSA_PP(b), 0x100 + (d), 0xF, 0xF, true) - (d) * COST_S;       \
    const unsigned long long okr_ = __builtin_amdgcn_ballot_w64(cr_ >= htl_) & (mr);                            \
    const unsigned long long okl_ = __builtin_amdgcn_ballot_w64(cl_ >= htr_) & (ml);                            \
    const int tr_ = __builtin_amdgcn_inverse_ballot_w64(okr_) ? cr_ : 0;                                         \
    const int tl_ = __builtin_amdgcn_inverse_ballot_w64(okl_) ? cl_ : 0;                                         \
    TSA_G(b) = max3i(TSA_G(b), tr_, tl_);                                                                        \
    TSA_PP(b) = TSA_G(b) >= ht_ ? TSA_G(b) : 0;                                                                  \
  }
#define TSA_SUPER(b, AGVAR)                                                                                      \
  if (left_ == 0u) {   /* the one-cell passes ran out while the row was still moving */                         \
    const unsigned long long o_ = __builtin_amdgcn_ballot_w64(open_ != 0);                                       \
    const unsigned long long mr2_ = o_ & (o_ << 1), ml2_ = o_ & (o_ >> 1);                                       \
    const unsigned long long mr4_ = mr2_ & (mr2_ << 2), ml4_ = ml2_ & (ml2_ >> 2);                               \
    const unsigned long long mr8_ = mr4_ & (mr4_ << 4), ml8_ = ml4_ & (ml4_ >> 4);                               \
    /* h + thr of the lane before / after, minus one step: candidate >= it  <=>  the cell in between passes on */ \
    const int htl_ = lane_m1(ht_, INF) - COST_S, htr_ = lane_p1(ht_, INF) - COST_S;                              \
    int sp_ = 0;                                                                                                 \
    for (;;) {                                                                                                   \
      const int gold_ = TSA_G(b);                                                                                \
      {                                                                                                          \
        const int c1_ = max(lane_m1(TSA_PP(b)) + nS, lane_p1(TSA_PP(b)) + nS) & open_;                           \
        TSA_G(b) = max(TSA_G(b), c1_);                                                                           \
      }                                                                                                          \
      /* no one-cell step moves anything: the row is at its fixed point (a longer shortcut that improved its     \
         target would have a first cell along its way that a one-cell step improves) */                          \
      const unsigned long long mv1_ = __builtin_amdgcn_ballot_w64(TSA_G(b) != gold_);                            \
      TSA_CNT(14, 1);                                                                                            \
      if (!mv1_) break;                                                                                          \
      TSA_PP(b) = TSA_G(b) >= ht_ ? TSA_G(b) : 0;                                                                \
      TSA_SUPER_STEP(b, 2, mr2_, ml2_)                                                                           \
      TSA_SUPER_STEP(b, 4, mr4_, ml4_)                                                                           \
      TSA_SUPER_STEP(b, 8, mr8_, ml8_)                                                                           \
      up_ |= __builtin_amdgcn_ballot_w64(TSA_G(b) != gold_);                                                     \
      if (++sp_ == 8) { AGVAR |= 1u << (b); break; }   /* (a value crosses the tile in five) */                   \
    }                                                                                                            \
  }
#else
#define TSA_HP_OUT_OF_PASSES "s_bitset1_b32 %[ag], %[bit]\n"   /* out of passes: the row is looked at again in the next sweep */
#define TSA_SUPER(b, AGVAR)
#endif
#define TSA_ROW(b, VERT, AGVAR, DIR)                                                                                \
  {                                                                                                              \
    TSA_STAT_INC(evals);                                                                                         \
    TSA_PAD;                                                                                                     \
    const int open_ = TSA_OPEN(fbits, b);                                                                        \
    const int mv_ = VERT;                                                                                        \
    const int m_ = max3i(mv_, lane_m1(TSA_PP(b)) + nS, lane_p1(TSA_PP(b)) + nS) & open_;                         \
    unsigned long long up_ = __builtin_amdgcn_ballot_w64(m_ > TSA_G(b));                                         \
    if (up_) {                                                                                                   \
      /* lanes that improved by a candidate from the row's own neighbours (it beats the vertical one) */          \
      unsigned long long hsrc_ = up_ & __builtin_amdgcn_ballot_w64(m_ > (mv_ & open_));                          \
      asm volatile("" : "+s"(hsrc_));   /* formed here: the pass loop then updates up_ in place, no copy */        \
      const int ht_ = TSA_HT(b);        /* passes on iff u - h >= thr */                                         \
      TSA_G(b) = max(TSA_G(b), m_);                                                                              \
      TSA_PP(b) = TSA_G(b) >= ht_ ? TSA_G(b) : 0;                                                                \
      /* the passes along the row, hand-scheduled: 8 vector + 4 scalar instructions per pass that moves something, 5 + 1 \
         for the last one (the compiler's version of this loop spent 7 scalar instructions and two s_nop per pass on the    \
         loop control).  `left_` holds one bit per pass still allowed. */                                                \
      unsigned left_ = 1u << (RNA_TSA_HPASS - 1);                                                                  \
      {                                                                                                          \
        int t1_, t2_;                                                                                            \
        asm volatile(                                                                                            \
            "s_nop 1\n"   /* pp was written by the instruction before: two wait states before a DPP read */      \
            ".Lhp_top%=:\n\t"                                                                                     \
            "v_add_u32_dpp %[t1], %[pp], %[nS] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"             \
            "v_add_u32_dpp %[t2], %[pp], %[nS] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"             \
            "v_max_i32 %[t1], %[t1], %[t2]\n\t"                                                                   \
            "v_and_b32 %[t1], %[t1], %[open]\n\t"                                                                 \
            "v_cmp_gt_i32 vcc, %[t1], %[g]\n\t"                                                                   \
            "s_cbranch_vccz .Lhp_done%=\n\t"                                                                      \
            "v_max_i32 %[g], %[g], %[t1]\n\t"                                                                     \
            "s_or_b64 %[all], %[all], vcc\n\t"                                                                    \
            "v_cmp_ge_i32 vcc, %[g], %[ht]\n\t"                                                                   \
            "s_nop 1\n\t"                                                                                         \
            "v_cndmask_b32 %[pp], 0, %[g], vcc\n\t"                                                               \
            "s_lshr_b32 %[left], %[left], 1\n\t"                                                                  \
            "s_cbranch_scc1 .Lhp_top%=\n\t"                                                                       \
            TSA_HP_OUT_OF_PASSES                                                                                  \
            ".Lhp_done%=:"                                                                                        \
            : [g] "+v"(TSA_G(b)), [pp] "+v"(TSA_PP(b)), [all] "+s"(up_), [left] "+s"(left_), [ag] "+s"(AGVAR),    \
              [t1] "=&v"(t1_), [t2] "=&v"(t2_)                                                                   \
            : [nS] "v"(nS), [open] "v"(open_), [ht] "v"(ht_), [bit] "n"(b)                                       \
            : "vcc", "scc");                                                                                     \
      }                                                                                                          \
      TSA_STAT_HP(left_);                                                                                        \
      TSA_SUPER(b, AGVAR)                                                                                        \
      TSA_ROW_CHANGED(b, up_, DIR, hsrc_, left_)                                                                 \
    }                                                                                                            \
  }
#define TSA_DOWN(b) if ((nd >> (b)) & 1u) { nd &= ~(1u << (b)); TSA_ROW(b, TSA_VERT(b, TSA_CAT(TSA_DEC_, b), 0, 2), nu, 0) }
#define TSA_UP(b) if ((nu >> (b)) & 1u) { nu &= ~(1u << (b)); TSA_ROW(b, TSA_VERT(b, TSA_CAT(TSA_INC_, b), 5, 7), nd, 1) }
  // (the rows are tested four at a time first: the flags of a job cluster -- a front touches neighbouring rows --, and
  // two scalar instructions per skipped row were a seventh of the kernel's scalar work)
  for (;;) {
    if (nd != 0u) {
      if (nd & 0x000fu) {
        if (nd & 1u) { nd &= ~1u; TSA_ROW(0, 0, nu, 0) }
        TSA_DOWN(1) TSA_DOWN(2) TSA_DOWN(3)
      }
      if (nd & 0x00f0u) { TSA_DOWN(4) TSA_DOWN(5) TSA_DOWN(6) TSA_DOWN(7) }
      if (nd & 0x0f00u) { TSA_DOWN(8) TSA_DOWN(9) TSA_DOWN(10) TSA_DOWN(11) }
      if (nd & 0xf000u) { TSA_DOWN(12) TSA_DOWN(13) TSA_DOWN(14) TSA_DOWN(15) }
    }
    if (!(nd | nu)) break;
    if (nu != 0u) {
      if (nu & 0xf000u) {
        if ((nu >> 15) & 1u) { nu &= ~(1u << 15); TSA_ROW(15, 0, nd, 1) }
        TSA_UP(14) TSA_UP(13) TSA_UP(12)
      }
      if (nu & 0x0f00u) { TSA_UP(11) TSA_UP(10) TSA_UP(9) TSA_UP(8) }
      if (nu & 0x00f0u) { TSA_UP(7) TSA_UP(6) TSA_UP(5) TSA_UP(4) }
      if (nu & 0x000fu) { TSA_UP(3) TSA_UP(2) TSA_UP(1) TSA_UP(0) }
    }
    if (!(nd | nu)) break;
  }
#undef TSA_DOWN
#undef TSA_UP
#undef TSA_ROW
#undef TSA_VERT
#undef TSA_STAT_INC
#undef TSA_STAT_HP
  TSA_T(t_c);
  TSA_ACC(1, t_b, t_c);
  TSA_CNT(8, evals);
  TSA_CNT(11, hpass);
  if (first) { TSA_CNT(18, 1); TSA_CNT(19, evals); TSA_CNT(20, __builtin_popcount(rowchg)); }   // first jobs: how many, their evaluations, the rows they changed
  TSA_CNT(21, __builtin_popcount(rowchg));
  asm volatile("; TSA_MARK results_begin");
  // ---- 6. results: rows that changed, the edge-column copies, the goal ----
  asm volatile("" : "+v"(lane));
  // FRESH WAKE TESTS (round 5).  A changed edge cell wakes the tile beyond it only if it beats what that tile holds -- and
  // what it holds is read again HERE, at the end of the job, not taken from the halo the job loaded when it started: the
  // sweeps in between are most of the job's time, the neighbour (where the front came from, as a rule) has often caught up
  // meanwhile, and the wake-up would cost a whole job that finds nothing (scripts/sim_async.c, SIM_FRESH: 17 % fewer
  // jobs, no-op share 37 -> 23 %).  Values only get better, so a newer snapshot can only suppress wake-ups that would have
  // found nothing; one that is read while the neighbour is still storing errs towards waking, as before.  The three loads
  // are issued before this job's stores and used after them; pages come from the look-up at the job's start (a neighbour
  // that had none then reads page 0, "unreached", and is woken as before).
#if RNA_TSA_FRESH
  int topf, botf, Xf = 0;
  {
    const unsigned pgN = scr[1], pgS = scr[6];
    const int xl = lane & 31;
    const bool xr = lane >= 32;
    const int xz = (xl > 0 ? 1 : 0) + (xl > TJ ? 1 : 0);
    const int xdir = 3 * xz - (xz >> 1) + (xr ? 2 - (xz & 1) : 0);
    const unsigned pgX = scr[xdir];
    const int xrow = xl == 0 ? TJ - 1 : (xl <= TJ ? xl - 1 : 0);
    topf = (int)ld_l2(&C.pages[(pgN << 10) + (TJ - 1) * TI + lane]);
    botf = (int)ld_l2(&C.pages[(pgS << 10) + lane]);
    if (xl <= TJ + 1) Xf = (int)ld_l2(&C.paux[pgX * AUX_WORDS + (xr ? 0 : 16) + xrow]);
  }
#endif
  if (rowchg) {
    if (TSA_UNLIKELY(pg == 0u)) {   // first change of this tile: it gets a page (this job is the tile's only writer)
      int p = 0;
      if (lane == 0) {
        p = atomicAdd(C.nalloc, 1) + 1;
        if (p > C.cap) { p = 0; sch.pool_exhausted(); }
        else { C.owner[p] = (unsigned)t; __hip_atomic_store(&C.tmap[t], (unsigned)p, __ATOMIC_RELAXED, RNA_TSA_SCOPE); }
      }
      pg = (unsigned)__builtin_amdgcn_readfirstlane(p);
      if (pg == 0u) return;   // (the search is being abandoned: status 5)
    }
    unsigned* own = C.pages + (pg << 10);
    unsigned* ax = C.paux + pg * AUX_WORDS + (lane ? 16 : 0);
    const bool edge_lane = __builtin_amdgcn_inverse_ballot_w64(0x8000000000000001ull);   // lanes 0 and 63 (as `lane == 0 || lane == 63` the compiler nests two exec-masked blocks)
    // A reached cell whose g is about to leave the 30-bit range of the field word: a value written by this job is at
    // most lim + 1414 (its source passed on, i.e. g + h < lim), so the rows are only looked at when the bound is that far out.
    const bool ovf_possible = lim_u > (unsigned)(KU - 5 * COST_D);
    unsigned long long ovfm = 0ull;
    // (the lane offset as an unsigned value of known range: the stores then take the page pointer as scalar base and
    // need no 64-bit vector address each)
    const unsigned ulane = (unsigned)lane & 63u;
#define TSA_STORE(b) if ((rowchg >> (b)) & 1u) own[(b) * TI + ulane] = (unsigned)TSA_G(b);
    TSA_R16(TSA_STORE)
#undef TSA_STORE
    if (TSA_UNLIKELY(ovf_possible)) {
      unsigned rowchg_o = rowchg;
      asm volatile("" : "+s"(rowchg_o));
#define TSA_OVF(b) if ((rowchg_o >> (b)) & 1u) ovfm |= __builtin_amdgcn_ballot_w64((unsigned)(TSA_G(b) - 1) < (unsigned)(4 * COST_D - 1));
      TSA_R16(TSA_OVF)
#undef TSA_OVF
    }
    // the copies of columns 0 and 63 for the neighbours: lanes 0 and 63 store their sixteen cells as four 16-byte words,
    // changed or not (an unchanged cell is rewritten with the value it has: this job is the only writer) -- one store
    // per changed row under sixteen scalar tests cost 32 scalar instructions and twice the store instructions
    if (edge_lane) {
      uint4* a4 = reinterpret_cast<uint4*>(ax);
      a4[0] = make_uint4((unsigned)g0, (unsigned)g1, (unsigned)g2, (unsigned)g3);
      a4[1] = make_uint4((unsigned)g4, (unsigned)g5, (unsigned)g6, (unsigned)g7);
      a4[2] = make_uint4((unsigned)g8, (unsigned)g9, (unsigned)g10, (unsigned)g11);
      a4[3] = make_uint4((unsigned)g12, (unsigned)g13, (unsigned)g14, (unsigned)g15);
    }
    if (ovfm && lane == 0) sch.overflow();   // path costs beyond 2^30 - 5656: the search is abandoned (status 4)
    if (TSA_UNLIKELY(t == C.tg)) {
      unsigned rowchg_g = rowchg & (1u << C.gb);   // (the goal's row, if it changed: see srow above)
      asm volatile("" : "+s"(rowchg_g));
#define TSA_GOAL(b)                                                                               \
  if ((rowchg_g >> (b)) & 1u) {                                                    \
    const int u = __builtin_amdgcn_readlane(TSA_G(b), C.ga); /* C.ga is wave-uniform */           \
    if (u != 0 && lane == 0) sch.improve_best(KU - u);                                            \
  }
      TSA_R16(TSA_GOAL)
#undef TSA_GOAL
    }
  }
  TSA_T(t_w0);
  asm volatile("; TSA_MARK wake_begin");
  // ---- 7. who has to run: neighbours whose halo got better (or may pass on now), this tile again in a later bucket ----
  {
    // (a) this tile again when the next bucket opens: it holds reached cells beyond this bucket's bound that may still
    //     matter (f <= best).  Looked for in the rows that changed (all rows in a first job), and not at all once the
    //     tile is flagged -- it runs several times per bucket.
    //     When the bound of this job is best + 1 (the goal has been reached and lies inside the bucket) every cell
    //     with f <= best passes on: nothing is held back that matters.
    unsigned look = first ? 0xffffu : rowchg;
    asm volatile("" : "+s"(look));
    if (look && best1 > bucket_end && !sch.is_far(t)) {
      unsigned long long farm = 0ull;
      if (!TSA_UNLIKELY(best_in != INF)) {
        // no path known yet: every reached cell matters, so "held back" is all there is to test -- and what a cell may
        // pass on is its value or 0, i.e. a reached cell is held back iff g ^ pp != 0: two vector instructions per row,
        // no heuristic (this is the case of almost every job: the goal is reached in a search's last bucket)
        int acc = 0;
#define TSA_END(b) if ((look >> (b)) & 1u) acc |= TSA_G(b) ^ TSA_PP(b);
        TSA_R16(TSA_END)
#undef TSA_END
        farm = __builtin_amdgcn_ballot_w64(acc != 0);
      } else {
        const int thr_best = KU - best_in;   // f <= best  <=>  u - h >= thr_best
#define TSA_END(b)                                                                                               \
  if ((look >> (b)) & 1u) farm |= __builtin_amdgcn_ballot_w64(TSA_G(b) != 0 && TSA_PP(b) == 0 && TSA_G(b) >= TSA_HC(b, thr_best));
        TSA_R16(TSA_END)
#undef TSA_END
      }
      if (farm && lane == 0) sch.act_far(t);
    }
    // (b) a first job: cells the previous bucket's bound held back may pass their values on now although they did not
    //     change -- every edge cell that may pass on takes part in the tests below
    if (TSA_UNLIKELY(first)) {
      q0 |= __builtin_amdgcn_ballot_w64(pp0 != 0);
      q15 |= __builtin_amdgcn_ballot_w64(pp15 != 0);
      qany |= (1ull << 63) | 1ull;
    }
    // A changed edge cell wakes the tile beyond it only if it beats what that tile held when this job loaded its halo
    // (its values only get better, so the test can only err towards waking): most wake-ups used to be echoes -- the
    // front enters this tile FROM the neighbour, the cells along that edge improve, and the neighbour would be woken
    // to find nothing new (37 % of all jobs).
    // Each step is tested with this cell's own mask bit for it (the move, its target and -- for a diagonal -- both
    // corner cells are free): with 30 % of the map blocked, "beats an unreached neighbour" is mostly a blocked neighbour.
    // A wake-up carries a KEY: the lowest f = g + h among the neighbour's cells this tile improves (kept as the highest
    // u - h; one wave reduction per side).  Free wavefronts take the queued tile with the lowest key, so tiles are
    // relaxed roughly in the order A* would settle their cells (scripts/sim_async.c: 7 % fewer jobs than red-black
    // rounds and no wavefront waiting at a round barrier).
    bool wakeN, wakeS;
    int kfN = 0, kfS = 0, kfW = 0, kfE = 0;
    unsigned colw, am, wm;
#if RNA_TSA_CONFIRM
    int confirm_w = 0;          // (a 32-bit scalar, like the job loop's flags)
    bool col_any = (qany & 1ull) || (qany >> 63) || (q0 & 1ull) || (q0 >> 63) || (q15 & 1ull) || (q15 >> 63);
    for (;;) {
#else
    const bool col_any = (qany & 1ull) || (qany >> 63) || (q0 & 1ull) || (q0 >> 63) || (q15 & 1ull) || (q15 >> 63);
    {
#endif
    wakeN = q0 != 0ull; wakeS = q15 != 0ull;
    if (wakeN) {
#if !RNA_TSA_FRESH
      const int topv = (int)scr[84 + lane];
#else
      const int topv = topf;
#endif
      const int c_ = __builtin_amdgcn_inverse_ballot_w64(q0) ? g0 : 0;
      const int cw_ = c_ & __builtin_amdgcn_sbfe((int)mk0, 0, 1), cn_ = c_ & __builtin_amdgcn_sbfe((int)mk0, 1, 1), ce_ = c_ & __builtin_amdgcn_sbfe((int)mk0, 2, 1);
      // (the shifted values are formed for ALL lanes first: inside a short-circuit `||` the wave shift would run with the
      // lanes whose first test succeeded switched off, and their neighbours would read nothing from them)
      const int from_w_ = lane_m1(ce_) + nD, from_e_ = lane_p1(cw_) + nD;
      const int un_ = max3i(cn_ + nS, from_w_, from_e_);
      const bool imp_ = un_ > topv;
      wakeN = __builtin_amdgcn_ballot_w64(imp_) != 0ull;
      if (wakeN) kfN = wave_max_i32(imp_ ? un_ - TSA_H(-1) : (int)0x80000000);
    }
    if (wakeS) {
#if !RNA_TSA_FRESH
      const int botv = (int)scr[84 + 64 + lane];
#else
      const int botv = botf;
#endif
      const int c_ = __builtin_amdgcn_inverse_ballot_w64(q15) ? g15 : 0;
      const int cw_ = c_ & __builtin_amdgcn_sbfe((int)mk3, 24 + 5, 1), cs_ = c_ & __builtin_amdgcn_sbfe((int)mk3, 24 + 6, 1), ce_ = c_ & __builtin_amdgcn_sbfe((int)mk3, 24 + 7, 1);
      const int from_w_ = lane_m1(ce_) + nD, from_e_ = lane_p1(cw_) + nD;
      const int un_ = max3i(cs_ + nS, from_w_, from_e_);
      const bool imp_ = un_ > botv;
      wakeS = __builtin_amdgcn_ballot_w64(imp_) != 0ull;
      if (wakeS) kfS = wave_max_i32(imp_ ? un_ - TSA_H(TJ) : (int)0x80000000);
    }
    // the same for the two edge columns and the four corners: lane 0 / 63 lay their 16 cells (what they may pass on)
    // out in LDS, the lanes that hold the halo column as loaded (lane = row + 1) compare.  Unchanged cells take part
    // too: they cannot beat a neighbour that has already seen them.
    colw = 0u;   // the six column / corner directions that wake, at their bits of `am`
    if (col_any) {
      if (__builtin_amdgcn_inverse_ballot_w64(0x8000000000000001ull)) {   // lanes 0 and 63
        uint4* cp = reinterpret_cast<uint4*>(&scr[276 + (lane ? 32 : 0)]);
        cp[0] = make_uint4((unsigned)pp0, (unsigned)pp1, (unsigned)pp2, (unsigned)pp3);
        cp[1] = make_uint4((unsigned)pp4, (unsigned)pp5, (unsigned)pp6, (unsigned)pp7);
        cp[2] = make_uint4((unsigned)pp8, (unsigned)pp9, (unsigned)pp10, (unsigned)pp11);
        cp[3] = make_uint4((unsigned)pp12, (unsigned)pp13, (unsigned)pp14, (unsigned)pp15);
      }
      __builtin_amdgcn_wave_barrier();
      const int xl_ = lane & 31;
      const bool xr_ = lane >= 32;
      const int v_ = (xl_ >= 1 && xl_ <= TJ) ? (int)scr[276 + (lane & 32) + xl_ - 1] : 0;
#if !RNA_TSA_FRESH
      const int xv = (int)scr[84 + 128 + lane];
#else
      const int xv = Xf;
#endif
      const unsigned eb_ = scr[84 + 192 + 48 + lane];   // the edge cell's mask, as loaded (lanes 1..16 / 33..48)
      // straight: k3 / k4; towards the row above: k0 / k2; towards the row below: k5 / k7
      const int vs_ = v_ & -(int)((eb_ >> (xr_ ? 4 : 3)) & 1u), vu_ = v_ & -(int)((eb_ >> (xr_ ? 2 : 0)) & 1u), vd_ = v_ & -(int)((eb_ >> (xr_ ? 7 : 5)) & 1u);
      // the halo cell in lane l is row l - 1 of the neighbour: it is reached straight from this lane's cell, from the
      // cell one lane up (row l) by its "row above" step and from the one lane down (row l - 2) by its "row below" step
      const int from_below_ = lane_p1(vu_) + nD, from_above_ = lane_m1(vd_) + nD;   // (formed for all lanes before any test, see above)
      const int ux_ = max3i(vs_ + nS, from_below_, from_above_);
      const bool imp = (xl_ <= TJ + 1) & (ux_ > xv);
      const unsigned long long im = __builtin_amdgcn_ballot_w64(imp);
      // straight into the direction bits of `am` below (0 NW, 3 W, 5 SW from the low half; 2 NE, 4 E, 7 SE from the high half):
      // through an intermediate bit set and sixteen selects this was 35 scalar instructions
      const unsigned lo = (unsigned)im, hi = (unsigned)(im >> 32);
      colw = (lo & 1u) | ((lo >> 12) & 32u) | ((hi & 1u) << 2) | ((hi >> 10) & 128u) | ((lo & 0x1fffeu) ? 8u : 0u) | ((hi & 0x1fffeu) ? 16u : 0u);
      if (im) {   // one key per side: the corner tiles of a side share it
        const int hX_ = tsa_octile24(xr_ ? i0 + TI : i0 - 1, j0 + xl_ - 1, gi, gj);
        wave_halves_max_i32(imp ? ux_ - hX_ : (int)0x80000000, kfW, kfE);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // directions: 0 NW, 1 N, 2 NE, 3 W, 4 E, 5 SW, 6 S, 7 SE
    am = colw | (wakeN ? 2u : 0u) | (wakeS ? 64u : 0u);
    wm = am & (unsigned)__builtin_amdgcn_ballot_w64(nb_t >= 0);   // (nb_t is -1 in every lane from 8 on)
#if RNA_TSA_CONFIRM
    // CONFIRM.  A wake-up costs the neighbour a whole job, and a third of all jobs find nothing: the neighbour has caught up
    // by itself since this job loaded its halo (both were reached by the same front).  So the sides that are about to wake are
    // looked at once more: the neighbour's edge as it is in memory NOW replaces the copy in the wavefront's scratch, and the
    // tests above run again for those sides only.  Values only get better, a row that is being stored while it is read shows a
    // mix of old and new words, and either way the test can only err towards waking.
    confirm_w = __builtin_amdgcn_readfirstlane(confirm_w);
    if (!wm || confirm_w) break;
    confirm_w = 1;
    // (scalar page bases and 32-bit lane offsets of known range: no 64-bit vector addresses -- the first version of this
    // block spilled the zero the compiler keeps for extending indices, at the top of every job)
    const unsigned ul_c = (unsigned)lane & 63u;
    if (!(wm & 2u)) q0 = 0ull;
    else {
      const unsigned* pN = C.pages + ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)scr[1]) << 10);
      scr[84 + lane] = ld_l2(&pN[(TJ - 1) * TI + ul_c]);
    }
    if (!(wm & 64u)) q15 = 0ull;
    else {
      const unsigned* pS = C.pages + ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)scr[6]) << 10);
      scr[84 + 64 + lane] = ld_l2(&pS[ul_c]);
    }
    col_any = (wm & 0xbdu) != 0u;
    if (col_any) {
      const unsigned xl = ul_c & 31u;
      const unsigned xr = ul_c >> 5;
      const unsigned xz = (xl > 0u ? 1u : 0u) + (xl > (unsigned)TJ ? 1u : 0u);
      const unsigned xdir = 3u * xz - (xz >> 1) + (xr ? 2u - (xz & 1u) : 0u);
      const unsigned xrow = xl == 0u ? (unsigned)(TJ - 1) : (xl <= (unsigned)TJ ? xl - 1u : 0u);
      if (xl <= (unsigned)(TJ + 1)) {
        const unsigned pgX = scr[xdir] & 0xfffffu;   // (a page number: at most 65 536 per query; the mask tells the compiler the offset fits 32 bits)
        scr[84 + 128 + lane] = ld_l2(&C.paux[pgX * AUX_WORDS + (xr ? 0u : 16u) + xrow]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    }   // once more, for the sides that claimed a wake-up
#else
    }
#endif
    if (wm) {
      // this job's stores are in L2 before anybody is told to look at them (a woken tile's job loads with sc1 from L2)
#ifndef RNA_TSA_UNSAFE_NOWAIT   /* (developer build that is NOT exact: what hiding the stores' round trip could be worth at most) */
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      // lane k < 8 speaks for direction k: its tile, its key ((f - f at the bucket's start), quantised)
      // (four selects under constant lane masks: the nested comparisons became four levels of exec-masked blocks)
      int kf = __builtin_amdgcn_inverse_ballot_w64(0x29ull) ? kfW : kfE;   // lanes 0, 3, 5: the western side
      kf = __builtin_amdgcn_inverse_ballot_w64(0x02ull) ? kfN : kf;
      kf = __builtin_amdgcn_inverse_ballot_w64(0x40ull) ? kfS : kf;
      int key = (key_base - kf) >> key_shift;
      key = key < 0 ? 0 : (key > TSA_NCLS - 1 ? TSA_NCLS - 1 : key);
      TSA_T(t_w1);
      sch.wake8(__builtin_amdgcn_inverse_ballot_w64((unsigned long long)wm), nb_t, (unsigned)key, lane, spare);
      TSA_T(t_w2);
      TSA_ACC(3, t_w1, t_w2);   // queueing the wake-ups
      TSA_CNT(13, 1);
    }
  }
  asm volatile("; TSA_MARK job_end");
  TSA_T(t_d);
  TSA_ACC(2, t_c, t_d);
  TSA_ACC(6, t_w0, t_d);   // wake tests + queueing
  if (lane_in == 0) { atomicAdd(&scr[SCR_CNT], (unsigned)__builtin_popcount(rowchg)); atomicAdd(&scr[SCR_CNT + 1], 1u); }   // rows written, jobs that got past the halo step (this wavefront's own words; adds that return nothing)
  }
tsa_job_done:
  // the job's stores are performed before the tile can be taken again (or is pulled again by this wavefront)
#ifndef RNA_TSA_UNSAFE_NOWAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  {
    int t_f = t_in;   // (opaque once more: the masks finish() forms from the tile number are not to be hoisted to the top of the job)
    asm volatile("" : "+s"(t_f));
    const int again = sch.finish(t_f, lane_in, spare, scr + SCR_CNT);   // 0: the tile is released, 1: a wake-up came in while it ran, 2: ... and it is due as a first job
    if (!again) break;
    first_w = again >> 1;
    sticky_w = 1;
    TSA_CNT(7, 1);    // (a sticky turn counts as a job of its own in the developer build's figures)
    TSA_CNT(22, 1);
  }
  }   // next turn on the same tile
#undef TSA_ROW_CHANGED
#undef TSA_H
}

// ---- scheduler of the one-workgroup-per-query kernel: an open list of tiles in LDS ----
// Per query (= workgroup), in LDS:
//   st2    two bits per tile: D "a wake-up is pending" (bit 0), R "a wavefront is running the tile" (bit 1)
//   head   TSA_NCLS stacks of queue nodes, one per key class (key = the lowest f the waker offers the tile, counted
//          from the bucket's start and quantised): tag << 16 | index of the top node (0xffff: empty)
//   node   TSA_NP queue nodes: next << 16 | tile
//   bm     one bit per node: free
//   qc     tile -> key class of the newest entry pushed for it (direct-mapped, TSA_QC lines; a hint)
//   open   tiles that have to run as "first" in this f-bucket (they hold cells the last bucket's bound held back);
//          free wavefronts drain this set before they take entries
//   far    the same for the next bucket
// A wake-up of a tile that is not running pushes a node on the stack of its key class -- also when the tile is pending
// already (its older entry goes stale: an entry whose tile has nothing pending when it is taken is dropped); a wake-up
// of a running tile only sets D -- the wavefront that runs it queues it again (class 0) when it ends.  A free wavefront
// takes the top node of the lowest class that has one, claims its tile (R, then D) and runs the job.  Every queue
// operation is a handful of LDS round trips whatever the queue holds (round 2's array of (key, tile) entries was scanned
// for its minimum by every pop and for the tile's entry by every wake-up of a pending tile: 2.8 us of wavefront time per
// job, a sixth of the kernel; scripts/sim_async.c: last-in-first-out inside a class costs no more jobs than the exact
// minimum did).  No round barriers: a bucket is at its fixed point when every wavefront of the workgroup is idle (an
// idle counter; a wavefront only leaves the idle state when the entry counter says there is something to take, and
// only an idle-counted wavefront looks at the counter, so "all idle" is stable and every wavefront sees it).
// Exactness does not depend on the order: every improvement of a tile's edge that can matter to a neighbour sets the
// neighbour's D bit after the improved words are in L2, and a bucket only ends when nothing is pending.
#ifndef RNA_TSA_NODES
#define RNA_TSA_NODES 3072
#endif
constexpr int TSA_NP = RNA_TSA_NODES;   // (scripts/sim_async.c: ~3 400 entries at once in the worst of the bench's queries, stale ones included; more: see `spill`)
constexpr int TSA_BMW = TSA_NP / 32;
#ifndef RNA_TSA_QC
#define RNA_TSA_QC 1024
#endif
constexpr int TSA_QC = RNA_TSA_QC;      // lines of the tile -> queued class table (a power of two)
constexpr unsigned POP_EMPTY = 0xffffffffu;
static_assert(TSA_NP % 32 == 0 && TSA_NP < 0xffff, "node indices are 16 bits, 0xffff = none");
static_assert(TSA_NP * 4 >= (TI + 2) * (TJ + 2) * 4 + TILE_WORDS, "the backtrace's LDS image lives in the queue memory");

__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct TsaLocalSched {
  int* best_;
  int* state_;
  unsigned* st2_;
  unsigned* far_;
  unsigned* node_;
  unsigned* head_;
  unsigned* bm_;
  unsigned* qc_;
  int* count_;   // entries in the queue + tiles of the open set nobody has taken yet (a hint for idle wavefronts)
  int* spill_;   // no free node: wake-ups were parked in `far` and this bucket runs again
  unsigned* open_;   // tiles due as first jobs of this bucket
  int* open_left_;   // ... how many of them nobody has taken yet
  __device__ __forceinline__ int best() const { return __hip_atomic_load(best_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  __device__ __forceinline__ void improve_best(int g) { atomicMin(best_, g); }
  __device__ __forceinline__ void overflow() { *state_ = 4; }
  __device__ __forceinline__ void pool_exhausted() { *state_ = 5; }
  __device__ __forceinline__ void act_far(int t) { atomicOr(&far_[t >> 5], 1u << (t & 31)); }
  __device__ __forceinline__ bool is_far(int t) const { return (lds_ld(&far_[t >> 5]) >> (t & 31)) & 1u; }
  // a free node (-1: none found).  Every lane that needs one probes on its own, from a start of its own.
  __device__ __forceinline__ int node_alloc(unsigned seed) {
    unsigned w = ((seed & 0xffffu) * (unsigned)TSA_BMW) >> 16;
    for (int tries = 0; tries < 32; ++tries) {
      const unsigned v = lds_ld(&bm_[w]);
      if (v) {
        const unsigned b = v & (0u - v);
        if (atomicAnd(&bm_[w], ~b) & b) return (int)(w * 32u) + __builtin_ctz(b);
      } else {
        w = w + 1u == (unsigned)TSA_BMW ? 0u : w + 1u;
      }
    }
    return -1;
  }
  __device__ __forceinline__ void node_free(int idx) { atomicOr(&bm_[idx >> 5], 1u << (idx & 31)); }
  // node idx (owned by the caller) on top of class cls
  __device__ __forceinline__ void link(int idx, unsigned t, unsigned cls) {
    for (;;) {
      const unsigned hv = lds_ld(&head_[cls]);
      __hip_atomic_store(&node_[idx], (hv << 16) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");   // (a wavefront's LDS operations are performed in the order they were issued)
      if (atomicCAS(&head_[cls], hv, ((hv + 0x10000u) & 0xffff0000u) | (unsigned)idx) == hv) break;
    }
  }
  // tile t gets an entry of class cls (called by every lane of the wavefront with uniform arguments; lane 0 acts)
  __device__ __forceinline__ void push(unsigned t, unsigned cls, int lane, int* spare) {
    if (lane == 0) {
      const int idx = *spare >= 0 ? *spare : node_alloc(t * 29u);
      if (idx >= 0) {
        link(idx, t, cls);
        atomicAdd(count_, 1);
        __hip_atomic_store(&qc_[t & (unsigned)(TSA_QC - 1)], (t << 8) | cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {   // no node: the tile (its D bit stays set) runs as a first job when this bucket is opened again
        atomicOr(&far_[t >> 5], 1u << (t & 31));
        *spill_ = 1;
      }
    }
    *spare = -1;
  }
  // the wake-ups of one job together: lane k < 8 with `w` set wakes tile t with key class `cls` (all lanes call)
  __device__ __forceinline__ void wake8(bool w, int t, unsigned cls, int lane, int* spare) {
    const unsigned sh = 2u * ((unsigned)t & 15u);
    unsigned old = 2u;
    if (w) old = atomicOr(&st2_[t >> 4], 1u << sh) >> sh;
    bool q = w && !(old & 2u);   // not running: an entry
    // ... unless the tile is pending with an entry of this class or a lower one already (`qc`, a direct-mapped table
    // tile -> class of its newest entry; a hint: another tile's line or a stale one only costs a duplicate or order)
    const unsigned qi = (unsigned)t & (unsigned)(TSA_QC - 1);
    if (q && (old & 1u)) {
      const unsigned c = lds_ld(&qc_[qi]);
      if ((c >> 8) == (unsigned)t && (c & 0xffu) <= cls) q = false;
    }
    if (q) __hip_atomic_store(&qc_[qi], ((unsigned)t << 8) | cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long qm = __builtin_amdgcn_ballot_w64(q);
    if (!qm) return;
    // the node this wavefront took off the queue with its job goes to the first entry
    const int reuse = *spare;
    *spare = -1;
    int idx = -1;
    if (q) {
      idx = (reuse >= 0 && lane == __builtin_ctzll(qm)) ? reuse : node_alloc(tsa_seed24((unsigned)t, (unsigned)lane));
      if (idx >= 0) {
        link(idx, (unsigned)t, cls);
      } else {
        atomicOr(&far_[t >> 5], 1u << (t & 31));
        *spill_ = 1;
      }
    }
    const unsigned long long okm = __builtin_amdgcn_ballot_w64(idx >= 0);
    if (okm && lane == __builtin_ctzll(okm)) atomicAdd(count_, __builtin_popcountll(okm));
  }
  // End of a job of tile t (all lanes call; the job's stores have been performed).  A wake-up that came in while the tile
  // ran leaves the tile with this wavefront: D is cleared, R stays, and the caller goes round its job loop once more
  // (returns 1; 2 when the tile is also due as a FIRST job of this bucket -- whoever wanted to take it from the open set
  // found it running, put the bit back and set D).  Otherwise the tile is released; a wake-up that slipped in between
  // the look and the release is queued as before (class 0).
  // D of a running tile is only ever SET by others (an entry is only claimed while R is clear), so the look cannot miss.
  // `cnt`: the wavefront's own counters in LDS -- [2] every turn that ends here (jobs of both kinds and sticky turns), [3] the
  // turns after which the tile stays (inside the lane-0 block that is there anyway: two LDS adds that return nothing).
  __device__ __forceinline__ int finish(int t, int lane, int* spare, unsigned* cnt) {
    const unsigned sh = 2u * ((unsigned)t & 15u);
    int r = 0;
    if (lane == 0) {
      atomicAdd(&cnt[2], 1u);
#ifdef RNA_TSA_NO_STICKY   /* developer build: the round-4 behaviour (release, queue again) in this round's code */
      if (false) {
#else
      if ((lds_ld(&st2_[t >> 4]) >> sh) & 1u) {
#endif
        atomicAnd(&st2_[t >> 4], ~(1u << sh));
        atomicAdd(&cnt[3], 1u);
        r = 1;
        if (lds_ldi(open_left_) > 0 && ((atomicAnd(&open_[t >> 5], ~(1u << (t & 31))) >> (t & 31)) & 1u)) {
          atomicSub(open_left_, 1);
          atomicSub(count_, 1);
          r = 2;
        }
      } else if ((atomicAnd(&st2_[t >> 4], ~(2u << sh)) >> sh) & 1u) {
        r = 3;
      }
    }
    r = __builtin_amdgcn_readfirstlane(r);
    if (r == 3) { push((unsigned)t, 0u, lane, spare); r = 0; }
    return r;
  }
  // take the top entry of the lowest class that has one: its tile, POP_EMPTY if the queue is empty.  The node becomes
  // the wavefront's spare one (*spare; the one it held goes back to the pool).
  __device__ __forceinline__ unsigned pop(int lane, int* spare) {
    for (;;) {
      asm volatile("" ::: "memory");
      uint4 h;   // classes 4 lane .. 4 lane + 3
      {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u hv4 = *reinterpret_cast<const volatile v4u*>(&head_[4 * lane]);
        h = make_uint4(hv4.x, hv4.y, hv4.z, hv4.w);
      }
      const bool e0 = (h.x & 0xffffu) != 0xffffu, e1 = (h.y & 0xffffu) != 0xffffu, e2 = (h.z & 0xffffu) != 0xffffu, e3 = (h.w & 0xffffu) != 0xffffu;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(e0 | e1 | e2 | e3);
      if (!m) return POP_EMPTY;
      const int f = __builtin_ctzll(m);
      unsigned got = POP_EMPTY;
      if (lane == f) {
        const int c = e0 ? 0 : (e1 ? 1 : (e2 ? 2 : 3));
        const unsigned hv = e0 ? h.x : (e1 ? h.y : (e2 ? h.z : h.w));
        const unsigned idx = hv & 0xffffu;
        const unsigned nd = lds_ld(&node_[idx]);
        if (atomicCAS(&head_[4 * lane + c], hv, ((hv + 0x10000u) & 0xffff0000u) | (nd >> 16)) == hv) {
          got = (idx << 16) | (nd & 0xffffu);
          atomicSub(count_, 1);
        }
      }
      got = (unsigned)__builtin_amdgcn_readlane((int)got, f);
      if (got != POP_EMPTY) {
        if (*spare >= 0 && lane == 0) node_free(*spare);
        *spare = (int)(got >> 16);
        return got & 0xffffu;
      }
      // another wavefront changed that stack in between: look again
    }
  }
};

// Kernel arguments that are the same for every query of a launch.
struct TsaLaunch {
  int rows, cols, tiles_i, tiles_j, s0, s1;
  const rna_astar_query* queries;
  TsaStage S;
  int bucket_width;
  int prio_first;   // the first prio_first workgroups to start (the longest expected searches) run at raised wave priority
  // Second pass over a batch (the RETRY instantiation of the kernel, a launch of its own; `retry` is informational): workgroup r serves the r-th query that ran out of pages (status 5) in the
  // first pass, in slot r of the stage's RETRY view S2 -- a few slots with one page per tile of the map, which a
  // search can never outgrow (a goal that cannot be reached floods its whole component; the reference answers "no
  // path", it does not fail).
  int retry, n;
  int* retry_count;   // (device memory, may be null) searches of this launch that ended with status 5 ...
  int* retry_list;    // ... and which, in the order they ended (device memory, [max_queries])
  int retry_base;     // second pass: workgroup r serves retry_list[retry_base + r]
  TsaStage S2;
  int32_t* paths;
  int max_path_len;
  int32_t* rev_all;
  int rev_cap;
  rna_astar_result* results;
};
constexpr int TSA_FOUND = -1000;        // provisional status inside the search kernel: found, path not traced yet

// Canonical backtrace, one wavefront per query: walk from the goal to the neighbour n with g[n] + w(n, c) == g[c],
// lowest linear index first (lane k probes neighbour k).  The walk runs in LDS: the 64 x 16 tile of the current
// cell plus its halo ring and the tile's neighbour masks are loaded once, then every step is one LDS round trip
// until the path leaves the tile -- a path of 2 000 cells is ~100 tile loads instead of 2 000 dependent
// HBM round trips.
constexpr int BW = TI + 2;   // LDS row pitch of the backtrace image (halo included)
// (run by the first wavefront of the query's search workgroup once the search has ended -- the other wavefronts have
// left, the workgroup's queue memory holds the LDS image; r = the provisional result)
__device__ __forceinline__ void tsa_backtrace_wave(const TsaLaunch& A, const TsaStage& S, const int sl, const int q, const rna_astar_result r,
                                                   const int lane, unsigned* s_tile, unsigned char* s_mask) {
  const int rows = A.rows, cols = A.cols, tiles_i = A.tiles_i, tiles_j = A.tiles_j;
  const int ncell = rows * cols, ntile = tiles_i * tiles_j;
  const unsigned* tmap = S.tmap + (size_t)sl * ntile;
  const size_t page_base = (size_t)sl * ((size_t)S.cap + 1);
  const int start = tsa_unwrap_lin(A.queries[q].start, rows, cols, A.s0, A.s1);
  const int goal = tsa_unwrap_lin(A.queries[q].goal, rows, cols, A.s0, A.s1);
  const int si = start % rows, sj = start / rows;
  int ci = goal % rows, cj = goal / rows;
  int* rev = A.rev_all + (size_t)q * A.rev_cap;
  const int k = lane & 7;
  const int wk = (k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D;
  const int di = kdi_of(k), dj = kdj_of(k);
  const int off = di + dj * BW;
  int len = 0;
  bool ok = true, done = false;
#ifdef RNA_TSA_STATS
  unsigned long long bt_loads = 0, bt_load_ticks = 0;
#endif
  while (ok && !done) {
#ifdef RNA_TSA_STATS
    const unsigned long long t_l0 = wall_clock64();
#endif
    // ---- tile of the current cell + halo ring + masks -> LDS ----
    const int ti = ci >> 6, tj = cj >> 4;
    const int t = tj * tiles_i + ti;
    unsigned pgl = 0u;   // lane k < 8: page of the neighbouring tile in direction k; lane 8: page of this tile
    {
      const int nti = lane < 8 ? ti + di : ti, ntj = lane < 8 ? tj + dj : tj;
      if (lane < 9 && nti >= 0 && ntj >= 0 && nti < tiles_i && ntj < tiles_j) pgl = ld_l2(&tmap[ntj * tiles_i + nti]);
    }
    auto gpage = [&](int src) -> size_t {
      const unsigned lp = (unsigned)__shfl((int)pgl, src);
      return page_base + lp;
    };
    {
      const int xl = lane & 31;
      const bool xr = lane >= 32;
      const unsigned* pown = S.pages + (gpage(8) << 10);
      const unsigned top = ld_l2(&S.pages[(gpage(1) << 10) + (TJ - 1) * TI + lane]);
      const unsigned bot = ld_l2(&S.pages[(gpage(6) << 10) + lane]);
      const int xdir = xl == 0 ? (xr ? 2 : 0) : (xl <= TJ ? (xr ? 4 : 3) : (xr ? 7 : 5));
      const int xrow = xl == 0 ? TJ - 1 : (xl <= TJ ? xl - 1 : 0);
      const size_t gpx = gpage(xdir);
      unsigned X = 0u;
      if (xl <= TJ + 1) X = ld_l2(&S.paux[gpx * AUX_WORDS + (xr ? 0 : 16) + xrow]);
      unsigned tv[TJ];
#pragma unroll
      for (int b = 0; b < TJ; ++b) tv[b] = ld_l2(&pown[b * TI + lane]);
      const uint4 mv = *reinterpret_cast<const uint4*>(S.nbr_tm + (size_t)t * MASK_STRIDE + lane * 16);
      __builtin_amdgcn_wave_barrier();   // the previous tile's walk has finished reading the LDS image
      s_tile[lane + 1] = top;
      s_tile[(TJ + 1) * BW + lane + 1] = bot;
      if (xl <= TJ + 1) s_tile[xl * BW + (xr ? TI + 1 : 0)] = X;
#pragma unroll
      for (int b = 0; b < TJ; ++b) s_tile[(b + 1) * BW + lane + 1] = tv[b];
      *reinterpret_cast<uint4*>(&s_mask[lane * 16]) = mv;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#ifdef RNA_TSA_STATS
    bt_loads += 1; bt_load_ticks += wall_clock64() - t_l0;
#endif
    // ---- walk inside the tile ----
    int il = ci & (TI - 1), jl = cj & (TJ - 1);
    unsigned uc = s_tile[(jl + 1) * BW + il + 1];
    for (;;) {
      if (lane == 0 && len < A.rev_cap) rev[len] = cj * rows + ci;
      ++len;
      if (ci == si && cj == sj) { done = true; break; }
      if (len > ncell || uc == 0u) { ok = false; break; }
      const unsigned mc = s_mask[il * 16 + jl];
      const unsigned un = s_tile[(jl + 1) * BW + il + 1 + off];
      const bool hit = (lane < 8) & (((mc >> k) & 1u) != 0u) & (un != 0u) & (un == uc + (unsigned)wk);   // g(n) + w == g(c)  (no short circuits: one chain of compares, no exec-masked blocks in the walk)
      const unsigned long long m = __ballot(hit);
      if (!m) { ok = false; break; }
      // (only lanes 0..7 can hit: a 32-bit bit search.  As __ffsll the index was carried as a 64-bit scalar pair whose upper half the
      // compiler took from a register it reused inside the loop -- the -DRNA_TSA_STATS build walked in circles until `len > ncell`)
      const int src = __builtin_ctz((unsigned)m);
      uc = (unsigned)__builtin_amdgcn_readlane((int)un, src);   // (src is wave-uniform: a v_readlane, not a trip through the LDS crossbar)
      const int sdi = kdi_of(src), sdj = kdj_of(src);
      ci += sdi; cj += sdj; il += sdi; jl += sdj;
      if (il < 0 || jl < 0 || il >= TI || jl >= TJ) break;   // left the tile: load that one
    }
  }
#ifdef RNA_TSA_STATS
  if (lane == 0) { atomicAdd(&g_tsa_stat[27], bt_loads); atomicAdd(&g_tsa_stat[28], bt_load_ticks); atomicAdd(&g_tsa_stat[29], (unsigned long long)len); }
#endif
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

// One workgroup of 8 wavefronts per query, four of them per CU.  The workgroup first resets the pages the slot's previous
// search used, then searches; its first wavefront traces the path at the end (tsa_backtrace_wave).
// WAVES = wavefronts per workgroup (= per query): 8 when batches are pipelined (four queries share a CU, the throughput
// configuration), 16 for a single batch on the engine's own stream and for batches of <= 32 queries (latency is what
// counts there).
template <int WAVES, bool RETRY>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(RNA_TSA_WAVES_PER_EU, RNA_TSA_WAVES_PER_EU))) tsa_search_kernel(const TsaLaunch A) {
  constexpr int TSA_THREADS = WAVES * 64;
  __shared__ __attribute__((aligned(16))) unsigned s_scr[WAVES][SCR_WORDS];
  extern __shared__ unsigned s_dyn[];   // sized by the launch: st2 (2 x nt_words) | open (nt_words) | far (nt_words)
  __shared__ __attribute__((aligned(16))) unsigned s_node[TSA_NP];
  __shared__ __attribute__((aligned(16))) unsigned s_head[TSA_NCLS];
  __shared__ unsigned s_bm[TSA_BMW];
  __shared__ unsigned s_qc[TSA_QC];
  __shared__ int s_best, s_state, s_bucket, s_bucket0, s_jobs_done, s_jobs_noop, s_jobs_sticky, s_expanded, s_nalloc, s_any;
  __shared__ int s_count, s_spill, s_idle, s_open_left, s_open_pos, s_reruns;

  // Workgroups take their query when they START (a ticket), not by blockIdx: the hardware deals workgroup
  // indices round-robin to the XCDs, so a fixed mapping lets one XCD with several long searches hold back
  // its share of the batch while the other XCDs idle.  With tickets a free CU anywhere takes the next
  // (longest remaining) query.
  __shared__ int s_q, s_rank;
  if (!RETRY) {
    if (threadIdx.x == 0) { const int k = atomicAdd(A.S.ticket, 1); s_rank = k; s_q = A.S.perm[k]; }
  } else {
    // (the list was written by the first pass: a scan of the results for status 5 here would race with the workgroups
    // of this pass that have already answered theirs)
    if (threadIdx.x == 0) { s_q = A.retry_list[A.retry_base + (int)blockIdx.x]; s_rank = 1 << 30; }
  }
  __syncthreads();
  const int q = __builtin_amdgcn_readfirstlane(s_q);   // wave-uniform values belong in SGPRs: the tile jobs need every VGPR
  if (q < 0) return;
  if (RETRY && threadIdx.x == 0) A.S2.ticket[8 + blockIdx.x] = q;   // (tsa_retry_served: the settled-cell count of q reads this slot)
  const TsaStage& S = RETRY ? A.S2 : A.S;
  const int sl = RETRY ? (int)blockIdx.x : q;   // the slot whose pages / tables this search uses
  // A batch lasts as long as its longest search, and a stage cannot take its next batch before: the searches expected
  // to be the longest (the first tickets) get the issue slots of their SIMDs first, the short ones fill in around them.
#ifndef RNA_TSA_PRIO_FIRST
#define RNA_TSA_PRIO_FIRST 0
#endif
  if (__builtin_amdgcn_readfirstlane(s_rank) < A.prio_first) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef RNA_TSA_STATS
  const unsigned long long t_kernel0 = wall_clock64();
#endif
  const int rows = A.rows, cols = A.cols, tiles_i = A.tiles_i, tiles_j = A.tiles_j;
  rna_astar_query qu = A.queries[q];   // buffer linear indices; the search itself runs in map space
  const int ncell = rows * cols;
  const int ntile = tiles_i * tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  unsigned* const s_st2 = s_dyn;
  unsigned* const s_open = s_dyn + 2 * nt_words;
  unsigned* const s_far = s_dyn + 3 * nt_words;
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
  C.tiles_magic = tiles_i > 1 ? (unsigned)(0x100000000ull / (unsigned)tiles_i) + 1u : 0u;
  C.pages = S.pages + (((size_t)sl * ((size_t)S.cap + 1)) << 10);
  C.paux = S.paux + (size_t)sl * ((size_t)S.cap + 1) * AUX_WORDS;
  C.tmap = S.tmap + (size_t)sl * ntile;
  C.owner = S.owner + (size_t)sl * (S.cap + 1);
  C.cap = S.cap;
  C.nalloc = &s_nalloc;
  C.nbr_tm = S.nbr_tm;
  C.gi = gi; C.gj = gj;
  C.ts = tile_of(si, sj, tiles_i); C.sa = si & (TI - 1); C.sb = sj & (TJ - 1);
  C.tg = tile_of(gi, gj, tiles_i); C.ga = gi & (TI - 1); C.gb = gj & (TJ - 1);

  for (int w = tid; w < 4 * nt_words; w += TSA_THREADS) s_dyn[w] = 0u;
  for (int w = tid; w < TSA_NCLS; w += TSA_THREADS) s_head[w] = 0xffffu;
  for (int w = tid; w < TSA_BMW; w += TSA_THREADS) s_bm[w] = 0xffffffffu;
  for (int w = tid; w < TSA_QC; w += TSA_THREADS) s_qc[w] = 0xffffffffu;
  if (lane >= 48) s_scr[wv][68 + lane - 48] = 0u;   // the zero tail of the wave's scratch
  if (lane < 4) s_scr[wv][SCR_CNT + lane] = 0u;     // ... and its four counters
  // a start or a goal without a single traversable neighbour: blocked or walled in; nothing has been written yet, so
  // no page is in use
  {
    const unsigned ms = C.nbr_tm[(size_t)C.ts * MASK_STRIDE + C.sa * 16 + C.sb], mg = C.nbr_tm[(size_t)C.tg * MASK_STRIDE + C.ga * 16 + C.gb];
    if (qu.goal != qu.start && (mg == 0u || ms == 0u)) {
      if (tid == 0) results[q] = rna_astar_result{1, 0, INF, 0, 0, 0};
      return;
    }
  }
  // This query's pages as the last search in this slot left them: back to "unreached" (the slot is this workgroup's
  // alone; its local pages 1..used are one contiguous run of the stage's page array).  Doing it here instead of in a
  // kernel of its own takes ~1 ms off every turn of the stage, which can only take its next batch when this one's
  // longest search has ended.
  {
    const int used_raw = S.nalloc[sl];
    const int used = used_raw < C.cap ? used_raw : C.cap;
    if (used > 0) {
      uint4* pg4 = reinterpret_cast<uint4*>(C.pages + (1 << 10));
      for (size_t w = tid; w < (size_t)used * (TILE_WORDS / 4); w += TSA_THREADS) pg4[w] = make_uint4(0u, 0u, 0u, 0u);
      uint4* ax4 = reinterpret_cast<uint4*>(C.paux + AUX_WORDS);
      for (size_t w = tid; w < (size_t)used * (AUX_WORDS / 4); w += TSA_THREADS) ax4[w] = make_uint4(0u, 0u, 0u, 0u);
      for (int p = 1 + tid; p <= used; p += TSA_THREADS) C.tmap[C.owner[p]] = 0u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  // key classes: (f - f at the bucket's start) >> key_shift, the last class takes everything beyond
  int key_shift = 0;
  while (((long long)A.bucket_width >> key_shift) > TSA_NCLS - 1) ++key_shift;
  if (tid == 0) {
    s_best = INF; s_state = 0; s_jobs_done = 0; s_jobs_noop = 0; s_jobs_sticky = 0; s_expanded = 0; s_nalloc = 0;
    s_bucket = tsa_octile(si, sj, gi, gj) / A.bucket_width;
    s_bucket0 = s_bucket;
    s_spill = 0; s_idle = 0; s_open_pos = 0; s_reruns = 0;
    s_count = 1; s_open_left = 1;
  }
  __syncthreads();
  if (tid == 0) s_open[C.ts >> 5] = 1u << (C.ts & 31);   // the start tile's first job plants g(start) = 0
  __syncthreads();

#ifdef RNA_TSA_STATS
  unsigned long long tsa_acc[24] = {};
  const unsigned long long t_life0 = wall_clock64();
  const unsigned long long c_life0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef RNA_TSA_IDLE
  // developer build: what share of a search wavefront's life has nothing to take (shader-clock ticks; counters [6] / [7] of
  // rna_astar_job_counters then hold life and idle ticks summed over the wavefronts instead of buckets / 0)
  unsigned long long idle_ticks = 0ull;
  const unsigned long long t_wave0 = __builtin_amdgcn_s_memtime();
#endif
  TsaLocalSched sch{&s_best, &s_state, s_st2, s_far, s_node, s_head, s_bm, s_qc, &s_count, &s_spill, s_open, &s_open_left};
  int spare = -1;   // a queue node this wavefront owns (the one its last job's entry sat in): its next wake-up uses it

  for (;;) {   // one pass per f-bucket (or per re-run of a bucket whose queue overflowed)
    const int bucket = __builtin_amdgcn_readfirstlane(lds_ldi(&s_bucket));
    const long long bucket_end = ((long long)bucket + 1) * A.bucket_width;
    const unsigned bucket_end_u = bucket_end > 0x80000000LL ? 0x80000000u : (unsigned)bucket_end;   // (everything >= 2^31 is "beyond any f")
    const long long key_base_ll = (long long)KU - (long long)bucket * A.bucket_width;   // KU - (f at the bucket's start)
    const int key_base = key_base_ll < -(long long)INF ? -INF : (int)key_base_ll;
    bool idle = false;   // this wavefront is counted in s_idle
    bool open_blocked = false;
#ifdef RNA_TSA_IDLE
    unsigned long long t_idle0 = 0ull;
#endif
    for (;;) {
      TSA_T(t_p0);
      if (lds_ldi(&s_state) >= 4) break;
      if (idle) {
        if (lds_ldi(&s_idle) == WAVES) break;          // every wavefront idle: nothing queued, nothing running
        if (lds_ldi(&s_count) <= 0) { __builtin_amdgcn_s_sleep(RNA_TSA_IDLE_SLEEP); continue; }
        if (lane == 0) atomicSub(&s_idle, 1);
        idle = false;
#ifdef RNA_TSA_IDLE
        idle_ticks += __builtin_amdgcn_s_memtime() - t_idle0;
#endif
      }
      int t = -1;
      int first = 0;   // (a 32-bit scalar, not a lane-mask bool: see tsa_job)
      // ---- 1. a tile of the open set (runs as "first") ----
      const int open_left = lds_ldi(&s_open_left);   // (a hint: read once per turn)
      if (!open_blocked && open_left > 0) {
        const int pos = lds_ldi(&s_open_pos);
        const unsigned w = pos + lane < nt_words ? lds_ld(&s_open[pos + lane]) : 0u;
        const unsigned long long nz = __builtin_amdgcn_ballot_w64(w != 0u);
        if (!nz) {   // (wraps around: a tile that was running when it was taken is put back)
          if (lane == 0) atomicCAS(&s_open_pos, pos, pos + 64 >= nt_words ? 0 : pos + 64);
          continue;
        }
        const int l0 = __builtin_ctzll(nz);
        const unsigned w0 = (unsigned)__builtin_amdgcn_readlane((int)w, l0);
        const int b0 = __builtin_ctz(w0);
        const int tt = ((pos + l0) << 5) + b0;
        const unsigned sh = 2u * ((unsigned)tt & 15u);
        int r = 0;   // 0: somebody else took the bit, 1: ours, 2: the tile is running right now
        if (lane == 0) {
          if ((atomicAnd(&s_open[pos + l0], ~(1u << b0)) >> b0) & 1u) {
            const unsigned old = atomicOr(&s_st2[tt >> 4], 2u << sh) >> sh;
            if (old & 2u) {   // running (as a woken tile): it has to run again as first -- bit back, D set, its wavefront queues it
              atomicOr(&s_open[pos + l0], 1u << b0);
              atomicOr(&s_st2[tt >> 4], 1u << sh);
              r = 2;
            } else {
              atomicAnd(&s_st2[tt >> 4], ~(1u << sh));   // whatever was pending is served by this job (its entry goes stale)
              atomicSub(&s_open_left, 1);
              atomicSub(&s_count, 1);
              r = 1;
            }
          }
        }
        r = __builtin_amdgcn_readfirstlane(r);
        if (r == 0) continue;
        if (r == 2) { open_blocked = true; continue; }   // take an entry meanwhile
        t = tt;
        first = 1;
      } else {
        // ---- 2. the queued tile with the lowest key ----
        open_blocked = false;
        const unsigned e = sch.pop(lane, &spare);
        if (e == POP_EMPTY) {
          if (open_left > 0) { __builtin_amdgcn_s_sleep(2); continue; }   // (the open tile that was running)
          if (lane == 0) atomicAdd(&s_idle, 1);
          idle = true;
#ifdef RNA_TSA_IDLE
          t_idle0 = __builtin_amdgcn_s_memtime();
#endif
          continue;
        }
        const int tt = (int)(e & 0xffffu);
        const unsigned sh = 2u * ((unsigned)tt & 15u);
        int r = 0;   // 0: not ours (running or stale), 1: ours, 2: ours and first
        if (lane == 0) {
          // pending and not running -> running, nothing pending, in ONE compare-and-swap on the tile's two bits (a look
          // first: most entries that are not ours are stale or their tile is running; an entry whose tile has nothing
          // pending is dropped -- a wake-up that arrives later brings its own)
          unsigned w = lds_ld(&s_st2[tt >> 4]);
          while (((w >> sh) & 3u) == 1u) {
            const unsigned got = atomicCAS(&s_st2[tt >> 4], w, (w | (2u << sh)) & ~(1u << sh));
            if (got == w) { r = 1; break; }
            w = got;   // (another tile of the word changed)
          }
          if (r == 1 && open_left > 0 && ((atomicAnd(&s_open[tt >> 5], ~(1u << (tt & 31))) >> (tt & 31)) & 1u)) {
            atomicSub(&s_open_left, 1);
            atomicSub(&s_count, 1);
            r = 2;
          }
        }
        r = __builtin_amdgcn_readfirstlane(r);
        if (r <= 0) continue;
        t = tt;
        first = r >> 1;
      }
#ifdef RNA_TSA_TAKE_SLEEP   /* developer build: extra latency (in units of 64 clocks) between taking a job and running it */
      __builtin_amdgcn_s_sleep(RNA_TSA_TAKE_SLEEP);
#endif
      TSA_T(t_p1);
      TSA_ACC(4, t_p0, t_p1);   // taking a job
      TSA_CNT(7, 1);
#if defined(RNA_TSA_JOBPRIO)
      __builtin_amdgcn_s_setprio(RNA_TSA_JOBPRIO);        // developer build: issue priority of a wavefront inside a job ...
#endif
      tsa_job(sch, s_scr[wv], lane, t, C, first, bucket_end_u, key_base, key_shift, &spare TSA_ACC_ARG);
#if defined(RNA_TSA_JOBPRIO)
      __builtin_amdgcn_s_setprio(RNA_TSA_IDLEPRIO);       // ... and while it takes the next one / polls
#endif
      // (the tile has been released by the job itself -- or kept for further turns while wake-ups kept coming: Sched::finish)
    }
    // ---- the bucket is at its fixed point (or the search is being abandoned) ----
    __syncthreads();
#ifdef RNA_TSA_IDLE
    if (idle) idle_ticks += __builtin_amdgcn_s_memtime() - t_idle0;   // (the wait for the last wavefront of the bucket included)
#endif
    if (s_state >= 4) break;
    int tid_r = threadIdx.x;
    asm volatile("" : "+v"(tid_r));
    const bool spilled = s_spill != 0;
    __syncthreads();
    if (tid_r == 0) {
      // every cell with f < (bucket + 1) * B has its exact g (unless wake-ups were parked: then this bucket runs again)
      if (!spilled && s_best != INF && (long long)s_best < bucket_end) s_state = 1;
      if (spilled) s_reruns += 1;   // (the queue ran out of nodes: wake-ups were parked and the bucket runs again)
      s_any = 0; s_spill = 0; s_idle = 0; s_open_pos = 0;
    }
    __syncthreads();
    if (s_state == 1) break;
    // tiles that hold cells of the next bucket (or parked wake-ups) become the open set
    int cnt = 0;
    for (int w = tid_r; w < nt_words; w += TSA_THREADS) {
      const unsigned b = s_far[w];
      s_open[w] = b;
      s_far[w] = 0u;
      cnt += __popc(b);
    }
    if (cnt) atomicAdd(&s_any, cnt);
    __syncthreads();
    if (tid_r == 0) {
      if (!s_any) s_state = (s_best != INF) ? 1 : 2;   // nothing left anywhere
      else { s_count = s_any; s_open_left = s_any; if (!spilled) s_bucket += 1; }
    }
    __syncthreads();
    if (s_state != 0) break;
  }
#ifdef RNA_TSA_IDLE
  if (lane == 0) {
    atomicAdd(&tsa_counters_of(S)[8], __builtin_amdgcn_s_memtime() - t_wave0);
    atomicAdd(&tsa_counters_of(S)[9], idle_ticks);
  }
#endif
  if (lane == 0) {   // cells written; jobs (those that changed something, those that found nothing), sticky turns among them
    atomicAdd(&s_expanded, (int)s_scr[wv][SCR_CNT] * TI);
    atomicAdd(&s_jobs_done, (int)s_scr[wv][SCR_CNT + 2]);
    atomicAdd(&s_jobs_noop, (int)(s_scr[wv][SCR_CNT + 2] - s_scr[wv][SCR_CNT + 1]));
    atomicAdd(&s_jobs_sticky, (int)s_scr[wv][SCR_CNT + 3]);
  }
  __syncthreads();
  // the launch's job counters (rna_astar_job_counters: what the bench reports as jobs per touched tile / no-op share,
  // observed in the run itself): six adds per SEARCH
  // (they live in the second half of the 256-byte block in front of the stage's page counts -- reached through S.nalloc, which
  // the lines below need anyway: as a kernel argument of its own the pointer was two more SGPRs alive across every tile job,
  // and the spills they caused cost the bench 1 %)
  if (tid == 0) {
    unsigned long long* const cnt = tsa_counters_of(S);
    atomicAdd(&cnt[0], 1ull);
    atomicAdd(&cnt[1], (unsigned long long)s_nalloc);
    atomicAdd(&cnt[2], (unsigned long long)s_jobs_done);
    atomicAdd(&cnt[3], (unsigned long long)s_jobs_noop);
    atomicAdd(&cnt[4], (unsigned long long)s_jobs_sticky);
    atomicAdd(&cnt[5], (unsigned long long)(s_expanded / TI));
    atomicAdd(&cnt[6], (unsigned long long)(s_bucket - s_bucket0 + 1));
    if (s_reruns) atomicAdd(&cnt[7], (unsigned long long)s_reruns);
  }
#ifdef RNA_TSA_STATS
  tsa_acc[5] = wall_clock64() - t_life0;   // wave lifetime inside the search loop
  tsa_acc[12] = __builtin_amdgcn_s_memtime() - c_life0;   // the same in shader clock ticks
  if (lane == 0) for (int k = 0; k < 16; ++k) atomicAdd(&g_tsa_stat[k], tsa_acc[k]);
  if (lane == 0) for (int k = 18; k < 24; ++k) atomicAdd(&g_tsa_stat[k], tsa_acc[k]);   // (16 / 17: per search, below)
  if (tid == 0) { atomicAdd(&g_tsa_stat[16], (unsigned long long)s_nalloc); atomicAdd(&g_tsa_stat[17], 1ull); }   // tiles this search touched; searches
#endif
  // what the next search in this slot has to reset
  if (tid == 0) S.nalloc[sl] = s_nalloc < C.cap ? s_nalloc : C.cap;
  // (`rounds` reports tile jobs per wavefront -- every job, also the ones that found nothing in their halo: there are no rounds any more)
  const int state = s_state;
  const rna_astar_result r{state == 1 ? TSA_FOUND : (state >= 4 ? state : 1), 0, state == 1 ? s_best : INF, s_expanded,
                           (s_jobs_done + WAVES - 1) / WAVES, s_bucket - s_bucket0 + 1};
  if (state != 1) {
    if (tid == 0) {
      results[q] = r;
      if (!RETRY && state == 5 && A.retry_count) A.retry_list[atomicAdd(A.retry_count, 1)] = q;
    }
    return;
  }
  // found: the first wavefront traces the path over the exact field (the queue memory holds the LDS image of the walk)
  __syncthreads();
  if (wv != 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // The walk is ONE chain of dependent instructions (a step: two LDS reads, a ballot, a lane read, a handful of scalar
  // instructions), run by one wavefront while its workgroup's LDS and query slot stay taken: at the priority of the seven
  // search wavefronts it shares its SIMD with, a step took 0.88 us and a path of 2 000 cells 2 ms -- 18 % of the workgroup's
  // residence (profiles/r05_search_job_stats.txt).  At the highest issue priority it gets its instructions in when it asks.
#ifndef RNA_TSA_BT_PRIO
#define RNA_TSA_BT_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(RNA_TSA_BT_PRIO);
#ifdef RNA_TSA_STATS
  const unsigned long long t_bt0 = wall_clock64();
#endif
  tsa_backtrace_wave(A, S, sl, q, r, lane, s_node, reinterpret_cast<unsigned char*>(s_node + BW * (TJ + 2)));
#ifdef RNA_TSA_STATS
  if (lane == 0) { atomicAdd(&g_tsa_stat[24], wall_clock64() - t_bt0); atomicAdd(&g_tsa_stat[25], 1ull); atomicAdd(&g_tsa_stat[26], wall_clock64() - t_kernel0); }
#endif
}

// |{n : g(n) + h(n) <= f*}| per query from the pages still resident in HBM (measurement utility)
__global__ void tsa_settled_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries,
                                   const rna_astar_result* __restrict__ results, TsaStage S_main, TsaStage S_retry, const int* __restrict__ served,
                                   const int* __restrict__ retry_list, int n_retried, int32_t* __restrict__ counts, int s0, int s1) {
  __shared__ int s_cnt;
  const int q = blockIdx.x;
  // a query that was searched again lives in the retry slot that served it
  int sl = q;
  bool retried = false;
  if (served)
    for (int r = 0; r < TSA_RETRY; ++r)
      if (served[r] == q) { sl = r; retried = true; }
  if (!retried)   // searched again in an earlier second pass whose slot has been reused since: the field is gone
    for (int r = 0; r < n_retried; ++r)
      if (retry_list[r] == q) { if (threadIdx.x == 0) counts[q] = -1; return; }
  const TsaStage& S = retried ? S_retry : S_main;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const rna_astar_result r = results[q];
  int cnt = 0;
  if (r.status == 0 || r.status == 3) {
    const int goal = tsa_unwrap_lin(queries[q].goal, rows, cols, s0, s1);
    const int gi = goal % rows, gj = goal / rows;
    const int used = S.nalloc[sl];
    for (size_t w = threadIdx.x; w < ((size_t)used << 10); w += blockDim.x) {
      const int p = 1 + (int)(w >> 10), l = (int)(w & 1023);
      const int t = (int)S.owner[(size_t)sl * (S.cap + 1) + p];
      const int i = (t % tiles_i) * TI + (l & (TI - 1)), j = (t / tiles_i) * TJ + (l >> 6);
      const unsigned u = S.pages[(((size_t)sl * ((size_t)S.cap + 1) + p) << 10) + l];
      if (u != 0u && i < rows && j < cols && (KU - (int)u) + tsa_octile(i, j, gi, gj) <= r.cost) ++cnt;
    }
  }
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[q] = s_cnt;
}

// ---- host entry points used by astar.hip ----
static inline int tsa_ntile(const rna_engine* e) {
  return ((e->geom.size[0] + TI - 1) / TI) * ((e->geom.size[1] + TJ - 1) / TJ);
}
bool tsa_supported(const rna_engine* e) { return (size_t)tsa_ntile(e) <= (size_t)TSA_MAX_TILE_WORDS * 32; }
int tsa_tiles(const rna_engine* e) { return tsa_ntile(e); }
// HBM of one pipeline stage: pages and their edge-column copies (cap per query + the shared page 0) ...
size_t tsa_pool_bytes(int max_queries, int cap) { return (size_t)max_queries * ((size_t)cap + 1) * (TILE_WORDS + AUX_WORDS) * sizeof(unsigned); }
// ... and one allocation that must start zeroed: ticket | nalloc | perm | mask snapshot | tmap | owner
size_t tsa_aux_bytes(const rna_engine* e, int max_queries, int cap) {
  const size_t ntile = (size_t)tsa_ntile(e);
  return 256 + 2 * tsa_align256((size_t)max_queries * sizeof(int)) + tsa_align256(ntile * MASK_STRIDE) +
         tsa_align256((size_t)max_queries * ntile * sizeof(unsigned)) + tsa_align256((size_t)max_queries * ((size_t)cap + 1) * sizeof(unsigned));
}
static TsaStage tsa_stage_view(const rna_engine* e, int slot) {
  const AstarDevice& a = e->astar;
  const size_t ntile = (size_t)tsa_ntile(e);
  char* base = static_cast<char*>(a.tsa_aux[slot]);
  TsaStage S;
  S.cap = a.page_cap;
  S.pages = reinterpret_cast<unsigned*>(a.g[slot]);
  S.paux = S.pages + (((size_t)a.max_queries * ((size_t)S.cap + 1)) << 10);
  S.ticket = reinterpret_cast<int*>(base);
  base += 256;
  S.nalloc = reinterpret_cast<int*>(base);
  base += tsa_align256((size_t)a.max_queries * sizeof(int));
  S.perm = reinterpret_cast<int*>(base);
  base += tsa_align256((size_t)a.max_queries * sizeof(int));
  S.nbr_tm = reinterpret_cast<uint8_t*>(base);
  base += tsa_align256(ntile * MASK_STRIDE);
  S.tmap = reinterpret_cast<unsigned*>(base);
  base += tsa_align256((size_t)a.max_queries * ntile * sizeof(unsigned));
  S.owner = reinterpret_cast<unsigned*>(base);
  return S;
}
// The retry view of a stage (see TsaLaunch::retry): TSA_RETRY slots with one page per tile; its own pages and
// ticket | nalloc | tmap | owner block, the mask snapshot of the stage itself.
size_t tsa_retry_pool_bytes(const rna_engine* e) { return tsa_pool_bytes(TSA_RETRY, tsa_ntile(e)); }
size_t tsa_retry_aux_bytes(const rna_engine* e) {
  const size_t ntile = (size_t)tsa_ntile(e);
  return 256 + tsa_align256(TSA_RETRY * sizeof(int)) + tsa_align256((size_t)TSA_RETRY * ntile * sizeof(unsigned)) +
         tsa_align256((size_t)TSA_RETRY * (ntile + 1) * sizeof(unsigned)) + tsa_align256((size_t)e->astar.max_queries * sizeof(int));
}
// the stage's list of searches to repeat (the last block of the retry aux)
static int* tsa_retry_list(const rna_engine* e, int slot) {
  const size_t ntile = (size_t)tsa_ntile(e);
  return reinterpret_cast<int*>(static_cast<char*>(e->astar.tsa_aux_retry[slot]) + 256 + tsa_align256(TSA_RETRY * sizeof(int)) +
                                tsa_align256((size_t)TSA_RETRY * ntile * sizeof(unsigned)) + tsa_align256((size_t)TSA_RETRY * (ntile + 1) * sizeof(unsigned)));
}
// [TSA_RETRY] ints in the retry view's ticket block: the query each retry slot served in the stage's current batch (-1: none)
static int* tsa_retry_served(const rna_engine* e, int slot) { return reinterpret_cast<int*>(e->astar.tsa_aux_retry[slot]) + 8; }
// the int behind them: searches of the stage's current batch that ended with status 5 (copied to pinned host memory
// behind the search, so that the host sees it when the stage's stream is idle -- no PCIe atomics needed)
static int* tsa_retry_count(const rna_engine* e, int slot) { return tsa_retry_served(e, slot) + TSA_RETRY; }
static TsaStage tsa_retry_view(const rna_engine* e, int slot, const TsaStage& main) {
  const AstarDevice& a = e->astar;
  const size_t ntile = (size_t)tsa_ntile(e);
  char* base = static_cast<char*>(a.tsa_aux_retry[slot]);
  TsaStage S = main;
  S.cap = (int)ntile;
  S.pages = reinterpret_cast<unsigned*>(a.g_retry[slot]);
  S.paux = S.pages + (((size_t)TSA_RETRY * ((size_t)S.cap + 1)) << 10);
  S.ticket = reinterpret_cast<int*>(base);
  base += 256;
  S.nalloc = reinterpret_cast<int*>(base);
  base += tsa_align256(TSA_RETRY * sizeof(int));
  S.tmap = reinterpret_cast<unsigned*>(base);
  base += tsa_align256((size_t)TSA_RETRY * ntile * sizeof(unsigned));
  S.owner = reinterpret_cast<unsigned*>(base);
  return S;
}
int tsa_retry_prepare(rna_engine* e, int slot) {   // fresh retry pages: "unreached"
  hipLaunchKernelGGL(tsa_fill_pages_kernel, dim3(2048), dim3(256), 0, e->stream, reinterpret_cast<uint4*>(e->astar.g_retry[slot]),
                     tsa_retry_pool_bytes(e) / sizeof(uint4));
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}
// fresh stage: pages "unreached" (everything else was zeroed by the caller's hipMemsetAsync)
int tsa_stage_prepare(rna_engine* e, int slot) {
  const TsaStage S = tsa_stage_view(e, slot);
  const size_t pages = (size_t)e->astar.max_queries * ((size_t)S.cap + 1);
  hipLaunchKernelGGL(tsa_fill_pages_kernel, dim3(8192), dim3(256), 0, e->stream, reinterpret_cast<uint4*>(S.pages),
                     pages * (TILE_WORDS + AUX_WORDS) / 4);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

// one entry of the launch ring (AstarDevice::ring_mem): ticket | launch order | mask snapshot
size_t tsa_ring_bytes(const rna_engine* e, int max_queries) {
  return 256 + tsa_align256((size_t)max_queries * sizeof(int)) + tsa_align256((size_t)tsa_ntile(e) * MASK_STRIDE);
}
static_assert(sizeof(TsaLaunch) <= sizeof(AstarDevice::last_launch[0]), "AstarDevice::last_launch holds a TsaLaunch");

// the second pass over the stage's last batch (the searches that outgrew their share of pages, on the stage's full-size
// retry slots; workgroups without such a query end at once)
int tsa_retry_launch(rna_engine* e, int slot, hipStream_t search_stream, int count) {
  AstarDevice& a = e->astar;
  TsaLaunch A;
  memcpy(&A, a.last_launch[slot], sizeof(A));
  A.retry = 1;
  A.retry_count = nullptr;
  A.S2 = tsa_retry_view(e, slot, A.S);
  // TSA_RETRY searches at a time share the stage's retry slots: one launch after the other on the stage's stream
  for (int base = 0; base < count; base += TSA_RETRY) {
    A.retry_base = base;
    const int wgs = std::min(TSA_RETRY, count - base);
    if (a.depth > 1) hipLaunchKernelGGL((tsa_search_kernel<TSA_WAVES, true>), dim3(wgs), dim3(TSA_WAVES * 64), a.last_lds[slot], search_stream, A);
    else hipLaunchKernelGGL((tsa_search_kernel<16, true>), dim3(wgs), dim3(16 * 64), a.last_lds[slot], search_stream, A);
    RNA_HIP(e, hipGetLastError());
  }
  return RNA_OK;
}

int tsa_launch(rna_engine* e, int slot, hipStream_t init_stream, hipStream_t search_stream, hipEvent_t ev_init,
               const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len, rna_astar_result* res_dev) {
  AstarDevice& a = e->astar;
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TI - 1) / TI, tj = (cols + TJ - 1) / TJ;
  TsaStage S = tsa_stage_view(e, slot);
  // Snapshot of the neighbour masks as they are NOW (a later map update must not disturb a search in flight) and the
  // launch order of the batch.  Pipelined: into an entry of the launch ring, on the side stream behind an event of the
  // engine stream (whatever changes the masks next waits for the snapshot, side_join) -- on the stage's own stream
  // these two short kernels waited up to a millisecond for CU slots among the searches' workgroups, and the engine
  // stream's next map update with them.  Single stream: in place.
  hipStream_t prep_stream = search_stream;
  int ring = -1;
  if (ev_init) {
    ring = (int)(a.launches % (unsigned long long)a.ring_n);
    char* base = a.ring_mem + (size_t)ring * a.ring_stride;
    S.ticket = reinterpret_cast<int*>(base);
    S.perm = reinterpret_cast<int*>(base + 256);
    S.nbr_tm = reinterpret_cast<uint8_t*>(base + 256 + tsa_align256((size_t)a.max_queries * sizeof(int)));
    { const int rc = side_stream(e, &prep_stream); if (rc != RNA_OK) return rc; }
    RNA_HIP(e, hipEventRecord(ev_init, init_stream));
    RNA_HIP(e, hipStreamWaitEvent(prep_stream, ev_init, 0));
    if (a.ring_used[ring]) RNA_HIP(e, hipStreamWaitEvent(prep_stream, a.ring_free[ring], 0));   // (thirteen launches ago at least: long over)
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_INIT, prep_stream);
    hipLaunchKernelGGL(tsa_snapshot_kernel, dim3(std::min(ti * tj, 4096)), dim3(256), 0, prep_stream, e->nbr, rows, cols, ti, tj, S.nbr_tm,
                       e->geom.start[0], e->geom.start[1]);
    RNA_HIP(e, hipGetLastError());
  }
  if (ev_init) {
    RNA_HIP(e, hipEventRecord(a.snap_done[slot], prep_stream));
    a.snap_pending[slot] = true;
  }
  {
    // ticket and launch order (the pages of a slot are reset by the workgroup that takes the slot)
    KernelTimer kt(e, RNA_K_ASTAR_RESET, prep_stream);
    const int ranked = n <= 2048 ? 1 : 0;
    hipLaunchKernelGGL(tsa_prepare_kernel, dim3(1), dim3(256), ranked ? (size_t)n * sizeof(int) : 0, prep_stream, S, q_dev, n, rows, cols, ranked,
                       a.g_retry[slot] ? tsa_retry_served(e, slot) : nullptr);
    RNA_HIP(e, hipGetLastError());
  }
  if (ev_init) {
    RNA_HIP(e, hipEventRecord(a.ev_prep, prep_stream));
    RNA_HIP(e, hipStreamWaitEvent(search_stream, a.ev_prep, 0));
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_SEARCH, search_stream);
    const size_t nt_bytes = (size_t)((ti * tj + 31) / 32) * sizeof(unsigned);
    TsaLaunch A;
    memset(&A, 0, sizeof(A));
    A.rows = rows; A.cols = cols; A.tiles_i = ti; A.tiles_j = tj; A.s0 = e->geom.start[0]; A.s1 = e->geom.start[1];
    A.queries = q_dev; A.S = S; A.bucket_width = a.bucket_width; A.paths = paths_dev; A.max_path_len = max_len;
    A.rev_all = a.rev[slot]; A.rev_cap = a.rev_cap; A.results = res_dev;
    A.retry = 0; A.n = n; A.S2 = S;
    A.prio_first = a.depth > 1 ? RNA_TSA_PRIO_FIRST : 0;
    if (const char* pf = getenv("RNA_TSA_PRIO_FIRST")) A.prio_first = atoi(pf);   // developer knob
    size_t lds_dyn = 4 * nt_bytes;
    if (const char* pad = getenv("RNA_TSA_LDS_PAD")) lds_dyn += (size_t)atoi(pad);   // developer knob: fewer search workgroups per CU
    // a search that outgrows its share of pages is searched again when the host sees the count (astar_settle)
    const bool can_retry = a.g_retry[slot] != nullptr && a.retry_flag != nullptr;
    A.retry_count = can_retry ? tsa_retry_count(e, slot) : nullptr;
    A.retry_list = can_retry ? tsa_retry_list(e, slot) : nullptr;
    A.retry_base = 0;
    if (can_retry) a.retry_flag[slot] = 0;
    a.last_retried[slot] = 0;
    // 16 wavefronts per query where latency counts (one stream, or a batch too small to fill the chip: a lone search
    // takes 9.1 instead of 12.9 ms), 8 where throughput does (16 in the pipeline: 91.6 k instead of 122.8 k cycles/s)
    static const bool wide_env = getenv("RNA_TSA_WIDE") != nullptr;   // developer knob: 16 wavefronts per query always
    const bool wide = wide_env || n <= 32;
    if (a.depth > 1 && !wide) hipLaunchKernelGGL((tsa_search_kernel<TSA_WAVES, false>), dim3(n), dim3(TSA_WAVES * 64), lds_dyn, search_stream, A);
    else hipLaunchKernelGGL((tsa_search_kernel<16, false>), dim3(n), dim3(16 * 64), lds_dyn, search_stream, A);
    RNA_HIP(e, hipGetLastError());
    memcpy(a.last_launch[slot], &A, sizeof(A));
    a.last_lds[slot] = lds_dyn;
    a.retry_armed[slot] = A.retry_count != nullptr;
    // the count travels to pinned host memory behind the search (before the stage's `done` event)
    if (A.retry_count) RNA_HIP(e, hipMemcpyAsync(a.retry_flag + slot, A.retry_count, sizeof(int), hipMemcpyDeviceToHost, search_stream));
  }
  if (ring >= 0) {
    RNA_HIP(e, hipEventRecord(a.ring_free[ring], search_stream));
    a.ring_used[ring] = true;
  }
  return RNA_OK;
}

// the counters of every stage view that exists, summed into out[TSA_COUNTER_WORDS] (the caller has waited for the searches)
int tsa_counters_read(rna_engine* e, unsigned long long* out, bool reset) {
  AstarDevice& a = e->astar;
  for (int k = 0; k < TSA_COUNTER_WORDS; ++k) out[k] = 0ull;
  for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d) {
    for (int view = 0; view < 2; ++view) {
      char* base = static_cast<char*>(view ? a.tsa_aux_retry[d] : a.tsa_aux[d]);
      if (!base) continue;
      unsigned long long w[TSA_COUNTER_WORDS];
      char* cnt = base + 256 - sizeof(w);   // (nalloc starts at base + 256 in both views: tsa_stage_view / tsa_retry_view)
      RNA_HIP(e, hipMemcpyAsync(w, cnt, sizeof(w), hipMemcpyDeviceToHost, e->stream));
      if (reset) RNA_HIP(e, hipMemsetAsync(cnt, 0, sizeof(w), e->stream));
      RNA_HIP(e, hipStreamSynchronize(e->stream));
      for (int k = 0; k < TSA_COUNTER_WORDS; ++k) out[k] += w[k];
    }
  }
  return RNA_OK;
}

#ifdef RNA_TSA_STATS
void tsa_stats_dump() {
  unsigned long long st[32];
  if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tsa_stat), sizeof(st)) != hipSuccess) return;
  const double jobs = (double)st[7];
  if (jobs <= 0) return;
  const double busy = (double)(st[0] + st[1] + st[2]);
  fprintf(stderr, "[tsa stats] per job us: loads issued + arrived %.2f (of load+halo), wake tests + queueing %.2f of which queueing %.2f (of results); jobs that queue wake-ups %.1f%%; a job that finds nothing %.2f us\n",
          st[9] * 0.01 / jobs, st[6] * 0.01 / jobs, st[3] * 0.01 / jobs, 100.0 * (double)st[13] / jobs, st[15] * 0.01 / (double)std::max<unsigned long long>(1, st[10]));
  fprintf(stderr, "[tsa stats] searches %.0f, tiles touched %.0f (%.1f per search), jobs per touched tile %.2f, jobs that find nothing %.3f of all\n", (double)st[17],
          (double)st[16], (double)st[16] / (double)std::max<unsigned long long>(1, st[17]), jobs / (double)std::max<unsigned long long>(1, st[16]), (double)st[10] / jobs);
  fprintf(stderr, "[tsa stats, tile kernel, all launches] jobs %.0f (%.1f%% no-op) | per job us: load+halo %.2f sweeps %.2f results %.2f | wave lifetime %.1f wave-ms, in jobs %.1f wave-ms (%.1f%%), taking jobs %.1f wave-ms (%.1f%%) | row evaluations per job %.1f, extra horizontal passes %.1f | shader clock while searching %.0f MHz\n",
          jobs, 100.0 * (double)st[10] / jobs, st[0] * 0.01 / jobs, st[1] * 0.01 / jobs, st[2] * 0.01 / jobs, st[5] * 1e-5, busy * 1e-5,
          100.0 * busy / (double)st[5], st[4] * 1e-5, 100.0 * (double)st[4] / (double)st[5], st[8] / jobs, st[11] / jobs,
          100.0 * (double)st[12] / (double)std::max<unsigned long long>(1, st[5]));
  if (st[25]) fprintf(stderr, "[tsa stats] backtrace (one wavefront, the workgroup's LDS held meanwhile): %.1f us per found path, %.1f %% of the workgroup's residence (%.2f ms from its first instruction to the end of the backtrace, found paths only)\n",
                      st[24] * 0.01 / (double)st[25], 100.0 * (double)st[24] / (double)std::max<unsigned long long>(1, st[26]), st[26] * 1e-5 / (double)st[25]);
  if (st[25]) fprintf(stderr, "[tsa stats] backtrace: %.0f cells and %.1f tile loads per path (%.1f cells per load), a tile load %.2f us (%.0f %% of the backtrace), a step %.3f us\n",
                      (double)st[29] / (double)st[25], (double)st[27] / (double)st[25], (double)st[29] / (double)std::max<unsigned long long>(1, st[27]),
                      st[28] * 0.01 / (double)std::max<unsigned long long>(1, st[27]), 100.0 * (double)st[28] / (double)std::max<unsigned long long>(1, st[24]),
                      (double)(st[24] - st[28]) * 0.01 / (double)std::max<unsigned long long>(1, st[29]));
  fprintf(stderr, "[tsa stats] sticky turns (the wavefront kept a tile that was woken while it ran) %.3f of all jobs, of which %.3f find nothing\n",
          (double)st[22] / jobs, (double)st[23] / (double)std::max<unsigned long long>(1, st[22]));
  fprintf(stderr, "[tsa stats] first jobs (a tile's first in a bucket: the rows a new bound releases, and what they flag) %.3f of all, %.1f row evaluations and %.1f changed rows each; rows changed per job %.1f; rows sent to the scan per job %.2f\n",
          (double)st[18] / jobs, (double)st[19] / (double)std::max<unsigned long long>(1, st[18]), (double)st[20] / (double)std::max<unsigned long long>(1, st[18]), (double)st[21] / jobs, (double)st[14] / jobs);
}
#endif

int tsa_settled(rna_engine* e, int slot, const rna_astar_query* q, const rna_astar_result* r, int n, int32_t* d_counts) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TI - 1) / TI, tj = (cols + TJ - 1) / TJ;
  const TsaStage S = tsa_stage_view(e, slot);
  const bool has_retry = e->astar.g_retry[slot] != nullptr;
  hipLaunchKernelGGL(tsa_settled_kernel, dim3(n), dim3(1024), 0, e->stream, rows, cols, ti, tj, q, r, S,
                     has_retry ? tsa_retry_view(e, slot, S) : S, has_retry ? tsa_retry_served(e, slot) : nullptr,
                     has_retry ? tsa_retry_list(e, slot) : nullptr, has_retry ? e->astar.last_retried[slot] : 0,
                     d_counts, e->geom.start[0], e->geom.start[1]);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

}  // namespace rna
