// astar_tile.hip -- tile-synchronous grid A* for gfx950 ("TSA"): the same contract and the same
// label-correcting argument as astar.hip, but the relaxation runs inside LDS.
//
// Why: the frontier kernel of astar.hip pays two HBM/L2 round trips plus two workgroup barriers per
// search GENERATION (~5 us), and a long query has >10^4 generations.  Here the search field is
// stored TILE-MAJOR (32 x 32 cells = 4 KiB per tile) and a wavefront owns one tile at a time:
//   1. grab-and-clear the tile's pending bits (cells improved since its last visit),
//   2. load the tile + a one-cell halo into LDS with coalesced loads,
//   3. relax to the tile-local fixed point of the current f-bucket entirely in LDS (ds_min_rtn,
//      wave-synchronous queue, ~0.3 us per generation instead of ~5 us),
//   4. write the tile back (inner cells: coalesced stores; edge ring and improved halo cells:
//      atomicMin, because neighbouring tiles may be in flight on other wavefronts), hand improved
//      halo cells to their tiles as pending bits and activate those tiles.
// A workgroup (8 wavefronts) serves one query; rounds of tile jobs are separated by one workgroup
// barrier.  Two pending bitmaps per query (current bucket / next bucket) replace the frontier
// queues, so nothing can overflow.  Exactness: every update is a min over lengths of real paths and
// the schedule runs every bucket to its fixed point, so at termination g is exact for f <= f*, which
// is all the canonical backtrace reads (DESIGN.md "Grid A* contract").
#include "engine.hpp"

using namespace rna;

namespace rna {

constexpr int TS = 32;                 // tile edge (cells)
constexpr int TW = TS + 2;             // LDS row pitch incl. halo
constexpr int TILE_WORDS = TS * TS;    // 1024
#ifndef RNA_TSA_WAVES
#define RNA_TSA_WAVES 16
#endif
#ifndef RNA_TSA_UNR
#define RNA_TSA_UNR 1
#endif
constexpr int TSA_WAVES = RNA_TSA_WAVES;
constexpr int TSA_THREADS = TSA_WAVES * 64;
constexpr int TSA_MAX_TILE_WORDS = 2048;   // active-tile bitset words -> up to 65536 tiles
constexpr int TSA_JOBS = 4096;             // tile jobs per round (more stay flagged for the next round)
constexpr int LQ = 1024;                   // per-wave local queue (u16 LDS positions): <= 1024 live entries (in-queue filter)
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
// word index of cell (i, j) in a tile-major field
__device__ __forceinline__ size_t tm_index(int i, int j, int tiles_i) {
  return ((size_t)((j >> 5) * tiles_i + (i >> 5)) << 10) + ((j & 31) << 5) + (i & 31);
}

// field[q][tile][jl][il] = (G_INF << 8) | nbr(i, j) (0 outside the map); pending bitmaps zeroed
__global__ void tsa_init_kernel(const uint8_t* __restrict__ nbr, int rows, int cols, int tiles_i, int tiles_j,
                                unsigned* __restrict__ field, size_t field_stride, unsigned* __restrict__ pend,
                                size_t pend_stride, int n) {
  const size_t nw = (size_t)tiles_i * tiles_j * TILE_WORDS;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += step) {
    const int t = (int)(w >> 10), l = (int)(w & 1023);
    const int i = (t % tiles_i) * TS + (l & 31), j = (t / tiles_i) * TS + (l >> 5);
    const unsigned m = (i < rows && j < cols) ? nbr[(size_t)j * rows + i] : 0u;
    const unsigned v = 0xFFFFFF00u | m;
    for (int q = 0; q < n; ++q) field[(size_t)q * field_stride + w] = v;
  }
  const size_t pw = (size_t)n * pend_stride;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < pw; w += step) pend[w] = 0u;
}

struct TsaWave {
  unsigned tile[TW * TW];        // (g << 8) | mask, halo included; index (jl+1)*TW + (il+1)
  unsigned short lq[LQ];         // local queue of LDS positions
  unsigned char flags[TW * TW + 4];  // per LDS position: bit0 in the local queue, bit1 halo cell improved by this
                                     // job, bit2 interior cell improved beyond the current bucket
};

__global__ void __launch_bounds__(TSA_THREADS)
tsa_search_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries,
                  unsigned* __restrict__ field_all, size_t field_stride, unsigned* __restrict__ pend_all,
                  size_t pend_stride, int bucket_width, int32_t* __restrict__ paths, int max_path_len,
                  int32_t* __restrict__ rev_all, int rev_cap, rna_astar_result* __restrict__ results) {
  __shared__ TsaWave s_w[TSA_WAVES];
  __shared__ unsigned s_act[2][TSA_MAX_TILE_WORDS];   // active tiles: [0] current bucket (next round), [1] next bucket
  __shared__ unsigned short s_jobs[TSA_JOBS];
  __shared__ int s_njobs, s_first_fail, s_job_next, s_best, s_state, s_bucket, s_bucket0, s_rounds, s_role, s_expanded, s_len;

  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const rna_astar_query qu = queries[q];
  const int ncell = rows * cols;
  const int ntile = tiles_i * tiles_j;
  const int nt_words = (ntile + 31) >> 5;
  unsigned* field = field_all + (size_t)q * field_stride;
  unsigned* pend0 = pend_all + (size_t)q * pend_stride;   // two bitmaps of ntile*32 words each
  const size_t pend_words = (size_t)ntile * TS;

  const bool valid = qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell;
  if (!valid) {
    if (tid == 0) results[q] = rna_astar_result{2, 0, INF, 0, 0, 0};
    return;
  }
  const int si = qu.start % rows, sj = qu.start / rows;
  const int gi = qu.goal % rows, gj = qu.goal / rows;

  for (int w = tid; w < nt_words; w += TSA_THREADS) { s_act[0][w] = 0u; s_act[1][w] = 0u; }
  if (tid == 0) {
    s_best = INF; s_state = 0; s_rounds = 0; s_role = 0; s_expanded = 0;
    s_bucket = tsa_octile(si, sj, gi, gj) / bucket_width;
    s_bucket0 = s_bucket;
    const size_t ws = tm_index(si, sj, tiles_i);
    const unsigned w0 = ld_l2(&field[ws]);
    __hip_atomic_store(&field[ws], w0 & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // g(start) = 0
    const int ts = (sj >> 5) * tiles_i + (si >> 5);
    atomicOr(&pend0[(size_t)ts * TS + (sj & 31)], 1u << (si & 31));
    // a goal without a single traversable neighbour cannot be reached (blocked or walled in)
    if (qu.goal != qu.start && (ld_l2(&field[tm_index(gi, gj, tiles_i)]) & 0xffu) == 0u) s_state = 2;
  }
  __syncthreads();
  if (tid == 0 && s_state == 0) {
    const int ts = (sj >> 5) * tiles_i + (si >> 5);
    s_act[0][ts >> 5] = 1u << (ts & 31);
  }
  __syncthreads();
  if (s_state == 2) {
    if (tid == 0) results[q] = rna_astar_result{1, 0, INF, 0, 0, 0};
    return;
  }

  TsaWave& W = s_w[wv];
  int my_expanded = 0;
#ifdef RNA_TSA_DEBUG
  __shared__ int s_dbg_jobs, s_dbg_iters; __shared__ long long s_dbg_t[4];
  if (tid == 0) { s_dbg_jobs = 0; s_dbg_iters = 0; s_dbg_t[0] = s_dbg_t[1] = s_dbg_t[2] = s_dbg_t[3] = 0; }
  long long d_t[4] = {0, 0, 0, 0}; int d_jobs = 0, d_iters = 0;
#endif
  const unsigned long long lane_lt = (1ull << lane) - 1ull;

  for (;;) {
    // ---- build this round's job list from the active-tile bitset ----
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
        while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; s_jobs[k++] = (unsigned short)((w << 5) + b); }
      } else {
        atomicMin(&s_first_fail, base);   // job list full: these tiles stay flagged for the next round
      }
    }
    __syncthreads();
    const int njobs = s_njobs < s_first_fail ? s_njobs : s_first_fail;
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
    for (;;) {
      int job = 0;
      if (lane == 0) job = atomicAdd(&s_job_next, 1);
      job = __shfl(job, 0);
      if (job >= njobs) break;
      const int t = s_jobs[job];
#ifdef RNA_TSA_DEBUG
      const long long t0 = wall_clock64(); ++d_jobs;
#endif
      const int ti = t % tiles_i, tj = t / tiles_i;
      const int i0 = ti * TS, j0 = tj * TS;
      unsigned* ftile = field + ((size_t)t << 10);

      // 1. grab-and-clear the pending bits of this tile (lane = column jl)
      unsigned seed = 0u;
      if (lane < TS) seed = atomicExch(&pend_cur[(size_t)t * TS + lane], 0u);
      for (int w = lane; w < (TW * TW + 4) / 4; w += 64) reinterpret_cast<unsigned*>(W.flags)[w] = 0u;
      // 2. tile + halo -> LDS (after the pending bits: every grabbed bit's value is already in L2)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      {
        unsigned tv[TILE_WORDS / 64];   // issue all 16 coalesced loads, then one wait, then the LDS stores
#pragma unroll
        for (int r = 0; r < TILE_WORDS / 64; ++r) tv[r] = ld_l2(&ftile[r * 64 + lane]);
#pragma unroll
        for (int r = 0; r < TILE_WORDS / 64; ++r) {
          const int l = r * 64 + lane;
          W.tile[((l >> 5) + 1) * TW + (l & 31) + 1] = tv[r];
        }
      }
      {
        // halo: left/right neighbour columns (contiguous), top/bottom rows (strided), 4 corners
        const int h = lane & 31;
        const bool second = lane >= 32;
        // columns: tile (ti, tj-1) column 31 -> LDS jl=-1 ; tile (ti, tj+1) column 0 -> LDS jl=32
        {
          const int ntj = second ? tj + 1 : tj - 1;
          unsigned v = 0xFFFFFF00u;
          if (ntj >= 0 && ntj < tiles_j) v = ld_l2(&field[((size_t)(ntj * tiles_i + ti) << 10) + ((second ? 0 : 31) << 5) + h]);
          W.tile[(second ? TS + 1 : 0) * TW + h + 1] = v;
        }
        // rows: tile (ti-1, tj) row 31 -> LDS il=-1 ; tile (ti+1, tj) row 0 -> LDS il=32
        {
          const int nti = second ? ti + 1 : ti - 1;
          unsigned v = 0xFFFFFF00u;
          if (nti >= 0 && nti < tiles_i) v = ld_l2(&field[((size_t)(tj * tiles_i + nti) << 10) + (h << 5) + (second ? 0 : 31)]);
          W.tile[(h + 1) * TW + (second ? TS + 1 : 0)] = v;
        }
        if (lane < 4) {
          const int di = (lane & 1) ? 1 : -1, dj = (lane & 2) ? 1 : -1;
          const int nti = ti + di, ntj = tj + dj;
          unsigned v = 0xFFFFFF00u;
          if (nti >= 0 && nti < tiles_i && ntj >= 0 && ntj < tiles_j)
            v = ld_l2(&field[((size_t)(ntj * tiles_i + nti) << 10) + ((dj > 0 ? 0 : 31) << 5) + (di > 0 ? 0 : 31)]);
          W.tile[(dj > 0 ? TS + 1 : 0) * TW + (di > 0 ? TS + 1 : 0)] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#ifdef RNA_TSA_DEBUG
      const long long t1 = wall_clock64();
#endif
      // 3. seed the local queue from the pending bits
      int head = 0, tail = 0;   // wave-uniform
      {
        unsigned bits = seed;   // lane jl holds the bits (il) of its column
        for (;;) {
          const bool has = bits != 0u;
          const unsigned long long m = __ballot(has);
          if (!m) break;
          if (has) {
            const int il = __ffs(bits) - 1;
            bits &= bits - 1;
            const int p = (lane + 1) * TW + il + 1;
            const int pos = tail + __popcll(m & lane_lt);
            W.lq[pos & (LQ - 1)] = (unsigned short)p;
            W.flags[p] = 1;
          }
          tail += __popcll(m);
        }
      }
      __builtin_amdgcn_wave_barrier();

      // 4. relax to the tile-local fixed point of the current bucket.  One lane per popped cell; the
      //    eight directions are visited one after the other.  Within one direction all lanes target
      //    DIFFERENT cells (target = own cell + the same offset), so the min-update of a neighbour and
      //    the test-and-set of its in-queue flag are plain LDS reads and writes -- no LDS atomics
      //    (ds_min_rtn on 16 waves turned out to be the bottleneck of an earlier version).
      const int best_in = s_best;
      while (tail != head) {
        const int n = tail - head;
        const int take = n < 64 ? n : 64;
#ifdef RNA_TSA_DEBUG
        ++d_iters;
#endif
        const bool act = lane < take;
        const int p = act ? (int)W.lq[(head + lane) & (LQ - 1)] : (TW + 1);
        head += take;
        const unsigned cw = W.tile[p];
        if (act) W.flags[p] &= (unsigned char)~1u;   // popped: may be queued again
        const int g = (int)(cw >> 8);
        const int pil = p % TW - 1, pjl = p / TW - 1;
        const int ci = i0 + pil, cj = j0 + pjl;
        const int sb = s_best;
        const int best_now = best_in < sb ? best_in : sb;
        bool ok = act && g + tsa_octile(ci, cj, gi, gj) <= best_now;
        if (ok) {
          ++my_expanded;
          if (ci == gi && cj == gj) { atomicMin(&s_best, g); ok = false; }
        }
        const unsigned m = ok ? (cw & 0xffu) : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int di = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0);
          const int dj = k < 3 ? -1 : (k > 4 ? 1 : 0);
          const int np_ = p + di + dj * TW;
          const int ng = g + ((k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D);
          bool doit = false;
          if ((m >> k) & 1u) {
            const unsigned nwv = W.tile[np_];
            const unsigned fl = W.flags[np_];
            if ((unsigned)ng < (nwv >> 8)) {
              W.tile[np_] = ((unsigned)ng << 8) | (nwv & 0xffu);
              const int ni = ci + di, nj = cj + dj;
              const int fn = ng + tsa_octile(ni, nj, gi, gj);
              if (ng >= (int)G_INF - 2 * COST_D) s_state = 4;            // 24-bit g exhausted
              else {
                if (ni == gi && nj == gj) atomicMin(&s_best, ng);
                if (fn <= best_now) {
                  const int nil = pil + di, njl = pjl + dj;
                  if (nil < 0 || nil >= TS || njl < 0 || njl >= TS) W.flags[np_] = (unsigned char)(fl | 2u);  // halo
                  else if (fn >= bucket_end) W.flags[np_] = (unsigned char)(fl | 4u);                          // next bucket
                  else if (!(fl & 1u)) { W.flags[np_] = (unsigned char)(fl | 1u); doit = true; }
                }
              }
            }
          }
          const unsigned long long bm = __ballot(doit);
          if (doit) W.lq[(tail + __popcll(bm & lane_lt)) & (LQ - 1)] = (unsigned short)np_;
          tail += __popcll(bm);
        }
        __builtin_amdgcn_wave_barrier();
      }

#ifdef RNA_TSA_DEBUG
      const long long t2 = wall_clock64();
#endif
      // 5. write back.  Inner 30 x 30 cells are private to this tile: coalesced stores.  Edge ring:
      //    atomicMin (a neighbouring tile's job may have improved them in HBM meanwhile).
#pragma unroll
      for (int r = 0; r < TILE_WORDS / 64; ++r) {
        const int l = r * 64 + lane;
        const int il = l & 31, jl = l >> 5;
        const unsigned v = W.tile[(jl + 1) * TW + il + 1];
        if (il == 0 || il == TS - 1 || jl == 0 || jl == TS - 1) {
          if ((v >> 8) != G_INF) (void)atomicMin(&ftile[l], v);
        } else {
          ftile[l] = v;
        }
      }
      //    far-bucket cells of this tile: column jl -> one pending word (bit il)
      {
        bool anyfar = false;
#pragma unroll 4
        for (int r = 0; r < TS / 2; ++r) {
          const int jl = 2 * r + (lane >> 5), il = lane & 31;
          const bool f = (W.flags[(jl + 1) * TW + il + 1] & 4u) != 0u;
          const unsigned long long bm = __ballot(f);
          const unsigned word = (unsigned)(bm >> (lane & 32));
          if ((lane & 31) == 0 && word) { atomicOr(&pend_far[(size_t)t * TS + jl], word); }
          anyfar |= bm != 0ull;
        }
        if (anyfar && lane == 0) atomicOr(&s_act[1][t >> 5], 1u << (t & 31));
      }
      //    improved halo cells -> their tiles (value first, then the pending bit, then the activation)
      for (int hh = lane; hh < 4 * TW; hh += 64) {
        // ring positions: hh in [0,TW): jl=-1 row; [TW,2TW): jl=32 row; [2TW,3TW): il=-1 col; [3TW,4TW): il=32 col
        const int side = hh / TW, u = hh % TW;
        int pil, pjl;
        if (side == 0) { pjl = -1; pil = u - 1; }
        else if (side == 1) { pjl = TS; pil = u - 1; }
        else if (side == 2) { pil = -1; pjl = u - 1; }
        else { pil = TS; pjl = u - 1; }
        if (side >= 2 && (pjl < 0 || pjl >= TS)) continue;   // corners are covered by the row sides
        const int p = (pjl + 1) * TW + pil + 1;
        if (!(W.flags[p] & 2u)) continue;
        const int ni = i0 + pil, nj = j0 + pjl;
        if (ni < 0 || nj < 0 || ni >= rows || nj >= cols) continue;
        const unsigned v = W.tile[p];
        const size_t nidx = tm_index(ni, nj, tiles_i);
        const unsigned old = atomicMin(&field[nidx], v);
        if (v < old) {
          const int nt = (int)(nidx >> 10);
          const int fn = (int)(v >> 8) + tsa_octile(ni, nj, gi, gj);
          const bool far = fn >= bucket_end;
          atomicOr(&(far ? pend_far : pend_cur)[(size_t)nt * TS + (nj & 31)], 1u << (ni & 31));
          atomicOr(&s_act[far ? 1 : 0][nt >> 5], 1u << (nt & 31));
        }
      }
      __builtin_amdgcn_wave_barrier();
#ifdef RNA_TSA_DEBUG
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long t3 = wall_clock64();
      d_t[0] += t1 - t0; d_t[1] += t2 - t1; d_t[2] += t3 - t2;
#endif
    }
#ifdef RNA_TSA_DEBUG
    const long long tb0 = wall_clock64();
#endif
    // all stores / atomics of this round are performed before any wave loads tiles in the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int stop = __syncthreads_or(s_state == 4);
#ifdef RNA_TSA_DEBUG
    d_t[3] += wall_clock64() - tb0;
#endif
    if (tid == 0) s_rounds += 1;
    if (stop) break;
  }
  atomicAdd(&s_expanded, my_expanded);
#ifdef RNA_TSA_DEBUG
  if (lane == 0) { atomicAdd(&s_dbg_jobs, d_jobs); atomicAdd(&s_dbg_iters, d_iters);
    if (wv == 0) { s_dbg_t[0] = d_t[0]; s_dbg_t[1] = d_t[1]; s_dbg_t[2] = d_t[2]; s_dbg_t[3] = d_t[3]; } }
#endif
  __syncthreads();
#ifdef RNA_TSA_DEBUG
  if (tid == 0 && s_state == 1) {
    // status=jobs, path_len=local iterations (all waves), cost/expanded/rounds/buckets = wave-0 time (10 ns ticks): load, local, writeback, barrier-wait
    results[q] = rna_astar_result{s_dbg_jobs, s_dbg_iters, (int)s_dbg_t[0], (int)s_dbg_t[1], (int)s_dbg_t[2], (int)s_dbg_t[3]};
    paths[(size_t)q * max_path_len] = s_rounds; paths[(size_t)q * max_path_len + 1] = s_expanded;
  }
  if (s_state == 1) return;
#endif

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
      const unsigned wn = ld_l2(&field[inb ? tm_index(ni, nj, tiles_i) : tm_index(ci, cj, tiles_i)]);
      const bool hit = tid < 8 && inb && ((wc >> k) & 1u) && ((wn >> 8) != G_INF) && ((wn >> 8) + (unsigned)w == (wc >> 8));
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
  for (int i = tid; i < len; i += TSA_THREADS) path[i] = rev[len - 1 - i];
  if (tid == 0) results[q] = rna_astar_result{0, len, s_best, s_expanded, s_rounds, n_buckets};
}

// |{n : g(n) + h(n) <= f*}| per query from the resident tile-major fields (measurement utility)
__global__ void tsa_settled_kernel(int rows, int cols, int tiles_i, int tiles_j, const rna_astar_query* __restrict__ queries,
                                   const rna_astar_result* __restrict__ results, const unsigned* __restrict__ field_all,
                                   size_t field_stride, int32_t* __restrict__ counts) {
  __shared__ int s_cnt;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const rna_astar_result r = results[q];
  int cnt = 0;
  if (r.status == 0 || r.status == 3) {
    const int goal = queries[q].goal;
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

int tsa_launch(rna_engine* e, hipStream_t init_stream, hipStream_t search_stream, hipEvent_t ev_init, unsigned* field,
               size_t field_stride, unsigned* pend, size_t pend_stride, int32_t* rev, int rev_cap,
               const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len, rna_astar_result* res_dev) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  {
    KernelTimer kt(e, RNA_K_ASTAR_INIT, init_stream);
    hipLaunchKernelGGL(tsa_init_kernel, dim3(4096), dim3(256), 0, init_stream, e->nbr, rows, cols, ti, tj, field,
                       field_stride, pend, pend_stride, n);
    RNA_HIP(e, hipGetLastError());
  }
  if (ev_init) {
    RNA_HIP(e, hipEventRecord(ev_init, init_stream));
    RNA_HIP(e, hipStreamWaitEvent(search_stream, ev_init, 0));
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_SEARCH, search_stream);
    hipLaunchKernelGGL(tsa_search_kernel, dim3(n), dim3(TSA_THREADS), 0, search_stream, rows, cols, ti, tj, q_dev, field,
                       field_stride, pend, pend_stride, e->astar.bucket_width, paths_dev, max_len, rev, rev_cap, res_dev);
    RNA_HIP(e, hipGetLastError());
  }
  return RNA_OK;
}

int tsa_settled(rna_engine* e, const unsigned* field, size_t field_stride, const rna_astar_query* q, const rna_astar_result* r,
                int n, int32_t* d_counts) {
  const int rows = e->geom.size[0], cols = e->geom.size[1];
  const int ti = (rows + TS - 1) / TS, tj = (cols + TS - 1) / TS;
  hipLaunchKernelGGL(tsa_settled_kernel, dim3(n), dim3(1024), 0, e->stream, rows, cols, ti, tj, q, r, field, field_stride,
                     d_counts);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

}  // namespace rna
