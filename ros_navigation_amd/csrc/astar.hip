// astar.hip -- batched grid A* on gfx950 (one workgroup per query).
//
// Grid A* (DESIGN.md "Grid A* contract", restated by oracle/astar.c).  The reference has no grid
// search (AStarPlanner::makePlan walks a 9-vertex graph, mc/src/astar_planner.cpp:63-127); the
// contract is: 8-connected, integer costs 1000/1414, octile heuristic, no corner cutting, every cell
// with f <= f* settled, canonical predecessor = optimal neighbour with the lowest linear index.
// Because the path is a function of the exact distance field only, any label-correcting schedule
// that converges gives the CPU oracle's path bit for bit.  Schedule used here:
//   * bucketed parallel frontier ("delta-stepping" on f = g + h): cells whose f lies in the current
//     bucket [k*B, (k+1)*B) are relaxed to a fixed point in rounds, then the search advances.  With
//     B >= 2828 (the largest f increase of one step) a relaxation only ever targets the current or
//     the next bucket: an LDS-resident double-buffered frontier (cur / nxt) plus one "far" queue in
//     HBM suffice.
//   * one search field per concurrent query in HBM, one 32-bit word per cell = (g << 8) | mask,
//     where mask is the cell's 8-bit traversable-neighbour mask.  atomicMin on the word is atomicMin
//     on g.  A single CU sustains only ~1 scattered memory lane-op per 4 cycles (measured,
//     scripts/ubench_atomics.hip), so the expansion is built around FEW, WIDE accesses: the 3x3
//     neighbourhood is three 12-byte buffer loads (one per column, lanes of a column are contiguous
//     in the column-major field), which deliver the staleness probe, the cell's own mask and all
//     eight neighbour values at once; an atomicMin is then issued only for neighbours it can improve.
//   * queue entries carry (cell, g); a popped entry whose g no longer matches the field is stale and
//     is dropped;  pruning: once the goal has a finite g, candidates with f > g(goal) are dropped;
//   * the path is rebuilt by one wavefront: lane k probes neighbour k, ballot + ffs picks the
//     lowest-index optimal predecessor.
// g is 24 bits: path cost < 16 777 215 (about 16 700 straight cells); longer searches fail loudly
// with status 4.
#include "engine.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <vector>

using namespace rna;

namespace {

constexpr int COST_S = 1000, COST_D = 1414;
constexpr int INF = 0x7fffffff;          // "no path" cost in results
constexpr unsigned G_INF = 0xFFFFFFu;    // unreached cell in the packed field
#ifndef RNA_ASTAR_LQ_CAP
#define RNA_ASTAR_LQ_CAP 4096
#endif
constexpr int LQ_CAP = RNA_ASTAR_LQ_CAP;   // entries of each LDS frontier queue (2 queues x 8 B x LQ_CAP)

typedef int v3i __attribute__((ext_vector_type(3)));

__device__ __forceinline__ int octile(int i, int j, int gi, int gj) {
  const int dx = abs(i - gi), dy = abs(j - gj);
  const int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return COST_S * mx + (COST_D - COST_S) * mn;
}

__device__ __forceinline__ unsigned field_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2-served (sc1)
}

// field[q][c] = (G_INF << 8) | nbr[c] for the n concurrent queries: 4 cells per thread, 16-byte stores
__global__ void astar_init_kernel(const uint8_t* __restrict__ nbr, unsigned* __restrict__ field, size_t stride,
                                  int n, size_t ncell) {
  const size_t n4 = ncell / 4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += step) {
    const uchar4 m = reinterpret_cast<const uchar4*>(nbr)[i];
    const uint4 v = make_uint4(0xFFFFFF00u | m.x, 0xFFFFFF00u | m.y, 0xFFFFFF00u | m.z, 0xFFFFFF00u | m.w);
    for (int q = 0; q < n; ++q) reinterpret_cast<uint4*>(field + (size_t)q * stride)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (ncell & 3)) {
    const size_t c = n4 * 4 + threadIdx.x;
    for (int q = 0; q < n; ++q) field[(size_t)q * stride + c] = 0xFFFFFF00u | nbr[c];
  }
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
astar_search_kernel(int rows, int cols, const rna_astar_query* __restrict__ queries, unsigned* __restrict__ field_all,
                    size_t field_stride, int2* __restrict__ queues, int queue_cap, int bucket_width,
                    int32_t* __restrict__ paths, int max_path_len, rna_astar_result* __restrict__ results) {
  __shared__ int2 l_q[2][LQ_CAP];  // frontier of the current bucket: (cell, g); overflow spills to HBM
  __shared__ int s_n[2];           // entries pushed into frontier queue b (LDS part + spill part)
  __shared__ int s_far_n, s_best, s_overflow, s_expanded, s_state, s_bucket, s_rounds, s_bucket0;
  __shared__ int s_cur;            // which LDS queue is being popped
  __shared__ int s_lds_n, s_glob_n;  // entries to pop this round from LDS / from the HBM source
  __shared__ int s_sel[3];         // HBM buffers playing: spill of cur, spill of nxt, far
  __shared__ int s_len;
#ifdef RNA_ASTAR_DEBUG
  __shared__ int s_dbg_iters, s_dbg_pops, s_dbg_glob;
  if (threadIdx.x == 0) { s_dbg_iters = 0; s_dbg_pops = 0; s_dbg_glob = 0; }
#endif
  const int q = blockIdx.x;
  const int tid = threadIdx.x;
  const rna_astar_query qu = queries[q];
  unsigned* field = field_all + (size_t)q * field_stride;
  int2* qbase = queues + (size_t)q * 3 * queue_cap;
  const int ncell = rows * cols;

  const bool valid = qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell;
  if (!valid) {
    if (tid == 0) results[q] = rna_astar_result{2, 0, INF, 0, 0, 0};
    return;
  }
  const int gi = qu.goal % rows, gj = qu.goal / rows;
  // wave-uniform buffer descriptor starting ONE WORD BEFORE this query's field (fields are padded),
  // so the 12-byte load at byte offset 4*c covers cells c-1, c, c+1; out-of-range loads return 0
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(field - 1, 0, (ncell + 2) * 4, 0x00020000);

  if (tid == 0) {
    s_sel[0] = 0; s_sel[1] = 1; s_sel[2] = 2;
    s_n[0] = 0; s_n[1] = 0; s_cur = 0; s_lds_n = 1; s_glob_n = 0;
    s_far_n = 0;
    s_best = INF; s_overflow = 0; s_expanded = 0; s_state = 0;
    const int f0 = octile(qu.start % rows, qu.start / rows, gi, gj);
    s_bucket = f0 / bucket_width;
    s_bucket0 = s_bucket;
    s_rounds = 0;
    const unsigned w0 = field_load(&field[qu.start]);
    __hip_atomic_store(&field[qu.start], w0 & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // g = 0
    l_q[0][0] = make_int2(qu.start, 0);
    // a goal without a single traversable neighbour (blocked, or walled in) cannot be reached:
    // answer "no path" without flooding the whole connected component
    if (qu.goal != qu.start && (field_load(&field[qu.goal]) & 0xffu) == 0u) s_state = 2;
  }
  __syncthreads();
  if (s_state == 2) {
    if (tid == 0) results[q] = rna_astar_result{1, 0, INF, 0, 0, 0};
    return;
  }

  int my_expanded = 0;
  for (;;) {
    const int cb = s_cur, nb = cb ^ 1;
    const int lds_n = s_lds_n, glob_n = s_glob_n;
    const int2* gsrc = qbase + (size_t)s_sel[0] * queue_cap;
    int2* spill = qbase + (size_t)s_sel[1] * queue_cap;
    int2* far = qbase + (size_t)s_sel[2] * queue_cap;
    const int best = s_best;
    const long long bucket_end = ((long long)s_bucket + 1) * bucket_width;  // exclusive

    auto expand = [&](const int cell, const int gv) {
      const int ci = cell % rows, cj = cell / rows;
      if (gv + octile(ci, cj, gi, gj) > best) return;
      if (cell == qu.goal) { atomicMin(&s_best, gv); ++my_expanded; return; }  // (also covers start == goal)
      // 3x3 neighbourhood = three 12-byte loads: rows (i-1, i, i+1) of columns j-1, j, j+1
      const int base = cell * 4;
      const v3i c0 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, base - rows * 4, 0, 16);
      const v3i c1 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, base, 0, 16);
      const v3i c2 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, base + rows * 4, 0, 16);
      const unsigned centre = (unsigned)c1.y;
      if ((int)(centre >> 8) != gv) return;  // stale: the cell was improved after this entry was queued
      ++my_expanded;
      const unsigned m = centre & 0xffu;
      const unsigned w[8] = {(unsigned)c0.x, (unsigned)c0.y, (unsigned)c0.z, (unsigned)c1.x,
                             (unsigned)c1.z, (unsigned)c2.x, (unsigned)c2.y, (unsigned)c2.z};
      const int off[8] = {-1 - rows, -rows, 1 - rows, -1, 1, rows - 1, rows, rows + 1};
      unsigned old[8], nw[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // issue every useful relaxation before consuming any result
        const int ng = gv + ((k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D);
        nw[k] = ((unsigned)ng << 8) | (w[k] & 0xffu);
        old[k] = 0;
        if (((m >> k) & 1u) && (unsigned)ng < (w[k] >> 8)) old[k] = atomicMin(&field[cell + off[k]], nw[k]);
      }
      const int best_now = s_best;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (nw[k] >= old[k]) continue;  // not issued (old = 0) or lost the race
        const int ng = (int)(nw[k] >> 8);
        const int nc = cell + off[k];
        const int ni = ci + ((k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0));
        const int nj = cj + (k < 3 ? -1 : (k > 4 ? 1 : 0));
        const int fn = ng + octile(ni, nj, gi, gj);
        if (fn > best_now) continue;
        if (ng >= (int)G_INF - 2 * COST_D) { s_overflow = 2; continue; }  // 24-bit g exhausted
        if (nc == qu.goal) atomicMin(&s_best, ng);
        if (fn < bucket_end) {
          const int pos = atomicAdd(&s_n[nb], 1);
          if (pos < LQ_CAP) l_q[nb][pos] = make_int2(nc, ng);
          else if (pos - LQ_CAP < queue_cap) spill[pos - LQ_CAP] = make_int2(nc, ng);
          else s_overflow = 1;
        } else {
          const int pos = atomicAdd(&s_far_n, 1);
          if (pos < queue_cap) far[pos] = make_int2(nc, ng); else s_overflow = 1;
        }
      }
    };
    for (int e = tid; e < lds_n; e += THREADS) {     // LDS-resident part of the frontier
      const int2 ent = l_q[cb][e];
      expand(ent.x, ent.y);
    }
    for (int e = tid; e < glob_n; e += THREADS) {    // spilled part / a freshly opened bucket (HBM)
      const int2 ent = gsrc[e];
      expand(ent.x, ent.y);
    }
    __syncthreads();
    if (tid == 0) {
      s_rounds += 1;
#ifdef RNA_ASTAR_DEBUG
      s_dbg_iters += (lds_n + glob_n + THREADS - 1) / THREADS; s_dbg_pops += lds_n + glob_n; if (glob_n) s_dbg_glob += 1;
#endif
      const int nn = s_n[nb];
      if (s_overflow) {
        s_state = 2 + s_overflow;          // 3 queue overflow, 4 cost overflow
      } else if (nn > 0) {                 // same bucket, next round: pop what was just pushed
        s_cur = nb;
        s_lds_n = nn < LQ_CAP ? nn : LQ_CAP;
        s_glob_n = nn - s_lds_n;
        s_n[cb] = 0;
        const int t = s_sel[0]; s_sel[0] = s_sel[1]; s_sel[1] = t;
      } else {
        // bucket k is at its fixed point: every cell with f < (k+1)*B has its exact g.
        const long long done_below = ((long long)s_bucket + 1) * bucket_width;
        if (s_best != INF && (long long)s_best < done_below) s_state = 1;       // goal settled, ties included
        else if (s_far_n == 0) s_state = (s_best != INF) ? 1 : 2;              // frontier exhausted
        else {                             // advance: the far queue (HBM) becomes the frontier
          const int t = s_sel[0]; s_sel[0] = s_sel[2]; s_sel[2] = t;
          s_lds_n = 0; s_glob_n = s_far_n; s_far_n = 0; s_bucket += 1;
          s_n[0] = 0; s_n[1] = 0;
        }
      }
    }
    __syncthreads();
    if (s_state != 0) break;
  }
  atomicAdd(&s_expanded, my_expanded);
  __syncthreads();

  const int state = s_state;
  const int n_buckets = s_bucket - s_bucket0 + 1;
  if (state != 1) {
    const int status = state == 3 ? (int)RNA_ECAPACITY : (state == 4 ? 4 : 1);
    if (tid == 0) results[q] = rna_astar_result{status, 0, INF, s_expanded, s_rounds, n_buckets};
    return;
  }

  // ---- canonical backtrace by the first wavefront; reversed path staged in the queue memory ----
  int* rev = reinterpret_cast<int*>(qbase);
  const int rev_cap = 3 * queue_cap * 2;
  if (tid < 64) {
    int c = qu.goal;
    int len = 0;
    bool ok = true;
    const int k = tid & 7;
    const int w = (k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D;
    const int offk = (k == 0 ? -1 - rows : k == 1 ? -rows : k == 2 ? 1 - rows : k == 3 ? -1 : k == 4 ? 1
                      : k == 5 ? rows - 1 : k == 6 ? rows : rows + 1);
    for (;;) {
      if (tid == 0 && len < rev_cap) rev[len] = c;
      ++len;
      if (c == qu.start) break;
      if (len > ncell) { ok = false; break; }
      // lane k probes neighbour k; g(c), its mask and the eight g(n) arrive in one round trip
      const int nc = c + offk;
      const bool inb = nc >= 0 && nc < ncell;
      const unsigned wc = field_load(&field[c]);
      const unsigned wn = field_load(&field[inb ? nc : c]);
      const bool hit = tid < 8 && ((wc >> k) & 1u) && ((wn >> 8) != G_INF) && ((wn >> 8) + (unsigned)w == (wc >> 8));
      const unsigned long long mask = __ballot(hit);
      if (!mask) { ok = false; break; }
      const int lane = __ffsll((long long)mask) - 1;
      c = __shfl(nc, lane);
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
  for (int i = tid; i < len; i += THREADS) path[i] = rev[len - 1 - i];
#ifdef RNA_ASTAR_DEBUG
  if (tid == 0) { results[q] = rna_astar_result{0, s_dbg_glob, s_dbg_pops, s_expanded, s_rounds, s_dbg_iters}; return; }
#endif
  if (tid == 0) results[q] = rna_astar_result{0, len, s_best, s_expanded, s_rounds, n_buckets};
}

// |{n : g(n) + h(n) <= f*}| per query, from the resident fields (measurement utility)
__global__ void astar_settled_kernel(int rows, int cols, const rna_astar_query* __restrict__ queries,
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
    const int ncell = rows * cols;
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
      const unsigned gv = field[c] >> 8;
      if (gv != G_INF && (int)gv + octile(c % rows, c / rows, gi, gj) <= r.cost) ++cnt;
    }
  }
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[q] = s_cnt;
}

// allocation of the configuration ensure_config settled on; everything is released again on failure
static int alloc_stages_impl(rna_engine* e) {
  AstarDevice& a = e->astar;
  int rc;
  // frontier kernel: each query's field is padded (the 3-cell column loads reach one word past either end) and 256-byte aligned
  a.field_stride = ((e->ncell + 64 + 63) / 64) * 64;
  for (int d = 0; d < a.depth; ++d) {
    if (a.mode != 0) {
      char* pool = nullptr;
      if ((rc = dev_alloc(e, &pool, tsa_pool_bytes(a.max_queries, a.page_cap))) != RNA_OK) { astar_release(e); return rc; }
      a.g[d] = reinterpret_cast<int32_t*>(pool);
      if ((rc = dev_alloc(e, &a.rev[d], (size_t)a.rev_cap * a.max_queries)) != RNA_OK) { astar_release(e); return rc; }
      char* aux = nullptr;
      const size_t aux_bytes = tsa_aux_bytes(e, a.max_queries, a.page_cap);
      if ((rc = dev_alloc(e, &aux, aux_bytes)) != RNA_OK) { astar_release(e); return rc; }
      a.tsa_aux[d] = aux;
      RNA_HIP(e, hipMemsetAsync(aux, 0, aux_bytes, e->stream));
      if ((rc = tsa_stage_prepare(e, d)) != RNA_OK) { astar_release(e); return rc; }
      if (a.page_cap < tsa_tiles(e)) {   // a search may outgrow its share of pages: retry slots with a page per tile
        char* rp = nullptr;
        if ((rc = dev_alloc(e, &rp, tsa_retry_pool_bytes(e))) != RNA_OK) { astar_release(e); return rc; }
        a.g_retry[d] = reinterpret_cast<int32_t*>(rp);
        char* ra = nullptr;
        if ((rc = dev_alloc(e, &ra, tsa_retry_aux_bytes(e))) != RNA_OK) { astar_release(e); return rc; }
        a.tsa_aux_retry[d] = ra;
        const hipError_t me = hipMemsetAsync(ra, 0, tsa_retry_aux_bytes(e), e->stream);
        if (me != hipSuccess) { astar_release(e); RNA_HIP(e, me); }
        if ((rc = tsa_retry_prepare(e, d)) != RNA_OK) { astar_release(e); return rc; }
      }
    } else {
      if ((rc = dev_alloc(e, &a.g[d], a.field_stride * (size_t)a.max_queries + 128)) != RNA_OK) { astar_release(e); return rc; }
      if ((rc = dev_alloc(e, &a.queues[d], (size_t)3 * a.queue_cap * a.max_queries)) != RNA_OK) { astar_release(e); return rc; }
    }
    if (a.depth > 1) {
      {
        // searches run at the lowest queue priority, the engine stream (map update, VFH+, field reset) at the highest:
        // its short kernels gate the next search launch and must not queue behind 1000 waiting search workgroups
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (getenv("RNA_NO_STREAM_PRIORITY")) prio_lo = prio_hi = 0;
        // The search streams may not use one CU in eight (ROCr deals the bits of a queue's CU mask round-robin to the
        // XCDs, so the first n bits are n / 8 CUs of every XCD).  A search workgroup holds its CU slot for a whole query
        // (milliseconds) and four of them fill a CU's registers, so a stream priority cannot make room for the engine
        // stream's short kernels (map update, VFH+, field reset) that gate the next search launch: without the reserve
        // they take 3 ms instead of 0.3 ms each and the step rate hangs on them (22 k instead of 33 k cycles/s when this
        // was introduced; with the final kernel 16 / 24 / 32 / 40 / 48 reserved CUs: 74.6 / 78.8 / 80.5 / 78.8 / 76.7 k).
        // RNA_SEARCH_CU_SKIP=n overrides the number of reserved CUs, 0 = no mask (stream priority only).
        int skip = e->cu_count / 8;
        if (const char* m = getenv("RNA_SEARCH_CU_SKIP")) skip = atoi(m);
        if (skip > 0 && skip < e->cu_count) {
          uint32_t mask[16] = {};
          const int words = (e->cu_count + 31) / 32 < 16 ? (e->cu_count + 31) / 32 : 16;
          for (int c = skip; c < e->cu_count && c < 512; ++c) mask[c >> 5] |= 1u << (c & 31);
          RNA_HIP(e, hipExtStreamCreateWithCUMask(&a.side[d], (uint32_t)words, mask));
        } else {
          RNA_HIP(e, hipStreamCreateWithPriority(&a.side[d], hipStreamNonBlocking, prio_lo));
        }
      }
      RNA_HIP(e, hipEventCreateWithFlags(&a.done[d], hipEventDisableTiming));
    }
    a.busy[d] = false;
    a.stage_seq[d] = 0;
  }
  if (a.depth > 1) RNA_HIP(e, hipEventCreateWithFlags(&a.ev_init, hipEventDisableTiming));
  a.launches = 0;
  if ((rc = dev_alloc(e, &a.queries_dev, (size_t)a.max_queries)) != RNA_OK) { astar_release(e); return rc; }
  if ((rc = dev_alloc(e, &a.results_dev, (size_t)a.max_queries)) != RNA_OK) { astar_release(e); return rc; }
  return RNA_OK;
}

// (whatever fails inside -- an allocation, a memset, a stream or an event -- nothing half-built is left behind: the next
// call starts from nothing again instead of finding a[0] allocated and taking the configuration for complete)
static int alloc_stages(rna_engine* e) {
  const int rc = alloc_stages_impl(e);
  if (rc != RNA_OK) {
    const std::string why = e->err;
    (void)astar_release(e);
    e->err = why;
  }
  return rc;
}

int ensure_config(rna_engine* e) {
  AstarDevice& a = e->astar;
  if (a.g[0]) return RNA_OK;
  if (a.max_queries <= 0) a.max_queries = 256;
  if (const char* t = getenv("RNA_ASTAR_THREADS")) a.threads = atoi(t);  // tuning knob of the frontier kernel: 256 / 512 / 1024
  a.mode = 1;
  if (const char* k = getenv("RNA_ASTAR_KERNEL")) a.mode = (k[0] == 'f') ? 0 : 1;  // frontier | tile (default)
  if (a.mode != 0 && !tsa_supported(e)) a.mode = 0;
  if (a.depth < 1) a.depth = 1;
  if (a.depth > AstarDevice::MAX_DEPTH) a.depth = AstarDevice::MAX_DEPTH;
  if (a.queue_cap <= 0) {
    // a bucket's queue holds the cells whose f falls into one bucket_width band of the search
    // ellipse (plus duplicates); 64 x (rows + cols) entries is a wide margin, checked at run time
    long long c = 64LL * (e->geom.size[0] + e->geom.size[1]);
    if (c < 65536) c = 65536;
    if (c > (1 << 22)) c = 1 << 22;
    a.queue_cap = (int)c;
  }
  if (e->ncell >= (1ull << 30)) return fail(e, RNA_EINVAL, "grid A*: more than 2^30 cells");
  // fit into free HBM (25 % headroom): (tile kernel) fewer pages per query, fewer pipeline stages, fewer pages still
  // -- a search that needs more than its share then ends with status 5 --, then fewer concurrent queries
  size_t free_b = 0, total_b = 0;
  RNA_HIP(e, hipMemGetInfo(&free_b, &total_b));
  const int ntile = a.mode != 0 ? tsa_tiles(e) : 0;
  a.page_cap = ntile;
  if (a.page_cap_request > 0 && a.page_cap_request < ntile) a.page_cap = a.page_cap_request;
  if (const char* c = getenv("RNA_ASTAR_PAGE_CAP")) { const int v = atoi(c); if (v > 0 && v < ntile) a.page_cap = v; }
  auto stage_bytes = [&]() -> double {
    if (a.mode != 0)
      return (double)tsa_pool_bytes(a.max_queries, a.page_cap) + (double)tsa_aux_bytes(e, a.max_queries, a.page_cap) +
             (double)a.rev_cap * 4.0 * a.max_queries +
             (a.page_cap < ntile ? (double)tsa_retry_pool_bytes(e) + (double)tsa_retry_aux_bytes(e) : 0.0);
    return ((double)(e->ncell + 128) * 4.0 + 3.0 * a.queue_cap * sizeof(int2)) * a.max_queries;
  };
  // (tile kernel) half a map's worth of pages per query first -- searches touch a few per cent of the map, and twelve
  // stages of 17.8 GB each at 4096^2 x 256 queries sit right at the budget --, then fewer stages, then fewer pages
  if (a.mode != 0 && a.page_cap > 64 && a.page_cap > ntile / 2 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.page_cap = (ntile + 1) / 2;
  while (a.depth > 1 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.depth -= 1;
  while (a.mode != 0 && a.page_cap > 64 && a.page_cap > ntile / 8 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.page_cap /= 2;
  while (a.max_queries > 1 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.max_queries /= 2;
  // HBM may have gone to another process between hipMemGetInfo and here: step down and try again
  for (;;) {
    const int rc = alloc_stages(e);
    if (rc != RNA_ENOMEM) return rc;
    (void)hipGetLastError();   // the failed hipMalloc must not surface at the next launch check
    if (a.mode != 0 && a.page_cap > 64 && a.page_cap > ntile / 2) a.page_cap = (ntile + 1) / 2;
    else if (a.depth > 1) a.depth = a.depth > 2 ? a.depth * 3 / 4 : 1;
    else if (a.mode != 0 && a.page_cap > 64 && a.page_cap > ntile / 8) a.page_cap /= 2;
    else if (a.max_queries > 1) a.max_queries /= 2;
    else return rc;
  }
}

// One batch (<= max_queries): field init on the engine stream (it reads the neighbour masks, which
// the next map update may overwrite), then the search on this pipeline stage's own stream.
int launch_chunk(rna_engine* e, const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len,
                 rna_astar_result* res_dev, int* slot_out) {
  AstarDevice& a = e->astar;
  // Which stage: the one that has been free the longest -- not simply the next in turn.  A batch lasts as long as its
  // longest search; taking the stages strictly in turn, the engine stream waits for the one slow batch while the stages
  // behind it have long finished (they sat idle 40 % of the time at thirteen stages).  If none is free, the oldest.
  int slot = 0;
  if (a.depth > 1) {
    int best = -1, oldest = 0;
    for (int d = 0; d < a.depth; ++d) {
      if (a.stage_seq[d] < a.stage_seq[oldest]) oldest = d;
      bool free_now = !a.busy[d];
      if (!free_now) {
        const hipError_t q = hipEventQuery(a.done[d]);
        if (q == hipSuccess) { a.busy[d] = false; free_now = true; }
        else (void)hipGetLastError();   // hipErrorNotReady is not an error
      }
      if (free_now && (best < 0 || a.stage_seq[d] < a.stage_seq[best])) best = d;
    }
    slot = best >= 0 ? best : oldest;
  }
  hipStream_t search_stream = a.depth > 1 ? a.side[slot] : e->stream;
  if (a.depth > 1 && a.busy[slot]) RNA_HIP(e, hipStreamWaitEvent(e->stream, a.done[slot], 0));  // wait until the stage is free again
  if (a.mode != 0) {
    int rc = tsa_launch(e, slot, e->stream, search_stream, a.depth > 1 ? a.ev_init : nullptr, q_dev, n, paths_dev, max_len, res_dev);
    if (rc != RNA_OK) return rc;
  } else {
    unsigned* field = reinterpret_cast<unsigned*>(a.g[slot]) + 64;
    {
      KernelTimer kt(e, RNA_K_ASTAR_INIT);
      hipLaunchKernelGGL(astar_init_kernel, dim3(4096), dim3(256), 0, e->stream, e->nbr, field, a.field_stride, n, e->ncell);
      RNA_HIP(e, hipGetLastError());
    }
    if (a.depth > 1) {
      RNA_HIP(e, hipEventRecord(a.ev_init, e->stream));
      RNA_HIP(e, hipStreamWaitEvent(search_stream, a.ev_init, 0));
    }
    {
      KernelTimer kt(e, RNA_K_ASTAR_SEARCH, search_stream);
#define RNA_LAUNCH_SEARCH(T)                                                                                        \
  hipLaunchKernelGGL(astar_search_kernel<T>, dim3(n), dim3(T), 0, search_stream, e->geom.size[0], e->geom.size[1], \
                     q_dev, field, a.field_stride, a.queues[slot], a.queue_cap, a.bucket_width, paths_dev, max_len, \
                     res_dev)
      if (a.threads == 256) RNA_LAUNCH_SEARCH(256);
      else if (a.threads == 512) RNA_LAUNCH_SEARCH(512);
      else RNA_LAUNCH_SEARCH(1024);
#undef RNA_LAUNCH_SEARCH
      RNA_HIP(e, hipGetLastError());
    }
  }
  if (a.depth > 1) {
    RNA_HIP(e, hipEventRecord(a.done[slot], search_stream));
    a.busy[slot] = true;
  }
  a.launches += 1;
  a.stage_seq[slot] = a.launches;
  a.last_queries = q_dev;
  a.last_results = res_dev;
  a.last_n = n;
  a.last_slot = slot;
  if (slot_out) *slot_out = slot;
  return RNA_OK;
}

}  // namespace

namespace rna {
#ifdef RNA_TSA_STATS
void tsa_stats_dump();
#endif
int astar_release(rna_engine* e) {
  AstarDevice& a = e->astar;
  (void)sync_all(e);
#ifdef RNA_TSA_STATS
  if (a.mode == 1) tsa_stats_dump();
#endif
  for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d) {
    dev_free(&a.g[d]); dev_free(&a.queues[d]); dev_free(&a.rev[d]);
    if (a.tsa_aux[d]) { (void)hipFree(a.tsa_aux[d]); a.tsa_aux[d] = nullptr; }
    dev_free(&a.g_retry[d]);
    if (a.tsa_aux_retry[d]) { (void)hipFree(a.tsa_aux_retry[d]); a.tsa_aux_retry[d] = nullptr; }
    if (a.side[d]) { (void)hipStreamDestroy(a.side[d]); a.side[d] = nullptr; }
    if (a.done[d]) { (void)hipEventDestroy(a.done[d]); a.done[d] = nullptr; }
    a.busy[d] = false;
  }
  if (a.ev_init) { (void)hipEventDestroy(a.ev_init); a.ev_init = nullptr; }
  dev_free(&a.queries_dev); dev_free(&a.results_dev); dev_free(&a.paths_dev);
  a.paths_cap = 0;
  a.last_queries = nullptr; a.last_results = nullptr; a.last_n = 0;
  return RNA_OK;
}
}  // namespace rna

extern "C" int rna_astar_configure(rna_engine* e, int max_queries, int queue_capacity, int bucket_width) {
  if (!e || max_queries < 0 || queue_capacity < 0 || bucket_width < 0) return RNA_EINVAL;
  if (bucket_width != 0 && bucket_width < 2 * COST_D) return fail(e, RNA_EINVAL, "bucket_width must be >= 2828");
  RNA_HIP(e, hipSetDevice(e->device));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  AstarDevice& a = e->astar;
  if (max_queries || queue_capacity) astar_release(e);
  if (const char* d = getenv("RNA_ASTAR_PIPELINE")) {  // pipeline stages (1 = searches run on the engine stream)
    const int dd = atoi(d);
    if (dd >= 1 && dd != a.depth) { astar_release(e); a.depth = dd; }
  }
  if (max_queries) a.max_queries = max_queries;
  if (queue_capacity) a.queue_cap = queue_capacity;
  if (bucket_width) a.bucket_width = bucket_width;
  return RNA_OK;
}

extern "C" int rna_astar_set_pipeline_depth(rna_engine* e, int depth) {
  if (!e || depth < 1 || depth > AstarDevice::MAX_DEPTH) return RNA_EINVAL;
  RNA_HIP(e, hipSetDevice(e->device));
  if (depth != e->astar.depth) {
    astar_release(e);
    e->astar.depth = depth;
  }
  return RNA_OK;
}

extern "C" int rna_astar_effective_config(const rna_engine* e, int* pipeline_depth, int* pages_per_query, int* max_queries) {
  if (!e) return RNA_EINVAL;
  const AstarDevice& a = e->astar;
  const bool live = a.g[0] != nullptr;
  if (pipeline_depth) *pipeline_depth = live ? a.depth : 0;
  if (pages_per_query) *pages_per_query = live ? a.page_cap : 0;
  if (max_queries) *max_queries = live ? a.max_queries : 0;
  return RNA_OK;
}

extern "C" int rna_astar_set_page_cap(rna_engine* e, int pages_per_query) {
  if (!e || pages_per_query < 0) return RNA_EINVAL;
  RNA_HIP(e, hipSetDevice(e->device));
  if (pages_per_query != e->astar.page_cap_request) {
    astar_release(e);
    e->astar.page_cap_request = pages_per_query;
  }
  return RNA_OK;
}

extern "C" int rna_astar_batch_device(rna_engine* e, const rna_astar_query* queries, int n, int32_t* paths,
                                      int max_path_len, rna_astar_result* results) {
  if (!e || n < 0 || max_path_len <= 0 || (n > 0 && (!queries || !paths || !results))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = ensure_config(e);
  if (rc != RNA_OK) return rc;
  // the tile kernels search in map space (unwrapped indices); the frontier kernel walks buffer indices
  if (e->astar.mode == 0 && (e->geom.start[0] != 0 || e->geom.start[1] != 0))
    return fail(e, RNA_ESTATE, "grid A* (frontier kernel) needs startIndex (0,0): buffer adjacency must be map adjacency");
  if ((rc = map_prepare_nbr(e)) != RNA_OK) return rc;
  const int chunk = e->astar.max_queries;
  for (int o = 0; o < n; o += chunk) {
    const int m = std::min(chunk, n - o);
    rc = launch_chunk(e, queries + o, m, paths + (size_t)o * max_path_len, max_path_len, results + o, nullptr);
    if (rc != RNA_OK) return rc;
  }
  return RNA_OK;
}

extern "C" int rna_astar_batch(rna_engine* e, const rna_astar_query* queries_host, int n, int32_t* paths_host,
                               int max_path_len, rna_astar_result* results_host) {
  if (!e || n < 0 || max_path_len <= 0 || (n > 0 && (!queries_host || !paths_host || !results_host))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = ensure_config(e);
  if (rc != RNA_OK) return rc;
  // the tile kernels search in map space (unwrapped indices); the frontier kernel walks buffer indices
  if (e->astar.mode == 0 && (e->geom.start[0] != 0 || e->geom.start[1] != 0))
    return fail(e, RNA_ESTATE, "grid A* (frontier kernel) needs startIndex (0,0): buffer adjacency must be map adjacency");
  if ((rc = map_prepare_nbr(e)) != RNA_OK) return rc;
  AstarDevice& a = e->astar;
  const int chunk = a.max_queries;
  const size_t need = (size_t)chunk * max_path_len;
  if ((size_t)a.paths_cap < need) {
    if (need > 0x7fffffffULL) return fail(e, RNA_EINVAL, "max_path_len too large");
    if ((rc = dev_alloc(e, &a.paths_dev, need)) != RNA_OK) return rc;
    a.paths_cap = (int)need;
  }
  for (int o = 0; o < n; o += chunk) {
    const int m = std::min(chunk, n - o);
    RNA_HIP(e, hipMemcpyAsync(a.queries_dev, queries_host + o, (size_t)m * sizeof(rna_astar_query),
                              hipMemcpyHostToDevice, e->stream));
    int slot = 0;
    rc = launch_chunk(e, a.queries_dev, m, a.paths_dev, max_path_len, a.results_dev, &slot);
    if (rc != RNA_OK) return rc;
    if (a.depth > 1) RNA_HIP(e, hipStreamWaitEvent(e->stream, a.done[slot], 0));
    RNA_HIP(e, hipMemcpyAsync(results_host + o, a.results_dev, (size_t)m * sizeof(rna_astar_result),
                              hipMemcpyDeviceToHost, e->stream));
    RNA_HIP(e, hipMemcpyAsync(paths_host + (size_t)o * max_path_len, a.paths_dev,
                              (size_t)m * max_path_len * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    RNA_HIP(e, hipStreamSynchronize(e->stream));
  }
  for (int i = 0; i < n; ++i)
    if (results_host[i].status == RNA_ECAPACITY)
      return fail(e, RNA_ECAPACITY, "A* frontier queue overflow: raise queue_capacity (rna_astar_configure)");
  return RNA_OK;
}

extern "C" int rna_astar_download_nbr_mask(rna_engine* e, uint8_t* host, size_t n) {
  if (!e || !host || n != e->ncell) return RNA_EINVAL;
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = map_prepare_nbr(e);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(host, e->nbr, n, hipMemcpyDeviceToHost, e->stream));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_astar_settled_counts(rna_engine* e, int32_t* counts_host, int n) {
  if (!e || !counts_host || n <= 0) return RNA_EINVAL;
  AstarDevice& a = e->astar;
  if (!a.g[0] || !a.last_queries || n != a.last_n) return fail(e, RNA_ESTATE, "no resident A* batch of that size");
  RNA_HIP(e, hipSetDevice(e->device));
  int32_t* d_counts = nullptr;
  int rc = sync_all(e);
  if (rc != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &d_counts, (size_t)n)) != RNA_OK) return rc;
  if (a.mode != 0)
    rc = tsa_settled(e, a.last_slot, a.last_queries, a.last_results, n, d_counts);
  else
    hipLaunchKernelGGL(astar_settled_kernel, dim3(n), dim3(1024), 0, e->stream, e->geom.size[0], e->geom.size[1],
                       a.last_queries, a.last_results, reinterpret_cast<const unsigned*>(a.g[a.last_slot]) + 64, a.field_stride, d_counts);
  if (rc != RNA_OK) { dev_free(&d_counts); return rc; }
  hipError_t st = hipGetLastError();
  if (st == hipSuccess) st = hipMemcpyAsync(counts_host, d_counts, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  dev_free(&d_counts);
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  return RNA_OK;
}
