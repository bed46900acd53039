// astar.hip -- batched grid A* on gfx950 (one workgroup per query) + the reference's waypoint-graph A*.
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
//     the next bucket, so three queues per query suffice: cur (this round), nxt (same bucket, next
//     round), far (next bucket).
//   * g lives in HBM (int32 per cell, one field per concurrent query) and is relaxed with
//     atomicMin; queue entries carry (cell, g) so stale entries are dropped when popped;
//   * queue tails are LDS counters; one 1024-thread workgroup (16 wavefronts) per query keeps the
//     whole round -- pop, 8 neighbour relaxations, push -- behind two workgroup barriers;
//   * pruning: once the goal has a finite g, candidates with f > g(goal) are dropped;
//   * the path is rebuilt by one wavefront: lane k tests neighbour k, ballot + ffs picks the
//     lowest-index optimal predecessor.
#include "engine.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

using namespace rna;

namespace {

constexpr int ASTAR_THREADS = 1024;
constexpr int COST_S = 1000, COST_D = 1414;
constexpr int INF = 0x7fffffff;

__device__ __forceinline__ int octile(int i, int j, int gi, int gj) {
  const int dx = abs(i - gi), dy = abs(j - gj);
  const int mx = dx > dy ? dx : dy, mn = dx > dy ? dy : dx;
  return COST_S * mx + (COST_D - COST_S) * mn;
}

__global__ void astar_fill_kernel(int4* __restrict__ p, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int4 v = make_int4(INF, INF, INF, INF);
  for (; i < n4; i += stride) p[i] = v;
}

__global__ void __launch_bounds__(ASTAR_THREADS)
astar_search_kernel(const uint8_t* __restrict__ nbr, int rows, int cols, const rna_astar_query* __restrict__ queries,
                    int32_t* __restrict__ g_all, size_t g_stride, int2* __restrict__ queues, int queue_cap,
                    int bucket_width, int32_t* __restrict__ paths, int max_path_len,
                    rna_astar_result* __restrict__ results) {
  __shared__ int s_cur_n, s_nxt_n, s_far_n, s_best, s_overflow, s_expanded, s_state, s_bucket;
  __shared__ int s_sel[3];  // which physical queue plays cur / nxt / far
  const int q = blockIdx.x;
  const int tid = threadIdx.x;
  const rna_astar_query qu = queries[q];
  int32_t* g = g_all + (size_t)q * g_stride;
  int2* qbase = queues + (size_t)q * 3 * queue_cap;
  const int ncell = rows * cols;

  const bool valid = qu.start >= 0 && qu.goal >= 0 && qu.start < ncell && qu.goal < ncell;
  if (!valid) {
    if (tid == 0) results[q] = rna_astar_result{2, 0, INF, 0};
    return;
  }
  const int gi = qu.goal % rows, gj = qu.goal / rows;
  const int off[8] = {-1 - rows, -rows, 1 - rows, -1, 1, rows - 1, rows, rows + 1};

  if (tid == 0) {
    s_sel[0] = 0; s_sel[1] = 1; s_sel[2] = 2;
    s_cur_n = 1; s_nxt_n = 0; s_far_n = 0;
    s_best = INF; s_overflow = 0; s_expanded = 0; s_state = 0;
    const int f0 = octile(qu.start % rows, qu.start / rows, gi, gj);
    s_bucket = f0 / bucket_width;
    __hip_atomic_store(&g[qu.start], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    qbase[0] = make_int2(qu.start, 0);
  }
  __syncthreads();

  int my_expanded = 0;
  for (;;) {
    const int n = s_cur_n;
    const int2* cur = qbase + (size_t)s_sel[0] * queue_cap;
    int2* nxt = qbase + (size_t)s_sel[1] * queue_cap;
    int2* far = qbase + (size_t)s_sel[2] * queue_cap;
    const int best = s_best;
    const long long bucket_end = ((long long)s_bucket + 1) * bucket_width;  // exclusive
    for (int e = tid; e < n; e += ASTAR_THREADS) {
      const int2 ent = cur[e];
      const int cell = ent.x, gv = ent.y;
      // stale entry: the cell was improved after this entry was queued
      if (__hip_atomic_load(&g[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gv) continue;
      const int ci = cell % rows, cj = cell / rows;
      if (gv + octile(ci, cj, gi, gj) > best) continue;
      ++my_expanded;
      if (cell == qu.goal) { atomicMin(&s_best, gv); continue; }  // (also covers start == goal)
      const unsigned m = nbr[cell];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!((m >> k) & 1u)) continue;
        const int nc = cell + off[k];
        const int ng = gv + ((k == 1 || k == 3 || k == 4 || k == 6) ? COST_S : COST_D);
        const int old = atomicMin(&g[nc], ng);
        if (ng >= old) continue;
        const int ni = ci + ((k == 0 || k == 3 || k == 5) ? -1 : ((k == 2 || k == 4 || k == 7) ? 1 : 0));
        const int nj = cj + (k < 3 ? -1 : (k > 4 ? 1 : 0));
        const int fn = ng + octile(ni, nj, gi, gj);
        if (fn > s_best) continue;
        if (nc == qu.goal) atomicMin(&s_best, ng);
        if (fn < bucket_end) {
          const int pos = atomicAdd(&s_nxt_n, 1);
          if (pos < queue_cap) nxt[pos] = make_int2(nc, ng); else s_overflow = 1;
        } else {
          const int pos = atomicAdd(&s_far_n, 1);
          if (pos < queue_cap) far[pos] = make_int2(nc, ng); else s_overflow = 1;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (s_overflow) {
        s_state = 3;
      } else if (s_nxt_n > 0) {            // same bucket, next round
        const int t = s_sel[0]; s_sel[0] = s_sel[1]; s_sel[1] = t;
        s_cur_n = s_nxt_n; s_nxt_n = 0;
      } else {
        // bucket k is at its fixed point: every cell with f < (k+1)*B has its exact g.
        const long long done_below = ((long long)s_bucket + 1) * bucket_width;
        if (s_best != INF && (long long)s_best < done_below) s_state = 1;       // goal settled, ties included
        else if (s_far_n == 0) s_state = (s_best != INF) ? 1 : 2;              // frontier exhausted
        else {
          const int t = s_sel[0]; s_sel[0] = s_sel[2]; s_sel[2] = t;
          s_cur_n = s_far_n; s_far_n = 0; s_bucket += 1;
        }
      }
    }
    __syncthreads();
    if (s_state != 0) break;
  }
  atomicAdd(&s_expanded, my_expanded);
  __syncthreads();

  const int state = s_state;
  if (state != 1) {
    if (tid == 0) results[q] = rna_astar_result{state == 3 ? (int)RNA_ECAPACITY : 1, 0, INF, s_expanded};
    return;
  }

  // ---- canonical backtrace by the first wavefront; reversed path staged in queue 0 ----
  int* rev = reinterpret_cast<int*>(qbase);
  const int rev_cap = 3 * queue_cap * 2;
  __shared__ int s_len;
  if (tid < 64) {
    int c = qu.goal;
    int len = 0;
    bool ok = true;
    for (;;) {
      if (tid == 0 && len < rev_cap) rev[len] = c;
      ++len;
      if (c == qu.start) break;
      if (len > ncell) { ok = false; break; }
      const unsigned m = nbr[c];
      const int gc = __hip_atomic_load(&g[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool hit = false;
      int nc = -1;
      if (tid < 8 && ((m >> tid) & 1u)) {
        nc = c + off[tid];
        const int gn = __hip_atomic_load(&g[nc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int w = (tid == 1 || tid == 3 || tid == 4 || tid == 6) ? COST_S : COST_D;
        hit = (gn != INF) && (gn + w == gc);
      }
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
    if (tid == 0) results[q] = rna_astar_result{1, 0, INF, s_expanded};
    return;
  }
  if (len > max_path_len || len > rev_cap) {
    if (tid == 0) results[q] = rna_astar_result{3, len, s_best, s_expanded};
    return;
  }
  int32_t* path = paths + (size_t)q * max_path_len;
  for (int k = tid; k < len; k += ASTAR_THREADS) path[k] = rev[len - 1 - k];
  if (tid == 0) results[q] = rna_astar_result{0, len, s_best, s_expanded};
}

// |{n : g(n) + h(n) <= f*}| per query, from the resident g fields (measurement utility)
__global__ void astar_settled_kernel(int rows, int cols, const rna_astar_query* __restrict__ queries,
                                     const rna_astar_result* __restrict__ results, const int32_t* __restrict__ g_all,
                                     size_t g_stride, int32_t* __restrict__ counts) {
  __shared__ int s_cnt;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const rna_astar_result r = results[q];
  int cnt = 0;
  if (r.status == 0 || r.status == 3) {
    const int goal = queries[q].goal;
    const int gi = goal % rows, gj = goal / rows;
    const int32_t* g = g_all + (size_t)q * g_stride;
    const int ncell = rows * cols;
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
      const int gv = g[c];
      if (gv != INF && gv + octile(c % rows, c / rows, gi, gj) <= r.cost) ++cnt;
    }
  }
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (threadIdx.x == 0) counts[q] = s_cnt;
}

int ensure_config(rna_engine* e) {
  AstarDevice& a = e->astar;
  if (a.g) return RNA_OK;
  if (a.max_queries <= 0) a.max_queries = 256;
  if (a.queue_cap <= 0) {
    // a bucket's queue holds the cells whose f falls into one bucket_width band of the search
    // ellipse (plus duplicates); 64 x (rows + cols) entries is a wide margin, checked at run time
    long long c = 64LL * (e->geom.size[0] + e->geom.size[1]);
    if (c < 65536) c = 65536;
    if (c > (1 << 22)) c = 1 << 22;
    a.queue_cap = (int)c;
  }
  // shrink the concurrent-query count until the g fields fit in free HBM (leave 25 % headroom)
  size_t free_b = 0, total_b = 0;
  RNA_HIP(e, hipMemGetInfo(&free_b, &total_b));
  const size_t per_query = e->ncell * sizeof(int32_t) + (size_t)3 * a.queue_cap * sizeof(int2);
  while (a.max_queries > 1 && (double)per_query * a.max_queries > 0.75 * (double)free_b) a.max_queries /= 2;
  int rc;
  if ((rc = dev_alloc(e, &a.g, e->ncell * (size_t)a.max_queries + 4)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &a.queues, (size_t)3 * a.queue_cap * a.max_queries)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &a.queries_dev, (size_t)a.max_queries)) != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &a.results_dev, (size_t)a.max_queries)) != RNA_OK) return rc;
  return RNA_OK;
}

int launch_chunk(rna_engine* e, const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len,
                 rna_astar_result* res_dev) {
  AstarDevice& a = e->astar;
  {
    KernelTimer kt(e, RNA_K_ASTAR_INIT);
    // g fields of the n concurrent queries are contiguous; the allocation carries 4 spare words so
    // the last int4 store may run past the used part
    const size_t n4 = (e->ncell * (size_t)n + 3) / 4;
    hipLaunchKernelGGL(astar_fill_kernel, dim3(8192), dim3(256), 0, e->stream, reinterpret_cast<int4*>(a.g), n4);
    RNA_HIP(e, hipGetLastError());
  }
  {
    KernelTimer kt(e, RNA_K_ASTAR_SEARCH);
    hipLaunchKernelGGL(astar_search_kernel, dim3(n), dim3(ASTAR_THREADS), 0, e->stream, e->nbr, e->geom.size[0],
                       e->geom.size[1], q_dev, a.g, e->ncell, a.queues, a.queue_cap, a.bucket_width, paths_dev,
                       max_len, res_dev);
    RNA_HIP(e, hipGetLastError());
  }
  a.last_queries = q_dev;
  a.last_results = res_dev;
  a.last_n = n;
  return RNA_OK;
}

}  // namespace

namespace rna {
int astar_release(rna_engine* e) {
  AstarDevice& a = e->astar;
  dev_free(&a.g); dev_free(&a.queues); dev_free(&a.queries_dev); dev_free(&a.results_dev); dev_free(&a.paths_dev);
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
  if (max_queries) a.max_queries = max_queries;
  if (queue_capacity) a.queue_cap = queue_capacity;
  if (bucket_width) a.bucket_width = bucket_width;
  return RNA_OK;
}

extern "C" int rna_astar_batch_device(rna_engine* e, const rna_astar_query* queries, int n, int32_t* paths,
                                      int max_path_len, rna_astar_result* results) {
  if (!e || n < 0 || max_path_len <= 0 || (n > 0 && (!queries || !paths || !results))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  if (e->geom.start[0] != 0 || e->geom.start[1] != 0)
    return fail(e, RNA_ESTATE, "grid A* needs startIndex (0,0): buffer adjacency must be map adjacency");
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = ensure_config(e);
  if (rc != RNA_OK) return rc;
  if ((rc = map_prepare_nbr(e)) != RNA_OK) return rc;
  const int chunk = e->astar.max_queries;
  for (int o = 0; o < n; o += chunk) {
    const int m = std::min(chunk, n - o);
    rc = launch_chunk(e, queries + o, m, paths + (size_t)o * max_path_len, max_path_len, results + o);
    if (rc != RNA_OK) return rc;
  }
  return RNA_OK;
}

extern "C" int rna_astar_batch(rna_engine* e, const rna_astar_query* queries_host, int n, int32_t* paths_host,
                               int max_path_len, rna_astar_result* results_host) {
  if (!e || n < 0 || max_path_len <= 0 || (n > 0 && (!queries_host || !paths_host || !results_host))) return RNA_EINVAL;
  if (n == 0) return RNA_OK;
  if (e->geom.start[0] != 0 || e->geom.start[1] != 0)
    return fail(e, RNA_ESTATE, "grid A* needs startIndex (0,0): buffer adjacency must be map adjacency");
  RNA_HIP(e, hipSetDevice(e->device));
  int rc = ensure_config(e);
  if (rc != RNA_OK) return rc;
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
    rc = launch_chunk(e, a.queries_dev, m, a.paths_dev, max_path_len, a.results_dev);
    if (rc != RNA_OK) return rc;
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
  if (!a.g || !a.last_queries || n != a.last_n) return fail(e, RNA_ESTATE, "no resident A* batch of that size");
  RNA_HIP(e, hipSetDevice(e->device));
  int32_t* d_counts = nullptr;
  int rc = dev_alloc(e, &d_counts, (size_t)n);
  if (rc != RNA_OK) return rc;
  hipLaunchKernelGGL(astar_settled_kernel, dim3(n), dim3(1024), 0, e->stream, e->geom.size[0], e->geom.size[1],
                     a.last_queries, a.last_results, a.g, e->ncell, d_counts);
  hipError_t st = hipGetLastError();
  if (st == hipSuccess) st = hipMemcpyAsync(counts_host, d_counts, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  dev_free(&d_counts);
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  return RNA_OK;
}
