// astar.hip -- host side of the batched grid A*: configuration (pipeline stages, pages per query, what fits HBM), the
// choice of a stage for each batch, chunking, the C ABI.  The search itself is astar_tile.hip (one workgroup per query,
// register-resident 64 x 16 tiles, an open list of tiles in LDS).
//
// Grid A* (DESIGN.md "Grid A* contract", restated by oracle/astar.c).  The reference has no grid
// search (AStarPlanner::makePlan walks a 9-vertex graph, mc/src/astar_planner.cpp:63-127); the
// contract is: 8-connected, integer costs 1000/1414, octile heuristic, no corner cutting, every cell
// with f <= f* settled, canonical predecessor = optimal neighbour with the lowest linear index.
// Because the path is a function of the exact distance field only, any label-correcting schedule
// that converges gives the CPU oracle's path bit for bit.
// Maps of more than 65 536 tiles (8192 x 8192 cells) are refused with RNA_EINVAL: tile numbers are 16 bits in the
// search's queue entries.  (Rounds 1-2 kept a cell-granular frontier kernel as a fallback for such maps; it did not
// handle moved maps, no configuration reached it, and it was removed in round 3.)
#include "engine.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

using namespace rna;

namespace {

constexpr int COST_D = 1414;

// developer switch RNA_HOST_TRACE=1: host microseconds inside the segments of a pipelined batch launch, printed when the search
// state is released (where the host's share of a pass goes)
struct HostTrace {
  bool on = getenv("RNA_HOST_TRACE") != nullptr;
  double us[8] = {};
  long n = 0, queries = 0;
  std::chrono::steady_clock::time_point t;
  void start() { if (on) t = std::chrono::steady_clock::now(); }
  void lap(int k) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    us[k] += std::chrono::duration<double, std::micro>(now - t).count();
    t = now;
  }
  void dump() {
    if (!on || n == 0) return;
    fprintf(stderr, "[host trace] %ld pipelined launches, us each: wait for a free stage %.1f (%.2f hipEventQuery calls) | ring entry's old batch %.1f | "
            "tsa_launch (events, snapshot, launch order, search, flag copy) %.1f | done event + bookkeeping %.1f\n", n, us[0] / n, (double)queries / n,
            us[1] / n, us[2] / n, us[3] / n);
    for (double& v : us) v = 0;
    n = 0; queries = 0;
  }
};
HostTrace g_trace;
// (RNA_HOST_TRACE) a stage's cycle in GPU time: event A when the stage's stream reaches the new batch (it is idle: at enqueue),
// C when the search and what follows it on the stream have ended; host clock: enqueue and "noticed free".  (With a third event
// where the search's last wait ends -- inside tsa_launch, a temporary hook -- round 6 measured 4.46 ms from enqueue until the
// search may start, 22.40 ms of search, 27.89 ms until the host has noticed: profiles/r06_host_trace.txt.)
struct StageTrace {
  hipEvent_t a[32] = {}, c[32] = {};
  std::chrono::steady_clock::time_point t_enq[32];
  bool armed[32] = {};
  double gpu_ms = 0, host_busy_ms = 0;
  long n = 0;
};
StageTrace g_stage;
void stage_trace_dump() {
  if (g_stage.n) fprintf(stderr, "[host trace] a stage's cycle, ms: from enqueue to the end of the search and what follows it on the stream %.2f (the search itself: roofline.avg_launch_ms) | "
                         "host: enqueue -> noticed free %.2f (%ld cycles)\n", g_stage.gpu_ms / g_stage.n, g_stage.host_busy_ms / g_stage.n, g_stage.n);
  g_stage.gpu_ms = g_stage.host_busy_ms = 0; g_stage.n = 0;
}

// allocation of the configuration ensure_config settled on; everything is released again on failure
static int alloc_stages_impl(rna_engine* e) {
  AstarDevice& a = e->astar;
  int rc;
  for (int d = 0; d < a.depth; ++d) {
    {
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
      RNA_HIP(e, hipEventCreateWithFlags(&a.snap_done[d], hipEventDisableTiming));
    }
    a.snap_pending[d] = false;
    a.busy[d] = false;
    a.stage_seq[d] = 0;
  }
  if (a.depth > 1) {
    RNA_HIP(e, hipEventCreateWithFlags(&a.ev_init, hipEventDisableTiming));
    RNA_HIP(e, hipEventCreateWithFlags(&a.ev_prep, hipEventDisableTiming));
    a.ring_n = 2 * a.depth;
    a.ring_stride = tsa_ring_bytes(e, a.max_queries);
    if ((rc = dev_alloc(e, &a.ring_mem, a.ring_stride * (size_t)a.ring_n)) != RNA_OK) return rc;
    RNA_HIP(e, hipMemsetAsync(a.ring_mem, 0, a.ring_stride * (size_t)a.ring_n, e->stream));
    for (int r = 0; r < a.ring_n; ++r) {
      RNA_HIP(e, hipEventCreateWithFlags(&a.ring_free[r], hipEventDisableTiming));
      a.ring_used[r] = false;
    }
  }
  if (a.g_retry[0]) {
    RNA_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&a.retry_flag), AstarDevice::MAX_DEPTH * sizeof(int), hipHostMallocCoherent));
    for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d) a.retry_flag[d] = 0;
  }
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
  if (!tsa_supported(e)) return fail(e, RNA_EINVAL, "grid A*: more than 65 536 tiles of 64 x 16 cells (8192 x 8192 cells): tile numbers are 16 bits");
  if (a.depth < 1) a.depth = 1;
  if (a.depth > AstarDevice::MAX_DEPTH) a.depth = AstarDevice::MAX_DEPTH;
  // fit into free HBM (25 % headroom): fewer pages per query (half a map's worth), fewer pipeline stages, fewer pages
  // still -- a search that needs more than its share is searched again on a full-size retry slot (astar_tile.hip: eight
  // at a time, as many passes as it takes; status 5 is transient and never final) --, then fewer concurrent queries
  size_t free_b = 0, total_b = 0;
  RNA_HIP(e, hipMemGetInfo(&free_b, &total_b));
  const int ntile = tsa_tiles(e);
  a.page_cap = ntile;
  if (a.page_cap_request > 0 && a.page_cap_request < ntile) a.page_cap = a.page_cap_request;
  if (const char* c = getenv("RNA_ASTAR_PAGE_CAP")) { const int v = atoi(c); if (v > 0 && v < ntile) a.page_cap = v; }
  auto stage_bytes = [&]() -> double {
    return (double)tsa_pool_bytes(a.max_queries, a.page_cap) + (double)tsa_aux_bytes(e, a.max_queries, a.page_cap) +
           (double)a.rev_cap * 4.0 * a.max_queries +
           (a.page_cap < ntile ? (double)tsa_retry_pool_bytes(e) + (double)tsa_retry_aux_bytes(e) : 0.0);
  };
  // (tile kernel) half a map's worth of pages per query first -- searches touch a few per cent of the map, and twelve
  // stages of 17.8 GB each at 4096^2 x 256 queries sit right at the budget --, then fewer stages, then fewer pages
  if (a.page_cap > 64 && a.page_cap > ntile / 2 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.page_cap = (ntile + 1) / 2;
  while (a.depth > 1 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.depth -= 1;
  while (a.page_cap > 64 && a.page_cap > ntile / 8 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.page_cap /= 2;
  while (a.max_queries > 1 && stage_bytes() * a.depth > 0.75 * (double)free_b) a.max_queries /= 2;
  // HBM may have gone to another process between hipMemGetInfo and here: step down and try again
  for (;;) {
    const int rc = alloc_stages(e);
    if (rc != RNA_ENOMEM) return rc;
    (void)hipGetLastError();   // the failed hipMalloc must not surface at the next launch check
    if (a.page_cap > 64 && a.page_cap > ntile / 2) a.page_cap = (ntile + 1) / 2;
    else if (a.depth > 1) a.depth = a.depth > 2 ? a.depth * 3 / 4 : 1;
    else if (a.page_cap > 64 && a.page_cap > ntile / 8) a.page_cap /= 2;
    else if (a.max_queries > 1) a.max_queries /= 2;
    else return rc;
  }
}

// The stage's search has finished (its `done` event is complete): if searches of that batch ran out of pages, their
// second pass goes out now and the stage stays busy (returns true).
static bool stage_settled(rna_engine* e, int d, int* rc) {
  AstarDevice& a = e->astar;
  *rc = RNA_OK;
  if (!a.retry_armed[d]) return true;
  a.retry_armed[d] = false;
  if (!a.retry_flag || a.retry_flag[d] == 0) return true;
  const int count = a.retry_flag[d];
  a.retry_flag[d] = 0;
  a.last_retried[d] = count;
  hipStream_t st = a.depth > 1 ? a.side[d] : e->stream;
  *rc = tsa_retry_launch(e, d, st, count);
  if (*rc != RNA_OK) return true;
  if (a.depth > 1 && hipEventRecord(a.done[d], st) != hipSuccess) { *rc = fail(e, RNA_EHIP, "hipEventRecord"); return true; }
  return false;
}

// One batch (<= max_queries): mask snapshot + launch order on the side stream (it reads the neighbour masks, which the
// next map update may overwrite), then the search on a pipeline stage's own stream.
int launch_chunk(rna_engine* e, const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len,
                 rna_astar_result* res_dev, int* slot_out) {
  AstarDevice& a = e->astar;
  // Which stage: one that IS free, the one that has been free the longest -- not simply the next in turn.  A batch lasts
  // as long as its longest search; taking the stages strictly in turn, a new batch waits for the one slow batch while
  // the stages behind it have long finished (they sat idle 40 % of the time at thirteen stages).  If none is free the
  // HOST waits for one (a few hundred microseconds at the bench's load): what the batch needs from the engine stream
  // is enqueued before the wait, so the search starts the moment a stage frees up -- enqueued blindly behind "the
  // oldest", as in round 2, a stage's stream sat idle 7 ms between two searches of 23 ms.
  int slot = 0;
  g_trace.start();
  if (a.depth > 1) {
    static const bool nowait = getenv("RNA_ASTAR_NOWAIT") != nullptr;   // developer knob: the round-2 behaviour
    // A stage known to be free is taken without asking anybody; otherwise the stages are asked in the order of their launches
    // (they end roughly in that order), at most RNA_ASTAR_QUERIES_PER_LOOK (3) per look, and the first one found complete is
    // taken.  (Round 6 measured this loop, RNA_HOST_TRACE=1: the host spends 1.38 ms of a 1.57 ms pass here -- not in the
    // queries, which cost 5 us each, but because every stage IS taken: a stage's cycle is 4.5 ms from enqueue until its search
    // may start (the pass's map-update chain, snapshot and launch order), 22.4 ms of search and 1 ms until the host has
    // noticed; a completion flag written by the stream instead of the event changes nothing.)
    static const int per_look = getenv("RNA_ASTAR_QUERIES_PER_LOOK") ? std::max(1, atoi(getenv("RNA_ASTAR_QUERIES_PER_LOOK"))) : 3;   // developer knob
    for (int spin = 0;; ++spin) {
      int best = -1, oldest = 0;
      for (int d = 0; d < a.depth; ++d) {
        if (a.stage_seq[d] < a.stage_seq[oldest]) oldest = d;
        if (!a.busy[d] && (best < 0 || a.stage_seq[d] < a.stage_seq[best])) best = d;
      }
      if (best < 0) {
        // the busy stages by age; every eighth look asks all of them (a batch with one very long search must not hide the others)
        int order[AstarDevice::MAX_DEPTH], nb = 0;
        for (int d = 0; d < a.depth; ++d) if (a.busy[d]) order[nb++] = d;
        std::sort(order, order + nb, [&](int x, int y) { return a.stage_seq[x] < a.stage_seq[y]; });
        const int ask = (spin & 7) == 7 ? nb : std::min(nb, per_look);
        for (int k = 0; k < ask && best < 0; ++k) {
          const int d = order[k];
          g_trace.queries += 1;
          const hipError_t q = hipEventQuery(a.done[d]);
          if (q == hipSuccess) {
            int rc = RNA_OK;
            if (stage_settled(e, d, &rc)) { a.busy[d] = false; best = d; }
            if (g_trace.on && g_stage.armed[d] && best == d) {
              float f = 0;
              if (hipEventElapsedTime(&f, g_stage.a[d], g_stage.c[d]) == hipSuccess) {
                g_stage.gpu_ms += f; g_stage.n += 1;
                g_stage.host_busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_stage.t_enq[d]).count();
              }
              g_stage.armed[d] = false;
            }
            if (rc != RNA_OK) return rc;
          } else {
            (void)hipGetLastError();   // hipErrorNotReady is not an error
          }
        }
      }
      if (best >= 0) { slot = best; break; }
      if (nowait) { slot = oldest; break; }
      // (no stage free: the host SLEEPS between looks -- with thirteen launches in flight a stage that frees up 50 us before
      // the host notices costs nothing, and eight ranks of one node share the host's cores; round 5 yielded 64 times and then
      // slept 20 us, which kept a core busy: config.host_cores_used 0.98)
      static const int wait_us = getenv("RNA_ASTAR_WAIT_US") ? atoi(getenv("RNA_ASTAR_WAIT_US")) : 50;   // developer knob
      if (spin >= 2 && wait_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
      else std::this_thread::yield();
    }
    g_trace.lap(0);
    // the ring entry this launch writes: the batch that read it 2 x depth launches ago is over and settled
    const unsigned long long prev = a.launches + 1 >= (unsigned long long)a.ring_n ? a.launches + 1 - (unsigned long long)a.ring_n : 0;
    if (prev > 0)
      for (int d = 0; d < a.depth; ++d)
        if (a.busy[d] && a.stage_seq[d] == prev) {
          int rc = RNA_OK;
          do { RNA_HIP(e, hipEventSynchronize(a.done[d])); } while (!stage_settled(e, d, &rc) && rc == RNA_OK);
          if (rc != RNA_OK) return rc;
          a.busy[d] = false;
        }
  }
  g_trace.lap(1);
  if (a.depth == 1 && a.retry_armed[0]) {   // one stream: the batch before this one has to be complete before its slots are reused
    RNA_HIP(e, hipStreamSynchronize(e->stream));
    int rc = RNA_OK;
    (void)stage_settled(e, 0, &rc);
    if (rc != RNA_OK) return rc;
  }
  hipStream_t search_stream = a.depth > 1 ? a.side[slot] : e->stream;
  if (a.depth > 1 && a.busy[slot]) RNA_HIP(e, hipStreamWaitEvent(search_stream, a.done[slot], 0));   // (RNA_ASTAR_NOWAIT only; the same stream anyway)
  if (g_trace.on && a.depth > 1) {
    if (!g_stage.a[slot]) { (void)hipEventCreate(&g_stage.a[slot]); (void)hipEventCreate(&g_stage.c[slot]); }
    (void)hipEventRecord(g_stage.a[slot], search_stream);
    g_stage.t_enq[slot] = std::chrono::steady_clock::now();
  }
  {
    int rc = tsa_launch(e, slot, e->stream, search_stream, a.depth > 1 ? a.ev_init : nullptr, q_dev, n, paths_dev, max_len, res_dev);
    if (rc != RNA_OK) return rc;
  }
  if (g_trace.on && a.depth > 1) { (void)hipEventRecord(g_stage.c[slot], search_stream); g_stage.armed[slot] = true; }
  g_trace.lap(2);
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
  g_trace.lap(3);
  if (a.depth > 1) g_trace.n += 1;
  return RNA_OK;
}

}  // namespace

namespace rna {
#ifdef RNA_TSA_STATS
void tsa_stats_dump();
#endif
// every stream is idle (sync_all): second passes that are due go out and are waited for
int astar_settle(rna_engine* e) {
  AstarDevice& a = e->astar;
  if (a.depth == 1 && a.retry_armed[0]) {
    int rc = RNA_OK;
    if (!stage_settled(e, 0, &rc) && rc == RNA_OK) RNA_HIP(e, hipStreamSynchronize(e->stream));
    return rc;
  }
  for (int d = 0; d < a.depth && d < AstarDevice::MAX_DEPTH; ++d) {
    if (!a.busy[d]) continue;
    int rc = RNA_OK;
    while (!stage_settled(e, d, &rc) && rc == RNA_OK) RNA_HIP(e, hipStreamSynchronize(a.side[d]));
    if (rc != RNA_OK) return rc;
    a.busy[d] = false;
  }
  return RNA_OK;
}
int astar_release(rna_engine* e) {
  AstarDevice& a = e->astar;
  (void)sync_all(e);
  stage_trace_dump();
  g_trace.dump();
#ifdef RNA_TSA_STATS
  tsa_stats_dump();
#endif
  for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d) {
    dev_free(&a.g[d]); dev_free(&a.rev[d]);
    if (a.tsa_aux[d]) { (void)hipFree(a.tsa_aux[d]); a.tsa_aux[d] = nullptr; }
    dev_free(&a.g_retry[d]);
    if (a.tsa_aux_retry[d]) { (void)hipFree(a.tsa_aux_retry[d]); a.tsa_aux_retry[d] = nullptr; }
    if (a.side[d]) { (void)hipStreamDestroy(a.side[d]); a.side[d] = nullptr; }
    if (a.done[d]) { (void)hipEventDestroy(a.done[d]); a.done[d] = nullptr; }
    if (a.snap_done[d]) { (void)hipEventDestroy(a.snap_done[d]); a.snap_done[d] = nullptr; }
    a.snap_pending[d] = false;
    a.busy[d] = false;
  }
  if (a.ev_init) { (void)hipEventDestroy(a.ev_init); a.ev_init = nullptr; }
  if (a.ev_prep) { (void)hipEventDestroy(a.ev_prep); a.ev_prep = nullptr; }
  for (int r = 0; r < AstarDevice::MAX_RING; ++r) {
    if (a.ring_free[r]) { (void)hipEventDestroy(a.ring_free[r]); a.ring_free[r] = nullptr; }
    a.ring_used[r] = false;
  }
  dev_free(&a.ring_mem);
  a.ring_n = 0;
  if (a.retry_flag) { (void)hipHostFree(a.retry_flag); a.retry_flag = nullptr; }
  for (int d = 0; d < AstarDevice::MAX_DEPTH; ++d) a.retry_armed[d] = false;
  dev_free(&a.queries_dev); dev_free(&a.results_dev); dev_free(&a.paths_dev);
  a.paths_cap = 0;
  a.last_queries = nullptr; a.last_results = nullptr; a.last_n = 0;
  return RNA_OK;
}
}  // namespace rna

extern "C" int rna_astar_configure(rna_engine* e, int max_queries, int queue_capacity, int bucket_width) {
  if (!e || max_queries < 0 || queue_capacity < 0 || bucket_width < 0) return RNA_EINVAL;
  if (bucket_width != 0 && bucket_width < 2 * COST_D) return fail(e, RNA_EINVAL, "bucket_width must be >= 2828");
  RNA_ENTER(e);
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
  RNA_ENTER(e);
  if (depth != e->astar.depth) {
    astar_release(e);
    e->astar.depth = depth;
  }
  char advice[448];   // (the longest advice line is about 330 characters)
  if (rna_hw_queue_advice(depth, advice, sizeof(advice))) e->err = advice;   // (still RNA_OK: correct, only slower)
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
  RNA_ENTER(e);
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
  RNA_ENTER_NOJOIN(e);   // (reads the neighbour masks only; map_prepare_nbr joins if it has to rebuild them)
  int rc = ensure_config(e);
  if (rc != RNA_OK) return rc;
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
  RNA_ENTER(e);
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
    int slot = 0;
    rc = launch_chunk(e, a.queries_dev, m, a.paths_dev, max_path_len, a.results_dev, &slot);
    if (rc != RNA_OK) return rc;
    if (a.depth > 1) {
      int src = RNA_OK;
      do { RNA_HIP(e, hipEventSynchronize(a.done[slot])); } while (!stage_settled(e, slot, &src) && src == RNA_OK);
      if (src != RNA_OK) return src;
      a.busy[slot] = false;
    } else if (a.retry_armed[0]) {
      RNA_HIP(e, hipStreamSynchronize(e->stream));
      int src = RNA_OK;
      (void)stage_settled(e, 0, &src);   // (a second pass, if one is due, is on the stream ahead of the copies below)
      if (src != RNA_OK) return src;
    }
    RNA_HIP(e, hipMemcpyAsync(results_host + o, a.results_dev, (size_t)m * sizeof(rna_astar_result),
                              hipMemcpyDeviceToHost, e->stream));
    RNA_HIP(e, hipMemcpyAsync(paths_host + (size_t)o * max_path_len, a.paths_dev,
                              (size_t)m * max_path_len * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    RNA_HIP(e, hipStreamSynchronize(e->stream));
  }
  return RNA_OK;
}

extern "C" int rna_astar_download_nbr_mask(rna_engine* e, uint8_t* host, size_t n) {
  if (!e || !host || n != e->ncell) return RNA_EINVAL;
  RNA_ENTER(e);
  int rc = map_prepare_nbr(e);
  if (rc != RNA_OK) return rc;
  RNA_HIP(e, hipMemcpyAsync(host, e->nbr, n, hipMemcpyDeviceToHost, e->stream));
  RNA_HIP(e, hipStreamSynchronize(e->stream));
  return RNA_OK;
}

extern "C" int rna_astar_job_counters(rna_engine* e, uint64_t* counters_host, int reset) {
  if (!e || !counters_host) return RNA_EINVAL;
  AstarDevice& a = e->astar;
  if (!a.g[0]) return fail(e, RNA_ESTATE, "no A* batch has run yet");
  RNA_ENTER(e);
  int rc = sync_all(e);
  if (rc != RNA_OK) return rc;
  unsigned long long all[16];
  if ((rc = tsa_counters_read(e, all, reset != 0)) != RNA_OK) return rc;
  for (int k = 0; k < 8; ++k) counters_host[k] = all[k];
#ifdef RNA_TSA_IDLE   /* developer build: wavefront life / idle ticks (shader clock) in the reserved word and the bucket count's */
  counters_host[6] = all[8];
  counters_host[7] = all[9];
#endif
  return RNA_OK;
}

extern "C" int rna_astar_settled_counts(rna_engine* e, int32_t* counts_host, int n) {
  if (!e || !counts_host || n <= 0) return RNA_EINVAL;
  AstarDevice& a = e->astar;
  if (!a.g[0] || !a.last_queries || n != a.last_n) return fail(e, RNA_ESTATE, "no resident A* batch of that size");
  RNA_ENTER(e);
  int32_t* d_counts = nullptr;
  int rc = sync_all(e);
  if (rc != RNA_OK) return rc;
  if ((rc = dev_alloc(e, &d_counts, (size_t)n)) != RNA_OK) return rc;
  rc = tsa_settled(e, a.last_slot, a.last_queries, a.last_results, n, d_counts);
  if (rc != RNA_OK) { dev_free(&d_counts); return rc; }
  hipError_t st = hipGetLastError();
  if (st == hipSuccess) st = hipMemcpyAsync(counts_host, d_counts, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  dev_free(&d_counts);
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  return RNA_OK;
}
