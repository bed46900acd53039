// engine.hpp -- internal state of one rna_engine (librna.so).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rna.h"
#include "gridmath.hpp"

namespace rna {

constexpr int TILE = 64;  // dirty-tracking / mask tile edge (cells)

struct HimmSlot { int cell; int head; int len; int offset; };

struct HimmScratch {
  int cap_rays = 0;
  int n_slots = 0;
  rna_ray* rays_dev = nullptr;   // staging for host-pointer calls
  int4* desc = nullptr;          // clipped start/end indices per ray
  int* ncells = nullptr;         // Bresenham cell count per ray (0 = no cells)
  int* next = nullptr;           // chain of marks per cell
  HimmSlot* slots = nullptr;     // open-addressing hash cell -> marks
  int* seqs = nullptr;           // sorted ray sequence numbers of marks, grouped per cell
  unsigned* before = nullptr;    // clears that land before mark k
  unsigned* after = nullptr;     // clears after the last mark of the cell
  int* total = nullptr;          // allocation cursor into seqs
  unsigned* mark_bitmap = nullptr;  // 1 bit per cell: cell holds >= 1 mark in the current batch
  int* pairs = nullptr;          // (tile, ray) pairs grouped by tile: the rays each 64 x 64 tile has to rasterise
  size_t pairs_cap = 0;          // ints allocated for them
  bool pairs_checked = false;    // pairs_cap is below the worst case: every batch's pair count is read back before it is used
  int* pairs_total_host = nullptr;   // pinned: the count of the batch being launched (pairs_checked only)
  int* tile_bins = nullptr;      // [3][ntile]: pair count, offset and fill cursor per tile
  int win[4] = {0, 0, 0, 0};     // owner window [i0, i1) x [j0, j1) in buffer indices; i1 == 0: whole map
};

struct VfhDevice {
  bool ready = false;
  rna_vfh_params p{};
  int n_robots = 0;
  int W = 0, H = 0, T = 0, CX = 0, CY = 0, NQ = 0, NQF = 0, NW = 0, max_speed = 0;
  // tables
  float *cell_dist = nullptr, *cell_base_mag = nullptr, *cell_dir = nullptr;  // [q] in (y outer, x inner) order
  int* range_idx = nullptr;        // [q] rint(2*dir)
  unsigned* memb = nullptr;        // [T][H][NW] sector-major membership bit masks over q
  unsigned* in_circle = nullptr;   // [max_speed+1][2][NW] cell inside right / left blocked circle
  int* min_turning_radius = nullptr;  // [max_speed+1]
  // per-robot state
  float *last_binary = nullptr, *hist = nullptr, *origin = nullptr;
  float *picked = nullptr, *last_picked = nullptr, *blocked_radius = nullptr;
  int* last_chosen_speed = nullptr;
  // staging
  rna_pose* poses_dev = nullptr;
  rna_vfh_out* out_dev = nullptr;
  double* ranges_dev = nullptr;
};

struct AstarDevice {
  int max_queries = 0;
  int queue_cap = 0;
  int bucket_width = 128000;        // f-range of one bucket (128 cells; round 6, on the bench: 72 / 96 / 128 / 160 / 200 / 256 k -> 159.8 / 160.8 / 162.2 / 159.7 / 158.5 / 159.3 k cycles/s, profiles/r06_sweep_bucket_width.txt); inside it free wavefronts take the tile with the lowest key (astar_tile.hip)
  int threads = 512;               // workgroup size of the search kernel (256 / 512 / 1024)
  // Pipelined batches: `depth` independent sets of search fields + queues, each with its own HIP
  // stream, so the tail of batch k (few long queries) overlaps the head of batch k+1.
#ifndef RNA_ASTAR_MAX_DEPTH
#define RNA_ASTAR_MAX_DEPTH 20   // (developer builds may raise it: scripts/r06_depth_hwq.sh)
#endif
  // From 22 stages on the rate falls by a fifth, with GPU_MAX_HW_QUEUES = 8 and = 24 alike (profiles/r04_sweep_depth_16_22_engine_max_24.txt,
  // r06_sweep_depth_hw_queues.txt): the process then holds 25 queues (the stages' streams, the engine stream, the side stream, the null stream).
  static constexpr int MAX_DEPTH = RNA_ASTAR_MAX_DEPTH;
  int depth = 4;
  int32_t* g[MAX_DEPTH] = {};      // frontier kernel: [max_queries][field_stride] search fields (g << 8) | mask;
                                   // tile kernel: the stage's page pool (pages, then pending bitmaps), see astar_tile.hip
  int2* queues[MAX_DEPTH] = {};    // [max_queries][3][queue_cap] (cell, g)
  int32_t* rev[MAX_DEPTH] = {};    // tile kernel: reversed-path staging per query
  int rev_cap = 16800;             // g < 2^24 at >= 1000 per step bounds a path to 16 777 cells
  int mode = 1;                    // 0 = frontier kernel (astar.hip, fallback for maps beyond 65 536 tiles), 1 = tile kernel (default)
  int page_cap = 0;                // tile kernel: 4 KiB pages per query (0 = one per tile: a search can never run out)
  int page_cap_request = 0;        // rna_astar_set_page_cap
  void* tsa_aux[MAX_DEPTH] = {};   // tile kernel: ticket, page counts, launch order, neighbour-mask snapshot, tile -> page tables
  int32_t* g_retry[MAX_DEPTH] = {};      // tile kernel, only when page_cap < tiles of the map: TSA_RETRY slots with a page per tile,
  void* tsa_aux_retry[MAX_DEPTH] = {};   // for the second pass over searches that outgrew their pages (astar_tile.hip)
  hipStream_t side[MAX_DEPTH] = {};
  hipEvent_t done[MAX_DEPTH] = {}; // search of the batch that last used this set has finished
  hipEvent_t snap_done[MAX_DEPTH] = {};   // the mask snapshot of the stage's batch (taken on the side stream) has been taken
  bool snap_pending[MAX_DEPTH] = {};
  bool busy[MAX_DEPTH] = {};
  unsigned long long stage_seq[MAX_DEPTH] = {};   // launch number of the stage's last batch (0: never used)
  hipEvent_t ev_init = nullptr;
  // Pipelined launches: the mask snapshot, the launch order and the ticket of a batch live in an entry of a ring, not in
  // the stage -- they are written on the engine's side stream (CUs the searches cannot take) as soon as the map is
  // composed, before any stage has to be free; the search itself goes to a stage that IS free (the host waits for one).
  static constexpr int MAX_RING = 2 * MAX_DEPTH;
  int ring_n = 0;
  char* ring_mem = nullptr;
  size_t ring_stride = 0;
  hipEvent_t ring_free[MAX_RING] = {};   // the search that read this entry last has finished
  bool ring_used[MAX_RING] = {};
  hipEvent_t ev_prep = nullptr;          // snapshot + launch order of the batch being launched are in place
  // Searches that outgrew their share of pages: counted by the kernel (the count is copied to pinned host memory behind
  // the search); the second pass over them is
  // launched when the host sees the count (a launch per batch "in case" cost every stage 0.2 - 4 ms: its workgroups wait
  // for CU slots like everybody else's)
  int* retry_flag = nullptr;             // [MAX_DEPTH], pinned host memory: copied from the stage's device counter behind each search
  bool retry_armed[MAX_DEPTH] = {};
  int last_retried[MAX_DEPTH] = {};      // searches of the stage's last batch that went through the second pass
  size_t last_lds[MAX_DEPTH] = {};
  alignas(8) unsigned char last_launch[MAX_DEPTH][384] = {};   // TsaLaunch of the stage's last batch
  unsigned long long launches = 0;
  int last_slot = 0;
  size_t field_stride = 0;         // words per query field incl. padding
  rna_astar_query* queries_dev = nullptr;
  rna_astar_result* results_dev = nullptr;
  int32_t* paths_dev = nullptr;
  int paths_cap = 0;               // max_queries * max_path_len currently allocated
  const rna_astar_query* last_queries = nullptr;   // device pointers of the last launched chunk
  const rna_astar_result* last_results = nullptr;
  int last_n = 0;
};

struct ProfSlot {
  double total_ms = 0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // recorded, not yet read back
};

}  // namespace rna

struct rna_engine {
  rna::Geom geom{};
  int device = 0;
  int cu_count = 256;              // compute units of the device (MI355X: 256 = 8 XCDs x 32)
  hipStream_t stream = nullptr;
  // Side work: the pipelined replan loop runs the VFH+ step and the mask snapshot + launch order of a search batch on
  // a stream of their own, so that neither sits in the engine stream's chain of map-update kernels.  Both only READ
  // the map (master layer / neighbour masks); whatever may WRITE what they read joins them first (RNA_ENTER).
  hipStream_t vfh_stream = nullptr;
  hipEvent_t ev_vfh_go = nullptr, ev_vfh_done = nullptr;
  bool vfh_pending = false;
  size_t ncell = 0;
  float* layer[RNA_NUM_LAYERS] = {nullptr, nullptr, nullptr};
  int tiles_i = 0, tiles_j = 0;
  unsigned* dirty_tiles = nullptr;   // one BYTE per TILE x TILE tile: laser changed since last compose
  unsigned* last_dirty = nullptr;    // the flags the last compose consumed (tiled mode: what this GPU has to hand on)
  int32_t* tile_list = nullptr;      // staging for rna_layer_pack_tiles / rna_layers_unpack_tiles
  int tile_list_cap = 0;
  bool laser_all_dirty = false;      // laser uploaded/filled: next compose is a whole-layer copy
  bool master_diverged = false;      // master written directly: next compose is a whole-layer copy
  uint8_t* nbr = nullptr;            // A* neighbour masks derived from master
  bool nbr_all_dirty = true;
  rna::HimmScratch himm;
  rna::VfhDevice vfh;
  rna::AstarDevice astar;
  int profiling = 0;                // 0 off, 1 every kernel slot, 2 only the slots of the A* pipeline stages' own streams
  rna::ProfSlot prof[RNA_K_COUNT];
  std::vector<hipEvent_t> free_events;   // recycled hipEvents of the profiler
  size_t pending_events = 0;
  std::string err;
};

namespace rna {

inline int fail(rna_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}

#define RNA_HIP(e, call)                                                                      \
  do {                                                                                        \
    hipError_t _st = (call);                                                                  \
    if (_st != hipSuccess)                                                                    \
      return ::rna::fail((e), RNA_EHIP, std::string(#call) + ": " + hipGetErrorString(_st)); \
  } while (0)

// Profiling (rna_profile_enable): every launch of a tracked kernel is bracketed by two hipEvents
// recorded on the engine stream WITHOUT a host sync; rna_profile_get() drains them.  The timed loop
// of bench.py therefore measures kernels live at negligible cost.
int profile_flush(rna_engine* e);

struct KernelTimer {
  rna_engine* e;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  static hipEvent_t take(rna_engine* e) {
    hipEvent_t ev = nullptr;
    if (!e->free_events.empty()) { ev = e->free_events.back(); e->free_events.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
    return ev;
  }
  hipStream_t st;
  KernelTimer(rna_engine* eng, int kid, hipStream_t stream = nullptr) : e(eng), id(kid), st(stream ? stream : eng->stream) {
    if (!wanted()) return;
    if (e->pending_events > 16384) (void)profile_flush(e);
    a = take(e);
    b = take(e);
    if (a) (void)hipEventRecord(a, st);
  }
  // mode 2: only the search and the launch-order kernel (stage / side stream).  An event record is a barrier packet of its own: on the
  // engine stream, whose chain of short kernels gates the next batch, the brackets of ten kernel slots cost ~1 ms per pass
  bool wanted() const { return e->profiling == 1 || (e->profiling == 2 && (id == RNA_K_ASTAR_SEARCH || id == RNA_K_ASTAR_RESET)); }
  ~KernelTimer() {
    if (!wanted() || !a || !b) return;
    (void)hipEventRecord(b, st);
    e->prof[id].pending.emplace_back(a, b);
    e->pending_events += 2;
  }
};

template <typename T>
inline int dev_alloc(rna_engine* e, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) return RNA_OK;
  hipError_t st = hipMalloc((void**)p, n * sizeof(T));
  if (st != hipSuccess) { *p = nullptr; return fail(e, RNA_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(st)); }
  return RNA_OK;
}

template <typename T>
inline void dev_free(T** p) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
}

// the engine stream waits for the side work in flight (see rna_engine::vfh_stream); cheap when there is none
int side_join(rna_engine* e);
// the side stream (created on first use)
int side_stream(rna_engine* e, hipStream_t* out);
// grid A*: second pass over the searches of finished batches that ran out of pages (astar.hip)
int astar_settle(rna_engine* e);
// entry of a C-ABI call that enqueues on the engine stream: select the device, join the side work.  Calls that cannot
// disturb it (the ray batch of the laser layer, the searches, the VFH+ step itself, getters) use RNA_ENTER_NOJOIN.
#define RNA_ENTER(e)                                        \
  do {                                                      \
    RNA_HIP(e, hipSetDevice((e)->device));                  \
    const int rc_join_ = rna::side_join(e);                 \
    if (rc_join_ != RNA_OK) return rc_join_;                \
  } while (0)
#define RNA_ENTER_NOJOIN(e) RNA_HIP(e, hipSetDevice((e)->device))

// module entry points used across translation units
int himm_release(rna_engine* e);
int vfh_release(rna_engine* e);
int astar_release(rna_engine* e);
int map_prepare_nbr(rna_engine* e);   // make e->nbr consistent with the master layer
int sync_all(rna_engine* e);          // main stream + every A* side stream
// tile-synchronous A* (astar_tile.hip)
bool tsa_supported(const rna_engine* e);
int tsa_tiles(const rna_engine* e);                                   // 32 x 32 tiles of the map
size_t tsa_pool_bytes(int max_queries, int cap);                      // pages + pending bitmaps of one pipeline stage
size_t tsa_aux_bytes(const rna_engine* e, int max_queries, int cap);  // per stage, must start zeroed
int tsa_stage_prepare(rna_engine* e, int slot);                       // fresh stage: every page "unreached"
constexpr int TSA_RETRY = 8;                                          // full-size retry slots per stage
size_t tsa_retry_pool_bytes(const rna_engine* e);
size_t tsa_retry_aux_bytes(const rna_engine* e);
int tsa_retry_prepare(rna_engine* e, int slot);
size_t tsa_ring_bytes(const rna_engine* e, int max_queries);            // one entry of the launch ring
int tsa_retry_launch(rna_engine* e, int slot, hipStream_t search_stream, int count);   // second pass over `count` listed searches
int tsa_launch(rna_engine* e, int slot, hipStream_t init_stream, hipStream_t search_stream, hipEvent_t ev_init,
               const rna_astar_query* q_dev, int n, int32_t* paths_dev, int max_len, rna_astar_result* res_dev);
int tsa_settled(rna_engine* e, int slot, const rna_astar_query* q, const rna_astar_result* r, int n, int32_t* d_counts);
int tsa_counters_read(rna_engine* e, unsigned long long* out, bool reset);   // sums the job counters of every stage view into out[16]

}  // namespace rna
