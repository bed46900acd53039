/*
 * rna.h -- C ABI of the MI355X-native occupancy-grid planning engine (librna.so).
 *
 * The reference (jmloveyj/ros_navigation) has no FFI layer: its replan hot path sits behind plain
 * C++ class methods called by the node mains.  Every entry point below replaces one of those call
 * sites; the C++ mirror classes in ros_navigation_amd/host/ (same names and argument meaning as
 * the reference) and the ctypes binding in ros_navigation_amd/capi.py are thin shims over it.
 * Citations: mc/ = move_control/, gmc/ = grid_map-master/grid_map_core/ in the reference tree.
 *
 * Conventions: every call returns 0 (RNA_OK) or a negative rna_status; no exceptions cross the
 * boundary; buffers are caller-owned; calls on one engine are serialised by the caller (one HIP
 * stream per engine).  Grids are column-major float32, linear index = i + j*rows, exactly the
 * Eigen::MatrixXf storage of grid_map::GridMap (gmc/include/grid_map_core/TypeDefs.hpp:16).
 * "_device" variants take device pointers (inputs/outputs already resident in HBM) and are
 * asynchronous on the engine's stream; the plain variants take host pointers and return after the
 * results are back in host memory.
 */
#ifndef RNA_H
#define RNA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rna_laser_scan carries the end pose (80 bytes), rna_astar_result.expanded / .rounds changed meaning (cells written,
 *    tile jobs per wavefront), statuses 4 / 5, profile slot astar_reset, default bucket width 96000.  A host checks
 *    rna_abi_version() == RNA_ABI_VERSION after loading the library (capi.py and move_control_amd.hpp do). */
/* 3: rna_synchronize_map, rna_hw_queue_advice (round 4); no existing signature changed.
 * 4: rna_scan_to_rays_tf[_device], rna_range_to_rays_tf (sensors with a full tf transform; round 4); no existing signature changed.
 * 5: rna_astar_job_counters (round 6); the default bucket width of the grid search is 128000, a pipeline may have up to 20 stages,
 *    rna_astar_result.rounds counts every job again (also the ones that find nothing); no existing signature changed. */
#define RNA_ABI_VERSION 5

typedef enum {
  RNA_OK = 0,
  RNA_EINVAL = -1,     /* bad argument */
  RNA_ENOMEM = -2,     /* device/host allocation failed */
  RNA_EHIP = -3,       /* HIP runtime error (rna_last_error has the text) */
  RNA_ECAPACITY = -4,  /* a device work queue overflowed; retry with a larger capacity */
  RNA_ESTATE = -5,     /* call order (e.g. vfh step before vfh init) */
  RNA_ENODEVICE = -6   /* no usable gfx950 device */
} rna_status;

typedef struct rna_engine rna_engine;

/* grid_map::GridMap geometry (gmc/include/grid_map_core/GridMap.hpp:493-516) */
typedef struct {
  double length[2];      /* length_   */
  double position[2];    /* position_ */
  double resolution;     /* resolution_ */
  int32_t size[2];       /* size_: rows = Index(0), cols = Index(1) */
  int32_t start_index[2];/* startIndex_ (circular buffer) */
} rna_geometry;

/* layers of MapProvider's GridMap (mc/src/map_provider.cpp:17-19, map_updater.h:12-13) */
typedef enum { RNA_LAYER_MASTER = 0, RNA_LAYER_LASER = 1, RNA_LAYER_RANGE = 2, RNA_NUM_LAYERS = 3 } rna_layer;

/* ---- engine / GridMap container -------------------------------------------------------------- */
/* GridMap::GridMap + setGeometry (gmc/src/GridMap.cpp:27-70): size = round(length/res), all layers
 * NaN.  MapProvider::initMap (mc/src/map_provider.cpp:145-149). */
int rna_create(rna_engine** out, double length_x, double length_y, double resolution,
               double position_x, double position_y, int device_id);
void rna_destroy(rna_engine* e);
const char* rna_last_error(const rna_engine* e);
int rna_abi_version(void);
int rna_get_geometry(const rna_engine* e, rna_geometry* out);
/* GridMap::operator[] / get (gmc/src/GridMap.cpp:125-151): whole-layer copies to/from the host */
int rna_layer_upload(rna_engine* e, int layer, const float* host, size_t n_cells);
int rna_layer_download(rna_engine* e, int layer, float* host, size_t n_cells);
int rna_layer_fill(rna_engine* e, int layer, float value);
/* device pointer of a layer (rows*cols float32, column-major) for zero-copy producers/consumers */
void* rna_layer_device_ptr(rna_engine* e, int layer);
/* hipStream_t of the engine's MAP stream: map updates, compose, pack / unpack, uploads and the host-pointer calls run
 * on it.  It is not the only stream: with a pipeline depth > 1 the A* searches run on the stages' own streams and
 * rna_vfh_step_batch_device on an internal side stream, so work a caller chains on rna_stream() is ordered with the
 * map, NOT with the outputs of those two calls -- they are valid after rna_synchronize() (or, for VFH+, after
 * rna_synchronize_map()). */
void* rna_stream(rna_engine* e);
/* everything the engine has in flight: the map stream, the VFH+ side stream, every A* pipeline stage, and the second
 * passes over searches that ran out of pages (issued here if they are still due) */
int rna_synchronize(rna_engine* e);
/* the map stream and the VFH+ side stream only: searches in flight keep running (a tiled host's per-pass exchange,
 * a consumer of the VFH+ commands).  Does not make A* outputs valid. */
int rna_synchronize_map(rna_engine* e);
/* GridMap::getIndex / getPosition / isInside (gmc/src/GridMap.cpp:227-240) -- host-side math */
int rna_get_index(const rna_engine* e, double x, double y, int32_t index[2]);   /* 1 inside, 0 outside */
int rna_get_position(const rna_engine* e, int32_t i, int32_t j, double position[2]);

/* The same math over a geometry alone -- no engine, no GPU (host iterators of the C++ mirror, callers that only
 * need indices).  1 inside / in range, 0 outside, < 0 = rna_status. */
int rna_geometry_index(const rna_geometry* g, double x, double y, int32_t index[2]);
int rna_geometry_position(const rna_geometry* g, int32_t i, int32_t j, double position[2]);
/* grid_map_core's iterators as cell lists ((i, j) buffer-index pairs in visiting order; return value = length of
 * the walk, only the first cap cells are written): LineIterator (gmc/src/iterators/LineIterator.cpp:16-150; the
 * cells a HIMM ray clears), CircleIterator (CircleIterator.cpp:16-93; the blocked-disc scan of GlobalPlanner::
 * ifBlocked), SubmapIterator (SubmapIterator.cpp:28-83). */
int rna_line_cells(const rna_geometry* g, double sx, double sy, double ex, double ey, int32_t* cells, int cap);
int rna_circle_cells(const rna_geometry* g, double cx, double cy, double radius, int32_t* cells, int cap);
int rna_submap_cells(const rna_geometry* g, const int32_t top_left[2], const int32_t size[2], int32_t* cells, int cap);
/* GridMap copy-assignment (`map = map_`, MapProvider::getMap, mc/src/map_provider.cpp:120-125): a new engine with
 * the same geometry, start index and layer contents; the caller destroys it. */
int rna_clone(rna_engine* src, rna_engine** out);

/* ---- HIMM map update ------------------------------------------------------------------------- */
/* RangeSample (mc/include/move_control/map_updater.h:28-32) */
typedef struct {
  double sx, sy;        /* start */
  double ex, ey;        /* end   */
  int32_t clear_end;    /* ifClearEnd */
  int32_t _pad;
} rna_ray;
/* LaserMapUpdater::updateMap / RangeMapUpdater::updateMap -> MapUpdater::lineOnMap for every
 * sample in order (mc/src/laser_map_updater.cpp:7-21, map_updater.h:38-71): order-faithful result. */
int rna_himm_update(rna_engine* e, int layer, const rna_ray* rays_host, int n);
int rna_himm_update_device(rna_engine* e, int layer, const rna_ray* rays_device, int n);
/* MapProvider::composeMasterMapFromLayerdMap (mc/src/map_provider.cpp:216-223): master = laser.
 * mode 0: only the 64x64 tiles the HIMM batches touched since the last compose (fused path);
 * mode 1: whole-layer copy exactly as the reference does every cycle. */
int rna_compose_master(rna_engine* e, int mode);
/* MapProvider::updateMap (mc/src/map_provider.cpp:190-205): laser HIMM batch, then compose(mode) */
int rna_update_map(rna_engine* e, const rna_ray* rays_host, int n, int compose_mode);
int rna_update_map_device(rna_engine* e, const rna_ray* rays_device, int n, int compose_mode);
/* Tiled single map (SURVEY.md 8e mode 2, BASELINE config 5): every GPU holds a full-size layer but
 * owns one window of it.  rna_himm_set_window restricts every later HIMM batch to the cells
 * [i0, i0+ni) x [j0, j0+nj) (buffer indices): lines are still clipped and walked on the whole map's
 * geometry (LineIterator.cpp:60-150), so inside the window the result is bit-identical to the
 * untiled update.  ni <= 0 or nj <= 0 restores the whole map. */
int rna_himm_set_window(rna_engine* e, int i0, int j0, int ni, int nj);
/* Rectangular block of a layer <-> dense column-major device buffer (ni*nj floats, i fastest):
 * halo strips and owner tiles exchanged between GPUs.  Both return after the copy has finished. */
int rna_layer_pack_region(rna_engine* e, int layer, int i0, int ni, int j0, int nj, float* dense_device);
int rna_layer_unpack_region(rna_engine* e, int layer, int i0, int ni, int j0, int nj, const float* dense_device);
/* Incremental hand-over of the tiled mode.  rna_last_dirty_tiles: one byte per 64 x 64 tile (index tj*tiles_i + ti,
 * tiles_i = ceil(rows/64)) = the tiles the last rna_compose_master consumed, i.e. what the last HIMM batch changed on
 * this GPU.  rna_layer_pack_tiles / rna_layers_unpack_tiles move the listed tiles, clipped to the window
 * [i0,i0+ni) x [j0,j0+nj), through a dense device buffer of n slots of 4096 floats; unpack writes layer_a and (if
 * >= 0) layer_b and flags the tiles for the next rna_compose_master(e, 0), which refreshes their A* neighbour masks
 * (+ring) instead of rebuilding all of them.  rna_layer_unpack_region_tracked is rna_layer_unpack_region with the
 * same per-tile bookkeeping.  Precondition of the tracked calls: the laser layer is complete on this GPU (every
 * owner's changes have been unpacked into laser AND master), so that composing a flagged tile is harmless. */
int rna_last_dirty_tiles(rna_engine* e, uint8_t* flags_host, size_t n_tiles);
int rna_layer_pack_tiles(rna_engine* e, int layer, const int32_t* tiles_host, int n, int i0, int ni, int j0, int nj,
                         float* dense_device);
int rna_layers_unpack_tiles(rna_engine* e, int layer_a, int layer_b, const int32_t* tiles_host, int n, int i0, int ni,
                            int j0, int nj, const float* dense_device);
int rna_layer_unpack_region_tracked(rna_engine* e, int layer, int i0, int ni, int j0, int nj, const float* dense_device);
/* The same with the tile lists resident on the device (a C/C++ host that drives RCCL itself, include/rna_rccl.h):
 * rna_last_dirty_tiles_device compacts the flagged tiles that intersect the window into list_device (room for every
 * tile of the map) and their number into *count_device; the pack / unpack forms take device lists.  Asynchronous on
 * the engine's stream. */
int rna_last_dirty_tiles_device(rna_engine* e, int i0, int ni, int j0, int nj, int32_t* list_device, int* count_device);
int rna_layer_pack_tiles_device(rna_engine* e, int layer, const int32_t* tiles_device, int n, int i0, int ni, int j0, int nj,
                                float* dense_device);
int rna_layers_unpack_tiles_device(rna_engine* e, int layer_a, int layer_b, const int32_t* tiles_device, int n, int i0, int ni,
                                   int j0, int nj, const float* dense_device);
/* GridMap::move (gmc/src/GridMap.cpp:346-412): recentre the circular buffer, dropped cells -> NaN */
int rna_move(rna_engine* e, double position_x, double position_y, int* moved);

/* MapProvider::getSubMap -> GridMap::getSubmap (mc/src/map_provider.cpp:93-100, gmc/src/GridMap.cpp:287-339,
 * getSubmapInformation gmc/src/GridMapMath.cpp:246-296): the requested window is clamped to the map, and the
 * submap's cells are gathered across the circular-buffer seam (the reference's <= 4 quadrant block copies) into
 * out (column-major, info->size[0]*info->size[1] floats, startIndex (0,0)).  Callers: Steerer (1.5 m window,
 * mc/src/steerer.cpp:158; fused into rna_vfh_step_batch here) and Nav::makePlan's planning window
 * (mc/src/nav_node.cpp:141).  Returns 1 = success, 0 = the reference's isSuccess == false, < 0 = rna_status
 * (RNA_ECAPACITY when the submap has more than cap_cells cells; info is filled in that case too). */
typedef struct {
  double length[2];      /* submap length_ (size * resolution) */
  double position[2];    /* submap position_ (centre) */
  int32_t size[2];
  int32_t top_left[2];   /* buffer index of the submap's first cell in the parent map */
} rna_submap_info;
int rna_get_submap(rna_engine* e, int layer, double position_x, double position_y, double length_x, double length_y,
                   float* out_host, size_t cap_cells, rna_submap_info* info);
int rna_get_submap_device(rna_engine* e, int layer, double position_x, double position_y, double length_x,
                          double length_y, float* out_device, size_t cap_cells, rna_submap_info* info);
/* The same as a GridMap of its own, which is what GridMap::getSubmap returns: a NEW engine (same device) with the
 * submap's geometry and all three layers, ready for rna_rrt_batch / rna_astar_batch on the planning window -- the
 * flow of Nav::makePlan (mc/src/nav_node.cpp:136-152: getSubMap, then RrtPlanner on the submap).  Returns 1 and the
 * engine in *out (caller destroys it), 0 when the reference's isSuccess is false, < 0 = rna_status. */
int rna_create_submap(rna_engine* parent, double position_x, double position_y, double length_x, double length_y,
                      rna_engine** out);

/* ---- VFH+ local avoidance -------------------------------------------------------------------- */
/* VFH constructor arguments + SetRobotRadius (mc/include/move_control/vfh.h:185-203,235;
 * defaults of Steerer::initVfh, mc/src/steerer.cpp:69-121) */
typedef struct {
  double cell_size;
  int32_t window_diameter;
  int32_t sector_angle;
  double safety_dist_0ms, safety_dist_1ms;
  int32_t max_speed, max_speed_narrow_opening, max_speed_wide_opening;
  int32_t max_acceleration, min_turnrate, max_turnrate_0ms, max_turnrate_1ms;
  double min_turn_radius_safety_factor;
  double free_space_cutoff_0ms, obs_cutoff_0ms, free_space_cutoff_1ms, obs_cutoff_1ms;
  double weight_desired_dir, weight_current_dir;
  double robot_radius;
} rna_vfh_params;
void rna_vfh_default_params(rna_vfh_params* p);
/* One robot pose + the per-step arguments of VFH::Update_VFH (mc/include/move_control/vfh.h:216-222)
 * as Steerer::update derives them (mc/src/steerer.cpp:221-263); dt replaces gettimeofday(). */
typedef struct {
  double x, y, yaw;            /* MapProvider::getRobotPos(pos, orient) */
  double dt;                   /* seconds since this robot's previous step */
  int32_t current_speed;       /* mm/s */
  float goal_direction;        /* deg, 90 = straight ahead */
  float goal_distance;         /* mm */
  float goal_tolerance;        /* mm */
} rna_pose;
typedef struct {
  int32_t chosen_speed;        /* mm/s  */
  int32_t chosen_turnrate;     /* deg/s */
  float picked_angle;          /* VFH::GetPickedAngle() */
  int32_t emergency;           /* 1 when something was inside the safety distance */
} rna_vfh_out;
/* VFH::VFH + SetRobotRadius + Init (mc/src/vfh.cpp:53-110,237-416) for `n_robots` independent,
 * stateful VFH instances (Last_Binary_Hist, Last_Picked_Angle, last_chosen_speed, ...). */
int rna_vfh_init(rna_engine* e, const rna_vfh_params* p, int n_robots);
int rna_vfh_reset(rna_engine* e);          /* re-Init every instance's state */
int rna_vfh_hist_size(const rna_engine* e);
/* Steerer::getRangesFromSubmap + VFH::Update_VFH for robots [0, n) (mc/src/steerer.cpp:147-191,
 * 260-263; mc/src/vfh.cpp:480-605).  origin_hist / hist: n x hist_size float32 (may be NULL):
 * VFH::OriginHist and VFH::Hist after the step. */
int rna_vfh_step_batch(rna_engine* e, const rna_pose* poses_host, int n, rna_vfh_out* out_host,
                       float* origin_hist_host, float* hist_host);
/* (_device: with a pipeline depth > 1 the step runs on an internal side stream behind the map stream -- its outputs are
 * valid after rna_synchronize_map() / rna_synchronize(), not for work merely enqueued on rna_stream() after the call) */
int rna_vfh_step_batch_device(rna_engine* e, const rna_pose* poses_device, int n, rna_vfh_out* out_device,
                              float* origin_hist_device, float* hist_device);
/* VFH::Update_VFH fed with caller-provided range scans (double[361][2] per robot, as the reference
 * signature) instead of the map -- the drop-in for code that owns its own scan (cd/src/vfh_node.cpp). */
int rna_vfh_update_batch(rna_engine* e, const double* ranges_host /* n*361*2 */, const rna_pose* poses_host,
                         int n, rna_vfh_out* out_host, float* origin_hist_host, float* hist_host);

/* ---- global planning: grid A* ------------------------------------------------------------------ */
/* The reference's AStarPlanner::makePlan (mc/src/astar_planner.cpp:63-96) searches a 9-vertex
 * waypoint graph; BASELINE.json asks for a grid A* over the GridMap, whose contract is defined in
 * DESIGN.md ("Grid A* contract") and restated by oracle/astar.c.  Cells are buffer linear indices
 * (i + j*rows, what GridMap::getIndex hands out); on a map that GridMap::move has recentred the
 * search runs in map space (unwrapped indices) and paths cross the circular-buffer seam. */
typedef struct { int32_t start, goal; } rna_astar_query;
typedef struct {
  int32_t status;      /* 0 found, 1 no path, 2 invalid query, 3 path longer than max_path_len,
                          4 path cost beyond the field's 30-bit g range (>= 1.07e9), 5 TRANSIENT: the query's share of search
                          pages was used up (pages are only limited after rna_astar_set_page_cap or when HBM is short) and it
                          waits for its second pass on a full-size retry slot -- visible only in a device result buffer read
                          before rna_synchronize(); never final (a goal that cannot be reached floods its component, is
                          searched again and answers 1) */
  int32_t path_len;    /* cells, start..goal inclusive */
  int32_t cost;        /* 1000/1414 integer cost of the path */
  int32_t expanded;    /* cells the device wrote (64 per row of a tile a job changed; >= the oracle's settled count) */
  int32_t rounds;      /* tile jobs per wavefront of the query's workgroup: every job, also the ones that find nothing better in their
                          halo, and every sticky turn (rounds 1-2: barrier-separated rounds of jobs) */
  int32_t buckets;     /* f-buckets visited */
} rna_astar_result;
/* max_queries: queries searched concurrently (one g-field each; larger batches are processed in
 * chunks); queue_capacity: ignored (the cell queues of the frontier kernel removed in round 3; the tile kernel's open
 * list has a fixed size in LDS and parks what does not fit); bucket_width: the f-range (cost units, >= 2828) inside
 * which free wavefronts take tiles by key before the search advances (default 128000; 96000 until round 5); 0 = keep/default.
 * Maps of more than 65 536 tiles of 64 x 16 cells (8192 x 8192 cells) are refused with RNA_EINVAL. */
int rna_astar_configure(rna_engine* e, int max_queries, int queue_capacity, int bucket_width);
/* Pipelined batches: with depth d > 1 consecutive rna_astar_batch_device calls run their searches on d
 * rotating internal streams (each with its own search fields), so the tail of one batch overlaps
 * the next batch and the next map update.  Outputs of a call are valid after rna_synchronize() -- not after a
 * bare hipDeviceSynchronize(): searches that ran out of their share of pages are searched again by a second launch
 * the host issues when it sees their count, at the latest inside rna_synchronize().  The call returns after
 * enqueueing; while every stage is busy it blocks until one is free.  The caller must give calls that may be in
 * flight together distinct output buffers.  Default 4; up to 20 (16 until round 5).
 * Hardware queues: every stage's stream wants a hardware queue of its own, and the HIP runtime multiplexes all of a
 * process's streams over GPU_MAX_HW_QUEUES (default 4) of them, read from the environment at its first call.  librna.so
 * sets GPU_MAX_HW_QUEUES=8 when it is loaded and the variable is unset (RNA_KEEP_HW_QUEUES=1 turns that off); when the
 * value in force is too small for `depth` the call still succeeds -- searches then share queues and overlap less --
 * and rna_last_error() holds the one-line advice of rna_hw_queue_advice(). */
int rna_astar_set_pipeline_depth(rna_engine* e, int depth);
/* Host-only (no engine, no GPU): 0 and an empty string when GPU_MAX_HW_QUEUES -- the value librna.so found (or set) WHEN IT
 * WAS LOADED, which is what the HIP runtime latches at its first call; later changes of the environment do not count --
 * leaves the stages of `pipeline_depth` a hardware queue each (depth <= 2, or >= 8 queues); 1 and one line of advice in
 * buf when it is too small; 2 and one line when the library set the variable itself but the process had already opened
 * the GPU by then (the setting probably came too late: export it before the first HIP call, e.g. before importing torch). */
int rna_hw_queue_advice(int pipeline_depth, char* buf, size_t cap);
/* The tile kernel keeps a search's distance field in 4 KiB pages (64 x 16 cells) handed out on first touch.  By
 * default every query may take one page per tile of the map (it can never run out; HBM is only touched where a
 * search goes; half a map's worth when the pipeline stages would not fit HBM otherwise).  A smaller share per query makes room for more queries / pipeline stages in flight; a search that
 * needs more is searched again on one of 8 full-size retry slots per stage (eight at a time, as many passes as it
 * takes): status 5 does not reach the caller.  0 = default. */
int rna_astar_set_page_cap(rna_engine* e, int pages_per_query);
/* What the first batch (or rna_astar_configure after a map is loaded) actually allocated: pipeline stages, pages per
 * query and concurrent queries may have been reduced to fit 75 % of the free HBM.  Any pointer may be NULL; all three
 * are 0 before the allocation. */
int rna_astar_effective_config(const rna_engine* e, int* pipeline_depth, int* pages_per_query, int* max_queries);
int rna_astar_batch(rna_engine* e, const rna_astar_query* queries_host, int n, int32_t* paths_host,
                    int max_path_len, rna_astar_result* results_host);
int rna_astar_batch_device(rna_engine* e, const rna_astar_query* queries_device, int n, int32_t* paths_device,
                           int max_path_len, rna_astar_result* results_device);
/* E of DESIGN.md's roofline: for each query of the LAST batch (n <= max_queries, i.e. one chunk) the
 * number of cells with g + h <= f*, counted from the g fields still resident in HBM.  Equals the CPU
 * oracle's settled count; measurement/test utility, not part of the timed path.  -1 for a query whose field is gone
 * (it outgrew its share of pages and was searched again in a second pass whose retry slot has been reused since). */
int rna_astar_settled_counts(rna_engine* e, int32_t* counts_host, int n);
/* What the grid-A* search kernels counted since the engine's search state was allocated (or since the last call with
 * reset != 0), summed over all searches of all batches; waits for the batches in flight.  counters_host[8]:
 *   [0] searches, [1] tiles that got a page (a search's "touched tiles"), [2] tile jobs -- every job, sticky turns included --,
 *   [3] of them jobs that found nothing better in their halo and left before the sweeps, [4] sticky turns (a wavefront kept a
 *   tile that was woken while its job ran), [5] rows of 64 cells written, [6] f-buckets, [7] buckets that were run again because the open list ran out of nodes.
 * Measurement utility (bench.py's roofline.work_inflation: jobs per touched tile and the no-op share, observed in the timed
 * run itself); the kernel's cost is seven atomic adds per SEARCH.  No reference counterpart (the reference has no grid search:
 * move_control/src/astar_planner.cpp:63-96 is the waypoint graph). */
int rna_astar_job_counters(rna_engine* e, uint64_t* counters_host, int reset);
/* per-cell traversable-neighbour mask derived from the master layer (rows*cols uint8) */
int rna_astar_download_nbr_mask(rna_engine* e, uint8_t* host, size_t n_cells);

/* ---- global planning: waypoint-graph A* (the reference's own AStarPlanner) ------------------- */
/* AStarPlanner::init + makePlan over a caller-supplied graph (astar_planner.cpp:63-145): start and
 * target positions are snapped to the closest vertex; path = start, vertex locations..., target. */
int rna_graph_astar_batch(rna_engine* e, int n_vertices, const double* vertex_xy, int n_edges,
                          const int32_t* edge_uv, const float* edge_weight /* NULL = 0 as the reference */,
                          const double* start_target_xy /* n*4 */, int n, double* paths_xy /* n*max_len*2 */,
                          int max_len, int32_t* path_len);

/* ---- global planning: RRT (mc/src/rrt_planner.cpp) -------------------------------------------- */
typedef struct {
  double start[2], target[2];
  double close_tolerance;      /* RrtPlanner ctor default 0.2 */
  uint32_t seed;               /* srand(seed) of this query's private glibc-compatible rand() */
  int32_t max_samples;         /* bound for extendTree's while(true) */
} rna_rrt_query;
typedef struct { int32_t status, path_len, tree_size, samples; } rna_rrt_result;
int rna_rrt_batch(rna_engine* e, const rna_rrt_query* queries_host, int n, double* paths_xy_host,
                  int max_path_len, rna_rrt_result* results_host);
int rna_rrt_batch_device(rna_engine* e, const rna_rrt_query* queries_device, int n, double* paths_xy_device,
                         int max_path_len, rna_rrt_result* results_device);

/* ---- laser ingestion ------------------------------------------------------------------------- */
/* One sensor_msgs/LaserScan plus the planar sensor poses tf reports for it: at header.stamp, and at the end time
 * laser_geometry's high-fidelity projection asks for, stamp + (beams - 1) * time_increment, where `beams` counts the
 * beams of the scan that is projected -- the decimated one when angle_increment < 0.017 (rna_scan_projected_beams).
 * Every beam is transformed with the pose interpolated for its index (position linearly, yaw along the shortest arc),
 * as transformLaserScanToPointCloud does with tf's start / end transforms.  End pose == start pose: a pose that is
 * constant over the scan. */
typedef struct {
  float angle_min, angle_max, angle_increment, range_min, range_max;
  int32_t n_ranges;
  int64_t ranges_offset;   /* first range of this scan in the concatenated ranges array */
  double x, y, yaw;        /* sensor pose in the map frame at header.stamp (also the origin of every ray) */
  double x_end, y_end, yaw_end;   /* sensor pose at the end time */
} rna_laser_scan;
/* number of beams of the scan LaserMapUpdater hands to the projector (simplifyLaserScan, mc/src/laser_map_updater.cpp:
 * 118-143, when angle_increment < 0.017; else n_ranges): the end time above is stamp + (this - 1) * time_increment */
int rna_scan_projected_beams(int n_ranges, float angle_increment);
/* LaserMapUpdater::bufferIncomingMsg (mc/src/laser_map_updater.cpp:37-75): simplifyLaserScan
 * (:118-143), laser_geometry's projection + tf transform of every valid beam (:78-99), ray origin
 * (:101-116) -> RangeSamples in scan order, beam order.  n_rays receives the number of rays produced;
 * RNA_ECAPACITY if it exceeds max_rays (the first max_rays rays are still written).  The device
 * variant leaves the count in device memory; max_beams_per_scan >= every scan's n_ranges (<= 8192). */
int rna_scan_to_rays(rna_engine* e, const rna_laser_scan* scans_host, int n_scans, const float* ranges_host,
                     size_t n_ranges_total, rna_ray* rays_host, int max_rays, int* n_rays);
int rna_scan_to_rays_device(rna_engine* e, const rna_laser_scan* scans_device, int n_scans, const float* ranges_device,
                            int max_beams_per_scan, rna_ray* rays_device, int max_rays, int* n_rays_device);

/* The same with the sensor's FULL pose (a tilted or rolled mount, a sensor above the base): what tf's lookupTransform
 * (map frame <- header.frame_id) returns at the two times, translation and rotation quaternion (x, y, z, w) as
 * StampedTransform::getOrigin / getRotation give them.  Every beam's point (x, y, 0 in the sensor frame, float32) goes
 * through the transform interpolated for its index -- translation by tf::Vector3::setInterpolate3, rotation by
 * tf::Quaternion::slerp (shortest arc) -- and the map keeps x and y of the result, as LaserMapUpdater reads only the
 * cloud's x and y fields (mc/src/laser_map_updater.cpp:53-60, 78-99).  The ray origin is the start translation's x, y
 * (tf::transformPoint of the frame's origin, :101-116).  For a planar pose (rotation about z) the result equals
 * rna_scan_to_rays' up to the rounding of the two formulations. */
typedef struct {
  float angle_min, angle_max, angle_increment, range_min, range_max;
  int32_t n_ranges;
  int64_t ranges_offset;   /* first range of this scan in the concatenated ranges array */
  double t[3], q[4];       /* sensor frame in the map frame at header.stamp: translation, quaternion x y z w */
  double t_end[3], q_end[4];   /* ... at the end time (see rna_laser_scan) */
} rna_laser_scan_tf;
int rna_scan_to_rays_tf(rna_engine* e, const rna_laser_scan_tf* scans_host, int n_scans, const float* ranges_host,
                        size_t n_ranges_total, rna_ray* rays_host, int max_rays, int* n_rays);
int rna_scan_to_rays_tf_device(rna_engine* e, const rna_laser_scan_tf* scans_device, int n_scans, const float* ranges_device,
                               int max_beams_per_scan, rna_ray* rays_device, int max_rays, int* n_rays_device);

/* One sensor_msgs/Range reading (sonar) plus the planar sensor pose tf reports for its stamp. */
typedef struct {
  float range, max_range;  /* msg->range, msg->max_range */
  double x, y, yaw;        /* sensor pose in the map frame */
} rna_range_reading;
/* RangeMapUpdater::bufferIncomingMsg (mc/src/range_map_updater.cpp:38-76), host only: one RangeSample per reading,
 * start = T(0,0,0), end = T(range,0,0) (tf::transformPoint in double), ifClearEnd = !(range < max_range).  The rays
 * go to rna_himm_update(e, RNA_LAYER_RANGE, ...) -- the "range" MapUpdater of MapProvider's factory
 * (mc/src/map_provider.cpp:12-15,262-266); master is composed from the laser layer only (:216-223). */
int rna_range_to_rays(const rna_range_reading* readings, int n, rna_ray* rays);
/* The same with the sensor's full pose (translation + quaternion x y z w, as above): start = T * (0, 0, 0),
 * end = T * (range, 0, 0), x and y kept. */
typedef struct {
  float range, max_range;
  double t[3], q[4];
} rna_range_reading_tf;
int rna_range_to_rays_tf(const rna_range_reading_tf* readings, int n, rna_ray* rays);

/* ---- message formats either side of the path ------------------------------------------------- */
/* GridMapRosConverter::toOccupancyGrid (grid_map-master/grid_map_ros/src/GridMapRosConverter.cpp:251-287)
 * as MapProvider::publishMap calls it with (0, 255) (mc/src/map_provider.cpp:113-118,206-213):
 * out[nCells-1-(ui + uj*rows)] = NaN or negative -> -1, else (int8)(clamp((v-min)/(max-min), 0, 1)*100),
 * (ui, uj) the unwrapped index -- info.width = rows, info.height = cols, origin = position - length/2
 * (rna_get_geometry).  out holds rows*cols int8. */
int rna_to_occupancy_grid(rna_engine* e, int layer, float data_min, float data_max, int8_t* out_host);
int rna_to_occupancy_grid_device(rna_engine* e, int layer, float data_min, float data_max, int8_t* out_device);
/* GridMapRosConverter::fromOccupancyGrid, data part (:238-246): layer(i) = data[n-1-i], -1 -> NaN.  The
 * geometry part (:208-236) is rna_create(resolution*size, resolution, origin + length/2). */
int rna_from_occupancy_grid(rna_engine* e, int layer, const int8_t* data_host);
/* Steerer::pubHist -> move_control/Histogram.msg (mc/src/steerer.cpp:201-220) for the first n VFH
 * instances: num_bin = hist_size/2 per robot; x_data[num_bin] (shared), y_data / y_bin_data
 * [n][num_bin] = (uint16)(int)OriginHist / Hist; thresholds = {yLowThreshold, yHighThreshold}. */
int rna_vfh_hist_msg_batch(rna_engine* e, int n, uint16_t* x_data_host, uint16_t* y_data_host,
                           uint16_t* y_bin_data_host, uint16_t thresholds[2]);
/* Nav::taileredPlan (mc/src/nav_node.cpp:192-204): walk the plan backwards keeping every stride-th
 * index and the last one (host only; out_xy holds up to n positions). */
int rna_tailor_plan(const double* plan_xy, int n, unsigned stride, double* out_xy, int* n_out);
/* Steerer::acceptPlan + the plan-following head of Steerer::update (mc/src/steerer.cpp:27-33,222-256), host only:
 * *plan_index is Steerer::planIndex_ (acceptPlan sets it to 1); way points closer than 250 mm are skipped, then
 * goal distance (mm, float hypot), goal direction (deg, RAD2DEG(normalize_angle_positive(atan2f(dy,dx) - yaw + M_PI/2)))
 * and current_speed = (int)(linear_velocity * 1000.0) are written to *pose together with x, y, yaw, dt -- ready for
 * rna_vfh_step_batch.  Returns 1 while the plan is being followed, 0 when it is finished (ifPlanReady_ = false;
 * also for a plan of fewer than 2 points, where the reference reads plan_[1] out of bounds), < 0 on bad arguments. */
int rna_follow_plan(const double* plan_xy, int n, int32_t* plan_index, double x, double y, double yaw,
                    double linear_velocity, double dt, rna_pose* pose);

/* ---- measurement ------------------------------------------------------------------------------ */
typedef enum {
  RNA_K_HIMM_PREP = 0, RNA_K_HIMM_RASTER, RNA_K_HIMM_APPLY, RNA_K_COMPOSE, RNA_K_NBRMASK,
  RNA_K_VFH_STEP, RNA_K_ASTAR_SEARCH, RNA_K_ASTAR_INIT, RNA_K_RRT, RNA_K_OCCUPANCY, RNA_K_ASTAR_RESET, RNA_K_COUNT
} rna_kernel_id;
/* when enabled, every launch of the kernels above is bracketed by hipEvents on the stream it runs on (the engine
 * stream; astar_search: the pipeline stage's own stream; astar_init / astar_reset / vfh_step: the side stream) */
/* on: 0 off, 1 every slot, 2 only astar_search / astar_reset (an event record is a packet of its own on the stream:
 * bracketing the ten slots of the engine stream costs its chain of short kernels about 1 ms per replan pass) */
int rna_profile_enable(rna_engine* e, int on);
int rna_profile_reset(rna_engine* e);
int rna_profile_get(rna_engine* e, int kernel_id, double* total_ms, int64_t* launches);
const char* rna_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* RNA_H */
