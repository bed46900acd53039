/*
 * rna_oracle.h -- CPU restatement of the move_control hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity ORACLE for the MI355X engine in ros_navigation_amd/.  It is a plain-C
 * restatement, written from the behaviour of the reference sources (cited per function as
 * mc/ = /root/reference/move_control, gmc/ = /root/reference/grid_map-master/grid_map_core),
 * and is only ever imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product (ros_navigation_amd) never links or calls anything in oracle/.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - grid index math, submaps, iterators, LineIterator, GridMap::move: pinned by the reference's
 *     own gtest known answers (tests/golden/gridmap_known_answers.json).
 *   - VFH+ (vfh.c): pinned against the reference vfh.cpp compiled here into oracle/_ref and
 *     against committed golden vectors generated from it (tests/golden/vfh_*.json).
 *   - HIMM, getRangesFromSubmap, RRT, graph A*: restated line by line; the reference has no test
 *     or buildable binary for them here -> "parity unpinned" (only RRT's glibc rand() replica is
 *     pinned, against this libc).
 *   - grid A*: no reference implementation exists (reference A* is a 9-vertex waypoint graph);
 *     the oracle *defines* the contract (DESIGN.md "Grid A* contract").
 */
#ifndef RNA_ORACLE_H
#define RNA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry of a grid_map::GridMap (gmc/include/grid_map_core/GridMap.hpp:493-516) ---- */
typedef struct {
  double len[2];   /* length_  (x, y) in metres                */
  double pos[2];   /* position_ of the map centre              */
  double res;      /* resolution_                              */
  int    size[2];  /* size_ (rows = Index(0), cols = Index(1)) */
  int    start[2]; /* startIndex_ of the circular buffer       */
} og_geom;

/* gmc/src/GridMap.cpp:51-70 */
void og_set_geometry(og_geom* g, double len_x, double len_y, double res, double pos_x, double pos_y);

/* gmc/src/GridMapMath.cpp:115-239, 414-488 */
int  og_position_from_index(const og_geom* g, const int idx[2], double pos[2]);
int  og_index_from_position(const og_geom* g, const double pos[2], int idx[2]);
int  og_position_within_map(const double pos[2], const double len[2], const double mpos[2]);
void og_index_shift_from_position_shift(const double shift[2], double res, int out[2]);
void og_position_shift_from_index_shift(const int shift[2], double res, double out[2]);
int  og_index_within_range(const int idx[2], const int size[2]);
int  og_wrap_index(int idx, int size);
void og_limit_position_to_range(double pos[2], const double len[2], const double mpos[2]);
void og_unwrap_index(const int bidx[2], const int size[2], const int start[2], int out[2]);
void og_buffer_index(const int idx[2], const int size[2], const int start[2], int out[2]);
int  og_increment_index(int idx[2], const int size[2], const int start[2]);
int  og_increment_index_for_submap(int sub_idx[2], int idx[2], const int sub_tl[2],
                                   const int sub_size[2], const int size[2], const int start[2]);
void og_index_from_linear(size_t lin, const int size[2], int row_major, int out[2]);
size_t og_linear_from_index(const int idx[2], const int size[2], int row_major);

/* gmc/src/GridMapMath.cpp:246-296 */
typedef struct {
  int    top_left[2];      /* buffer index of the submap's top-left cell */
  int    size[2];
  double pos[2];
  double len[2];
  int    requested_index[2];
} og_submap_info;
int og_submap_information(const og_geom* g, const double req_pos[2], const double req_len[2],
                          og_submap_info* out);

/* gmc/src/GridMapMath.cpp:306-412 ; quadrant codes follow BufferRegion::Quadrant order
 * (0 Undefined, 1 TopLeft, 2 TopRight, 3 BottomLeft, 4 BottomRight) */
typedef struct { int index[2]; int size[2]; int quadrant; } og_region;
int og_buffer_regions_for_submap(const int sub_index[2], const int sub_size[2], const int size[2],
                                 const int start[2], og_region out[4]); /* returns count, -1 = false */

/* gmc/src/GridMap.cpp:287-339 : copy a submap of one column-major layer.
 * sub_out must hold info.size[0]*info.size[1] floats.  Returns 1 on success. */
int og_get_submap(const og_geom* g, const float* layer, const double pos[2], const double len[2],
                  og_geom* sub_geom, float* sub_out, int sub_cap);

/* gmc/src/GridMap.cpp:346-412 : recentre (circular buffer). layers = n_layers column-major
 * buffers that all get their dropped rows/cols set to NaN.  regions_out (cap 4) receives the
 * cleared regions in the reference's order; returns the number of regions, moved flag in *moved. */
int og_move(og_geom* g, float** layers, int n_layers, const double new_pos[2], og_region* regions_out,
            int* moved);

/* gmc/src/iterators/LineIterator.cpp:16-150.  Writes up to cap (i,j) pairs; returns the number of
 * cells of the line (0 when the line misses the map). */
int og_line_cells(const og_geom* g, const double start[2], const double end[2], int* cells, int cap);
/* Index-based constructor (LineIterator.cpp:25-28). */
int og_line_cells_index(const int start[2], const int end[2], int* cells, int cap);

/* gmc/src/iterators/CircleIterator.cpp:16-93 (+ SubmapIterator.cpp:28-83). */
int og_circle_cells(const og_geom* g, const double center[2], double radius, int* cells, int cap);
/* gmc/src/iterators/SubmapIterator.cpp ; lists buffer indices + submap indices (4 ints / cell). */
int og_submap_cells(const og_geom* g, const int tl[2], const int size[2], int* out, int cap);

/* ---- HIMM (mc/include/move_control/map_updater.h:28-71) ---- */
typedef struct {
  double sx, sy;      /* RangeSample.start */
  double ex, ey;      /* RangeSample.end   */
  int32_t clear_end;  /* RangeSample.ifClearEnd */
  int32_t _pad;
} og_ray;
float og_himm_clear(float v);                     /* map_updater.h:61-71 */
float og_himm_mark(float v);                      /* map_updater.h:52-59 */
/* lineOnMap for every ray in order (laser_map_updater.cpp:7-21); bbox[4]=minX,minY,maxX,maxY is
 * grown by touch() exactly as the reference does (and, like there, never used). */
void og_himm_update(const og_geom* g, float* layer, const og_ray* rays, int n, double bbox[4]);
/* MapProvider::composeMasterMapFromLayerdMap (map_provider.cpp:216-223): master = laser */
void og_compose_master(float* master, const float* laser, size_t n_cells);

/* ---- Steerer::getRangesFromSubmap (mc/src/steerer.cpp:147-191) ---- */
/* ranges[361][2]; only [i][0] is written. master is the full map's "master" layer. */
int og_ranges_from_submap(const og_geom* g, const float* master, const double robot_pos[2],
                          double yaw, double ranges[361][2]);

/* ---- VFH+ (mc/src/vfh.cpp, mc/include/move_control/vfh.h) ---- */
typedef struct {
  double cell_size;
  int    window_diameter;
  int    sector_angle;
  double safety_dist_0ms, safety_dist_1ms;
  int    max_speed, max_speed_narrow_opening, max_speed_wide_opening;
  int    max_acceleration, min_turnrate, max_turnrate_0ms, max_turnrate_1ms;
  double min_turn_radius_safety_factor;
  double free_space_cutoff_0ms, obs_cutoff_0ms, free_space_cutoff_1ms, obs_cutoff_1ms;
  double weight_desired_dir, weight_current_dir;
  double robot_radius;            /* SetRobotRadius() before Init(), steerer.cpp:131 */
} og_vfh_params;
void og_vfh_default_params(og_vfh_params* p);     /* Steerer::initVfh defaults, steerer.cpp:69-121 */

typedef struct og_vfh og_vfh;
og_vfh* og_vfh_create(const og_vfh_params* p);    /* ctor + SetRobotRadius + Init */
void    og_vfh_destroy(og_vfh* v);
/* Update_VFH with the wall-clock delta passed in explicitly (vfh.cpp:519-531 uses gettimeofday) */
int og_vfh_update(og_vfh* v, double ranges[361][2], int current_speed, float goal_direction,
                  float goal_distance, float goal_distance_tolerance, double dt_seconds,
                  int* chosen_speed, int* chosen_turnrate);
int          og_vfh_hist_size(const og_vfh* v);
const float* og_vfh_hist(const og_vfh* v);        /* Hist (binary, then masked) */
const float* og_vfh_origin_hist(const og_vfh* v); /* OriginHist */
float        og_vfh_picked_angle(const og_vfh* v);
float        og_vfh_last_picked_angle(const og_vfh* v);
int          og_vfh_max_speed_for_picked_angle(const og_vfh* v);
/* table access for host/device table checks */
int          og_vfh_num_tables(const og_vfh* v);
const float* og_vfh_cell_direction(const og_vfh* v); /* [x*W + y] */
const float* og_vfh_cell_dist(const og_vfh* v);
const float* og_vfh_cell_base_mag(const og_vfh* v);
int          og_vfh_cell_sector_count(const og_vfh* v, int table, int x, int y);
const int*   og_vfh_cell_sector_list(const og_vfh* v, int table, int x, int y);
int          og_vfh_min_turning_radius(const og_vfh* v, int speed);

/* One full Steerer step for a pose over a map: ranges from submap, then Update_VFH. */
int og_vfh_step_pose(og_vfh* v, const og_geom* g, const float* master, const double robot_pos[2],
                     double yaw, int current_speed, float goal_direction, float goal_distance,
                     float goal_tol, double dt, int* chosen_speed, int* chosen_turnrate);

/* ---- grid A* oracle (build-defined contract) ---- */
#define OG_ASTAR_COST_STRAIGHT 1000
#define OG_ASTAR_COST_DIAG     1414
#define OG_ASTAR_INF           0x7fffffff
/* blocked iff finite and > 0 (GlobalPlanner::ifBlocked predicate, map_global_planner.h:47-50) */
void og_astar_blocked_mask(const float* master, size_t n, uint8_t* blocked);
/* 8-bit traversable-neighbour mask per cell; bit k <-> neighbour k in the fixed order
 * (di,dj) = (-1,-1),(0,-1),(1,-1),(-1,0),(1,0),(-1,1),(0,1),(1,1) (ascending linear index) */
void og_astar_nbr_mask(const uint8_t* blocked, int rows, int cols, uint8_t* nbr);
typedef struct {
  int32_t status;      /* 0 found, 1 no path, 2 start/goal invalid or blocked, 3 path_cap too small */
  int32_t path_len;    /* number of cells start..goal inclusive */
  int32_t cost;        /* g(goal) */
  int32_t settled;     /* E: cells with f <= f* that were closed */
} og_astar_result;
/* g_work: rows*cols int32 scratch (filled with the final g field); path: linear indices i + j*rows */
void og_astar_query(const uint8_t* nbr, int rows, int cols, int start_lin, int goal_lin,
                    int32_t* g_work, int32_t* path, int path_cap, og_astar_result* res);

/* cells the calling thread's last og_astar_query had closed when the goal came off the heap (a measurement aid) */
int32_t og_astar_last_settled_at_goal(void);

/* the same contract in map space for a moved (circular-buffer) map: start/goal/path are buffer linear
 * indices, adjacency and tie-breaking use unwrapped indices; g_work is left in unwrapped order */
void og_astar_query_on_map(const og_geom* g, const float* master, int start_lin, int goal_lin, int32_t* g_work,
                           int32_t* path, int path_cap, og_astar_result* res);

/* ---- 9-vertex waypoint-graph A* (mc/src/astar_planner.cpp:63-145) ---- */
/* Generic small-graph version with BGL astar_search semantics (boost/graph/astar_search.hpp,
 * Boost 1.54 as shipped with ROS Indigo): undirected CSR graph, float edge weights.
 * Returns number of vertices on the path (0 = goal not reached). */
int og_graph_astar(int n_vertices, const double* loc_xy, int n_edges, const int* edge_uv,
                   const float* edge_w, int start, int goal, int* path_vertices, int cap);
int og_graph_closest_vertex(int n_vertices, const double* loc_xy, const double pos[2]);
/* The reference's hard-coded graph (astar_planner.cpp:98-127) and makePlan() path shape. */
int og_reference_graph(double* loc_xy /*18*/, int* edge_uv /*20*/);
int og_graph_make_plan(const double start[2], const double target[2], double* path_xy, int cap);

/* ---- RRT (mc/src/rrt_planner.cpp, mc/include/move_control/map_global_planner.h) ---- */
typedef struct { int32_t r[34]; int f, b; } og_rand_state; /* glibc TYPE_3 random() replica */
void og_srand(og_rand_state* s, unsigned seed);
int  og_rand(og_rand_state* s);
int  og_if_blocked(const og_geom* g, const float* master, const double p[2]); /* map_global_planner.h:39-54 */
typedef struct {
  int32_t status;     /* 1 = makePlan returned true, 0 = false (iteration budget), -1 = sample budget */
  int32_t path_len;   /* positions, goal -> start order as in the reference */
  int32_t tree_size;
  int32_t samples;    /* extendTree loop iterations in total */
} og_rrt_result;
/* max_samples bounds the reference's unbounded while(true) in extendTree (rrt_planner.cpp:28). */
void og_rrt_plan(const og_geom* g, const float* master, const double start[2], const double target[2],
                 double close_tol, unsigned seed, int max_samples, double* path_xy, int path_cap,
                 og_rrt_result* res);
/* the same with the steering step's formulation chosen (oracle/rrt.c): 0 = the reference's atan2 / cos / sin, 1 = normalised offset */
void og_rrt_plan_steer(const og_geom* g, const float* master, const double start[2], const double target[2],
                       double close_tol, unsigned seed, int max_samples, int steer, double* path_xy, int path_cap,
                       og_rrt_result* res);

/* ---- laser ingestion: LaserScan -> RangeSamples (scan.c; laser_map_updater.cpp:37-143) ---- */
typedef struct {
  float angle_min, angle_max, angle_increment, range_min, range_max;   /* sensor_msgs/LaserScan fields */
  int32_t n_ranges;
  int64_t ranges_offset;   /* first range of this scan in the concatenated ranges array */
  double x, y, yaw;        /* sensor pose in the map frame (tf) at header.stamp */
  double x_end, y_end, yaw_end;   /* ... at stamp + (projected beams - 1) * time_increment */
} og_scan;
int og_simplify_scan(int n, float angle_increment, int* sel, int cap, float* out_increment);
int og_scan_to_rays(const og_scan* s, const float* ranges, og_ray* out, int cap);   /* returns the number of rays */
typedef struct {
  float angle_min, angle_max, angle_increment, range_min, range_max;
  int32_t n_ranges;
  int64_t ranges_offset;
  double t[3], q[4];           /* tf: sensor frame in the map frame at header.stamp (translation, quaternion x y z w) */
  double t_end[3], q_end[4];   /* ... at the end time */
} og_scan_tf;
int og_scan_to_rays_tf(const og_scan_tf* s, const float* ranges, og_ray* out, int cap);
/* sonar: RangeMapUpdater::bufferIncomingMsg (range_map_updater.cpp:38-76) */
void og_range_to_ray(float range, float max_range, double x, double y, double yaw, og_ray* out);
void og_range_to_ray_tf(float range, float max_range, const double t[3], const double q[4], og_ray* out);

/* ---- message formats either side of the path (msgs.c) ---- */
/* GridMapRosConverter::toOccupancyGrid / fromOccupancyGrid (grid_map_ros/src/GridMapRosConverter.cpp:205-287) */
void og_to_occupancy_grid(const og_geom* g, const float* layer, float data_min, float data_max, int8_t* out);
void og_from_occupancy_grid(int rows, int cols, const int8_t* data, float* layer);
/* Steerer::pubHist (steerer.cpp:201-220); returns num_bin */
int og_hist_msg(const float* hist, const float* origin_hist, int hist_size, int sector_angle, uint16_t* x_data,
                uint16_t* y_data, uint16_t* y_bin_data, uint16_t thresholds[2]);
/* Nav::taileredPlan (nav_node.cpp:192-204); returns the number of kept positions */
int og_tailor_plan(const double* plan_xy, int n, unsigned stride, double* out_xy);
/* mc/src/steerer.cpp:27-33,222-256 */
int og_follow_plan(const double* plan_xy, int n, int* plan_index, double x, double y, double yaw, float out[2]);

#ifdef __cplusplus
}
#endif
#endif
