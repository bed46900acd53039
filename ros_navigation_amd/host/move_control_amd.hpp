// move_control_amd.hpp -- C++ host-side mirror of the reference's planner-facing classes, implemented
// over the C ABI of include/rna.h (librna.so).  Header-only, ROS-free, C++11.
//
// Same names, argument meaning and error behaviour as the reference so that the node code of
// move_control (nav_graph_node.cpp, nav_node.cpp, steerer.cpp) can switch by changing includes:
//
//   grid_map::Position / Index / Length / Size   gmc/include/grid_map_core/TypeDefs.hpp:16-25
//   grid_map::GridMap (subset used by mc/)        gmc/include/grid_map_core/GridMap.hpp:39-520
//   move_control::VFH                             mc/include/move_control/vfh.h:182-361
//   move_control::MapProvider (update/compose/getMap/getSubMap core, no ROS I/O)
//                                                 mc/include/move_control/map_provider.h:21-29
//   move_control::AStarPlanner                    mc/include/move_control/astar_planner.h:30-44
//   move_control::GridAStarPlanner                (new: grid A* of BASELINE.json, same makePlan shape)
//   move_control::RrtPlanner                      mc/include/move_control/rrt_planner.h:7-41
//
// (mc/ = move_control, gmc/ = grid_map-master/grid_map_core in the reference tree.)
// Data lives on the GPU inside one rna_engine; these classes own no algorithmic code.
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rna.h"

namespace grid_map {

// Minimal stand-ins for the Eigen typedefs of TypeDefs.hpp (x()/y()/operator[] like Eigen::Vector2d)
struct Position {
  double v[2];
  Position() : v{0, 0} {}
  Position(double x, double y) : v{x, y} {}
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
};
typedef Position Length;
struct Index {
  int v[2];
  Index() : v{0, 0} {}
  Index(int i, int j) : v{i, j} {}
  int& operator[](int i) { return v[i]; }
  int operator[](int i) const { return v[i]; }
  int& operator()(int i) { return v[i]; }
  int operator()(int i) const { return v[i]; }
};
typedef Index Size;

inline void rna_check(int rc, rna_engine* e, const char* what) {
  if (rc != RNA_OK) throw std::runtime_error(std::string(what) + ": " + (e ? rna_last_error(e) : "rna error"));
}

// grid_map::Matrix stand-in: a host copy of one layer (column-major, linear = i + j*rows as Eigen::MatrixXf),
// what GridMap::operator[] hands out here (copy-out; write a changed copy back with GridMap::set).
class Matrix {
 public:
  Matrix() : rows_(0), cols_(0) {}
  Matrix(int rows, int cols, float v = std::numeric_limits<float>::quiet_NaN()) : rows_(rows), cols_(cols), d_((size_t)rows * cols, v) {}
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  size_t size() const { return d_.size(); }
  float& operator()(int i, int j) { return d_[(size_t)j * rows_ + i]; }
  float operator()(int i, int j) const { return d_[(size_t)j * rows_ + i]; }
  float& operator()(size_t linear) { return d_[linear]; }
  float operator()(size_t linear) const { return d_[linear]; }
  void setConstant(float v) { d_.assign(d_.size(), v); }
  const float* data() const { return d_.data(); }
  float* data() { return d_.data(); }
  const std::vector<float>& values() const { return d_; }
 private:
  int rows_, cols_;
  std::vector<float> d_;
};

// GridMap: geometry + the three layers MapProvider uses ("master", "laser", "range"), resident in HBM.
class GridMap {
 public:
  GridMap() : e_(nullptr), frameId_() {}
  // GridMap(const std::vector<std::string>& layers) (gmc/src/GridMap.cpp:27-40): the engine always carries the three
  // layers of move_control; any other name is refused as GridMap::get refuses it
  explicit GridMap(const std::vector<std::string>& layers) : e_(nullptr), frameId_() {
    for (size_t k = 0; k < layers.size(); ++k) (void)layerId(layers[k]);
  }
  ~GridMap() { if (e_) rna_destroy(e_); }
  // copy = deep copy (`map = map_` of MapProvider::getMap, map_provider.cpp:120-125)
  GridMap(const GridMap& o) : e_(nullptr), frameId_(o.frameId_) {
    if (o.e_) rna_check(rna_clone(o.e_, &e_), o.e_, "GridMap copy");
  }
  GridMap& operator=(const GridMap& o) {
    if (this != &o) {
      rna_engine* c = nullptr;
      if (o.e_) rna_check(rna_clone(o.e_, &c), o.e_, "GridMap copy");
      if (e_) rna_destroy(e_);
      e_ = c;
      frameId_ = o.frameId_;
    }
    return *this;
  }
  GridMap(GridMap&& o) : e_(o.e_), frameId_(o.frameId_) { o.e_ = nullptr; }
  GridMap& operator=(GridMap&& o) {   // `map = map_.getSubmap(...)` of MapProvider::getSubMap (map_provider.cpp:97)
    if (this != &o) { if (e_) rna_destroy(e_); e_ = o.e_; o.e_ = nullptr; frameId_ = o.frameId_; }
    return *this;
  }
  void setFrameId(const std::string& frameId) { frameId_ = frameId; }
  const std::string& getFrameId() const { return frameId_; }
  // GridMap::add(layer, value) (gmc/src/GridMap.cpp:77-95) for the layers the engine carries
  void add(const std::string& layer, float value = std::numeric_limits<float>::quiet_NaN()) {
    rna_check(rna_layer_fill(e_, layerId(layer), value), e_, "GridMap::add");
  }
  // GridMap::operator[] / get (gmc/src/GridMap.cpp:125-151): std::out_of_range for unknown layers; copy-out
  Matrix operator[](const std::string& layer) const {
    const int id = layerId(layer);
    rna_geometry g = geometry();
    Matrix m(g.size[0], g.size[1]);
    rna_check(rna_layer_download(e_, id, m.data(), m.size()), e_, "GridMap::operator[]");
    return m;
  }
  void set(const std::string& layer, const Matrix& m) {
    rna_check(rna_layer_upload(e_, layerId(layer), m.data(), m.size()), e_, "GridMap::set");
  }
  Index getStartIndex() const { rna_geometry g = geometry(); return Index(g.start_index[0], g.start_index[1]); }

  // GridMap::setGeometry (gmc/src/GridMap.cpp:51-70)
  void setGeometry(const Length& length, double resolution, const Position& position = Position(0.0, 0.0), int device = 0) {
    if (e_) { rna_destroy(e_); e_ = nullptr; }
    if (rna_abi_version() != RNA_ABI_VERSION) throw std::runtime_error("librna.so was built from another include/rna.h (ABI version)");
    int rc = rna_create(&e_, length[0], length[1], resolution, position[0], position[1], device);
    if (rc != RNA_OK) throw std::runtime_error("GridMap::setGeometry: rna_create failed (no MI355X / bad geometry)");
  }
  static int layerId(const std::string& layer) {
    if (layer == "master") return RNA_LAYER_MASTER;
    if (layer == "laser") return RNA_LAYER_LASER;
    if (layer == "range") return RNA_LAYER_RANGE;
    // GridMap::get throws std::out_of_range for unknown layers (gmc/src/GridMap.cpp:125-141)
    throw std::out_of_range("GridMap::get(...) : No map layer '" + layer + "' available.");
  }
  bool exists(const std::string& layer) const { return layer == "master" || layer == "laser" || layer == "range"; }
  rna_geometry geometry() const { rna_geometry g; rna_check(rna_get_geometry(e_, &g), e_, "geometry"); return g; }
  Length getLength() const { rna_geometry g = geometry(); return Length(g.length[0], g.length[1]); }
  Position getPosition() const { rna_geometry g = geometry(); return Position(g.position[0], g.position[1]); }
  Size getSize() const { rna_geometry g = geometry(); return Size(g.size[0], g.size[1]); }
  double getResolution() const { return geometry().resolution; }
  // getIndex / getPosition / isInside (gmc/src/GridMap.cpp:227-240)
  bool getIndex(const Position& p, Index& idx) const {
    int32_t o[2];
    if (rna_get_index(e_, p[0], p[1], o) != 1) return false;
    idx = Index(o[0], o[1]);
    return true;
  }
  bool getPosition(const Index& idx, Position& p) const {
    double o[2];
    if (rna_get_position(e_, idx[0], idx[1], o) != 1) return false;
    p = Position(o[0], o[1]);
    return true;
  }
  bool isInside(const Position& p) const { Index i; return getIndex(p, i); }
  bool move(const Position& p) { int moved = 0; rna_check(rna_move(e_, p[0], p[1], &moved), e_, "move"); return moved != 0; }
  // whole-layer access (column-major, linear = i + j*rows) replacing operator[] copies
  void set(const std::string& layer, const std::vector<float>& data) {
    rna_check(rna_layer_upload(e_, layerId(layer), data.data(), data.size()), e_, "GridMap::set");
  }
  std::vector<float> get(const std::string& layer) const {
    rna_geometry g = geometry();
    std::vector<float> out((size_t)g.size[0] * g.size[1]);
    rna_check(rna_layer_download(e_, layerId(layer), out.data(), out.size()), e_, "GridMap::get");
    return out;
  }
  // GridMap::getSubmap(position, length, isSuccess) (gmc/src/GridMap.cpp:287-339): a GridMap of its own (new engine on
  // the same device, all layers gathered on the device, startIndex (0,0)); an empty GridMap when isSuccess is false.
  GridMap getSubmap(const Position& position, const Length& length, bool& isSuccess) const {
    GridMap sub;
    const int rc = rna_create_submap(e_, position[0], position[1], length[0], length[1], &sub.e_);
    if (rc < 0) rna_check(rc, e_, "GridMap::getSubmap");
    isSuccess = rc == 1;
    return sub;
  }
  // One layer of the same submap as host data with its geometry (no new engine)
  struct SubMap {
    Length length; Position position; Size size; double resolution;
    std::vector<float> data;   // column-major size[0] x size[1]
    float at(int i, int j) const { return data[(size_t)j * size[0] + i]; }
  };
  SubMap getSubmapData(const Position& position, const Length& length, bool& isSuccess, const std::string& layer = "master") const {
    SubMap sm;
    rna_geometry g = geometry();
    sm.resolution = g.resolution;
    size_t cap = ((size_t)std::ceil(length[0] / g.resolution) + 2) * ((size_t)std::ceil(length[1] / g.resolution) + 2);
    const size_t all = (size_t)g.size[0] * g.size[1];
    if (cap > all || cap == 0) cap = all;
    sm.data.resize(cap);
    rna_submap_info info;
    const int rc = rna_get_submap(e_, layerId(layer), position[0], position[1], length[0], length[1], sm.data.data(), cap, &info);
    if (rc < 0) rna_check(rc, e_, "GridMap::getSubmapData");
    isSuccess = rc == 1;
    if (!isSuccess) { sm.data.clear(); return sm; }
    sm.length = Length(info.length[0], info.length[1]);
    sm.position = Position(info.position[0], info.position[1]);
    sm.size = Size(info.size[0], info.size[1]);
    sm.data.resize((size_t)info.size[0] * info.size[1]);
    return sm;
  }
  rna_engine* engine() const { return e_; }

 private:
  rna_engine* e_;
  std::string frameId_;
};

// grid_map_core's iterators over a GridMap's geometry (host side; the same walks the kernels do, through
// rna_line_cells / rna_circle_cells / rna_submap_cells).  Usage as in the reference:
//   for (LineIterator it(map, start, end); !it.isPastEnd(); ++it) { const Index& index = *it; ... }
class CellListIterator {
 public:
  bool isPastEnd() const { return k_ >= n_; }
  CellListIterator& operator++() { ++k_; return *this; }
  const Index& operator*() const { cur_ = Index(cells_[2 * k_], cells_[2 * k_ + 1]); return cur_; }
  bool operator!=(const CellListIterator& o) const { return k_ != o.k_; }
  size_t size() const { return n_; }
 protected:
  CellListIterator() : k_(0), n_(0) {}
  template <class F> void fill(F&& walk) {
    int cap = 1024;
    for (;;) {
      cells_.resize(2 * (size_t)cap);
      const int n = walk(cells_.data(), cap);
      if (n < 0) throw std::invalid_argument("grid_map iterator: bad geometry");
      if (n <= cap) { n_ = (size_t)n; return; }
      cap = n;
    }
  }
  std::vector<int32_t> cells_;
  size_t k_, n_;
  mutable Index cur_;
};
class LineIterator : public CellListIterator {      // gmc/src/iterators/LineIterator.cpp:16-150
 public:
  LineIterator(const GridMap& map, const Position& start, const Position& end) {
    const rna_geometry g = map.geometry();
    fill([&](int32_t* c, int cap) { return rna_line_cells(&g, start[0], start[1], end[0], end[1], c, cap); });
  }
};
class CircleIterator : public CellListIterator {    // gmc/src/iterators/CircleIterator.cpp:16-93
 public:
  CircleIterator(const GridMap& map, const Position& center, double radius) {
    const rna_geometry g = map.geometry();
    fill([&](int32_t* c, int cap) { return rna_circle_cells(&g, center[0], center[1], radius, c, cap); });
  }
};
class SubmapIterator : public CellListIterator {    // gmc/src/iterators/SubmapIterator.cpp:28-83
 public:
  SubmapIterator(const GridMap& map, const Index& submapStartIndex, const Size& submapSize) {
    const rna_geometry g = map.geometry();
    const int32_t tl[2] = {submapStartIndex[0], submapStartIndex[1]}, sz[2] = {submapSize[0], submapSize[1]};
    fill([&](int32_t* c, int cap) { return rna_submap_cells(&g, tl, sz, c, cap); });
  }
};
class GridMapIterator {                             // gmc/src/iterators/GridMapIterator.cpp:14-83: linear order of the buffer
 public:
  explicit GridMapIterator(const GridMap& map) : k_(0) { const Size s = map.getSize(); rows_ = s[0]; n_ = (size_t)s[0] * s[1]; }
  bool isPastEnd() const { return k_ >= n_; }
  GridMapIterator& operator++() { ++k_; return *this; }
  const Index& operator*() const { cur_ = Index((int)(k_ % rows_), (int)(k_ / rows_)); return cur_; }
  size_t getLinearIndex() const { return k_; }
 private:
  size_t k_, n_;
  int rows_;
  mutable Index cur_;
};

}  // namespace grid_map

// With RNA_REFERENCE_SIGNATURES (set by move_control_api.hpp) the compute cores below live in move_control::core and
// move_control itself holds the classes with the reference's exact constructor / method signatures.
#ifdef RNA_REFERENCE_SIGNATURES
namespace move_control { namespace core {
#else
namespace move_control {
#endif

using grid_map::GridMap;
using grid_map::Length;
using grid_map::Position;

// MapUpdater::RangeSample (mc/include/move_control/map_updater.h:28-32)
struct RangeSample {
  Position start, end;
  bool ifClearEnd;
};

// The fields of nav_msgs/OccupancyGrid that GridMapRosConverter::toOccupancyGrid fills
// (grid_map_ros/src/GridMapRosConverter.cpp:251-287); header/stamp stay with the ROS shim.
struct OccupancyGrid {
  float resolution;
  unsigned width, height;          // info.width = size(0), info.height = size(1)
  double origin_x, origin_y;       // position - length/2
  std::vector<int8_t> data;        // -1 unknown, 0..100
};

// move_control/Histogram.msg as Steerer::pubHist fills it (mc/src/steerer.cpp:201-220)
struct Histogram {
  uint8_t num_bin;
  std::vector<uint16_t> xData, yData, yBinData;
  uint16_t yLowThreshold, yHighThreshold;
};

// The compute core of MapProvider (mc/src/map_provider.cpp:190-223): buffered samples are applied
// in arrival order (laser_map_updater.cpp:7-21) and master is composed from the laser layer.
class MapProvider {
 public:
  explicit MapProvider(Length mapLength = Length(30, 30), double resolution = 0.05, int device = 0) {
    map_.setGeometry(mapLength, resolution, Position(0.0, 0.0), device);  // initParameter/initMap, :130-149
  }
  // typeName selects the MapUpdater as MapProvider's factory does (map_provider.cpp:12-15,262-266): "laser" or "range"
  void bufferSample(const RangeSample& s, const std::string& typeName = "laser") {  // bufferIncomingMsg's push_back
    rna_ray r;
    r.sx = s.start[0]; r.sy = s.start[1]; r.ex = s.end[0]; r.ey = s.end[1];
    r.clear_end = s.ifClearEnd ? 1 : 0; r._pad = 0;
    if (typeName == "laser") buffer_.push_back(r);
    else if (typeName == "range") range_buffer_.push_back(r);
    else throw std::invalid_argument("MapProvider: unknown MapUpdater type '" + typeName + "'");
  }
  // RangeMapUpdater::bufferIncomingMsg (range_map_updater.cpp:38-76) for sonar readings with their tf pose
  void bufferRanges(const std::vector<rna_range_reading>& readings) {
    std::vector<rna_ray> rays(readings.size());
    if (rna_range_to_rays(readings.data(), (int)readings.size(), rays.data()) != RNA_OK) throw std::invalid_argument("bufferRanges");
    range_buffer_.insert(range_buffer_.end(), rays.begin(), rays.end());
  }
  // ... and with the sensor's full tf pose (a tilted or raised mount)
  void bufferRanges(const std::vector<rna_range_reading_tf>& readings) {
    std::vector<rna_ray> rays(readings.size());
    if (rna_range_to_rays_tf(readings.data(), (int)readings.size(), rays.data()) != RNA_OK) throw std::invalid_argument("bufferRanges");
    range_buffer_.insert(range_buffer_.end(), rays.begin(), rays.end());
  }
  // LaserMapUpdater::bufferIncomingMsg (laser_map_updater.cpp:37-75) for whole scans: decimation, projection and the
  // tf transform of every beam (interpolated between the scan's start and end) run on the device -- for planar sensor
  // poses (x, y, yaw) and, below, for the full transforms tf reports
  void bufferScans(const std::vector<rna_laser_scan>& scans, const std::vector<float>& ranges) {
    size_t cap = 1;
    for (size_t k = 0; k < scans.size(); ++k) cap += (size_t)scans[k].n_ranges;
    std::vector<rna_ray> rays(cap);
    int n = 0;
    grid_map::rna_check(rna_scan_to_rays(map_.engine(), scans.data(), (int)scans.size(), ranges.data(), ranges.size(), rays.data(),
                                         (int)cap, &n), map_.engine(), "LaserMapUpdater::bufferIncomingMsg");
    buffer_.insert(buffer_.end(), rays.begin(), rays.begin() + n);
  }
  void bufferScans(const std::vector<rna_laser_scan_tf>& scans, const std::vector<float>& ranges) {
    size_t cap = 1;
    for (size_t k = 0; k < scans.size(); ++k) cap += (size_t)scans[k].n_ranges;
    std::vector<rna_ray> rays(cap);
    int n = 0;
    grid_map::rna_check(rna_scan_to_rays_tf(map_.engine(), scans.data(), (int)scans.size(), ranges.data(), ranges.size(), rays.data(),
                                            (int)cap, &n), map_.engine(), "LaserMapUpdater::bufferIncomingMsg");
    buffer_.insert(buffer_.end(), rays.begin(), rays.begin() + n);
  }
  // MapProvider::updateMap: drain the buffer through HIMM, then compose master (fused, dirty tiles)
  void updateMap(bool wholeLayerCopy = false) {
    // every updater writes its own layer (map_updater.h:12-13); master is composed from "laser" alone (:216-223)
    grid_map::rna_check(rna_himm_update(map_.engine(), RNA_LAYER_RANGE, range_buffer_.data(), (int)range_buffer_.size()),
                        map_.engine(), "RangeMapUpdater::updateMap");
    range_buffer_.clear();
    grid_map::rna_check(rna_update_map(map_.engine(), buffer_.data(), (int)buffer_.size(), wholeLayerCopy ? 1 : 0),
                        map_.engine(), "MapProvider::updateMap");
    buffer_.clear();
  }
  GridMap& getMap() { return map_; }
  // MapProvider::publishMap (:113-118,206-213): toOccupancyGrid(map, "master", 0.0, 255.0, msg) on the device
  void publishMap(OccupancyGrid& msg) { publishMap(map_, msg); }
  // MapProvider::publishMap(GridMap&) (:113-118): the "local_map" topic, e.g. the planning window of Nav::makePlan
  void publishMap(GridMap& map, OccupancyGrid& msg) {
    rna_geometry g;
    grid_map::rna_check(rna_get_geometry(map.engine(), &g), map.engine(), "MapProvider::publishMap");
    msg.resolution = (float)g.resolution;
    msg.width = (unsigned)g.size[0]; msg.height = (unsigned)g.size[1];
    msg.origin_x = g.position[0] - 0.5 * g.length[0]; msg.origin_y = g.position[1] - 0.5 * g.length[1];
    msg.data.resize((size_t)g.size[0] * g.size[1]);
    grid_map::rna_check(rna_to_occupancy_grid(map.engine(), RNA_LAYER_MASTER, 0.0f, 255.0f, msg.data.data()), map.engine(),
                        "MapProvider::publishMap");
  }
  // MapProvider::getSubMap (:93-100)
  bool getSubMap(GridMap& map, const Position& center, const Length& length) {
    bool ok = false;
    map = map_.getSubmap(center, length, ok);
    return ok;
  }
  bool getSubMap(GridMap::SubMap& map, const Position& center, const Length& length, const std::string& layer = "master") {
    bool ok = false;
    map = map_.getSubmapData(center, length, ok, layer);
    return ok;
  }
  bool ifCloseToPostion(const Position& robot, const Position& pos, double tolerance) const {  // :102-111
    return std::hypot(pos[0] - robot[0], pos[1] - robot[1]) < tolerance;
  }

 private:
  GridMap map_;
  std::vector<rna_ray> buffer_, range_buffer_;
};

// VFH with the reference's constructor signature and Update_VFH contract (vfh.h:185-253).  One
// instance drives `robots` independent stateful VFH instances on the GPU (robots = 1 is the drop-in).
class VFH {
 public:
  VFH(double cell_size, int window_diameter, int sector_angle, double safety_dist_0ms, double safety_dist_1ms,
      int max_speed, int max_speed_narrow_opening, int max_speed_wide_opening, int max_acceleration, int min_turnrate,
      int max_turnrate_0ms, int max_turnrate_1ms, double min_turn_radius_safety_factor, double free_space_cutoff_0ms,
      double obs_cutoff_0ms, double free_space_cutoff_1ms, double obs_cutoff_1ms, double weight_desired_dir,
      double weight_current_dir)
      : map_(nullptr), robots_(1), picked_(90.0f) {
    p_.cell_size = cell_size; p_.window_diameter = window_diameter; p_.sector_angle = sector_angle;
    p_.safety_dist_0ms = safety_dist_0ms; p_.safety_dist_1ms = safety_dist_1ms; p_.max_speed = max_speed;
    p_.max_speed_narrow_opening = max_speed_narrow_opening; p_.max_speed_wide_opening = max_speed_wide_opening;
    p_.max_acceleration = max_acceleration; p_.min_turnrate = min_turnrate; p_.max_turnrate_0ms = max_turnrate_0ms;
    p_.max_turnrate_1ms = max_turnrate_1ms; p_.min_turn_radius_safety_factor = min_turn_radius_safety_factor;
    p_.free_space_cutoff_0ms = free_space_cutoff_0ms; p_.obs_cutoff_0ms = obs_cutoff_0ms;
    p_.free_space_cutoff_1ms = free_space_cutoff_1ms; p_.obs_cutoff_1ms = obs_cutoff_1ms;
    p_.weight_desired_dir = weight_desired_dir; p_.weight_current_dir = weight_current_dir;
    p_.robot_radius = 0.0;
    Hist = nullptr; OriginHist = nullptr;
  }
  void SetRobotRadius(float robot_radius) { p_.robot_radius = robot_radius; }
  // Init() needs the map the instance works on (the reference passes ranges in; here the engine
  // also offers the fused map -> ranges -> VFH path)
  int Init(GridMap& map, int robots = 1) {
    map_ = &map; robots_ = robots;
    grid_map::rna_check(rna_vfh_init(map.engine(), &p_, robots), map.engine(), "VFH::Init");
    hist_.assign((size_t)robots * getHistSize(), 0.0f);
    origin_.assign((size_t)robots * getHistSize(), 0.0f);
    Hist = hist_.data(); OriginHist = origin_.data();
    return 1;
  }
  // VFH::Update_VFH (vfh.h:216-222); dt = seconds since the previous call (the reference reads gettimeofday)
  int Update_VFH(double laser_ranges[361][2], int current_speed, float goal_direction, float goal_distance,
                 float goal_distance_tolerance, int& chosen_speed, int& chosen_turnrate, double dt = 0.2) {
    rna_pose pose = {0, 0, 0, dt, current_speed, goal_direction, goal_distance, goal_distance_tolerance};
    rna_vfh_out out;
    grid_map::rna_check(rna_vfh_update_batch(map_->engine(), &laser_ranges[0][0], &pose, 1, &out, origin_.data(), hist_.data()),
                        map_->engine(), "VFH::Update_VFH");
    chosen_speed = out.chosen_speed; chosen_turnrate = out.chosen_turnrate; picked_ = out.picked_angle;
    return 1;
  }
  // Steerer::update's getRangesFromSubmap + Update_VFH for a batch of robot poses (steerer.cpp:221-270)
  int Update_VFH(const std::vector<rna_pose>& poses, std::vector<rna_vfh_out>& out) {
    out.resize(poses.size());
    grid_map::rna_check(rna_vfh_step_batch(map_->engine(), poses.data(), (int)poses.size(), out.data(), origin_.data(), hist_.data()),
                        map_->engine(), "VFH::Update_VFH(batch)");
    if (!out.empty()) picked_ = out[0].picked_angle;
    return 1;
  }
  float GetPickedAngle() const { return picked_; }
  // Steerer::pubHist (steerer.cpp:201-220) for robot 0, packed on the device from the resident histograms
  void pubHist(Histogram& msg) {
    const int bins = getHistSize() / 2;
    msg.num_bin = (uint8_t)bins;
    msg.xData.resize(bins); msg.yData.resize(bins); msg.yBinData.resize(bins);
    uint16_t th[2];
    grid_map::rna_check(rna_vfh_hist_msg_batch(map_->engine(), 1, msg.xData.data(), msg.yData.data(), msg.yBinData.data(), th),
                        map_->engine(), "Steerer::pubHist");
    msg.yLowThreshold = th[0]; msg.yHighThreshold = th[1];
  }
  int getHistSize() const { return rna_vfh_hist_size(map_->engine()); }
  int getSectorAngle() const { return p_.sector_angle; }
  float* Hist;        // public as in the reference (vfh.h:239-244); refreshed by every Update_VFH
  float* OriginHist;

 private:
  rna_vfh_params p_;
  GridMap* map_;
  int robots_;
  float picked_;
  std::vector<float> hist_, origin_;
};

// Steerer (mc/include/move_control/steerer.h:18-47, mc/src/steerer.cpp): acceptPlan + one update() per VFH period.
// The ROS side (odometry monitor, velocity / histogram publishers, the 5 Hz thread) stays with the node; update()
// takes what Steerer::update reads from them (robot pose, odom linear velocity) and returns what pubVel publishes.
class Steerer {
 public:
  Steerer(MapProvider& mapProvider, VFH& vfh) : mapProvider_(mapProvider), vfh_(vfh), ifPlanReady_(false), planIndex_(1) {}
  void acceptPlan(std::vector<Position>& plan) {   // steerer.cpp:27-33
    plan_.resize(2 * plan.size());
    for (size_t k = 0; k < plan.size(); ++k) { plan_[2 * k] = plan[k][0]; plan_[2 * k + 1] = plan[k][1]; }
    ifPlanReady_ = true;
    planIndex_ = 1;
  }
  bool ifPlanReady() const { return ifPlanReady_; }
  // steerer.cpp:222-270.  false: no plan / plan finished (nothing published, as in the reference).
  // linear_x / angular_z are the geometry_msgs/Twist fields of pubVel (steerer.cpp:193-199).
  bool update(const Position& currentPos, double currentDir, double odomLinearX, double dt, int& chosenSpeed,
              int& chosenTurnrate, double& linear_x, double& angular_z) {
    if (!ifPlanReady_) return false;
    rna_pose pose;
    const int rc = rna_follow_plan(plan_.data(), (int)(plan_.size() / 2), &planIndex_, currentPos[0], currentPos[1],
                                   currentDir, odomLinearX, dt, &pose);
    if (rc < 0) throw std::invalid_argument("Steerer::update");
    if (rc == 0) { ifPlanReady_ = false; return false; }
    std::vector<rna_pose> poses(1, pose);
    std::vector<rna_vfh_out> out;
    vfh_.Update_VFH(poses, out);        // getRangesFromSubmap + Update_VFH, fused on the device
    chosenSpeed = out[0].chosen_speed; chosenTurnrate = out[0].chosen_turnrate;
    linear_x = (float)(chosenSpeed) / 1000.0;
    angular_z = (chosenTurnrate) * M_PI / 180.0;
    return true;
  }
  int planIndex() const { return planIndex_; }

 private:
  MapProvider& mapProvider_;
  VFH& vfh_;
  std::vector<double> plan_;
  bool ifPlanReady_;
  int32_t planIndex_;
};

// AStarPlanner::makePlan over the reference's hard-coded waypoint graph (astar_planner.cpp:63-127)
class AStarPlanner {
 public:
  explicit AStarPlanner(GridMap& map) : map_(map) {  // init(), :98-127
    const int m = 8, n = 5;
    const double L[9][2] = {{0.5 * m, 0.0 * n}, {1.5 * m, 0.0 * n}, {2.5 * m, 0.0 * n}, {2.5 * m, 1.0 * n}, {1.5 * m, 1.0 * n},
                            {0.5 * m, 1.0 * n}, {0.5 * m, 2.0 * n}, {1.5 * m, 2.0 * n}, {2.5 * m, 2.0 * n}};
    const int E[10][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {0, 5}, {3, 8}};
    for (auto& l : L) { locations_.push_back(l[0]); locations_.push_back(l[1]); }
    for (auto& ed : E) { edges_.push_back(ed[0]); edges_.push_back(ed[1]); }
  }
  // appends to `path` (start, vertex locations..., target); leaves it untouched when no route exists
  bool makePlan(Position& start, Position& target, std::vector<Position>& path) {
    const double st[4] = {start[0], start[1], target[0], target[1]};
    std::vector<double> out(2 * 16);
    int32_t len = 0;
    grid_map::rna_check(rna_graph_astar_batch(map_.engine(), 9, locations_.data(), 10, edges_.data(), nullptr, st, 1,
                                              out.data(), 16, &len), map_.engine(), "AStarPlanner::makePlan");
    for (int k = 0; k < len; ++k) path.push_back(Position(out[2 * k], out[2 * k + 1]));
    return len > 0;
  }

 private:
  GridMap& map_;
  std::vector<double> locations_;
  std::vector<int32_t> edges_;
};

// Grid A* over the GridMap's master layer (BASELINE.json's planner): same makePlan shape.
class GridAStarPlanner {
 public:
  // One synchronous query at a time is the reference's usage (nav_graph_node.cpp): one search field, no
  // pipelining -- 4 B per cell of HBM instead of the batch default (256 fields x 4 stages).
  explicit GridAStarPlanner(GridMap& map, int concurrent_queries = 1) : map_(map) {
    grid_map::rna_check(rna_astar_set_pipeline_depth(map.engine(), 1), map.engine(), "GridAStarPlanner");
    grid_map::rna_check(rna_astar_configure(map.engine(), concurrent_queries, 0, 0), map.engine(), "GridAStarPlanner");
  }
  bool makePlan(Position& start, Position& target, std::vector<Position>& path) {
    grid_map::Index s, t;
    if (!map_.getIndex(start, s) || !map_.getIndex(target, t)) return false;
    const int rows = map_.getSize()[0];
    rna_astar_query q = {s[0] + s[1] * rows, t[0] + t[1] * rows};
    std::vector<int32_t> cells(1 << 16);
    rna_astar_result r;
    grid_map::rna_check(rna_astar_batch(map_.engine(), &q, 1, cells.data(), (int)cells.size(), &r), map_.engine(),
                        "GridAStarPlanner::makePlan");
    if (r.status != 0) return false;
    for (int k = 0; k < r.path_len; ++k) {
      Position p;
      map_.getPosition(grid_map::Index(cells[k] % rows, cells[k] / rows), p);
      path.push_back(p);
    }
    return true;
  }

 private:
  GridMap& map_;
};

// RrtPlanner(GridMap&, start, target, closeTolerance).makePlan(path) (rrt_planner.h:17-28): clears
// then fills `path` goal -> start; returns false (with the best-effort path) after 2000 iterations.
// Nav::taileredPlan (mc/src/nav_node.cpp:192-204): the plan handed to the Steerer keeps every
// tailerPlanStride_-th position (5, nav_node.cpp:84) of the detailed plan walked backwards, and its last one
inline void taileredPlan(const std::vector<Position>& detailedPlan, std::vector<Position>& pathPlan, unsigned stride = 5) {
  std::vector<double> in(2 * detailedPlan.size()), out(2 * detailedPlan.size());
  for (size_t k = 0; k < detailedPlan.size(); ++k) { in[2 * k] = detailedPlan[k][0]; in[2 * k + 1] = detailedPlan[k][1]; }
  int m = 0;
  if (rna_tailor_plan(in.data(), (int)detailedPlan.size(), stride, out.data(), &m) != RNA_OK) throw std::invalid_argument("taileredPlan");
  pathPlan.clear();
  for (int k = 0; k < m; ++k) pathPlan.push_back(Position(out[2 * k], out[2 * k + 1]));
}

class RrtPlanner {
 public:
  RrtPlanner(GridMap& map, Position& start, Position& target, double closeTolerance = 0.2, unsigned seed = 1)
      : map_(map), start_(start), target_(target), tol_(closeTolerance), seed_(seed) {}
  bool makePlan(std::vector<Position>& path) {
    rna_rrt_query q;
    q.start[0] = start_[0]; q.start[1] = start_[1]; q.target[0] = target_[0]; q.target[1] = target_[1];
    q.close_tolerance = tol_; q.seed = seed_; q.max_samples = 1000000;
    std::vector<double> out(2 * 2048);
    rna_rrt_result r;
    grid_map::rna_check(rna_rrt_batch(map_.engine(), &q, 1, out.data(), 2048, &r), map_.engine(), "RrtPlanner::makePlan");
    path.clear();
    for (int k = 0; k < r.path_len && k < 2048; ++k) path.push_back(Position(out[2 * k], out[2 * k + 1]));
    return r.status == 1;
  }

 private:
  GridMap& map_;
  Position start_, target_;
  double tol_;
  unsigned seed_;
};

#ifdef RNA_REFERENCE_SIGNATURES
}  // namespace core
#endif
}  // namespace move_control
