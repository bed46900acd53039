// move_control_api.hpp -- move_control's planner-facing classes with the REFERENCE'S OWN SIGNATURES, so that the node
// mains (mc/src/nav_graph_node.cpp:44-47,75-76, mc/src/nav_node.cpp) compile against this header unchanged:
//
//   MapProvider(ros::NodeHandle&, Length = Length(30.0, 30.0), bool ifMoving = false)      mc/include/move_control/map_provider.h:21-29
//     getRobotPos(Position&) / getRobotPos(Position&, double&) / getSubMap(GridMap&, Position&, Length) /
//     ifCloseToPostion(Position&, double) / publishMap(GridMap&) / getMap(GridMap&)
//   Steerer(ros::NodeHandle&, MapProvider&), acceptPlan(std::vector<Position>&)              mc/include/move_control/steerer.h:20-22
//   VFH(19 arguments), int Init(), int Update_VFH(double[361][2], int, float, float, float, int&, int&)   vfh.h:185-253
//   AStarPlanner(ros::NodeHandle&), makePlan(Position&, Position&, vector<Position>&)        astar_planner.h:32-34
//   RrtPlanner(GridMap&, Position&, Position&, double closeTolerance = 0.2), makePlan(vector<Position>&)   rrt_planner.h:17-28
//
// The compute cores (move_control::core::*, move_control_amd.hpp) run on the GPU through the C ABI; this layer adds what
// the reference's classes own besides: the map geometry of initParameter/initMap, the VFH defaults of Steerer::initVfh,
// the recentring of loopMoveMap, the wall clock of Update_VFH.  ROS itself is not needed to compile it: the classes
// only keep the `ros::NodeHandle&` they are handed (forward declaration below; with ROS headers included first the
// real class is used), and the three places where the reference talks to ROS are seams a node fills:
//   robot pose      tf lookup odom -> base_link (map_provider.cpp:151-175)     MapProvider::setRobotPoseSource
//   odometry        ContinuousSensorHelperRos<nav_msgs::Odometry> (steerer.h)   Steerer::setOdomSource
//   publishers      global_map / local_map / velocity / hist                    MapProvider::setMapSink, Steerer::setVelocitySink
// and the two threads of MapProvider and the VFH thread of Steerer are explicit ticks (spinUpdateOnce, spinMoveOnce,
// Steerer::spinOnce) that the node's threads -- or a test -- call at the reference's rates (5 Hz, 2 Hz, 5 Hz).
#pragma once

#define RNA_REFERENCE_SIGNATURES 1
#include "move_control_amd.hpp"

#include <chrono>
#include <functional>
#include <memory>
#include <mutex>

namespace ros { class NodeHandle; }

namespace move_control {

using grid_map::GridMap;
using grid_map::Index;
using grid_map::Length;
using grid_map::Position;
#if !__has_include(<move_control/Histogram.h>)
using core::Histogram;   // (under catkin `move_control::Histogram` is the generated message: the API below says core::Histogram)
#endif
using core::OccupancyGrid;
using core::RangeSample;
typedef core::RrtPlanner RrtPlanner;            // rrt_planner.h:17-28: the signature is the reference's already
typedef core::GridAStarPlanner GridAStarPlanner;
using core::taileredPlan;

class MapProvider;
class Steerer;
namespace detail {
// What the reference's MapProvider / Steerer constructors do through ROS (subscriptions, publishers, tf, their three
// threads) lives in ros/ros_seams.cpp.  A node either wires it itself (ros/*_amd.cpp: `RosSeams seams(nh, mapProvider,
// &steerer); seams.start();`) or -- to run move_control's OWN node mains unchanged -- is compiled with
// -DRNA_ROS_AUTOWIRE and linked with rna_ros_seams: the constructors below then call rosAutowire(), which creates the
// seams, wires them and starts the loops, as the reference's constructors do.
struct RosWiring {
  virtual ~RosWiring() {}
  virtual void attachSteerer(Steerer& s) = 0;   // odometry, velocity / hist publishers, the 5 Hz VFH loop
  virtual void detachSteerer() = 0;             // (the Steerer is going away: its loop stops first)
};
#if defined(RNA_ROS_AUTOWIRE)
std::shared_ptr<RosWiring> rosAutowire(ros::NodeHandle& nh, MapProvider& mapProvider);   // ros/ros_seams.cpp
#endif
}  // namespace detail

class MapProvider {
 public:
  typedef std::function<bool(Position&, double&)> PoseSource;                       // tf: position and yaw of base_link in odom
  typedef std::function<void(const char* topic, const OccupancyGrid&)> MapSink;     // "global_map" / "local_map"

  MapProvider(ros::NodeHandle& nh, Length mapLength = Length(30.0, 30.0), bool ifMoving = false)
      : nh_(nh), core_(mapLength, 0.05 /* initParameter: resolution_ */), ifMovingWithRobot_(ifMoving),
        updateRate_(5), publishRate_(1), moveMapRate_(2), cyclesSincePublish_(0) {
    core_.getMap().setFrameId("odom");   // initParameter / initMap (map_provider.cpp:130-149): centre (0, 0), frame odom
#if defined(RNA_ROS_AUTOWIRE)
    wiring_ = detail::rosAutowire(nh, *this);   // (last: everything the loops touch exists)
#endif
  }
  ~MapProvider() { wiring_.reset(); }            // the loops stop before the map goes
  detail::RosWiring* rosWiring() { return wiring_.get(); }

  bool getRobotPos(Position& pos) { double a; return getRobotPos(pos, a); }
  bool getRobotPos(Position& pos, double& orientAngle) {   // map_provider.cpp:151-175 (false when tf has no transform)
    if (!poseSource_) return false;
    return poseSource_(pos, orientAngle);
  }
  bool getSubMap(GridMap& map, Position& center, Length length) {   // :93-100
    std::lock_guard<std::mutex> lock(mapMutex_);
    return core_.getSubMap(map, center, length);
  }
  bool ifCloseToPostion(Position& pos, double tolerance) {   // :102-111
    Position robot;
    getRobotPos(robot);
    return core_.ifCloseToPostion(robot, pos, tolerance);
  }
  void publishMap(GridMap& map) {   // :113-118: the "local_map" topic
    OccupancyGrid msg;
    core_.publishMap(map, msg);
    if (mapSink_) mapSink_("local_map", msg);
  }
  bool getMap(GridMap& map) {       // :120-125: a copy under the lock
    std::lock_guard<std::mutex> lock(mapMutex_);
    map = core_.getMap();
    return true;
  }

  // ---- seams of the ROS shim -------------------------------------------------------------------------------------
  void setRobotPoseSource(PoseSource f) { poseSource_ = f; }
  void setMapSink(MapSink f) { mapSink_ = f; }
  // the sensor callbacks (bufferIncomingMsg of the "laser" / "range" MapUpdaters, map_provider.cpp:12-15,262-266)
  void bufferSample(const RangeSample& s, const std::string& typeName = "laser") {
    std::lock_guard<std::mutex> lock(mapMutex_);
    core_.bufferSample(s, typeName);
  }
  void bufferScans(const std::vector<rna_laser_scan>& scans, const std::vector<float>& ranges) {
    std::lock_guard<std::mutex> lock(mapMutex_);
    core_.bufferScans(scans, ranges);
  }
  void bufferRanges(const std::vector<rna_range_reading>& readings) {
    std::lock_guard<std::mutex> lock(mapMutex_);
    core_.bufferRanges(readings);
  }
  // the same with the sensors' full tf transforms (tilted or raised mounts)
  void bufferScans(const std::vector<rna_laser_scan_tf>& scans, const std::vector<float>& ranges) {
    std::lock_guard<std::mutex> lock(mapMutex_);
    core_.bufferScans(scans, ranges);
  }
  void bufferRanges(const std::vector<rna_range_reading_tf>& readings) {
    std::lock_guard<std::mutex> lock(mapMutex_);
    core_.bufferRanges(readings);
  }
  // one pass of loopUpdateAndPublishMap's body (:151-175 of map_provider.cpp, 5 Hz): updateMap(), and publishMap()
  // ("global_map") once per publishRate_
  void spinUpdateOnce() {
    {
      std::lock_guard<std::mutex> lock(mapMutex_);
      core_.updateMap(false);
    }
    if (++cyclesSincePublish_ >= updateRate_ / publishRate_) {
      cyclesSincePublish_ = 0;
      if (mapSink_) {
        OccupancyGrid msg;
        std::lock_guard<std::mutex> lock(mapMutex_);
        core_.publishMap(msg);
        mapSink_("global_map", msg);
      }
    }
  }
  // one pass of loopMoveMap's body (:177-188, 2 Hz, only started when ifMoving): map_.move(robotPos)
  bool spinMoveOnce() {
    if (!ifMovingWithRobot_) return false;
    Position robotPos;
    getRobotPos(robotPos);
    std::lock_guard<std::mutex> lock(mapMutex_);
    return core_.getMap().move(robotPos);
  }
  unsigned updateRate() const { return updateRate_; }
  unsigned moveMapRate() const { return moveMapRate_; }
  bool ifMovingWithRobot() const { return ifMovingWithRobot_; }
  core::MapProvider& core() { return core_; }
  ros::NodeHandle& nodeHandle() { return nh_; }

 private:
  ros::NodeHandle& nh_;
  core::MapProvider core_;
  std::mutex mapMutex_;
  bool ifMovingWithRobot_;
  unsigned updateRate_, publishRate_, moveMapRate_;
  unsigned cyclesSincePublish_;
  PoseSource poseSource_;
  MapSink mapSink_;
  std::shared_ptr<detail::RosWiring> wiring_;
};

// VFH with the reference's constructor, Init() and 7-argument Update_VFH (vfh.h:185-253).  The instance owns a small
// engine of its own (VFH needs no map: it works on the ranges it is handed); the time step of the acceleration limit
// comes from a steady clock as the reference reads gettimeofday (vfh.cpp:573-579), replaceable for tests.
class VFH {
 public:
  VFH(double cell_size, int window_diameter, int sector_angle, double safety_dist_0ms, double safety_dist_1ms,
      int max_speed, int max_speed_narrow_opening, int max_speed_wide_opening, int max_acceleration, int min_turnrate,
      int max_turnrate_0ms, int max_turnrate_1ms, double min_turn_radius_safety_factor, double free_space_cutoff_0ms,
      double obs_cutoff_0ms, double free_space_cutoff_1ms, double obs_cutoff_1ms, double weight_desired_dir,
      double weight_current_dir)
      : Hist(nullptr), OriginHist(nullptr),
        core_(cell_size, window_diameter, sector_angle, safety_dist_0ms, safety_dist_1ms, max_speed, max_speed_narrow_opening,
              max_speed_wide_opening, max_acceleration, min_turnrate, max_turnrate_0ms, max_turnrate_1ms,
              min_turn_radius_safety_factor, free_space_cutoff_0ms, obs_cutoff_0ms, free_space_cutoff_1ms, obs_cutoff_1ms,
              weight_desired_dir, weight_current_dir),
        haveLast_(false), sectorAngle_(sector_angle) {}
  ~VFH() {}
  int Init() {
    own_.setGeometry(Length(1.0, 1.0), 0.05);
    return Init(own_);
  }
  // with the map the Steerer works on, the fused map -> ranges -> VFH path of the engine is available too
  int Init(GridMap& map) {
    core_.Init(map, 1);
    Hist = core_.Hist; OriginHist = core_.OriginHist;
    return 1;
  }
  int Update_VFH(double laser_ranges[361][2], int current_speed, float goal_direction, float goal_distance,
                 float goal_distance_tolerance, int& chosen_speed, int& chosen_turnrate) {
    const int rc = core_.Update_VFH(laser_ranges, current_speed, goal_direction, goal_distance, goal_distance_tolerance,
                                    chosen_speed, chosen_turnrate, tick());
    Hist = core_.Hist; OriginHist = core_.OriginHist;
    return rc;
  }
  void SetRobotRadius(float robot_radius) { core_.SetRobotRadius(robot_radius); }
  float GetPickedAngle() { return core_.GetPickedAngle(); }
  int getHistSize() { return core_.getHistSize(); }
  int getSectorAngle() { return sectorAngle_; }
  void setClock(std::function<double()> secondsNow) { clock_ = secondsNow; haveLast_ = false; }
  // seconds since the previous call (first call: 0.3 s, i.e. beyond the reference's 0.3 s clamp of the speed increment)
  double tick() {
    const double now = clock_ ? clock_() : std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const double dt = haveLast_ ? now - last_ : 0.3;
    last_ = now; haveLast_ = true;
    return dt;
  }
  core::VFH& core() { return core_; }
  float* Hist;        // public as in the reference (vfh.h:239-244)
  float* OriginHist;

 private:
  core::VFH core_;
  GridMap own_;
  bool haveLast_;
  double last_;
  int sectorAngle_;
  std::function<double()> clock_;
};

class Steerer {
 public:
  typedef std::function<bool(double& linear_x)> OdomSource;                         // nav_msgs/Odometry twist.twist.linear.x
  typedef std::function<void(double linear_x, double angular_z)> VelocitySink;      // geometry_msgs/Twist of pubVel
  typedef std::function<void(const core::Histogram&)> HistSink;                           // move_control/Histogram of pubHist

  Steerer(ros::NodeHandle& nh, MapProvider& mapProvider) : nh_(nh), mapProvider_(mapProvider), lastOdom_(0.0) {
    initVfh();
    steer_.reset(new core::Steerer(mapProvider_.core(), vfhP_->core()));
    if (detail::RosWiring* w = mapProvider_.rosWiring()) w->attachSteerer(*this);   // (-DRNA_ROS_AUTOWIRE only)
  }
  ~Steerer() { if (detail::RosWiring* w = mapProvider_.rosWiring()) w->detachSteerer(); }
  void acceptPlan(std::vector<Position>& plan) {   // steerer.cpp:27-33
    std::lock_guard<std::mutex> lock(planMutex_);
    steer_->acceptPlan(plan);
  }

  void setOdomSource(OdomSource f) { odomSource_ = f; }
  void setVelocitySink(VelocitySink f) { velSink_ = f; }
  void setHistSink(HistSink f) { histSink_ = f; }
  // one pass of vfhLoop's body (steerer.cpp:123-145, 5 Hz): `if (ifPlanReady_) update();`.  true: a velocity command
  // was produced (and handed to the sinks)
  bool spinOnce(int* chosenSpeed = nullptr, int* chosenTurnrate = nullptr) {
    std::lock_guard<std::mutex> lock(planMutex_);
    if (!steer_->ifPlanReady()) return false;
    Position currentPos;
    double currentDir = 0.0;
    mapProvider_.getRobotPos(currentPos, currentDir);
    if (odomSource_) (void)odomSource_(lastOdom_);
    int speed = 0, turnrate = 0;
    double lin = 0.0, ang = 0.0;
    if (!steer_->update(currentPos, currentDir, lastOdom_, vfhP_->tick(), speed, turnrate, lin, ang)) return false;
    if (chosenSpeed) *chosenSpeed = speed;
    if (chosenTurnrate) *chosenTurnrate = turnrate;
    if (velSink_) velSink_(lin, ang);
    if (histSink_) { core::Histogram h; vfhP_->core().pubHist(h); histSink_(h); }
    return true;
  }
  bool ifPlanReady() const { return steer_->ifPlanReady(); }
  VFH& vfh() { return *vfhP_; }

 private:
  // Steerer::initVfh (steerer.cpp:46-121): the defaults of every ROS parameter (rna_vfh_default_params holds them)
  void initVfh() {
    rna_vfh_params p;
    rna_vfh_default_params(&p);
    vfhP_.reset(new VFH(p.cell_size, p.window_diameter, p.sector_angle, p.safety_dist_0ms, p.safety_dist_1ms, p.max_speed,
                        p.max_speed_narrow_opening, p.max_speed_wide_opening, p.max_acceleration, p.min_turnrate,
                        p.max_turnrate_0ms, p.max_turnrate_1ms, p.min_turn_radius_safety_factor, p.free_space_cutoff_0ms,
                        p.obs_cutoff_0ms, p.free_space_cutoff_1ms, p.obs_cutoff_1ms, p.weight_desired_dir, p.weight_current_dir));
    vfhP_->SetRobotRadius((float)p.robot_radius);
    vfhP_->Init(mapProvider_.core().getMap());
  }
  ros::NodeHandle& nh_;
  MapProvider& mapProvider_;
  std::unique_ptr<VFH> vfhP_;
  std::unique_ptr<core::Steerer> steer_;
  std::mutex planMutex_;
  double lastOdom_;
  OdomSource odomSource_;
  VelocitySink velSink_;
  HistSink histSink_;
};

// AStarPlanner(ros::NodeHandle&) (astar_planner.h:32-34): the hard-coded 9-vertex waypoint graph of init(); the search
// kernel needs an engine, not a map: the planner owns a small one.
class AStarPlanner {
 public:
  explicit AStarPlanner(ros::NodeHandle& nh) : nh_(nh) {
    own_.setGeometry(Length(1.0, 1.0), 0.05);
    core_.reset(new core::AStarPlanner(own_));
  }
  ~AStarPlanner() {}
  bool makePlan(Position& start, Position& target, std::vector<Position>& path) { return core_->makePlan(start, target, path); }

 private:
  ros::NodeHandle& nh_;
  GridMap own_;
  std::unique_ptr<core::AStarPlanner> core_;
};

}  // namespace move_control
