// A node main shaped like the reference's mc/src/nav_node.cpp (mapTest: members :37-60, constructor :62-74,
// initParameters :77-85, makePlan :134-154, ifGoalAchieved :174-183, steer :185-190, taileredPlan :192-204) compiled
// against ros_navigation_amd/host/move_control_api.hpp and run on the GPU: the default 30 m map (600 x 600 cells), a
// 10 m planning window around the robot handed to RrtPlanner, the detailed plan thinned to every fifth way point and
// given to the Steerer -- the window, the local_map message, the RRT plan and the tailored plan checked against the
// CPU oracle, then the Steerer follows the plan.  Built and run by tests/test_host_mirror.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ros { class NodeHandle { public: bool ok() const { return true; } }; }

#include "../../oracle/rna_oracle.h"
#include "../../ros_navigation_amd/host/move_control_api.hpp"

using namespace std;
using namespace grid_map;
using namespace move_control;

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
static bool same_bits(float a, float b) { return (std::isnan(a) && std::isnan(b)) || std::memcmp(&a, &b, 4) == 0; }

// ---- nav_node.cpp:15-60: the members that take part in planning, in the reference's order ----
class Nav {
 public:
  Nav(ros::NodeHandle& nh);
  void initParameters();
  bool makePlan();
  bool ifGoalAchieved();
  void steer();
  void taileredPlan(vector<grid_map::Position>& detailedPlan);
  void goalCb(double x, double y);   // goalCb(const geometry_msgs::PoseStamped::ConstPtr&) without the message type

  ros::NodeHandle& nh_;
  MapProvider mapProvider_;
  Steerer steerer_;
  bool ifNaving_;
  grid_map::Position target_;
  std::vector<grid_map::Position> pathPlan_;
  double planInterval_;  //s
  double mapPlanLength_;
  unsigned tailerPlanStride_;
  double closeTolerance_;
  double robotVel_;
  std::vector<grid_map::Position> lastDetailedPlan_;   // (test only: what RrtPlanner returned)
  grid_map::GridMap lastWindow_;                       // (test only: the planning window)
};

// nav_node.cpp:62-64, the reference's initialiser list (mapProvider_(nh): the 30 m default map, not moving)
Nav::Nav(ros::NodeHandle& nh):nh_(nh),mapProvider_(nh),
    ifNaving_(false),robotVel_(0.0),steerer_(nh, mapProvider_)
{
    initParameters();
}

void Nav::initParameters()   // nav_node.cpp:77-85
{
    planInterval_ = 20; //s
    mapPlanLength_ = 10.0;
    tailerPlanStride_ = 5;
    closeTolerance_ = 0.2;
}

void Nav::goalCb(double x, double y)   // :94-101
{
    target_[0] =  x;
    target_[1] =  y;
    ifNaving_ = true;
}

bool Nav::makePlan()   // :134-154
{
    grid_map::GridMap mapForPlan;
    Position start;
    mapProvider_.getRobotPos(start);
    mapProvider_.getSubMap(mapForPlan,start,Length(mapPlanLength_,mapPlanLength_));
    mapProvider_.publishMap(mapForPlan);

    std::vector<grid_map::Position> detailedPlan;

    RrtPlanner planner(mapForPlan,start,target_,closeTolerance_);
    if(!planner.makePlan(detailedPlan))
        return false;
    lastDetailedPlan_ = detailedPlan;
    lastWindow_ = mapForPlan;

    taileredPlan(detailedPlan);

    return true;
}

bool Nav::ifGoalAchieved()   // :174-183
{
    Position currentPos;
    mapProvider_.getRobotPos(currentPos);
    double deltaX= (target_[0] - currentPos[0]);
    double deltaY= (target_[1] - currentPos[1]);
    double distance = hypot(deltaX,deltaY);
    return (distance<closeTolerance_);
}

void Nav::steer()   // :185-190
{
    steerer_.acceptPlan(pathPlan_);
}

void Nav::taileredPlan(vector<grid_map::Position> &detailedPlan)   // :192-204 (publishPlan is the node's)
{
    move_control::taileredPlan(detailedPlan, pathPlan_, tailerPlanStride_);
}

int main() {
  ros::NodeHandle nh;
  Nav node(nh);
  double robot_x = -3.0, robot_y = 1.0, robot_yaw = 0.2, t = 0.0;
  node.mapProvider_.setRobotPoseSource([&](Position& p, double& yaw) { p = Position(robot_x, robot_y); yaw = robot_yaw; return true; });
  node.steerer_.setOdomSource([&](double& v) { v = 0.2; return true; });
  node.steerer_.vfh().setClock([&]() { return t; });
  int local_maps = 0, vel_msgs = 0;
  OccupancyGrid last_local;
  node.mapProvider_.setMapSink([&](const char* topic, const OccupancyGrid& m) {
    if (std::strcmp(topic, "local_map") == 0) { ++local_maps; last_local = m; }
  });
  node.steerer_.setVelocitySink([&](double, double) { ++vel_msgs; });

  GridMap view;
  CHECK(node.mapProvider_.getMap(view) && view.getSize()[0] == 600 && view.getSize()[1] == 600 && !node.mapProvider_.ifMovingWithRobot());

  // a few walls seen by the sensors (hits on a line of cells, several updates so that the cells saturate)
  og_geom g;
  og_set_geometry(&g, 30.0, 30.0, 0.05, 0.0, 0.0);
  std::vector<float> laser((size_t)600 * 600, NAN);
  std::srand(5);
  for (int rep = 0; rep < 4; ++rep) {
    std::vector<og_ray> rays;
    for (int k = 0; k < 400; ++k) {
      RangeSample s;
      const double wx = -1.0 + 0.004 * (k % 100), wy = -1.5 + 0.03 * k * 0.25;          // a wall x ~ -1, y in [-1.5, 1.5]
      s.start = Position(robot_x + 0.1 * std::rand() / RAND_MAX, robot_y);
      s.end = Position(wx, wy);
      s.ifClearEnd = false;
      node.mapProvider_.bufferSample(s);
      og_ray r = {s.start[0], s.start[1], s.end[0], s.end[1], 0, 0};
      rays.push_back(r);
    }
    node.mapProvider_.spinUpdateOnce();
    og_himm_update(&g, laser.data(), rays.data(), (int)rays.size(), nullptr);
  }
  std::vector<float> master = laser;

  node.goalCb(1.5, -0.5);           // behind the wall, inside the 10 m window
  CHECK(!node.ifGoalAchieved());
  CHECK(node.makePlan());
  node.steer();
  CHECK(local_maps == 1 && node.steerer_.ifPlanReady());

  // the oracle's planning window, local_map payload, RRT plan (srand(1) sequence) and tailoring
  og_geom sg;
  std::vector<float> sub(master.size());
  const double p[2] = {robot_x, robot_y}, l[2] = {10.0, 10.0};
  CHECK(og_get_submap(&g, master.data(), p, l, &sg, sub.data(), (int)sub.size()));
  CHECK(node.lastWindow_.getSize()[0] == sg.size[0] && node.lastWindow_.getSize()[1] == sg.size[1]);
  const Matrix W = node.lastWindow_["master"];
  for (int c = 0; c < sg.size[0] * sg.size[1]; ++c) CHECK(same_bits(W(c), sub[c]));
  std::vector<int8_t> occ((size_t)sg.size[0] * sg.size[1]);
  og_to_occupancy_grid(&sg, sub.data(), 0.0f, 255.0f, occ.data());
  CHECK(last_local.width == (unsigned)sg.size[0] && last_local.height == (unsigned)sg.size[1] && last_local.data == occ);
  std::vector<double> out(2 * 2048);
  og_rrt_result r;
  const double a[2] = {robot_x, robot_y}, b[2] = {1.5, -0.5};
  og_rrt_plan(&sg, sub.data(), a, b, 0.2, 1, 1000000, out.data(), 2048, &r);
  CHECK(r.status == 1 && (size_t)r.path_len == node.lastDetailedPlan_.size() && r.path_len >= 8);
  for (int k = 0; k < r.path_len; ++k)
    CHECK(std::fabs(out[2 * k] - node.lastDetailedPlan_[k][0]) < 1e-9 && std::fabs(out[2 * k + 1] - node.lastDetailedPlan_[k][1]) < 1e-9);
  // Nav::taileredPlan: the detailed plan is goal -> start; walked backwards, every 5th index and the last one are kept
  std::vector<Position> want;
  for (int i = r.path_len - 1; i >= 0; --i)
    if ((i % 5 == 0) || i == r.path_len - 1) want.push_back(node.lastDetailedPlan_[i]);
  CHECK(want.size() == node.pathPlan_.size());
  for (size_t k = 0; k < want.size(); ++k) CHECK(want[k][0] == node.pathPlan_[k][0] && want[k][1] == node.pathPlan_[k][1]);
  CHECK(std::fabs(node.pathPlan_.front()[0] - robot_x) < 1e-12 && std::fabs(node.pathPlan_.front()[1] - robot_y) < 1e-12);

  // the Steerer follows the tailored plan (5 Hz), against the oracle's plan following + VFH+ on the same map
  og_vfh_params ovp;
  og_vfh_default_params(&ovp);
  og_vfh* ov = og_vfh_create(&ovp);
  std::vector<double> plan;
  for (size_t k = 0; k < node.pathPlan_.size(); ++k) { plan.push_back(node.pathPlan_[k][0]); plan.push_back(node.pathPlan_[k][1]); }
  int oidx = 1, steps = 0;
  double t_prev = 0.0;
  for (int cycle = 0; cycle < 25; ++cycle) {
    t = 0.2 * cycle;
    int speed = -1, turn = -1;
    if (!node.steerer_.spinOnce(&speed, &turn)) break;
    float cmd[2];
    CHECK(og_follow_plan(plan.data(), (int)node.pathPlan_.size(), &oidx, robot_x, robot_y, robot_yaw, cmd) == 1);
    int ospeed = 0, oturn = 0;
    const double rp[2] = {robot_x, robot_y};
    og_vfh_step_pose(ov, &g, master.data(), rp, robot_yaw, 200, cmd[0], cmd[1], 250.0f, steps == 0 ? 0.3 : t - t_prev, &ospeed, &oturn);
    t_prev = t;
    CHECK(speed == ospeed && turn == oturn);
    ++steps;
    // drive towards the way point the Steerer is heading for
    robot_yaw += turn * M_PI / 180.0 * 0.2;
    robot_x += 0.001 * speed * 0.2 * std::cos(robot_yaw);
    robot_y += 0.001 * speed * 0.2 * std::sin(robot_yaw);
  }
  CHECK(steps >= 10 && vel_msgs == steps);
  og_vfh_destroy(ov);
  std::printf("nav_node-shaped main OK: RRT plan of %d positions in a %d x %d window, %zu way points, %d VFH steps\n", r.path_len, sg.size[0],
              sg.size[1], node.pathPlan_.size(), steps);
  return 0;
}
