// nav_graph_node_main.cpp -- the reference's mc/src/nav_graph_node.cpp (members :27-40, constructor :44-47, goalCb
// :70-80) against ros_navigation_amd/host/move_control_api.hpp, product library only.  A real node includes
// <ros/ros.h> and fills the three seams with tf / odometry / publishers; this main drives them with a scripted robot.
#include <cmath>
#include <cstdio>
#include <vector>

namespace ros { class NodeHandle { public: bool ok() const { return true; } }; }

#include "move_control_api.hpp"

using namespace grid_map;
using namespace move_control;

class NavGraph {
 public:
  NavGraph(ros::NodeHandle& nh);
  void goalCb(double x, double y);
  ros::NodeHandle& nh_;
  MapProvider mapProvider_;
  Steerer steerer_;
  AStarPlanner planner_;
  grid_map::Position target_;
  std::vector<grid_map::Position> pathPlan_;
};

NavGraph::NavGraph(ros::NodeHandle& nh):nh_(nh),
    planner_(nh),
    mapProvider_(nh,Length(4,4),true),
    steerer_(nh, mapProvider_)
{
}

void NavGraph::goalCb(double x, double y)
{
    target_[0] =  x;
    target_[1] =  y;
    pathPlan_.clear();
    Position currentPos;
    mapProvider_.getRobotPos(currentPos);
    planner_.makePlan(currentPos,target_,pathPlan_);
    steerer_.acceptPlan(pathPlan_);
}

int main() {
  ros::NodeHandle nh;
  NavGraph node(nh);
  double x = 0.3, y = -0.2, yaw = 0.1;
  node.mapProvider_.setRobotPoseSource([&](Position& p, double& a) { p = Position(x, y); a = yaw; return true; });
  node.steerer_.setOdomSource([&](double& v) { v = 0.2; return true; });
  node.steerer_.setVelocitySink([&](double lin, double ang) { std::printf("cmd_vel linear.x %.3f angular.z %.3f\n", lin, ang); });
  node.goalCb(12.0, 5.0);
  for (int cycle = 0; cycle < 10; ++cycle) {
    x += 0.04;
    for (int k = 0; k < 60; ++k) {
      RangeSample s;
      s.start = Position(x, y);
      s.end = Position(x + 1.5 * std::cos(yaw + (k - 30) * 0.045), y + 1.5 * std::sin(yaw + (k - 30) * 0.045));
      s.ifClearEnd = (k % 4 == 0);
      node.mapProvider_.bufferSample(s);
    }
    node.mapProvider_.spinUpdateOnce();
    if (cycle % 2) node.mapProvider_.spinMoveOnce();
    node.steerer_.spinOnce();
  }
  std::printf("plan of %zu way points followed\n", node.pathPlan_.size());
  return 0;
}
