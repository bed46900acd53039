// nav_graph_node_amd.cpp -- mapTest_graph (the node testMap.launch starts under the name mapTest_vfh,
// mc/launch/testMap.launch:30): the reference's NavGraph (mc/src/nav_graph_node.cpp: members :27-40, constructor
// :44-56, goalCb :69-80, publishPlan :82-98) on the MI355X engine.  MapProvider / Steerer / AStarPlanner are the classes of
// move_control_api.hpp (the reference's signatures over librna.so); what the reference's classes do with ROS inside
// their constructors -- subscriptions, publishers, tf, the three threads -- is RosSeams.  Topics, frames and rates are
// the reference's: move_base_simple/goal in; plan, global_map, local_map, hist, /mobile_base/commands/velocity out.
#if __has_include(<ros/ros.h>)
#include <cstdlib>
#include <ros/ros.h>
#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/Path.h>

#include "ros_seams.hpp"

namespace move_control {

class NavGraph {
 public:
  explicit NavGraph(ros::NodeHandle& nh);

 private:
  void goalCb(const geometry_msgs::PoseStamped::ConstPtr& goal);
  void publishPlan();

  ros::NodeHandle& nh_;
  MapProvider mapProvider_;
  Steerer steerer_;
  AStarPlanner planner_;
  RosSeams seams_;
  ros::Subscriber goalSub_;
  ros::Publisher planPublisher_;
  grid_map::Position target_;
  std::vector<grid_map::Position> pathPlan_;
};

NavGraph::NavGraph(ros::NodeHandle& nh)
    : nh_(nh), mapProvider_(nh, Length(4, 4), true), steerer_(nh, mapProvider_), planner_(nh), seams_(nh, mapProvider_, &steerer_) {
  goalSub_ = nh_.subscribe("move_base_simple/goal", 1, &NavGraph::goalCb, this);
  planPublisher_ = nh_.advertise<nav_msgs::Path>("plan", 1);
  seams_.start();
}

void NavGraph::goalCb(const geometry_msgs::PoseStamped::ConstPtr& msg) {
  target_[0] = msg->pose.position.x;
  target_[1] = msg->pose.position.y;
  pathPlan_.clear();
  Position currentPos;
  mapProvider_.getRobotPos(currentPos);
  planner_.makePlan(currentPos, target_, pathPlan_);
  publishPlan();
  steerer_.acceptPlan(pathPlan_);
}

void NavGraph::publishPlan() {
  nav_msgs::Path gui_path;
  gui_path.poses.resize(pathPlan_.size());
  gui_path.header.frame_id = "odom";
  gui_path.header.stamp = ros::Time::now();
  for (size_t i = 0; i < pathPlan_.size(); i++) {
    gui_path.poses[i].pose.position.x = pathPlan_[i][0];
    gui_path.poses[i].pose.position.y = pathPlan_[i][1];
  }
  planPublisher_.publish(gui_path);
}

}  // namespace move_control

int main(int argc, char* argv[]) {
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before the first HIP call: the A* pipeline stages want a hardware queue each
  ros::init(argc, argv, "mapTester");
  ros::NodeHandle nh;
  move_control::NavGraph nav(nh);
  ros::spin();
  return 0;
}
#else
#error "nav_graph_node_amd.cpp is the ROS node: build it in a catkin workspace (CMakeLists.txt, catkin branch)"
#endif
