// nav_only_vfh_node_amd.cpp -- mapTest_vfh: the reference's NavVfh (mc/src/nav_only_vfh_node.cpp:40-80) on the MI355X
// engine: a 4 m map that follows the robot, no global planner -- the plan handed to the Steerer is the straight line
// from the robot to the goal, VFH+ does the rest.  ROS wiring of MapProvider / Steerer: RosSeams.
#if __has_include(<ros/ros.h>)
#include <cstdlib>
#include <ros/ros.h>
#include <geometry_msgs/PoseStamped.h>

#include "ros_seams.hpp"

namespace move_control {

class NavVfh {
 public:
  explicit NavVfh(ros::NodeHandle& nh)
      : nh_(nh), mapProvider_(nh, Length(4, 4), true), steerer_(nh, mapProvider_), seams_(nh, mapProvider_, &steerer_) {
    goalSub_ = nh_.subscribe("move_base_simple/goal", 1, &NavVfh::goalCb, this);
    seams_.start();
  }

 private:
  void goalCb(const geometry_msgs::PoseStamped::ConstPtr& msg) {
    target_[0] = msg->pose.position.x;
    target_[1] = msg->pose.position.y;
    pathPlan_.clear();
    Position currentPos;
    mapProvider_.getRobotPos(currentPos);
    pathPlan_.push_back(currentPos);
    pathPlan_.push_back(target_);
    steerer_.acceptPlan(pathPlan_);
  }
  ros::NodeHandle& nh_;
  MapProvider mapProvider_;
  Steerer steerer_;
  RosSeams seams_;
  ros::Subscriber goalSub_;
  grid_map::Position target_;
  std::vector<grid_map::Position> pathPlan_;
};

}  // namespace move_control

int main(int argc, char* argv[]) {
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before the first HIP call: the A* pipeline stages want a hardware queue each
  ros::init(argc, argv, "mapTester");
  ros::NodeHandle nh;
  move_control::NavVfh nav(nh);
  ros::spin();
  return 0;
}
#else
#error "nav_only_vfh_node_amd.cpp is the ROS node: build it in a catkin workspace (CMakeLists.txt, catkin branch)"
#endif
