// nav_node_amd.cpp -- mapTest: the reference's Nav (mc/src/nav_node.cpp) on the MI355X engine: a goal wakes the planning
// thread, which plans with the goal-biased RRT over a 10 m window of the map around the robot (makePlan :134-154), thins
// the plan (taileredPlan :192-204), publishes it and hands it to the Steerer; it plans again every planInterval_ (20 s)
// until the goal is reached (loopPlan :109-126).  ROS wiring of MapProvider / Steerer: RosSeams.
#if __has_include(<ros/ros.h>)
#include <cstdlib>
#include <ros/ros.h>
#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/Path.h>

#include <condition_variable>
#include <mutex>
#include <thread>

#include "ros_seams.hpp"

namespace move_control {

class Nav {
 public:
  explicit Nav(ros::NodeHandle& nh);
  ~Nav();

 private:
  void initParameters();
  void loopPlan();
  bool makePlan();
  void publishPlan();
  bool ifGoalAchieved();
  void wakePlanner(const ros::TimerEvent& event);
  void goalCb(const geometry_msgs::PoseStamped::ConstPtr& goal);

  ros::NodeHandle& nh_;
  MapProvider mapProvider_;
  Steerer steerer_;
  RosSeams seams_;
  bool ifNaving_;
  ros::Subscriber goalSub_;
  ros::Publisher planPublisher_;
  grid_map::Position target_;
  std::vector<grid_map::Position> pathPlan_;
  std::mutex plannerMutex_;
  std::condition_variable plannerCond_;
  std::thread planThread_;
  bool wake_, quit_;
  std::string goalTopic_;
  double planInterval_;   // s
  double mapPlanLength_;
  unsigned tailerPlanStride_;
  double closeTolerance_;
};

Nav::Nav(ros::NodeHandle& nh)
    : nh_(nh), mapProvider_(nh), steerer_(nh, mapProvider_), seams_(nh, mapProvider_, &steerer_), ifNaving_(false), wake_(false), quit_(false) {
  initParameters();
  goalSub_ = nh_.subscribe(goalTopic_, 1, &Nav::goalCb, this);
  planPublisher_ = nh_.advertise<nav_msgs::Path>("plan", 1);
  seams_.start();
  planThread_ = std::thread([this] { loopPlan(); });
}

Nav::~Nav() {
  {
    std::lock_guard<std::mutex> lock(plannerMutex_);
    quit_ = true;
  }
  plannerCond_.notify_one();
  if (planThread_.joinable()) planThread_.join();
}

void Nav::initParameters() {   // nav_node.cpp:77-85
  goalTopic_ = "move_base_simple/goal";
  planInterval_ = 20;
  mapPlanLength_ = 10.0;
  tailerPlanStride_ = 5;
  closeTolerance_ = 0.2;
}

void Nav::goalCb(const geometry_msgs::PoseStamped::ConstPtr& msg) {
  std::lock_guard<std::mutex> lock(plannerMutex_);
  target_[0] = msg->pose.position.x;
  target_[1] = msg->pose.position.y;
  ifNaving_ = true;
  wake_ = true;
  plannerCond_.notify_one();
}

void Nav::loopPlan() {
  ros::Timer timer;
  std::unique_lock<std::mutex> lock(plannerMutex_);
  while (ros::ok() && !quit_) {
    plannerCond_.wait(lock, [this] { return wake_ || quit_; });
    if (quit_) break;
    wake_ = false;
    timer.stop();
    if (ifGoalAchieved()) continue;
    if (makePlan()) steerer_.acceptPlan(pathPlan_);
    else ROS_ERROR("can't get plan here");
    timer = nh_.createTimer(ros::Duration(planInterval_), &Nav::wakePlanner, this, true);
  }
}

void Nav::wakePlanner(const ros::TimerEvent&) {
  std::lock_guard<std::mutex> lock(plannerMutex_);
  wake_ = true;
  plannerCond_.notify_one();
}

bool Nav::makePlan() {
  grid_map::GridMap mapForPlan;
  Position start;
  mapProvider_.getRobotPos(start);
  mapProvider_.getSubMap(mapForPlan, start, Length(mapPlanLength_, mapPlanLength_));
  mapProvider_.publishMap(mapForPlan);
  std::vector<grid_map::Position> detailedPlan;
  RrtPlanner planner(mapForPlan, start, target_, closeTolerance_);
  if (!planner.makePlan(detailedPlan)) return false;
  taileredPlan(detailedPlan, pathPlan_, tailerPlanStride_);
  publishPlan();
  return true;
}

void Nav::publishPlan() {
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

bool Nav::ifGoalAchieved() {
  Position currentPos;
  mapProvider_.getRobotPos(currentPos);
  return std::hypot(target_[0] - currentPos[0], target_[1] - currentPos[1]) < closeTolerance_;
}

}  // namespace move_control

int main(int argc, char* argv[]) {
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // before the first HIP call: the A* pipeline stages want a hardware queue each
  ros::init(argc, argv, "mapTester");
  ros::NodeHandle nh;
  move_control::Nav nav(nh);
  ros::spin();
  return 0;
}
#else
#error "nav_node_amd.cpp is the ROS node: build it in a catkin workspace (CMakeLists.txt, catkin branch)"
#endif
