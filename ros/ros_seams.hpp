// ros_seams.hpp -- everything the reference's classes do with ROS, for the classes of move_control_api.hpp:
//
//   reference (inside the classes)                                         here
//   tf odom -> base_link: position / yaw of the robot                      RosSeams: tf::TransformListener -> MapProvider::setRobotPoseSource
//     MapProvider::getRobotPos  mc/src/map_provider.cpp:43-91
//   "/laser_scan" through tf::MessageFilter (target /odom, queue 50)        -> MapProvider::bufferScans (sensor pose at the stamp and at
//     LaserMapUpdater::addMonitorTopic / bufferIncomingMsg                    the end of the scan; the projection runs on the GPU)
//     mc/src/laser_map_updater.cpp:23-75
//   "/left_range" ... "/front_range" through tf::MessageFilter              -> MapProvider::bufferRanges
//     RangeMapUpdater  mc/src/range_map_updater.cpp:23-76, registered at mc/src/map_provider.cpp:27-32
//   publishers global_map / local_map (nav_msgs/OccupancyGrid)             MapProvider::setMapSink
//     mc/src/map_provider.cpp:34-35,113-118,206-213
//   "/odom" latest message (ContinuousSensorHelperRos)                      Steerer::setOdomSource
//     mc/src/steerer.cpp:37, mc/include/move_control/continuous_sensor_helper_ros.h
//   publishers /mobile_base/commands/velocity (Twist), hist (Histogram)    Steerer::setVelocitySink / setHistSink
//     mc/src/steerer.cpp:41-42,193-220
//   threads: update + publish 5 Hz / 1 Hz, move 2 Hz, VFH 5 Hz             RateLoop (ros/rate_loop.hpp) on ros::Time
//     mc/src/map_provider.cpp:151-188,225-233, mc/src/steerer.cpp:43,135-144
//
// Compiled only where ROS is (catkin branch of CMakeLists.txt); the rate keeping, the latest-message cache and the
// scan rate limit are ros/rate_loop.hpp and are tested without ROS (tests/cpp/rate_loop_test.cpp).
#pragma once
#if __has_include(<ros/ros.h>)

#include <ros/ros.h>
#include <tf/transform_listener.h>
#include <tf/message_filter.h>
#include <message_filters/subscriber.h>
#include <sensor_msgs/LaserScan.h>
#include <sensor_msgs/Range.h>
#include <nav_msgs/OccupancyGrid.h>
#include <nav_msgs/Odometry.h>
#include <geometry_msgs/Twist.h>
#include <move_control/Histogram.h>

#include <memory>
#include <string>
#include <vector>

#include "move_control_api.hpp"
#include "rate_loop.hpp"

namespace move_control {

class RosSeams : public detail::RosWiring {
 public:
  // steerer may be null (a node that only maps, or a Steerer attached later)
  RosSeams(ros::NodeHandle& nh, MapProvider& mapProvider, Steerer* steerer);
  ~RosSeams();
  // the reference starts its threads in the constructors of MapProvider / Steerer; here the node does, once it is wired
  // (or rosAutowire() / attachSteerer() do, for move_control's own node mains built with -DRNA_ROS_AUTOWIRE)
  void start();
  void stop();
  void attachSteerer(Steerer& s) override;
  void detachSteerer() override;
  tf::TransformListener& tf() { return tf_; }

 private:
  bool robotPose(Position& pos, double& yaw);
  bool sensorPose(const std::string& frame, const ros::Time& stamp, double& x, double& y, double& yaw);
  bool sensorTransform(const std::string& frame, const ros::Time& stamp, double t[3], double q[4]);
  void laserCb(const sensor_msgs::LaserScanConstPtr& msg);
  void rangeCb(const sensor_msgs::RangeConstPtr& msg);
  void odomCb(const nav_msgs::OdometryConstPtr& msg);
  void publishGrid(const char* topic, const OccupancyGrid& grid);
  void addLaserTopic(const std::string& topic);
  void addRangeTopic(const std::string& topic);
  void wireSteerer();
  bool started_ = false;

  ros::NodeHandle& nh_;
  MapProvider& mapProvider_;
  Steerer* steerer_;
  tf::TransformListener tf_;
  std::string mapFrameId_, robotFrameId_;   // initParameter: "odom", "base_link" (map_provider.cpp:137-138)
  ros::Publisher globalGridPub_, localGridPub_, velPublisher_, histPublisher_;
  ros::Subscriber odomSub_;
  std::vector<std::shared_ptr<message_filters::Subscriber<sensor_msgs::LaserScan> > > laserSubs_;
  std::vector<std::shared_ptr<tf::MessageFilter<sensor_msgs::LaserScan> > > laserFilters_;
  std::vector<std::shared_ptr<message_filters::Subscriber<sensor_msgs::Range> > > rangeSubs_;
  std::vector<std::shared_ptr<tf::MessageFilter<sensor_msgs::Range> > > rangeFilters_;
  LatestValue<double> odomLinearX_;
  MinInterval laserEvery_;
  std::unique_ptr<RateLoop> updateLoop_, moveLoop_, vfhLoop_;
};

}  // namespace move_control

#endif  // __has_include(<ros/ros.h>)
