// ros_seams.cpp -- see ros_seams.hpp.  Built by the catkin branch of CMakeLists.txt only.
#include "ros_seams.hpp"
#if __has_include(<ros/ros.h>)

#include <tf/transform_datatypes.h>

#include <cmath>

namespace move_control {

namespace {
double rosNow() { return ros::Time::now().toSec(); }
void rosSleep(double s) { ros::Duration(s).sleep(); }
}  // namespace

RosSeams::RosSeams(ros::NodeHandle& nh, MapProvider& mapProvider, Steerer* steerer)
    : nh_(nh), mapProvider_(mapProvider), steerer_(steerer), tf_(ros::Duration(10)),   // map_provider.cpp:18: tf_(ros::Duration(10))
      mapFrameId_("odom"), robotFrameId_("base_link"), laserEvery_(0.2) {              // laser_map_updater.h: msgProcessCycle_ 0.2 s
  mapProvider_.setRobotPoseSource([this](Position& p, double& yaw) { return robotPose(p, yaw); });
  globalGridPub_ = nh_.advertise<nav_msgs::OccupancyGrid>("global_map", 1);
  localGridPub_ = nh_.advertise<nav_msgs::OccupancyGrid>("local_map", 1);
  mapProvider_.setMapSink([this](const char* topic, const OccupancyGrid& g) { publishGrid(topic, g); });
  // the updaters MapProvider's constructor registers (map_provider.cpp:27-32)
  addRangeTopic("/left_range");
  addRangeTopic("/right_range");
  addRangeTopic("/front_left_range");
  addRangeTopic("/front_right_range");
  addRangeTopic("/front_range");
  addLaserTopic("/laser_scan");
  if (steerer_) wireSteerer();
}

void RosSeams::wireSteerer() {
  odomSub_ = nh_.subscribe<nav_msgs::Odometry>("/odom", 1, &RosSeams::odomCb, this);
  velPublisher_ = nh_.advertise<geometry_msgs::Twist>("/mobile_base/commands/velocity", 5);
  histPublisher_ = nh_.advertise<move_control::Histogram>("hist", 5);
  steerer_->setOdomSource([this](double& v) { return odomLinearX_.get(v); });
  steerer_->setVelocitySink([this](double lin, double ang) {   // Steerer::pubVel (steerer.cpp:193-199)
    geometry_msgs::Twist cmd;
    cmd.linear.x = lin;
    cmd.angular.z = ang;
    velPublisher_.publish(cmd);
  });
  steerer_->setHistSink([this](const core::Histogram& h) {     // Steerer::pubHist (steerer.cpp:201-220)
    move_control::Histogram msg;
    msg.num_bin = h.num_bin;
    msg.xData = h.xData;
    msg.yData = h.yData;
    msg.yBinData = h.yBinData;
    msg.yLowThreshold = h.yLowThreshold;
    msg.yHighThreshold = h.yHighThreshold;
    histPublisher_.publish(msg);
  });
}

// a Steerer constructed after the seams (move_control's own node mains, -DRNA_ROS_AUTOWIRE): wire it, and if the map
// loops run already start its 5 Hz loop as Steerer's constructor does in the reference (steerer.cpp:35-44)
void RosSeams::attachSteerer(Steerer& s) {
  steerer_ = &s;
  wireSteerer();
  if (started_ && !vfhLoop_) {
    const RateLoop::Ok ok = [this] { return nh_.ok(); };
    vfhLoop_.reset(new RateLoop(5.0, [this] { steerer_->spinOnce(); }, ok, rosNow, rosSleep));   // steerer.cpp:137
    vfhLoop_->start();
  }
}

void RosSeams::detachSteerer() {
  if (vfhLoop_) { vfhLoop_->stop(); vfhLoop_.reset(); }
  steerer_ = nullptr;
}

namespace detail {
// see move_control_api.hpp: what MapProvider's constructor calls when the node was compiled with -DRNA_ROS_AUTOWIRE
std::shared_ptr<RosWiring> rosAutowire(ros::NodeHandle& nh, MapProvider& mapProvider) {
  std::shared_ptr<RosSeams> seams(new RosSeams(nh, mapProvider, nullptr));
  seams->start();
  return seams;
}
}  // namespace detail

RosSeams::~RosSeams() { stop(); }

void RosSeams::start() {
  started_ = true;
  const RateLoop::Ok ok = [this] { return nh_.ok(); };
  updateLoop_.reset(new RateLoop(mapProvider_.updateRate(), [this] { mapProvider_.spinUpdateOnce(); }, ok, rosNow, rosSleep));
  updateLoop_->start();
  if (mapProvider_.ifMovingWithRobot()) {
    moveLoop_.reset(new RateLoop(mapProvider_.moveMapRate(), [this] { mapProvider_.spinMoveOnce(); }, ok, rosNow, rosSleep));
    moveLoop_->start();
  }
  if (steerer_) {
    vfhLoop_.reset(new RateLoop(5.0, [this] { steerer_->spinOnce(); }, ok, rosNow, rosSleep));   // steerer.cpp:137
    vfhLoop_->start();
  }
}

void RosSeams::stop() {
  if (vfhLoop_) vfhLoop_->stop();
  if (moveLoop_) moveLoop_->stop();
  if (updateLoop_) updateLoop_->stop();
}

// MapProvider::getRobotPos(pos, angle) (map_provider.cpp:69-91): the latest transform, waited for at most 1 s
bool RosSeams::robotPose(Position& pos, double& yaw) {
  return sensorPose(robotFrameId_, ros::Time(), pos[0], pos[1], yaw);
}

bool RosSeams::sensorPose(const std::string& frame, const ros::Time& stamp, double& x, double& y, double& yaw) {
  double t[3], q[4];
  if (!sensorTransform(frame, stamp, t, q)) return false;
  x = t[0];
  y = t[1];
  yaw = tf::getYaw(tf::Quaternion(q[0], q[1], q[2], q[3]));
  return true;
}

// the whole transform map <- frame as tf reports it (translation, quaternion x y z w): sensors are not assumed to be mounted
// level -- the engine interpolates and applies it the way laser_geometry / tf::transformPoint do (rna_scan_to_rays_tf)
bool RosSeams::sensorTransform(const std::string& frame, const ros::Time& stamp, double t[3], double q[4]) {
  if (!tf_.waitForTransform(mapFrameId_, frame, stamp, ros::Duration(1))) {
    ROS_ERROR_THROTTLE(1.0, "map provider can't transform from %s to %s at %f", mapFrameId_.c_str(), frame.c_str(), stamp.toSec());
    return false;
  }
  tf::StampedTransform st;
  try {
    tf_.lookupTransform(mapFrameId_, frame, stamp, st);
  } catch (tf::TransformException& ex) {
    ROS_WARN("Failure %s\n", ex.what());
    return false;
  }
  const tf::Vector3& o = st.getOrigin();
  const tf::Quaternion r = st.getRotation();
  t[0] = o.x(); t[1] = o.y(); t[2] = o.z();
  q[0] = r.x(); q[1] = r.y(); q[2] = r.z(); q[3] = r.w();
  return true;
}

void RosSeams::addLaserTopic(const std::string& topic) {   // LaserMapUpdater::addMonitorTopic (laser_map_updater.cpp:23-35)
  std::shared_ptr<message_filters::Subscriber<sensor_msgs::LaserScan> > sub(new message_filters::Subscriber<sensor_msgs::LaserScan>(nh_, topic, 50));
  std::shared_ptr<tf::MessageFilter<sensor_msgs::LaserScan> > filter(new tf::MessageFilter<sensor_msgs::LaserScan>(*sub, tf_, "/odom", 50));
  filter->registerCallback(boost::bind(&RosSeams::laserCb, this, _1));
  laserSubs_.push_back(sub);
  laserFilters_.push_back(filter);
}

void RosSeams::addRangeTopic(const std::string& topic) {   // RangeMapUpdater::addMonitorTopic (range_map_updater.cpp:23-35)
  std::shared_ptr<message_filters::Subscriber<sensor_msgs::Range> > sub(new message_filters::Subscriber<sensor_msgs::Range>(nh_, topic, 50));
  std::shared_ptr<tf::MessageFilter<sensor_msgs::Range> > filter(new tf::MessageFilter<sensor_msgs::Range>(*sub, tf_, "/odom", 50));
  filter->registerCallback(boost::bind(&RosSeams::rangeCb, this, _1));
  rangeSubs_.push_back(sub);
  rangeFilters_.push_back(filter);
}

// LaserMapUpdater::bufferIncomingMsg (laser_map_updater.cpp:37-75): at most one scan per 0.2 s; the scan goes to the
// engine as it is -- decimation, projection and the per-beam pose interpolation between tf's transforms at the stamp
// and at the end of the scan run on the GPU (rna_scan_to_rays)
void RosSeams::laserCb(const sensor_msgs::LaserScanConstPtr& msg) {
  if (!laserEvery_.take(rosNow())) return;
  rna_laser_scan_tf s;
  s.angle_min = msg->angle_min;
  s.angle_max = msg->angle_max;
  s.angle_increment = msg->angle_increment;
  s.range_min = msg->range_min;
  s.range_max = msg->range_max;
  s.n_ranges = (int32_t)msg->ranges.size();
  s.ranges_offset = 0;
  if (!sensorTransform(msg->header.frame_id, msg->header.stamp, s.t, s.q)) return;
  // laser_geometry's high-fidelity projection asks tf for the end of the (decimated) scan as well
  const int beams = rna_scan_projected_beams(s.n_ranges, s.angle_increment);
  const ros::Time end = msg->header.stamp + ros::Duration((beams > 0 ? beams - 1 : 0) * (double)msg->time_increment);
  if (msg->time_increment == 0.0f || !sensorTransform(msg->header.frame_id, end, s.t_end, s.q_end)) {
    for (int k = 0; k < 3; ++k) s.t_end[k] = s.t[k];
    for (int k = 0; k < 4; ++k) s.q_end[k] = s.q[k];
  }
  std::vector<rna_laser_scan_tf> scans(1, s);
  mapProvider_.bufferScans(scans, msg->ranges);
}

// RangeMapUpdater::bufferIncomingMsg (range_map_updater.cpp:38-76)
void RosSeams::rangeCb(const sensor_msgs::RangeConstPtr& msg) {
  rna_range_reading_tf r;
  r.range = msg->range;
  r.max_range = msg->max_range;
  if (!sensorTransform(msg->header.frame_id, msg->header.stamp, r.t, r.q)) return;
  std::vector<rna_range_reading_tf> one(1, r);
  mapProvider_.bufferRanges(one);
}

void RosSeams::odomCb(const nav_msgs::OdometryConstPtr& msg) { odomLinearX_.set(msg->twist.twist.linear.x); }

// the header GridMapRosConverter::toOccupancyGrid fills besides the data (GridMapRosConverter.cpp:251-270)
void RosSeams::publishGrid(const char* topic, const OccupancyGrid& g) {
  nav_msgs::OccupancyGrid msg;
  msg.header.frame_id = mapFrameId_;
  msg.header.stamp = ros::Time::now();
  msg.info.map_load_time = msg.header.stamp;
  msg.info.resolution = g.resolution;
  msg.info.width = g.width;
  msg.info.height = g.height;
  msg.info.origin.position.x = g.origin_x;
  msg.info.origin.position.y = g.origin_y;
  msg.info.origin.position.z = 0.0;
  msg.info.origin.orientation.w = 1.0;
  msg.data.assign(g.data.begin(), g.data.end());
  (std::string(topic) == "local_map" ? localGridPub_ : globalGridPub_).publish(msg);
}

}  // namespace move_control

#endif  // __has_include(<ros/ros.h>)
