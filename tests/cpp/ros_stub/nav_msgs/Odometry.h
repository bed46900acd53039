#pragma once
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/Twist.h>
namespace nav_msgs {
struct PoseWithCovariance { geometry_msgs::Pose pose; double covariance[36]; };
struct TwistWithCovariance { geometry_msgs::Twist twist; double covariance[36]; };
struct Odometry { std_msgs::Header header; std::string child_frame_id; PoseWithCovariance pose; TwistWithCovariance twist; };
typedef boost::shared_ptr<Odometry const> OdometryConstPtr;
}
