#pragma once
#include <std_msgs/Header.h>
#include <boost/bind.hpp>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; typedef boost::shared_ptr<PoseStamped const> ConstPtr; };
}
