#pragma once
#include <std_msgs/Header.h>
#include <boost/bind.hpp>
namespace sensor_msgs {
struct Range { std_msgs::Header header; uint8_t radiation_type = 0; float field_of_view = 0, min_range = 0, max_range = 0, range = 0; };
typedef boost::shared_ptr<Range const> RangeConstPtr;
}
