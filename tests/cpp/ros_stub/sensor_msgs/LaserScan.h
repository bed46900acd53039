#pragma once
#include <vector>
#include <std_msgs/Header.h>
#include <boost/bind.hpp>
namespace sensor_msgs {
struct LaserScan {
  std_msgs::Header header;
  float angle_min = 0, angle_max = 0, angle_increment = 0, time_increment = 0, scan_time = 0, range_min = 0, range_max = 0;
  std::vector<float> ranges, intensities;
};
typedef boost::shared_ptr<LaserScan const> LaserScanConstPtr;
}
