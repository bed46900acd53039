#pragma once
#include <vector>
#include <geometry_msgs/PoseStamped.h>
namespace nav_msgs {
struct MapMetaData { ros::Time map_load_time; float resolution = 0; uint32_t width = 0, height = 0; geometry_msgs::Pose origin; };
struct OccupancyGrid { std_msgs::Header header; MapMetaData info; std::vector<int8_t> data; };
}
