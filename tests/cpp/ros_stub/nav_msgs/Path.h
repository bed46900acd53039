#pragma once
#include <vector>
#include <geometry_msgs/PoseStamped.h>
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }
