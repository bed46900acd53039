// Forwarding header for "move_control/steerer.h" (see map_provider.h here).  The reference's steerer.h brings
// nav_msgs/Odometry.h -- and with it geometry_msgs/Twist.h -- into its node mains (steerer.h:9).
#pragma once
#if __has_include(<nav_msgs/Odometry.h>)
#include <nav_msgs/Odometry.h>
#endif
#if __has_include(<geometry_msgs/Twist.h>)
#include <geometry_msgs/Twist.h>
#endif
#include "move_control/map_provider.h"
#include "move_control/vfh.h"
