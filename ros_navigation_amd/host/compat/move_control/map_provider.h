// Forwarding header: the reference's node mains say #include "move_control/map_provider.h"; with this directory on the
// include path (catkin branch of CMakeLists.txt) they get the MI355X-backed MapProvider instead.  The using-directives
// are the reference header's own (map_provider.h:8-9): its node mains rely on them (`vector<Position>` unqualified).
#pragma once
#if __has_include(<ros/ros.h>)
#include <ros/ros.h>
#endif
#if __has_include(<tf/transform_listener.h>)
#include <tf/transform_listener.h>
#endif
#include "move_control_api.hpp"
using namespace std;
using namespace grid_map;
