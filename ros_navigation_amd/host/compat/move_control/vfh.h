// Forwarding header: the reference's node mains say #include "move_control/vfh.h"; with this directory on the include
// path (catkin branch of CMakeLists.txt) they get the MI355X-backed classes of the same names instead.
#pragma once
#include "move_control_api.hpp"
