// Forwarding header for "move_control/rrt_planner.h" (see map_provider.h here; rrt_planner.h:5-6 has the same
// using-directives).
#pragma once
#include "move_control/map_provider.h"
