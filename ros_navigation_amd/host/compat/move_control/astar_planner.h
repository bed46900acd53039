// Forwarding header for "move_control/astar_planner.h" (see map_provider.h here).
#pragma once
#include "move_control/map_provider.h"
