// Forwarding header: grid_map::GridMap and its iterators as the reference's node mains include them.
#pragma once
#include "move_control_api.hpp"
