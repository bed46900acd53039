/*
 * oracle/msgs.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into the product).
 *
 * The data formats either side of the replan path:
 *   - GridMapRosConverter::toOccupancyGrid / fromOccupancyGrid
 *     (grid_map-master/grid_map_ros/src/GridMapRosConverter.cpp:205-287), as MapProvider::publishMap
 *     calls it with dataMin 0, dataMax 255 (move_control/src/map_provider.cpp:113-118,206-213);
 *   - Steerer::pubHist -> move_control/Histogram.msg (move_control/src/steerer.cpp:201-220,
 *     move_control/msg/Histogram.msg);
 *   - Nav::taileredPlan (move_control/src/nav_node.cpp:192-204).
 * Pinned by the reference's own known answers in grid_map_ros/test/GridMapRosTest.cpp:116-184
 * (tests/test_oracle_msgs.py); pubHist and taileredPlan have no reference test (parity unpinned).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "rna_oracle.h"

/* GridMapRosConverter.cpp:251-287.  Every buffer cell is visited (GridMapIterator); its place in
 * the message is the column-major linear index of its UNWRAPPED index, reversed. */
void og_to_occupancy_grid(const og_geom* g, const float* layer, float data_min, float data_max, int8_t* out) {
  const int rows = g->size[0], cols = g->size[1];
  const size_t n = (size_t)rows * cols;
  const float cell_min = 0, cell_max = 100, cell_range = cell_max - cell_min;
  for (int j = 0; j < cols; ++j) {
    for (int i = 0; i < rows; ++i) {
      const int b[2] = {i, j};
      int u[2];
      og_unwrap_index(b, g->size, g->start, u);
      float value = (layer[(size_t)j * rows + i] - data_min) / (data_max - data_min);
      if (isnan(value) || value < 0) value = -1;
      else value = cell_min + fminf(fmaxf(0.0f, value), 1.0f) * cell_range;
      const size_t index = og_linear_from_index(u, g->size, 0);
      out[n - index - 1] = (int8_t)value;   /* float -> int8 as the implicit conversion of the assignment */
    }
  }
}

/* GridMapRosConverter.cpp:238-246: reverse iteration, -1 -> NaN; the geometry part (:208-236) is the
 * caller's setGeometry(resolution*size, resolution, origin + length/2). */
void og_from_occupancy_grid(int rows, int cols, const int8_t* data, float* layer) {
  const size_t n = (size_t)rows * cols;
  for (size_t i = 0; i < n; ++i) {
    const int8_t v = data[n - 1 - i];
    layer[i] = v != -1 ? (float)v : NAN;
  }
}

/* steerer.cpp:201-220: num_bin = HIST_SIZE/2; xData = i*sectorAngle; yBinData = (int)Hist[i],
 * yData = (int)OriginHist[i], each narrowed to the message's uint16; thresholds (uint)(2e6/1000),
 * (uint)(4e6/1000) (steerer.cpp:13-14,207-208). */
int og_hist_msg(const float* hist, const float* origin_hist, int hist_size, int sector_angle, uint16_t* x_data,
                uint16_t* y_data, uint16_t* y_bin_data, uint16_t thresholds[2]) {
  const int bin_num = hist_size / 2;
  thresholds[0] = (uint16_t)(unsigned)(2000000.0 / 1000.0);
  thresholds[1] = (uint16_t)(unsigned)(4000000.0 / 1000.0);
  for (int i = 0; i < bin_num; ++i) {
    x_data[i] = (uint16_t)(i * sector_angle);
    y_bin_data[i] = (uint16_t)(int)hist[i];
    y_data[i] = (uint16_t)(int)origin_hist[i];
  }
  return bin_num;
}

/* nav_node.cpp:192-204: walk the detailed plan backwards, keep every stride-th index and the last one */
int og_tailor_plan(const double* plan_xy, int n, unsigned stride, double* out_xy) {
  int m = 0;
  for (int i = n - 1; i >= 0; --i) {
    if (((unsigned)i % stride == 0) || i == n - 1) {
      out_xy[2 * m] = plan_xy[2 * i];
      out_xy[2 * m + 1] = plan_xy[2 * i + 1];
      ++m;
    }
  }
  return m;
}

/* Steerer::acceptPlan (steerer.cpp:27-33: planIndex_ = 1) and Steerer::update up to Update_VFH (steerer.cpp:222-256).
 * float deltaX, deltaY, desiredDist, desiredAngle; hypot / atan2 on floats are the float overloads there
 * (map_provider.h:8 `using namespace std`, -std=c++11 in mc/CMakeLists.txt:20).  out = {desiredAngle, desiredDist};
 * returns 1 while following, 0 when planIndex_ runs past the plan (ifPlanReady_ = false).  A plan shorter than
 * 2 points is finished at once (the reference reads plan_[1] out of bounds). */
int og_follow_plan(const double* plan_xy, int n, int* plan_index, double x, double y, double yaw, float out[2]) {
  float deltaX, deltaY, desiredDist;
  const float currGoalDistanceTolerance = 250;
  while (1) {
    if (*plan_index >= n) return 0;
    deltaX = (plan_xy[2 * *plan_index] - x) * 1000.0;
    deltaY = (plan_xy[2 * *plan_index + 1] - y) * 1000.0;
    desiredDist = hypotf(deltaX, deltaY);
    if (desiredDist < currGoalDistanceTolerance) {
      (*plan_index)++;
      if (*plan_index >= n) return 0;
    } else
      break;
  }
  double a = atan2f(deltaY, deltaX) - yaw + M_PI / 2;
  a = fmod(fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI); /* angles::normalize_angle_positive */
  out[0] = (a) * 180.0 / M_PI;                               /* RAD2DEG */
  out[1] = desiredDist;
  return 1;
}
