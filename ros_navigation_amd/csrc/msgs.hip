// msgs.hip -- the data formats either side of the replan path, produced
// on the device from the resident layers / VFH state so the 1 Hz map publish and the 5 Hz histogram
// publish do not have to copy float layers to the host first:
//   rna_to_occupancy_grid*   GridMapRosConverter::toOccupancyGrid
//                            (grid_map-master/grid_map_ros/src/GridMapRosConverter.cpp:251-287) as
//                            MapProvider::publishMap calls it (mc/src/map_provider.cpp:113-118,206-213)
//   rna_from_occupancy_grid  GridMapRosConverter::fromOccupancyGrid, data part (:238-246)
//   rna_vfh_hist_msg_batch   Steerer::pubHist -> Histogram.msg (mc/src/steerer.cpp:201-220)
//   rna_tailor_plan          Nav::taileredPlan (mc/src/nav_node.cpp:192-204), host only
//   rna_follow_plan          Steerer::acceptPlan / head of Steerer::update (mc/src/steerer.cpp:27-33,222-256), host only
#include "engine.hpp"

using namespace rna;

namespace {

// One thread per message cell k (int8 stores are contiguous; layer reads run backwards through
// memory, still whole cache lines).  k = nCells - 1 - (ui + uj*rows) with (ui, uj) the unwrapped
// index of the buffer cell, so a moved (circular-buffer) map is emitted in map order.
__global__ void to_occupancy_kernel(Geom g, const float* __restrict__ layer, float data_min, float data_max,
                                    int8_t* __restrict__ out) {
  const size_t n = (size_t)g.size[0] * g.size[1];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    const size_t index = n - 1 - k;
    const int u[2] = {(int)(index % g.size[0]), (int)(index / g.size[0])};
    int b[2];
    buffer_index(g, u, b);
    float value = (layer[(size_t)b[1] * g.size[0] + b[0]] - data_min) / (data_max - data_min);
    if (isnan(value) || value < 0) value = -1;
    else value = 0.0f + fminf(fmaxf(0.0f, value), 1.0f) * 100.0f;
    out[k] = (int8_t)value;
  }
}

__global__ void from_occupancy_kernel(const int8_t* __restrict__ data, size_t n, float* __restrict__ layer) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int8_t v = data[n - 1 - i];
    layer[i] = v != -1 ? (float)v : __int_as_float(0x7fc00000);
  }
}

// robot r, bin i < H/2: yBinData = (uint16)(int)Hist, yData = (uint16)(int)OriginHist
__global__ void hist_msg_kernel(const float* __restrict__ hist, const float* __restrict__ origin, int H, int n,
                                uint16_t* __restrict__ y_data, uint16_t* __restrict__ y_bin_data) {
  const int bins = H / 2;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n * bins) return;
  const int r = k / bins, i = k % bins;
  y_bin_data[k] = (uint16_t)(int)hist[(size_t)r * H + i];
  y_data[k] = (uint16_t)(int)origin[(size_t)r * H + i];
}

int grid_for(size_t n, int block) {
  size_t b = (n + block - 1) / block;
  if (b > 16384) b = 16384;
  return b ? (int)b : 1;
}

}  // namespace

extern "C" int rna_to_occupancy_grid_device(rna_engine* e, int layer, float data_min, float data_max, int8_t* out_dev) {
  if (!e || !out_dev || layer < 0 || layer >= RNA_NUM_LAYERS) return RNA_EINVAL;
  RNA_ENTER(e);
  KernelTimer kt(e, RNA_K_OCCUPANCY);
  hipLaunchKernelGGL(to_occupancy_kernel, dim3(grid_for(e->ncell, 256)), dim3(256), 0, e->stream, e->geom, e->layer[layer],
                     data_min, data_max, out_dev);
  RNA_HIP(e, hipGetLastError());
  return RNA_OK;
}

extern "C" int rna_to_occupancy_grid(rna_engine* e, int layer, float data_min, float data_max, int8_t* out_host) {
  if (!e || !out_host) return RNA_EINVAL;
  RNA_ENTER(e);
  int8_t* d = nullptr;
  int rc = dev_alloc(e, &d, e->ncell);
  if (rc != RNA_OK) return rc;
  rc = rna_to_occupancy_grid_device(e, layer, data_min, data_max, d);
  hipError_t st = hipSuccess;
  if (rc == RNA_OK) st = hipMemcpyAsync(out_host, d, e->ncell, hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  dev_free(&d);
  if (rc != RNA_OK) return rc;
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  return RNA_OK;
}

extern "C" int rna_from_occupancy_grid(rna_engine* e, int layer, const int8_t* data_host) {
  if (!e || !data_host || layer < 0 || layer >= RNA_NUM_LAYERS) return RNA_EINVAL;
  if (e->geom.start[0] != 0 || e->geom.start[1] != 0)   // the reference resets the geometry in that case (:230-236)
    return fail(e, RNA_ESTATE, "rna_from_occupancy_grid: map has been moved (start index != 0)");
  RNA_ENTER(e);
  int8_t* d = nullptr;
  int rc = dev_alloc(e, &d, e->ncell);
  if (rc != RNA_OK) return rc;
  hipError_t st = hipMemcpyAsync(d, data_host, e->ncell, hipMemcpyHostToDevice, e->stream);
  if (st == hipSuccess) {
    hipLaunchKernelGGL(from_occupancy_kernel, dim3(grid_for(e->ncell, 256)), dim3(256), 0, e->stream, d, e->ncell,
                       e->layer[layer]);
    st = hipGetLastError();
  }
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  dev_free(&d);
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  if (layer == RNA_LAYER_MASTER) { e->nbr_all_dirty = true; e->master_diverged = true; }
  if (layer == RNA_LAYER_LASER) e->laser_all_dirty = true;
  return RNA_OK;
}

extern "C" int rna_vfh_hist_msg_batch(rna_engine* e, int n, uint16_t* x_data_host, uint16_t* y_data_host,
                                      uint16_t* y_bin_data_host, uint16_t thresholds[2]) {
  if (!e || n < 0 || !x_data_host || !y_data_host || !y_bin_data_host || !thresholds) return RNA_EINVAL;
  if (!e->vfh.ready) return fail(e, RNA_ESTATE, "rna_vfh_hist_msg_batch before rna_vfh_init");
  if (n > e->vfh.n_robots) return fail(e, RNA_EINVAL, "more robots than VFH instances");
  const int H = e->vfh.H, bins = H / 2;
  thresholds[0] = (uint16_t)(unsigned)(2000000.0 / 1000.0);   // LOW_OBSTACLE_THRESHOLD, steerer.cpp:13,207
  thresholds[1] = (uint16_t)(unsigned)(4000000.0 / 1000.0);   // HIGH_OBSTACLE_THRESHOLD, steerer.cpp:14,208
  for (int i = 0; i < bins; ++i) x_data_host[i] = (uint16_t)(i * e->vfh.p.sector_angle);
  if (n == 0) return RNA_OK;
  RNA_ENTER(e);
  uint16_t* d = nullptr;
  const size_t cnt = (size_t)n * bins;
  int rc = dev_alloc(e, &d, 2 * cnt);
  if (rc != RNA_OK) return rc;
  hipLaunchKernelGGL(hist_msg_kernel, dim3(grid_for(cnt, 256)), dim3(256), 0, e->stream, e->vfh.hist, e->vfh.origin, H, n, d,
                     d + cnt);
  hipError_t st = hipGetLastError();
  if (st == hipSuccess) st = hipMemcpyAsync(y_data_host, d, cnt * 2, hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipMemcpyAsync(y_bin_data_host, d + cnt, cnt * 2, hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  dev_free(&d);
  if (st != hipSuccess) return fail(e, RNA_EHIP, hipGetErrorString(st));
  return RNA_OK;
}

extern "C" int rna_tailor_plan(const double* plan_xy, int n, unsigned stride, double* out_xy, int* n_out) {
  if (!plan_xy || !out_xy || !n_out || n < 0 || stride == 0) return RNA_EINVAL;
  int m = 0;
  for (int i = n - 1; i >= 0; --i) {
    if (((unsigned)i % stride == 0) || i == n - 1) {
      out_xy[2 * m] = plan_xy[2 * i];
      out_xy[2 * m + 1] = plan_xy[2 * i + 1];
      ++m;
    }
  }
  *n_out = m;
  return RNA_OK;
}

// Steerer::update up to the Update_VFH call (mc/src/steerer.cpp:222-256).  deltaX/deltaY/desiredDist/desiredAngle
// are floats there; map_provider.h:8 has `using namespace std` and the package builds with -std=c++11, so
// hypot(float, float) and atan2(float, float) resolve to the float overloads (hypotf / atan2f of the host libm).
extern "C" int rna_follow_plan(const double* plan_xy, int n, int32_t* plan_index, double x, double y, double yaw,
                               double linear_velocity, double dt, rna_pose* pose) {
  if (!plan_index || !pose || n < 0 || (n > 0 && !plan_xy) || *plan_index < 0) return RNA_EINVAL;
  const float tolerance = 250.0f;   // currGoalDistanceTolerance
  float dx = 0.0f, dy = 0.0f, dist = 0.0f;
  for (;;) {
    if (*plan_index >= n) return 0;   // plan finished (reference: ifPlanReady_ = false)
    dx = (float)((plan_xy[2 * (size_t)*plan_index] - x) * 1000.0);
    dy = (float)((plan_xy[2 * (size_t)*plan_index + 1] - y) * 1000.0);
    dist = hypotf(dx, dy);
    if (dist < tolerance) ++*plan_index;
    else break;
  }
  const double a = (double)atan2f(dy, dx) - yaw + M_PI / 2;
  const double two_pi = 2.0 * M_PI;
  const double norm = fmod(fmod(a, two_pi) + two_pi, two_pi);   // angles::normalize_angle_positive
  pose->x = x; pose->y = y; pose->yaw = yaw; pose->dt = dt;
  pose->current_speed = (int32_t)(linear_velocity * 1000.0);
  pose->goal_direction = (float)(norm * 180.0 / M_PI);
  pose->goal_distance = dist;
  pose->goal_tolerance = tolerance;
  return 1;
}
