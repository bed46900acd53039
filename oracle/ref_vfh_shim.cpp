/*
 * ref_vfh_shim.cpp -- ORACLE SUPPORT (test infrastructure): C entry points around the REFERENCE's
 * own VFH class, compiled from the sources where they lie under /root/reference (never copied):
 *   g++ -I/root/reference/move_control/include -Dgettimeofday=rna_ref_clock \
 *       /root/reference/move_control/src/vfh.cpp oracle/ref_vfh_shim.cpp -> oracle/_ref/libref_vfh.so
 * The only build-time intervention is pinning the wall clock: vfh.cpp reads gettimeofday() to
 * derive the acceleration step (vfh.cpp:413,521-531); -Dgettimeofday=rna_ref_clock routes that
 * call to the settable clock below so outputs are reproducible.
 * Used by tests/ and tests/golden/gen_vfh_golden.py only.
 */
#include <sys/time.h>
#include <cstring>

#define private public /* table read-back for tests (Cell_Direction, Cell_Sector, ...) */
#include "move_control/vfh.h"
#undef private

static double g_clock = 1000.0;

extern "C" int rna_ref_clock(struct timeval* tv, void*) noexcept {
  tv->tv_sec = (time_t)g_clock;
  tv->tv_usec = (suseconds_t)((g_clock - (double)tv->tv_sec) * 1e6 + 0.5);
  if (tv->tv_usec >= 1000000) { tv->tv_sec += 1; tv->tv_usec -= 1000000; }
  return 0;
}

struct ref_params { /* same layout as og_vfh_params (oracle/rna_oracle.h) */
  double cell_size; int window_diameter; int sector_angle;
  double safety_dist_0ms, safety_dist_1ms;
  int max_speed, max_speed_narrow_opening, max_speed_wide_opening;
  int max_acceleration, min_turnrate, max_turnrate_0ms, max_turnrate_1ms;
  double min_turn_radius_safety_factor;
  double free_space_cutoff_0ms, obs_cutoff_0ms, free_space_cutoff_1ms, obs_cutoff_1ms;
  double weight_desired_dir, weight_current_dir;
  double robot_radius;
};

using move_control::VFH;

extern "C" {

void refvfh_set_clock(double t) { g_clock = t; }

void* refvfh_create(const ref_params* p) {
  VFH* v = new VFH(p->cell_size, p->window_diameter, p->sector_angle, p->safety_dist_0ms,
                   p->safety_dist_1ms, p->max_speed, p->max_speed_narrow_opening,
                   p->max_speed_wide_opening, p->max_acceleration, p->min_turnrate,
                   p->max_turnrate_0ms, p->max_turnrate_1ms, p->min_turn_radius_safety_factor,
                   p->free_space_cutoff_0ms, p->obs_cutoff_0ms, p->free_space_cutoff_1ms,
                   p->obs_cutoff_1ms, p->weight_desired_dir, p->weight_current_dir);
  v->SetRobotRadius((float)p->robot_radius);
  v->Blocked_Circle_Radius = 0.0f; /* uninitialised in the reference; pinned to the oracle's definition */
  v->Init();
  return v;
}

void refvfh_destroy(void* h) { delete (VFH*)h; }

/* advances the pinned clock by dt (exactly representable steps keep diffSeconds == dt) */
int refvfh_update(void* h, double ranges[361][2], int current_speed, float goal_direction,
                  float goal_distance, float goal_tol, double dt, int* chosen_speed,
                  int* chosen_turnrate) {
  g_clock += dt;
  return ((VFH*)h)->Update_VFH(ranges, current_speed, goal_direction, goal_distance, goal_tol,
                               *chosen_speed, *chosen_turnrate);
}

int refvfh_hist_size(void* h) { return ((VFH*)h)->getHistSize(); }
const float* refvfh_hist(void* h) { return ((VFH*)h)->Hist; }
const float* refvfh_origin_hist(void* h) { return ((VFH*)h)->OriginHist; }
float refvfh_picked_angle(void* h) { return ((VFH*)h)->GetPickedAngle(); }
int refvfh_num_tables(void* h) { return ((VFH*)h)->NUM_CELL_SECTOR_TABLES; }
float refvfh_cell_direction(void* h, int x, int y) { return ((VFH*)h)->Cell_Direction[x][y]; }
float refvfh_cell_dist(void* h, int x, int y) { return ((VFH*)h)->Cell_Dist[x][y]; }
float refvfh_cell_base_mag(void* h, int x, int y) { return ((VFH*)h)->Cell_Base_Mag[x][y]; }
int refvfh_cell_sector_count(void* h, int t, int x, int y) { return (int)((VFH*)h)->Cell_Sector[t][x][y].size(); }
int refvfh_cell_sector(void* h, int t, int x, int y, int k) { return ((VFH*)h)->Cell_Sector[t][x][y][k]; }
int refvfh_min_turning_radius(void* h, int speed) { return ((VFH*)h)->Min_Turning_Radius[speed]; }

}
