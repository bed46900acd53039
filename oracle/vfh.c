/*
 * vfh.c -- ORACLE (test infrastructure): VFH+ of move_control restated in plain C.
 * Follows mc/src/vfh.cpp and mc/include/move_control/vfh.h (mc/ = /root/reference/move_control).
 * Pinned: tests compare this file bit-for-bit with the reference vfh.cpp compiled into
 * oracle/_ref (tests/test_oracle_vfh.py) and with the committed golden vectors.
 *
 * The reference is C++ and includes <math.h>, so calls such as atan(float), asin(float),
 * hypot(float,float) and fabs(float) resolve to the FLOAT overloads (atanf, asinf, hypotf, fabsf)
 * while pow(int,int) / pow(double,int) promote to double.  This file spells those out explicitly.
 * The wall-clock delta of Update_VFH (gettimeofday, vfh.cpp:521-531) is an explicit argument.
 * Blocked_Circle_Radius is uninitialised in the reference until the first non-emergency update
 * (vfh.h:333, vfh.cpp:1148); the oracle defines it as 0 until then.
 */
#include "rna_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MINI(a, b) (((a) < (b)) ? (a) : (b))

struct og_vfh {
  /* vfh.h:301-360 */
  float ROBOT_RADIUS;
  int CENTER_X, CENTER_Y, HIST_SIZE;
  float CELL_WIDTH;
  int WINDOW_DIAMETER, SECTOR_ANGLE;
  float SAFETY_DIST_0MS, SAFETY_DIST_1MS;
  int Current_Max_Speed, MAX_SPEED, MAX_SPEED_NARROW_OPENING, MAX_SPEED_WIDE_OPENING;
  int MAX_ACCELERATION, MIN_TURNRATE, NUM_CELL_SECTOR_TABLES;
  int MAX_TURNRATE_0MS, MAX_TURNRATE_1MS;
  double MIN_TURN_RADIUS_SAFETY_FACTOR;
  float Binary_Hist_Low_0ms, Binary_Hist_High_0ms, Binary_Hist_Low_1ms, Binary_Hist_High_1ms;
  float U1, U2;
  float Desired_Angle, Dist_To_Goal, Goal_Distance_Tolerance;
  float Picked_Angle, Last_Picked_Angle;
  int Max_Speed_For_Picked_Angle;
  float Blocked_Circle_Radius;
  float *Cell_Direction, *Cell_Base_Mag, *Cell_Mag, *Cell_Dist, *Cell_Enlarge; /* [x*W + y] */
  int* Cell_Sector;       /* [((t*W + x)*W + y)*HIST + k] */
  int* Cell_Sector_Count; /* [(t*W + x)*W + y] */
  float *Hist, *OriginHist, *Last_Binary_Hist;
  int* Min_Turning_Radius;
  int last_chosen_speed;
};

/* steerer.cpp:69-121 */
void og_vfh_default_params(og_vfh_params* p) {
  p->cell_size = 100;
  p->window_diameter = 30;
  p->sector_angle = 5;
  p->safety_dist_0ms = 10;
  p->safety_dist_1ms = 50;
  p->max_speed = 200;
  p->max_speed_narrow_opening = 200;
  p->max_speed_wide_opening = 300;
  p->max_acceleration = 200;
  p->min_turnrate = 40;
  p->max_turnrate_0ms = 40;
  p->max_turnrate_1ms = 40;
  p->min_turn_radius_safety_factor = 1.0;
  p->free_space_cutoff_0ms = 2000000.0;
  p->obs_cutoff_0ms = 4000000.0;
  p->free_space_cutoff_1ms = 2000000.0;
  p->obs_cutoff_1ms = 4000000.0;
  p->weight_desired_dir = 10.0;
  p->weight_current_dir = 1.0;
  p->robot_radius = 178.0;
}

/* vfh.cpp:130-138 */
static int get_max_turnrate(const og_vfh* v, int speed) {
  int val = (v->MAX_TURNRATE_0MS - (int)(speed * (v->MAX_TURNRATE_0MS - v->MAX_TURNRATE_1MS) / 1000.0));
  if (val < 0) val = 0;
  return val;
}

/* vfh.cpp:144-166 */
static void set_current_max_speed(og_vfh* v, int max_speed) {
  v->Current_Max_Speed = MINI(max_speed, v->MAX_SPEED);
  free(v->Min_Turning_Radius);
  v->Min_Turning_Radius = (int*)calloc((size_t)v->Current_Max_Speed + 1, sizeof(int));
  for (int x = 0; x <= v->Current_Max_Speed; x++) {
    double dx = (double)x / 1e6;
    double dtheta = ((M_PI / 180) * (double)(get_max_turnrate(v, x))) / 1000.0;
    v->Min_Turning_Radius[x] = (int)(((dx / tan(dtheta)) * 1000.0) * v->MIN_TURN_RADIUS_SAFETY_FACTOR);
  }
}

/* vfh.cpp:175-186 */
static int get_speed_index(const og_vfh* v, int speed) {
  int val = (int)floorf(((float)speed / (float)v->Current_Max_Speed) * v->NUM_CELL_SECTOR_TABLES);
  if (val >= v->NUM_CELL_SECTOR_TABLES) val = v->NUM_CELL_SECTOR_TABLES - 1;
  return val;
}

/* vfh.cpp:194-205 */
static int get_safety_dist(const og_vfh* v, int speed) {
  int val = (int)(v->SAFETY_DIST_0MS + (int)(speed * (v->SAFETY_DIST_1MS - v->SAFETY_DIST_0MS) / 1000.0));
  if (val < 0) val = 0;
  return val;
}

/* vfh.cpp:214-231 */
static float get_binary_hist_low(const og_vfh* v, int speed) {
  return (float)(v->Binary_Hist_Low_0ms - (speed * (v->Binary_Hist_Low_0ms - v->Binary_Hist_Low_1ms) / 1000.0));
}
static float get_binary_hist_high(const og_vfh* v, int speed) {
  return (float)(v->Binary_Hist_High_0ms - (speed * (v->Binary_Hist_High_0ms - v->Binary_Hist_High_1ms) / 1000.0));
}

/* vfh.cpp:673-686 */
static float delta_angle(float a1, float a2) {
  float diff = a2 - a1;
  if (diff > 180) diff -= 360;
  else if (diff < -180) diff += 360;
  return diff;
}

/* one of the four wrap-aware signed differences of vfh.cpp:343-381 */
static float sector_to_dir(float sector, float dir) {
  if ((sector - dir) > 180) return dir - (sector - 360);
  if ((dir - sector) > 180) return sector - (dir + 360);
  return dir - sector;
}

/* vfh.cpp:53-110 (ctor), :421-467 (VFH_Allocate), :237-416 (Init) */
og_vfh* og_vfh_create(const og_vfh_params* p) {
  og_vfh* v = (og_vfh*)calloc(1, sizeof(og_vfh));
  v->CELL_WIDTH = (float)p->cell_size;
  v->WINDOW_DIAMETER = p->window_diameter;
  v->SECTOR_ANGLE = p->sector_angle;
  v->SAFETY_DIST_0MS = (float)p->safety_dist_0ms;
  v->SAFETY_DIST_1MS = (float)p->safety_dist_1ms;
  v->Current_Max_Speed = p->max_speed;
  v->MAX_SPEED = p->max_speed;
  v->MAX_SPEED_NARROW_OPENING = p->max_speed_narrow_opening;
  v->MAX_SPEED_WIDE_OPENING = p->max_speed_wide_opening;
  v->MAX_ACCELERATION = p->max_acceleration;
  v->MIN_TURNRATE = p->min_turnrate;
  v->MAX_TURNRATE_0MS = p->max_turnrate_0ms;
  v->MAX_TURNRATE_1MS = p->max_turnrate_1ms;
  v->MIN_TURN_RADIUS_SAFETY_FACTOR = p->min_turn_radius_safety_factor;
  v->Binary_Hist_Low_0ms = (float)p->free_space_cutoff_0ms;
  v->Binary_Hist_High_0ms = (float)p->obs_cutoff_0ms;
  v->Binary_Hist_Low_1ms = (float)p->free_space_cutoff_1ms;
  v->Binary_Hist_High_1ms = (float)p->obs_cutoff_1ms;
  v->U1 = (float)p->weight_desired_dir;
  v->U2 = (float)p->weight_current_dir;
  v->Desired_Angle = 90;
  v->Picked_Angle = 90;
  v->Last_Picked_Angle = v->Picked_Angle;
  v->last_chosen_speed = 0;
  v->NUM_CELL_SECTOR_TABLES = (v->SAFETY_DIST_0MS == v->SAFETY_DIST_1MS) ? 1 : 20;
  v->ROBOT_RADIUS = (float)p->robot_radius; /* SetRobotRadius, vfh.h:235 */
  v->Blocked_Circle_Radius = 0.0f;           /* uninitialised in the reference */
  v->Max_Speed_For_Picked_Angle = 0;         /* uninitialised in the reference; always set before use */

  /* Init */
  const int W = v->WINDOW_DIAMETER;
  v->CENTER_X = (int)floor(W / 2.0);
  v->CENTER_Y = v->CENTER_X;
  v->HIST_SIZE = (int)rint(360.0 / v->SECTOR_ANGLE);
  const int H = v->HIST_SIZE, T = v->NUM_CELL_SECTOR_TABLES;
  const int CX = v->CENTER_X, CY = v->CENTER_Y;

  v->Cell_Direction = (float*)calloc((size_t)W * W, sizeof(float));
  v->Cell_Base_Mag = (float*)calloc((size_t)W * W, sizeof(float));
  v->Cell_Mag = (float*)calloc((size_t)W * W, sizeof(float));
  v->Cell_Dist = (float*)calloc((size_t)W * W, sizeof(float));
  v->Cell_Enlarge = (float*)calloc((size_t)W * W, sizeof(float));
  v->Cell_Sector = (int*)calloc((size_t)T * W * W * H, sizeof(int));
  v->Cell_Sector_Count = (int*)calloc((size_t)T * W * W, sizeof(int));
  v->Hist = (float*)calloc((size_t)H, sizeof(float));
  v->OriginHist = (float*)calloc((size_t)H, sizeof(float));
  v->Last_Binary_Hist = (float*)calloc((size_t)H, sizeof(float));
  set_current_max_speed(v, v->MAX_SPEED);

  for (int x = 0; x < H; x++) { v->Hist[x] = 0; v->OriginHist[x] = 0; v->Last_Binary_Hist[x] = 1; }

  for (int x = 0; x < W; x++) {
    for (int y = 0; y < W; y++) {
      const int c = x * W + y;
      v->Cell_Mag[c] = 0;
      v->Cell_Dist[c] = (float)(sqrt(pow((double)(CX - x), 2.0) + pow((double)(CY - y), 2.0)) * v->CELL_WIDTH);
      v->Cell_Base_Mag[c] = (float)(15 * pow((3000.0 - v->Cell_Dist[c]), 4.0) / 100000000.0);

      float d = 0.0f; /* calloc'ed vector element when no branch assigns (cannot happen) */
      if (x < CX) {
        if (y < CY) {
          d = atanf((float)(CY - y) / (float)(CX - x));
          d = (float)(d * (360.0 / 6.28));
          d = (float)(180.0 - d);
        } else if (y == CY) {
          d = 180.0f;
        } else {
          d = atanf((float)(y - CY) / (float)(CX - x));
          d = (float)(d * (360.0 / 6.28));
          d = (float)(180.0 + d);
        }
      } else if (x == CX) {
        if (y < CY) d = 90.0f;
        else if (y == CY) d = -1.0f;
        else d = 270.0f;
      } else {
        if (y < CY) {
          d = atanf((float)(CY - y) / (float)(x - CX));
          d = (float)(d * (360.0 / 6.28));
        } else if (y == CY) {
          d = 0.0f;
        } else {
          d = atanf((float)(y - CY) / (float)(x - CX));
          d = (float)(d * (360.0 / 6.28));
          d = (float)(360.0 - d);
        }
      }
      v->Cell_Direction[c] = d;

      for (int t = 0; t < T; t++) {
        int max_speed_this_table = (int)(((float)(t + 1) / (float)T) * (float)v->MAX_SPEED);
        if (v->Cell_Dist[c] > 0) {
          float r = v->ROBOT_RADIUS + get_safety_dist(v, max_speed_this_table);
          v->Cell_Enlarge[c] = (float)((float)asinf(r / v->Cell_Dist[c]) * (180 / M_PI));
        } else {
          v->Cell_Enlarge[c] = 0;
        }
        int* list = &v->Cell_Sector[(((size_t)t * W + x) * W + y) * H];
        int cnt = 0;
        float plus_dir = v->Cell_Direction[c] + v->Cell_Enlarge[c];
        float neg_dir = v->Cell_Direction[c] - v->Cell_Enlarge[c];
        for (int i = 0; i < (360 / v->SECTOR_ANGLE); i++) {
          float plus_sector = (i + 1) * (float)v->SECTOR_ANGLE;
          float neg_sector = i * (float)v->SECTOR_ANGLE;
          float neg_sector_to_neg_dir = sector_to_dir(neg_sector, neg_dir);
          float plus_sector_to_neg_dir = sector_to_dir(plus_sector, neg_dir);
          float plus_sector_to_plus_dir = sector_to_dir(plus_sector, plus_dir);
          float neg_sector_to_plus_dir = sector_to_dir(neg_sector, plus_dir);
          int plus_dir_bw = 0, neg_dir_bw = 0, dir_around_sector = 0;
          if ((neg_sector_to_neg_dir >= 0) && (plus_sector_to_neg_dir <= 0)) neg_dir_bw = 1;
          if ((neg_sector_to_plus_dir >= 0) && (plus_sector_to_plus_dir <= 0)) plus_dir_bw = 1;
          if ((neg_sector_to_neg_dir <= 0) && (neg_sector_to_plus_dir >= 0)) dir_around_sector = 1;
          if ((plus_sector_to_neg_dir <= 0) && (plus_sector_to_plus_dir >= 0)) plus_dir_bw = 1;
          if (plus_dir_bw || neg_dir_bw || dir_around_sector) list[cnt++] = i;
        }
        v->Cell_Sector_Count[((size_t)t * W + x) * W + y] = cnt;
      }
    }
  }
  return v;
}

void og_vfh_destroy(og_vfh* v) {
  if (!v) return;
  free(v->Cell_Direction); free(v->Cell_Base_Mag); free(v->Cell_Mag); free(v->Cell_Dist);
  free(v->Cell_Enlarge); free(v->Cell_Sector); free(v->Cell_Sector_Count);
  free(v->Hist); free(v->OriginHist); free(v->Last_Binary_Hist); free(v->Min_Turning_Radius);
  free(v);
}

/* vfh.cpp:986-1049 */
static int calculate_cells_mag(og_vfh* v, double ranges[361][2], int speed) {
  const int W = v->WINDOW_DIAMETER;
  float safeSpeed = (float)get_safety_dist(v, speed);
  float r = v->ROBOT_RADIUS + safeSpeed;
  for (int x = 0; x < W; x++) {
    for (int y = 0; y < (int)ceil(W / 2.0); y++) {
      const int c = x * W + y;
      if ((v->Cell_Dist[c] + v->CELL_WIDTH / 2.0) > ranges[(int)rint(v->Cell_Direction[c] * 2.0)][0]) {
        if (v->Cell_Dist[c] < r && !(x == v->CENTER_X && y == v->CENTER_Y)) return 0;
        v->Cell_Mag[c] = v->Cell_Base_Mag[c];
      } else {
        v->Cell_Mag[c] = 0.0;
      }
    }
  }
  return 1;
}

/* vfh.cpp:1057-1095 */
static int build_primary_polar_histogram(og_vfh* v, double ranges[361][2], int speed) {
  const int W = v->WINDOW_DIAMETER, H = v->HIST_SIZE;
  int speed_index = get_speed_index(v, speed);
  for (int x = 0; x < H; x++) v->OriginHist[x] = 0;
  if (calculate_cells_mag(v, ranges, speed) == 0) {
    for (int x = 0; x < H; x++) v->OriginHist[x] = 1;
    return 0;
  }
  for (int y = 0; y <= (int)ceil(W / 2.0); y++) {
    for (int x = 0; x < W; x++) {
      const size_t e = ((size_t)speed_index * W + x) * W + y;
      const int* list = &v->Cell_Sector[e * H];
      for (int i = 0; i < v->Cell_Sector_Count[e]; i++) v->OriginHist[list[i]] += v->Cell_Mag[x * W + y];
    }
  }
  return 1;
}

/* vfh.cpp:1102-1121 */
static void build_binary_polar_histogram(og_vfh* v, int speed) {
  for (int x = 0; x < v->HIST_SIZE; x++) {
    if (v->OriginHist[x] > get_binary_hist_high(v, speed)) v->Hist[x] = 1.0;
    else if (v->OriginHist[x] < get_binary_hist_low(v, speed)) v->Hist[x] = 0.0;
    else v->Hist[x] = v->Last_Binary_Hist[x];
  }
  for (int x = 0; x < v->HIST_SIZE; x++) v->Last_Binary_Hist[x] = v->Hist[x];
}

/* vfh.cpp:1131-1213 */
static void build_masked_polar_histogram(og_vfh* v, int speed) {
  const int W = v->WINDOW_DIAMETER;
  /* A current speed above Current_Max_Speed indexes Min_Turning_Radius out of bounds in the reference (Update_VFH
   * does not clamp, vfh.cpp:495-515).  Defined here and in the HIP kernel: the per-speed quantities of this function
   * are taken at Current_Max_Speed; valid inputs are unaffected. */
  const int ts = speed > v->Current_Max_Speed ? v->Current_Max_Speed : speed;
  float center_x_right = v->CENTER_X + (v->Min_Turning_Radius[ts] / (float)v->CELL_WIDTH);
  float center_x_left = v->CENTER_X - (v->Min_Turning_Radius[ts] / (float)v->CELL_WIDTH);
  float center_y = v->CENTER_Y;
  float angle_ahead = 90, phi_left = 180, phi_right = 0;

  v->Blocked_Circle_Radius = v->Min_Turning_Radius[ts] + v->ROBOT_RADIUS + get_safety_dist(v, ts);

  for (int y = 0; y < (int)ceil(W / 2.0); y++) {
    for (int x = 0; x < W; x++) {
      const int c = x * W + y;
      if (v->Cell_Mag[c] == 0) continue;
      if ((delta_angle(v->Cell_Direction[c], angle_ahead) > 0) && (delta_angle(v->Cell_Direction[c], phi_right) <= 0)) {
        float dist_r = hypotf(center_x_right - x, center_y - y) * v->CELL_WIDTH;
        if (dist_r < v->Blocked_Circle_Radius) phi_right = v->Cell_Direction[c];
      } else if ((delta_angle(v->Cell_Direction[c], angle_ahead) <= 0) && (delta_angle(v->Cell_Direction[c], phi_left) > 0)) {
        float dist_l = hypotf(center_x_left - x, center_y - y) * v->CELL_WIDTH;
        if (dist_l < v->Blocked_Circle_Radius) phi_left = v->Cell_Direction[c];
      }
    }
  }

  for (int x = 0; x < v->HIST_SIZE; x++) {
    float angle = x * v->SECTOR_ANGLE;
    if ((v->Hist[x] == 0) && (((delta_angle(angle, phi_right) <= 0) && (delta_angle(angle, angle_ahead) >= 0)) ||
                              ((delta_angle(angle, phi_left) >= 0) && (delta_angle(angle, angle_ahead) <= 0))))
      v->Hist[x] = 0;
    else
      v->Hist[x] = 1;
  }
}

/* vfh.cpp:715-749 */
static void select_candidate_angle(og_vfh* v, const float* cand_angle, const int* cand_speed, int n) {
  if (n == 0) {
    v->Picked_Angle = v->Last_Picked_Angle;
    v->Max_Speed_For_Picked_Angle = 0;
    v->Last_Picked_Angle = v->Picked_Angle;
    return;
  }
  v->Picked_Angle = 90;
  float min_weight = 10000000;
  for (int i = 0; i < n; i++) {
    float weight = v->U1 * fabsf(delta_angle(v->Desired_Angle, cand_angle[i])) +
                   v->U2 * fabsf(delta_angle(v->Last_Picked_Angle, cand_angle[i]));
    if (weight < min_weight) {
      min_weight = weight;
      v->Picked_Angle = cand_angle[i];
      v->Max_Speed_For_Picked_Angle = cand_speed[i];
    }
  }
  v->Last_Picked_Angle = v->Picked_Angle;
}

/* vfh.cpp:755-870 */
static void select_direction(og_vfh* v) {
  const int H = v->HIST_SIZE, SA = v->SECTOR_ANGLE;
  float* cand_angle = (float*)malloc(sizeof(float) * 4 * (size_t)(H + 2));
  int* cand_speed = (int*)malloc(sizeof(int) * 4 * (size_t)(H + 2));
  int* border = (int*)malloc(sizeof(int) * 2 * (size_t)(H + 2));
  int ncand = 0, nborder = 0;

  int start = -1;
  for (int i = 0; i < H / 2; i++) {
    if (v->Hist[i] == 1) { start = i; break; }
  }
  if (start == -1) {
    v->Picked_Angle = v->Desired_Angle;
    v->Last_Picked_Angle = v->Picked_Angle;
    v->Max_Speed_For_Picked_Angle = v->Current_Max_Speed;
    free(cand_angle); free(cand_speed); free(border);
    return;
  }

  int left = 1, nb_first = 0, nb_second = 0;
  for (int i = start; i <= (start + H); i++) {
    if ((v->Hist[i % H] == 0) && left) { nb_first = (i % H) * SA; left = 0; }
    if ((v->Hist[i % H] == 1) && !left) {
      nb_second = ((i % H) - 1) * SA;
      if (nb_second < 0) nb_second += 360;
      border[2 * nborder] = nb_first; border[2 * nborder + 1] = nb_second; nborder++;
      left = 1;
    }
  }

  for (int i = 0; i < nborder; i++) {
    const int b1 = border[2 * i], b2 = border[2 * i + 1];
    float angle = delta_angle((float)b1, (float)b2);
    if (fabsf(angle) < 10) continue;
    if (fabsf(angle) < 80) {
      float new_angle = (float)(b1 + (b2 - b1) / 2.0);
      cand_angle[ncand] = new_angle;
      cand_speed[ncand++] = MINI(v->Current_Max_Speed, v->MAX_SPEED_NARROW_OPENING);
    } else {
      float new_angle = (float)(b1 + (b2 - b1) / 2.0);
      cand_angle[ncand] = new_angle;
      cand_speed[ncand++] = v->Current_Max_Speed;

      new_angle = (float)((b1 + 40) % 360);
      cand_angle[ncand] = new_angle;
      cand_speed[ncand++] = MINI(v->Current_Max_Speed, v->MAX_SPEED_WIDE_OPENING);

      new_angle = (float)(b2 - 40);
      if (new_angle < 0) new_angle += 360;
      cand_angle[ncand] = new_angle;
      cand_speed[ncand++] = MINI(v->Current_Max_Speed, v->MAX_SPEED_WIDE_OPENING);

      if ((delta_angle(v->Desired_Angle, cand_angle[ncand - 2]) < 0) &&
          (delta_angle(v->Desired_Angle, cand_angle[ncand - 1]) > 0)) {
        cand_angle[ncand] = v->Desired_Angle;
        cand_speed[ncand++] = MINI(v->Current_Max_Speed, v->MAX_SPEED_WIDE_OPENING);
      }
    }
  }
  select_candidate_angle(v, cand_angle, cand_speed, ncand);
  free(cand_angle); free(cand_speed); free(border);
}

/* vfh.cpp:612-654 */
static int cant_turn_to_goal(const og_vfh* v) {
  float goal_x = (float)(v->Dist_To_Goal * cos(((v->Desired_Angle) * M_PI / 180)));
  float goal_y = (float)(v->Dist_To_Goal * sin(((v->Desired_Angle) * M_PI / 180)));
  float dist_between_centres = hypotf(goal_x - v->Blocked_Circle_Radius, goal_y);
  if (dist_between_centres + v->Goal_Distance_Tolerance < v->Blocked_Circle_Radius) return 1;
  dist_between_centres = hypotf(-goal_x - v->Blocked_Circle_Radius, goal_y);
  if (dist_between_centres + v->Goal_Distance_Tolerance < v->Blocked_Circle_Radius) return 1;
  return 0;
}

/* vfh.cpp:1222-1261 */
static void set_motion(const og_vfh* v, int* speed, int* turnrate, int actual_speed) {
  if (*speed <= 0) {
    *turnrate = get_max_turnrate(v, actual_speed);
    *speed = 0;
  } else {
    if ((v->Picked_Angle > 270) && (v->Picked_Angle < 360)) {
      *turnrate = -1 * get_max_turnrate(v, actual_speed);
    } else if ((v->Picked_Angle < 270) && (v->Picked_Angle > 180)) {
      *turnrate = get_max_turnrate(v, actual_speed);
    } else {
      *turnrate = (int)rint(((float)(v->Picked_Angle - 90) / 75.0) * get_max_turnrate(v, actual_speed));
      if (*turnrate > get_max_turnrate(v, actual_speed)) *turnrate = get_max_turnrate(v, actual_speed);
      else if (*turnrate < (-1 * get_max_turnrate(v, actual_speed))) *turnrate = -1 * get_max_turnrate(v, actual_speed);
    }
  }
}

/* vfh.cpp:480-605 */
int og_vfh_update(og_vfh* v, double ranges[361][2], int current_speed, float goal_direction,
                  float goal_distance, float goal_distance_tolerance, double diffSeconds,
                  int* chosen_speed, int* chosen_turnrate) {
  v->Desired_Angle = goal_direction;
  v->Dist_To_Goal = goal_distance;
  v->Goal_Distance_Tolerance = goal_distance_tolerance;

  int current_pos_speed = current_speed < 0 ? 0 : current_speed;
  if (current_pos_speed < v->last_chosen_speed) current_pos_speed = v->last_chosen_speed;

  if (build_primary_polar_histogram(v, ranges, current_pos_speed) == 0) {
    v->Picked_Angle = v->Last_Picked_Angle;
    v->Max_Speed_For_Picked_Angle = 0;
    v->Last_Picked_Angle = v->Picked_Angle;
  } else {
    build_binary_polar_histogram(v, current_pos_speed);
    build_masked_polar_histogram(v, current_pos_speed);
    select_direction(v);
  }

  int speed_incr;
  if ((diffSeconds > 0.3) || (diffSeconds < 0)) speed_incr = 10;
  else speed_incr = (int)(v->MAX_ACCELERATION * diffSeconds);

  if (cant_turn_to_goal(v)) speed_incr = -speed_incr;

  *chosen_speed = MINI(v->last_chosen_speed + speed_incr, v->Max_Speed_For_Picked_Angle);
  set_motion(v, chosen_speed, chosen_turnrate, current_pos_speed);
  v->last_chosen_speed = *chosen_speed;
  return 1;
}

int og_vfh_hist_size(const og_vfh* v) { return v->HIST_SIZE; }
const float* og_vfh_hist(const og_vfh* v) { return v->Hist; }
const float* og_vfh_origin_hist(const og_vfh* v) { return v->OriginHist; }
float og_vfh_picked_angle(const og_vfh* v) { return v->Picked_Angle; }
float og_vfh_last_picked_angle(const og_vfh* v) { return v->Last_Picked_Angle; }
int og_vfh_max_speed_for_picked_angle(const og_vfh* v) { return v->Max_Speed_For_Picked_Angle; }
int og_vfh_num_tables(const og_vfh* v) { return v->NUM_CELL_SECTOR_TABLES; }
const float* og_vfh_cell_direction(const og_vfh* v) { return v->Cell_Direction; }
const float* og_vfh_cell_dist(const og_vfh* v) { return v->Cell_Dist; }
const float* og_vfh_cell_base_mag(const og_vfh* v) { return v->Cell_Base_Mag; }
int og_vfh_cell_sector_count(const og_vfh* v, int t, int x, int y) {
  return v->Cell_Sector_Count[((size_t)t * v->WINDOW_DIAMETER + x) * v->WINDOW_DIAMETER + y];
}
const int* og_vfh_cell_sector_list(const og_vfh* v, int t, int x, int y) {
  return &v->Cell_Sector[(((size_t)t * v->WINDOW_DIAMETER + x) * v->WINDOW_DIAMETER + y) * v->HIST_SIZE];
}
int og_vfh_min_turning_radius(const og_vfh* v, int speed) { return v->Min_Turning_Radius[speed]; }

/* steerer.cpp:260-263 : getRangesFromSubmap() then Update_VFH() */
int og_vfh_step_pose(og_vfh* v, const og_geom* g, const float* master, const double robot_pos[2],
                     double yaw, int current_speed, float goal_direction, float goal_distance,
                     float goal_tol, double dt, int* chosen_speed, int* chosen_turnrate) {
  double ranges[361][2];
  if (!og_ranges_from_submap(g, master, robot_pos, yaw, ranges)) {
    /* getSubMap failure leaves ranges_ at 5000 everywhere (steerer.cpp:149-158) */
    for (int i = 0; i < 361; ++i) ranges[i][0] = 5000.0;
  }
  return og_vfh_update(v, ranges, current_speed, goal_direction, goal_distance, goal_tol, dt,
                       chosen_speed, chosen_turnrate);
}
