// GPU test of the C++ host mirror (ros_navigation_amd/host/move_control_amd.hpp): the calls read like
// the reference's node code (mc/src/nav_graph_node.cpp, steerer.cpp), results are checked against
// the CPU oracle.  Built and run by tests/test_host_mirror.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/rna_oracle.h"
#include "../../ros_navigation_amd/host/move_control_amd.hpp"

using namespace grid_map;
using namespace move_control;

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static bool same_bits(float a, float b) { return (std::isnan(a) && std::isnan(b)) || std::memcmp(&a, &b, 4) == 0; }

int main() {
  // ---- MapProvider: buffered range samples -> HIMM -> master (map_provider.cpp:190-223) ----
  MapProvider provider(Length(12.8, 12.8));
  GridMap& map = provider.getMap();
  og_geom g;
  og_set_geometry(&g, 12.8, 12.8, 0.05, 0.0, 0.0);
  CHECK(map.getSize()[0] == g.size[0] && map.getSize()[1] == g.size[1]);
  std::vector<float> ref((size_t)g.size[0] * g.size[1], NAN);
  std::vector<og_ray> rays;
  std::srand(7);
  for (int k = 0; k < 800; ++k) {
    const double a = 6.283 * std::rand() / RAND_MAX, l = 0.5 + 4.0 * std::rand() / RAND_MAX;
    RangeSample s;
    s.start = Position(1.0, -0.5);
    s.end = Position(1.0 + l * std::cos(a), -0.5 + l * std::sin(a));
    s.ifClearEnd = (k % 5 == 0);
    provider.bufferSample(s);
    og_ray r = {s.start[0], s.start[1], s.end[0], s.end[1], s.ifClearEnd ? 1 : 0, 0};
    rays.push_back(r);
  }
  // the "range" updater (sonars) works on its own layer and leaves master alone (map_provider.cpp:216-223)
  std::vector<float> range_ref((size_t)g.size[0] * g.size[1], NAN);
  {
    std::vector<rna_range_reading> sonar;
    std::vector<og_ray> sonar_rays;
    for (int k = 0; k < 40; ++k) {
      rna_range_reading m = {0.3f + 0.09f * k, 3.5f, -2.0 + 0.05 * k, 1.0, 0.16 * k};
      sonar.push_back(m);
      og_ray r;
      og_range_to_ray(m.range, m.max_range, m.x, m.y, m.yaw, &r);
      sonar_rays.push_back(r);
    }
    provider.bufferRanges(sonar);
    og_himm_update(&g, range_ref.data(), sonar_rays.data(), (int)sonar_rays.size(), nullptr);
    bool bad_type = false;
    try { provider.bufferSample(RangeSample(), "lidar3d"); } catch (const std::invalid_argument&) { bad_type = true; }
    CHECK(bad_type);
  }
  provider.updateMap();
  og_himm_update(&g, ref.data(), rays.data(), (int)rays.size(), nullptr);
  {
    std::vector<float> range_layer = map.get("range");
    for (size_t i = 0; i < range_ref.size(); ++i) CHECK(same_bits(range_layer[i], range_ref[i]));
  }
  std::vector<float> master = map.get("master");
  for (size_t i = 0; i < ref.size(); ++i) CHECK(same_bits(master[i], ref[i]));
  bool threw = false;
  try { map.get("elevation"); } catch (const std::out_of_range&) { threw = true; }
  CHECK(threw);  // unknown layer -> std::out_of_range as GridMap::get

  // ---- VFH with the Steerer's parameters (steerer.cpp:69-132) ----
  VFH vfh(100, 30, 5, 10, 50, 200, 200, 300, 200, 40, 40, 40, 1.0, 2000000.0, 4000000.0, 2000000.0, 4000000.0, 10.0, 1.0);
  vfh.SetRobotRadius(178.0);
  vfh.Init(map);
  og_vfh_params op;
  og_vfh_default_params(&op);
  og_vfh* ov = og_vfh_create(&op);
  double ranges[361][2];
  for (int step = 0; step < 5; ++step) {
    for (int i = 0; i < 361; ++i) { ranges[i][0] = 5000.0; ranges[i][1] = 0; }
    for (int i = 60 + 10 * step; i < 120; i += 2) ranges[i][0] = 700.0 + 13.0 * i;
    int cs = 0, ct = 0, ocs = 0, oct = 0;
    vfh.Update_VFH(ranges, 50 * step, 80.0f + 10 * step, 2500.0f, 250.0f, cs, ct, 0.2);
    og_vfh_update(ov, ranges, 50 * step, 80.0f + 10 * step, 2500.0f, 250.0f, 0.2, &ocs, &oct);
    CHECK(cs == ocs && ct == oct);
    CHECK(vfh.GetPickedAngle() == og_vfh_picked_angle(ov));
    for (int s = 0; s < vfh.getHistSize(); ++s) {
      CHECK(same_bits(vfh.Hist[s], og_vfh_hist(ov)[s]));
      CHECK(same_bits(vfh.OriginHist[s], og_vfh_origin_hist(ov)[s]));
    }
  }
  {  // Steerer::pubHist (steerer.cpp:201-220) from the resident histograms of the last step
    Histogram msg;
    vfh.pubHist(msg);
    uint16_t ox[36], oy[36], oyb[36], oth[2];
    CHECK(og_hist_msg(og_vfh_hist(ov), og_vfh_origin_hist(ov), 72, 5, ox, oy, oyb, oth) == 36 && msg.num_bin == 36);
    CHECK(msg.yLowThreshold == oth[0] && msg.yHighThreshold == oth[1]);
    for (int i = 0; i < 36; ++i) CHECK(msg.xData[i] == ox[i] && msg.yData[i] == oy[i] && msg.yBinData[i] == oyb[i]);
  }
  og_vfh_destroy(ov);

  {  // MapProvider::publishMap (map_provider.cpp:113-118,206-213)
    OccupancyGrid occ;
    provider.publishMap(occ);
    std::vector<int8_t> want(ref.size());
    og_to_occupancy_grid(&g, ref.data(), 0.0f, 255.0f, want.data());
    CHECK(occ.width == (unsigned)g.size[0] && occ.height == (unsigned)g.size[1] && occ.origin_x == -6.4 && occ.origin_y == -6.4);
    CHECK(occ.data == want);
  }

  {  // MapProvider::getSubMap (map_provider.cpp:93-100) -> GridMap::getSubmap: the Steerer's 1.5 m window and a
     // planning window that hangs over the map border (nav_node.cpp:141)
    const double wins[3][4] = {{1.0, -0.5, 1.5, 1.5}, {5.9, -6.1, 4.0, 4.0}, {0.0, 0.0, 100.0, 100.0}};
    for (int w = 0; w < 3; ++w) {
      GridMap::SubMap sm;
      Position c(wins[w][0], wins[w][1]);
      const bool ok = provider.getSubMap(sm, c, Length(wins[w][2], wins[w][3]));
      og_geom sg;
      std::vector<float> want(ref.size());
      const double p[2] = {wins[w][0], wins[w][1]}, l[2] = {wins[w][2], wins[w][3]};
      const int ook = og_get_submap(&g, ref.data(), p, l, &sg, want.data(), (int)want.size());
      CHECK(ok == (ook != 0));
      if (!ok) continue;
      CHECK(sm.size[0] == sg.size[0] && sm.size[1] == sg.size[1] && sm.position[0] == sg.pos[0] && sm.position[1] == sg.pos[1]);
      CHECK(sm.length[0] == sg.len[0] && sm.length[1] == sg.len[1] && sm.data.size() == (size_t)sg.size[0] * sg.size[1]);
      for (size_t i = 0; i < sm.data.size(); ++i) CHECK(same_bits(sm.data[i], want[i]));
    }
  }
  {  // Nav::makePlan (nav_node.cpp:136-152): planning window as a GridMap of its own, RRT on that window
    GridMap mapForPlan;
    Position rstart(1.0, -0.5), rtarget(9.0, 3.0);            // target outside the 8 m window: finish at its border
    CHECK(provider.getSubMap(mapForPlan, rstart, Length(8.0, 8.0)));
    og_geom sg;
    std::vector<float> sub(ref.size());
    const double p[2] = {rstart[0], rstart[1]}, l[2] = {8.0, 8.0};
    CHECK(og_get_submap(&g, ref.data(), p, l, &sg, sub.data(), (int)sub.size()));
    CHECK(mapForPlan.getSize()[0] == sg.size[0] && mapForPlan.getSize()[1] == sg.size[1]);
    CHECK(mapForPlan.getPosition()[0] == sg.pos[0] && mapForPlan.getPosition()[1] == sg.pos[1]);
    CHECK(mapForPlan.getLength()[0] == sg.len[0] && mapForPlan.getLength()[1] == sg.len[1]);
    std::vector<float> got = mapForPlan.get("master"), got_laser = mapForPlan.get("laser");
    for (size_t i = 0; i < got.size(); ++i) CHECK(same_bits(got[i], sub[i]) && same_bits(got_laser[i], sub[i]));
    {  // mapProvider_.publishMap(mapForPlan): the window as the local_map OccupancyGrid (nav_node.cpp:142)
      OccupancyGrid local;
      provider.publishMap(mapForPlan, local);
      std::vector<int8_t> want(got.size());
      og_to_occupancy_grid(&sg, sub.data(), 0.0f, 255.0f, want.data());
      CHECK(local.width == (unsigned)sg.size[0] && local.height == (unsigned)sg.size[1] && local.data == want);
      CHECK(local.origin_x == sg.pos[0] - 0.5 * sg.len[0] && local.origin_y == sg.pos[1] - 0.5 * sg.len[1]);
    }
    RrtPlanner wplanner(mapForPlan, rstart, rtarget, 0.2);
    std::vector<Position> wpath;
    const bool wok = wplanner.makePlan(wpath);
    std::vector<double> out(2 * 2048);
    og_rrt_result r;
    const double a[2] = {rstart[0], rstart[1]}, b[2] = {rtarget[0], rtarget[1]};
    og_rrt_plan(&sg, sub.data(), a, b, 0.2, 1, 1000000, out.data(), 2048, &r);
    CHECK(wok == (r.status == 1) && (size_t)r.path_len == wpath.size() && wpath.size() >= 2);
    for (int k = 0; k < r.path_len; ++k)
      CHECK(std::fabs(out[2 * k] - wpath[k][0]) < 1e-9 && std::fabs(out[2 * k + 1] - wpath[k][1]) < 1e-9);
    bool ok_far = true;
    GridMap none = map.getSubmap(Position(500.0, 500.0), Length(1.0, 1.0), ok_far);   // clamped window does not hold the centre
    CHECK(!ok_far && none.engine() == nullptr);
  }
  {  // Steerer::acceptPlan / update (steerer.cpp:27-33,222-270) on a fresh VFH instance, against the oracle
    VFH svfh(100, 30, 5, 10, 50, 200, 200, 300, 200, 40, 40, 40, 1.0, 2000000.0, 4000000.0, 2000000.0, 4000000.0, 10.0, 1.0);
    svfh.SetRobotRadius(178.0);
    svfh.Init(map);
    Steerer steerer(provider, svfh);
    std::vector<Position> plan;
    plan.push_back(Position(1.0, -0.5)); plan.push_back(Position(1.1, -0.45)); plan.push_back(Position(2.0, 0.5));
    plan.push_back(Position(2.1, 2.0));
    steerer.acceptPlan(plan);
    std::vector<double> pxy;
    for (size_t k = 0; k < plan.size(); ++k) { pxy.push_back(plan[k][0]); pxy.push_back(plan[k][1]); }
    og_vfh* sv = og_vfh_create(&op);
    int oidx = 1, steps = 0;
    double x = 1.0, y = -0.5, yaw = 0.3, vel = 0.0;
    for (int it = 0; it < 40; ++it) {
      int cs = 0, ct = 0;
      double lin = 0, ang = 0;
      const bool following = steerer.update(Position(x, y), yaw, vel, 0.2, cs, ct, lin, ang);
      float od[2];
      const int ofollow = og_follow_plan(pxy.data(), (int)plan.size(), &oidx, x, y, yaw, od);
      CHECK(following == (ofollow != 0) && steerer.planIndex() == oidx);
      if (!following) break;
      int ocs = 0, oct = 0;
      const double rp[2] = {x, y};
      og_vfh_step_pose(sv, &g, ref.data(), rp, yaw, (int)(vel * 1000.0), od[0], od[1], 250.0f, 0.2, &ocs, &oct);
      CHECK(cs == ocs && ct == oct);
      CHECK(lin == (float)cs / 1000.0 && ang == ct * M_PI / 180.0);
      vel = lin; yaw += ang * 0.2;
      x += 0.5 * (plan[oidx][0] - x); y += 0.5 * (plan[oidx][1] - y);   // the robot closes in on its way point
      ++steps;
    }
    CHECK(steps >= 3 && !steerer.ifPlanReady());
    og_vfh_destroy(sv);
  }

  // ---- planners ----
  Position start(-5.0, -5.0), target(5.5, 4.0);
  std::vector<Position> path;
  GridAStarPlanner grid_planner(map);
  CHECK(grid_planner.makePlan(start, target, path));
  {
    std::vector<uint8_t> blocked(ref.size()), nbr(ref.size());
    og_astar_blocked_mask(ref.data(), ref.size(), blocked.data());
    og_astar_nbr_mask(blocked.data(), g.size[0], g.size[1], nbr.data());
    int si[2], ti[2];
    const double sp[2] = {start[0], start[1]}, tp[2] = {target[0], target[1]};
    og_index_from_position(&g, sp, si);
    og_index_from_position(&g, tp, ti);
    std::vector<int32_t> gw(ref.size()), op_(ref.size());
    og_astar_result r;
    og_astar_query(nbr.data(), g.size[0], g.size[1], si[0] + si[1] * g.size[0], ti[0] + ti[1] * g.size[0], gw.data(),
                   op_.data(), (int)op_.size(), &r);
    CHECK(r.status == 0 && (size_t)r.path_len == path.size());
    for (int k = 0; k < r.path_len; ++k) {
      const int idx[2] = {op_[k] % g.size[0], op_[k] / g.size[0]};
      double p[2];
      og_position_from_index(&g, idx, p);
      CHECK(p[0] == path[k][0] && p[1] == path[k][1]);
    }
  }
  {  // Nav::taileredPlan (nav_node.cpp:192-204) on the detailed grid plan
    std::vector<Position> tailored;
    taileredPlan(path, tailored);
    std::vector<double> in(2 * path.size()), out(2 * path.size());
    for (size_t k = 0; k < path.size(); ++k) { in[2 * k] = path[k][0]; in[2 * k + 1] = path[k][1]; }
    const int m = og_tailor_plan(in.data(), (int)path.size(), 5, out.data());
    CHECK((size_t)m == tailored.size() && m >= 2);
    for (int k = 0; k < m; ++k) CHECK(out[2 * k] == tailored[k][0] && out[2 * k + 1] == tailored[k][1]);
  }
  AStarPlanner graph_planner(map);
  Position gs(3.0, 0.5), gt(19.0, 10.5);
  std::vector<Position> gpath;
  CHECK(graph_planner.makePlan(gs, gt, gpath));
  {
    double out[64];
    const double a[2] = {gs[0], gs[1]}, b[2] = {gt[0], gt[1]};
    const int n = og_graph_make_plan(a, b, out, 32);
    CHECK((size_t)n == gpath.size());
    for (int k = 0; k < n; ++k) CHECK(out[2 * k] == gpath[k][0] && out[2 * k + 1] == gpath[k][1]);
  }
  RrtPlanner rrt(map, start, target);
  std::vector<Position> rpath;
  const bool rrt_ok = rrt.makePlan(rpath);
  {
    std::vector<double> out(2 * 2048);
    og_rrt_result r;
    const double a[2] = {start[0], start[1]}, b[2] = {target[0], target[1]};
    og_rrt_plan(&g, ref.data(), a, b, 0.2, 1, 1000000, out.data(), 2048, &r);
    CHECK(rrt_ok == (r.status == 1) && (size_t)r.path_len == rpath.size());
    for (int k = 0; k < r.path_len; ++k)
      CHECK(std::fabs(out[2 * k] - rpath[k][0]) < 1e-9 && std::fabs(out[2 * k + 1] - rpath[k][1]) < 1e-9);
  }
  std::printf("host mirror OK (%zu grid-A* cells, %zu graph waypoints, %zu RRT waypoints)\n", path.size(), gpath.size(), rpath.size());
  return 0;
}
