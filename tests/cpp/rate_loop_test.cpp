// rate_loop_test.cpp -- the ROS-free parts of the ROS seams (ros/rate_loop.hpp) with a simulated clock: the three loops
// of the reference keep 5 / 2 / 5 Hz the way ros::Rate does, overruns are counted and reset the schedule, the latest
// /odom message is what a reader sees, scans closer than 0.2 s to the last accepted one are dropped.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../ros/rate_loop.hpp"

using move_control::LatestValue;
using move_control::MinInterval;
using move_control::RateLoop;

int main() {
  // 1. a body that takes 30 ms at 5 Hz: cycles are due every 200 ms exactly (no drift)
  {
    double t = 100.0;
    std::vector<double> starts;
    int n = 0;
    RateLoop loop(5.0, [&] { starts.push_back(t); t += 0.030; ++n; }, [&] { return n < 50; }, [&] { return t; }, [&](double s) { t += s; });
    loop.run();
    assert(starts.size() == 50);
    for (size_t k = 1; k < starts.size(); ++k) assert(std::fabs(starts[k] - starts[k - 1] - 0.2) < 1e-9);
    assert(loop.overruns() == 0 && loop.cycles() == 50 && std::fabs(loop.lastCycleTime() - 0.030) < 1e-9);
  }
  // 2. one body call takes 0.27 s (less than one extra period): the next cycle starts at once, the schedule is kept
  {
    double t = 0.0;
    std::vector<double> starts;
    int n = 0;
    RateLoop loop(5.0, [&] { starts.push_back(t); t += (n == 3 ? 0.27 : 0.01); ++n; }, [&] { return n < 8; }, [&] { return t; }, [&](double s) { t += s; });
    loop.run();
    assert(std::fabs(starts[4] - (starts[3] + 0.27)) < 1e-9);      // late start
    assert(std::fabs(starts[5] - (starts[3] + 0.4)) < 1e-9);       // back on the 200 ms grid of cycle 3
    assert(loop.overruns() == 0);
  }
  // 3. a body call of 0.65 s (more than a period late): counted, and the schedule restarts from "now"
  {
    double t = 0.0;
    std::vector<double> starts;
    int n = 0;
    RateLoop loop(5.0, [&] { starts.push_back(t); t += (n == 2 ? 0.65 : 0.01); ++n; }, [&] { return n < 6; }, [&] { return t; }, [&](double s) { t += s; });
    loop.run();
    assert(loop.overruns() == 1);
    assert(std::fabs(starts[3] - (starts[2] + 0.65)) < 1e-9 && std::fabs(starts[4] - (starts[3] + 0.2)) < 1e-9);
  }
  // 4. 2 Hz (loopMoveMap) next to 5 Hz on the same simulated clock: 2 : 5 calls
  {
    double t = 0.0;
    int a = 0, b = 0;
    RateLoop move(2.0, [&] { ++a; }, [&] { return a < 4; }, [&] { return t; }, [&](double s) { t += s; });
    move.run();
    const double span = t;
    t = 0.0;
    RateLoop upd(5.0, [&] { ++b; }, [&] { return t < span - 1e-9; }, [&] { return t; }, [&](double s) { t += s; });
    upd.run();
    assert(a == 4 && b == 10);
  }
  // 5. start() / stop() on a real thread
  {
    std::atomic<int> n(0);
    RateLoop loop(200.0, [&] { ++n; });
    loop.start();
    while (n < 5) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    loop.stop();
    const int at_stop = n;
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    assert(n == at_stop);
  }
  // 6. latest value, minimum interval
  {
    LatestValue<double> odom;
    double v = -1;
    assert(!odom.get(v));
    odom.set(0.1); odom.set(0.25);
    assert(odom.get(v) && v == 0.25);
    MinInterval every(0.2);
    assert(every.take(10.0) && !every.take(10.1) && !every.take(10.19) && every.take(10.2) && every.take(11.0));
  }
  std::printf("rate loop ok\n");
  return 0;
}
