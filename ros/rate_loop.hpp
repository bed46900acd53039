// rate_loop.hpp -- the body of the reference's three loops as a thread of its own, without ROS:
//   MapProvider::loopUpdateAndPublishMap (5 Hz) / loopMoveMap (2 Hz)   mc/src/map_provider.cpp:151-188, started :225-233
//   Steerer::vfhLoop (5 Hz)                                            mc/src/steerer.cpp:135-144
// each of them `ros::Rate r(hz); while (nh.ok()) { body(); r.sleep(); }`.  RateLoop keeps the rate the way ros::Rate
// does -- the next cycle is due one period after the previous one was DUE, not after it ended; a cycle that overran by
// more than a period resets the schedule (and is counted, the reference logs it: map_provider.cpp:170-173) -- with an
// injectable clock and sleep, so that the rate keeping is unit-tested with a simulated clock (tests/cpp/rate_loop_test.cpp)
// and the ROS build passes ros::Time / ros::Duration::sleep (simulated time under use_sim_time).
#pragma once

#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>

namespace move_control {

class RateLoop {
 public:
  typedef std::function<double()> Clock;          // seconds
  typedef std::function<void(double)> Sleep;      // seconds
  typedef std::function<bool()> Ok;               // nh.ok()

  RateLoop(double hz, std::function<void()> body, Ok ok = Ok(), Clock now = Clock(), Sleep sleep = Sleep())
      : period_(1.0 / hz), body_(body), ok_(ok), now_(now ? now : Clock(&RateLoop::steadyNow)),
        sleep_(sleep ? sleep : Sleep(&RateLoop::steadySleep)), stop_(false), cycles_(0), overruns_(0), lastCycle_(0.0) {}
  ~RateLoop() { stop(); }
  RateLoop(const RateLoop&) = delete;
  RateLoop& operator=(const RateLoop&) = delete;

  void start() {
    if (thread_.joinable()) return;
    stop_ = false;
    thread_ = std::thread([this] { run(); });
  }
  void stop() {
    stop_ = true;
    if (thread_.joinable()) thread_.join();
  }
  // the loop itself (start() runs it on a thread; a test calls it with an `ok` that ends it)
  void run() {
    double due = now_();                 // ros::Rate: start_ = now at construction
    while (!stop_ && (!ok_ || ok_())) {
      body_();
      due = sleepUntilNextCycle(due);
    }
  }
  // ros::Rate::sleep: the cycle that began at `start` is due at start + period.  Returns the start of the next cycle.
  double sleepUntilNextCycle(double start) {
    double expectedEnd = start + period_;
    const double actualEnd = now_();
    if (actualEnd < start) expectedEnd = actualEnd + period_;   // the clock went backwards (simulated time was reset)
    const double remaining = expectedEnd - actualEnd;
    lastCycle_ = actualEnd - start;
    cycles_ += 1;
    if (remaining <= 0.0) {
      if (actualEnd > expectedEnd + period_) { overruns_ += 1; return actualEnd; }   // more than a cycle late: reset
      return expectedEnd;
    }
    sleep_(remaining);
    return expectedEnd;
  }
  double period() const { return period_; }
  long cycles() const { return cycles_; }
  long overruns() const { return overruns_; }
  double lastCycleTime() const { return lastCycle_; }   // ros::Rate::cycleTime()

 private:
  static double steadyNow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  static void steadySleep(double s) { std::this_thread::sleep_for(std::chrono::duration<double>(s)); }
  double period_;
  std::function<void()> body_;
  Ok ok_;
  Clock now_;
  Sleep sleep_;
  std::atomic<bool> stop_;
  std::atomic<long> cycles_, overruns_;
  double lastCycle_;
  std::thread thread_;
};

// The latest message of a topic, as ContinuousSensorHelperRos keeps it for /odom
// (mc/include/move_control/continuous_sensor_helper_ros.h:9-63): the callback overwrites, readers copy.
template <class T>
class LatestValue {
 public:
  LatestValue() : have_(false) {}
  void set(const T& v) { std::lock_guard<std::mutex> l(m_); v_ = v; have_ = true; }
  bool get(T& out) const { std::lock_guard<std::mutex> l(m_); if (!have_) return false; out = v_; return true; }
 private:
  mutable std::mutex m_;
  T v_;
  bool have_;
};

// LaserMapUpdater::bufferIncomingMsg drops scans that arrive sooner than msgProcessCycle_ (0.2 s) after the last one it
// took (mc/src/laser_map_updater.cpp:40-44)
class MinInterval {
 public:
  explicit MinInterval(double seconds) : min_(seconds), last_(-1e300) {}
  bool take(double now) {
    if (last_ + min_ > now) return false;
    last_ = now;
    return true;
  }
 private:
  double min_, last_;
};

}  // namespace move_control
