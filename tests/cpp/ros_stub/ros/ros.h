#pragma once
#include <cstdio>
#include <string>
#include <boost/bind.hpp>
#include <ros/time.h>
namespace ros {
struct Publisher {
  template <class M> void publish(const M&) const {}
};
struct Subscriber {};
struct TimerEvent {};
struct Timer { void stop() {} };
struct Rate { explicit Rate(double) {} bool sleep() { return true; } };
struct NodeHandle {
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  bool ok() const;
  template <class M> Publisher advertise(const std::string&, unsigned, bool latch = false);
  template <class M, class T> Subscriber subscribe(const std::string&, unsigned, void (T::*)(const boost::shared_ptr<M const>&), T*);
  template <class T> Timer createTimer(Duration, void (T::*)(const TimerEvent&), T*, bool oneshot = false);
  template <class V> bool param(const std::string&, V&, const V&) const;
  template <class V> bool getParam(const std::string&, V&) const;
};
void init(int&, char**, const std::string&);
void spin();
bool ok();
void shutdown();
}
#define ROS_INFO(...) std::printf(__VA_ARGS__)
#define ROS_WARN(...) std::printf(__VA_ARGS__)
#define ROS_ERROR(...) std::printf(__VA_ARGS__)
#define ROS_ERROR_THROTTLE(period, ...) std::printf(__VA_ARGS__)
#define ROS_INFO_THROTTLE(period, ...) std::printf(__VA_ARGS__)
