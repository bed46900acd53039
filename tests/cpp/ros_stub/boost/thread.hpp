#pragma once
// the handful of boost::thread names the reference's node mains use
#include <condition_variable>
#include <mutex>
#include <thread>
#include <boost/bind.hpp>
namespace boost {
typedef std::mutex mutex;
typedef std::recursive_mutex recursive_mutex;
template <class M> using unique_lock = std::unique_lock<M>;
typedef std::condition_variable_any condition_variable_any;
typedef std::condition_variable condition_variable;
struct thread {
  thread() {}
  template <class F> explicit thread(F f) {}
  void join() {}
  void interrupt() {}
};
}
