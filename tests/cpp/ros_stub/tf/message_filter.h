#pragma once
#include <string>
#include <tf/transform_listener.h>
namespace tf {
template <class M>
struct MessageFilter {
  template <class F> MessageFilter(F&, TransformListener&, const std::string&, unsigned) {}
  template <class C> void registerCallback(const C&) {}
};
}
