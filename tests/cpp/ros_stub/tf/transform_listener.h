#pragma once
#include <tf/transform_datatypes.h>
namespace tf {
struct TransformListener {
  explicit TransformListener(ros::Duration = ros::Duration(10.0)) {}
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) const;
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const;
};
}
