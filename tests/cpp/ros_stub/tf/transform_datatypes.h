#pragma once
#include <stdexcept>
#include <string>
#include <ros/time.h>
namespace tf {
struct Vector3 { double x() const; double y() const; double z() const; };
struct Quaternion {};
double getYaw(const Quaternion&);
struct StampedTransform {
  Vector3 getOrigin() const;
  Quaternion getRotation() const;
  ros::Time stamp_;
};
struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; };
}
