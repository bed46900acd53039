#pragma once
#include <stdexcept>
#include <string>
#include <ros/time.h>
namespace tf {
struct Vector3 { double x() const; double y() const; double z() const; };
struct Quaternion {
  Quaternion();
  Quaternion(const double& x, const double& y, const double& z, const double& w);
  const double& x() const; const double& y() const; const double& z() const; const double& w() const;
};
double getYaw(const Quaternion&);
struct StampedTransform {
  Vector3 getOrigin() const;
  Quaternion getRotation() const;
  ros::Time stamp_;
};
struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; };
}
