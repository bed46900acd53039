#pragma once
namespace ros {
struct Duration {
  Duration() {}
  explicit Duration(double) {}
  bool sleep() const;
  double toSec() const;
};
struct Time {
  Time() {}
  explicit Time(double) {}
  static Time now();
  double toSec() const;
  Time operator+(const Duration&) const;
  bool operator==(const Time&) const;
};
}
