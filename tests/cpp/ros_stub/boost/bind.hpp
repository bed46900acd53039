#pragma once
#include <functional>
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
using std::bind;
}
using std::placeholders::_1;
