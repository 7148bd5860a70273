// Per-device "already done" words for one-time, per-device launch preparation (hipFuncSetAttribute grants of dynamic LDS
// above 64 KiB are a property of the (function, device) pair): a function-local `static bool done` would make the first
// device's grant hide the missing one on every other device of the process.  Racing writers store the same value.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace gps_dev {

constexpr int kMaxDevices = 32;

inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return d;
}

// N words per device; row() = the calling thread's current device
template <typename T, int N>
struct PerDevice {
  T v[kMaxDevices][N] = {};
  T *row() { return v[current_device()]; }
};

}  // namespace gps_dev
