// device_once.h — per-DEVICE (not per-process) bookkeeping of kernel attributes.  Plain C++ (no HIP types), so that the CPU test
// suite can exercise it (tests/cpp/device_once_test.cpp).
#pragma once
#include <atomic>
#include <cstddef>

namespace bodyfit {

// A kernel's dynamic-LDS grant (hipFuncSetAttribute) is an attribute of the kernel ON ONE DEVICE: a process that creates
// models on several devices needs it on each of them.  Bookkeeping per device, not per process.
struct DeviceOnce {
  std::atomic<unsigned long long> done[4] = {};       // bit d of word d / 64: granted on device d (d < 256)
  bool first(int device) {                             // true exactly once per device
    const unsigned d = (unsigned)device & 255u;
    const unsigned long long bit = 1ull << (d & 63u);
    return (done[d >> 6].fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
  }
};
struct DeviceMax {                                     // a grant that grows: true when `want` exceeds what device d has
  std::atomic<size_t> granted[256] = {};
  bool raise(int device, size_t want, size_t initial) {
    std::atomic<size_t>& g = granted[(unsigned)device & 255u];
    size_t cur = g.load(std::memory_order_acquire);
    if (cur == 0) cur = initial;
    if (want <= cur) return false;
    g.store(want, std::memory_order_release);
    return true;
  }
};

}  // namespace bodyfit
