// device_once.h — per-DEVICE (not per-process) bookkeeping of kernel attributes.  Plain C++ (no HIP types), so that the CPU test
// suite can exercise it (tests/cpp/device_once_test.cpp).
#pragma once
#include <atomic>
#include <cstddef>
#include <mutex>

namespace bodyfit {

// A kernel's dynamic-LDS grant (hipFuncSetAttribute) is an attribute of the kernel ON ONE DEVICE: a process that creates
// models on several devices needs it on each of them.  Bookkeeping per device, not per process — with ONCE semantics that
// cover the grant itself: the `done` bit is set only after the caller's grant function has returned, under the same lock,
// so a second thread can never see "already granted" and launch before the attribute exists.
struct DeviceOnce {
  std::atomic<unsigned long long> done[4] = {};       // bit d of word d / 64: granted on device d (d < 256)
  std::mutex mu;
  template <typename Grant>
  bool run(int device, Grant grant) {                  // true for the one caller per device that ran `grant`
    const unsigned d = (unsigned)device & 255u;
    const unsigned long long bit = 1ull << (d & 63u);
    if (done[d >> 6].load(std::memory_order_acquire) & bit) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (done[d >> 6].load(std::memory_order_relaxed) & bit) return false;
    grant();
    done[d >> 6].fetch_or(bit, std::memory_order_release);
    return true;
  }
};
// A grant that grows: `grant(want)` runs when `want` exceeds what device d has, under the lock that also publishes the new
// value (two threads with different wants leave the attribute and the record at the larger one).
struct DeviceMax {
  std::atomic<size_t> granted[256] = {};
  std::mutex mu;
  template <typename Grant>
  bool raise(int device, size_t want, size_t initial, Grant grant) {
    std::atomic<size_t>& g = granted[(unsigned)device & 255u];
    size_t cur = g.load(std::memory_order_acquire);
    if (want <= (cur ? cur : initial)) return false;
    std::lock_guard<std::mutex> lock(mu);
    cur = g.load(std::memory_order_relaxed);
    if (want <= (cur ? cur : initial)) return false;
    grant(want);
    g.store(want, std::memory_order_release);
    return true;
  }
};

}  // namespace bodyfit
