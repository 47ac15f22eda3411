// CPU-side unit test of the per-device kernel-attribute bookkeeping (3dbodyanimation_amd/csrc/device_once.h): a grant made on one
// device must not be taken for granted on another (hipFuncSetAttribute is per device), and under concurrent first use no
// thread may come back from run() / raise() before the grant it relies on has been made.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../3dbodyanimation_amd/csrc/device_once.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
  using namespace bodyfit;
  DeviceOnce once;
  int grants = 0;
  auto g = [&] { ++grants; };
  CHECK(once.run(0, g) && grants == 1);
  CHECK(!once.run(0, g) && grants == 1);
  CHECK(once.run(3, g) && grants == 2);        // another device: its own first time
  CHECK(!once.run(3, g));
  CHECK(once.run(63, g) && once.run(64, g) && once.run(255, g));   // word boundaries of the bit set
  CHECK(!once.run(64, g) && !once.run(255, g) && !once.run(0, g) && grants == 5);
  DeviceMax grant;
  size_t last = 0;
  auto gm = [&](size_t want) { last = want; };
  CHECK(!grant.raise(0, 48 * 1024, 48 * 1024, gm) && last == 0);     // within what every device starts with
  CHECK(grant.raise(0, 72 * 1024, 48 * 1024, gm) && last == 72 * 1024);      // needs more: set the attribute
  CHECK(!grant.raise(0, 64 * 1024, 48 * 1024, gm) && last == 72 * 1024);     // covered by the earlier grant
  CHECK(grant.raise(1, 64 * 1024, 48 * 1024, gm) && last == 64 * 1024);      // device 1 has not been granted anything yet
  CHECK(grant.raise(0, 80 * 1024, 48 * 1024, gm) && last == 80 * 1024);
  // concurrent first use: exactly one thread per device runs the (slow) grant, and NO thread returns before it is complete
  DeviceOnce race;
  std::atomic<int> total{0}, early{0};
  std::atomic<int> granted[8] = {};
  std::vector<std::thread> th;
  for (int t = 0; t < 64; ++t)
    th.emplace_back([&, t] {
      const int d = t % 8;
      race.run(d, [&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(20));   // hipFuncSetAttribute takes its time
        granted[d].store(1);
        total.fetch_add(1);
      });
      if (granted[d].load() != 1) early.fetch_add(1);                 // "would launch without the attribute"
    });
  for (auto& x : th) x.join();
  CHECK(total.load() == 8);
  CHECK(early.load() == 0);
  // concurrent raises with different wants: the record and the "attribute" both end at the largest
  DeviceMax mx;
  std::atomic<size_t> attr{48 * 1024};
  std::atomic<int> short_of{0};
  th.clear();
  for (int t = 0; t < 32; ++t)
    th.emplace_back([&, t] {
      const size_t want = (size_t)(49 + t) * 1024;
      mx.raise(2, want, 48 * 1024, [&](size_t w) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        attr.store(w);
      });
      if (attr.load() < want) short_of.fetch_add(1);                  // "would launch with more LDS than granted"
    });
  for (auto& x : th) x.join();
  CHECK(short_of.load() == 0);
  CHECK(attr.load() == 80 * 1024 && mx.granted[2].load() == 80 * 1024);
  std::printf("device_once_test ok\n");
  return 0;
}
