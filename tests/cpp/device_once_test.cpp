// CPU-side unit test of the per-device kernel-attribute bookkeeping (3dbodyanimation_amd/csrc/device_once.h): a grant made on one
// device must not be taken for granted on another (hipFuncSetAttribute is per device), also under concurrent first use.
#include <cstdio>
#include <thread>
#include <vector>

#include "../../3dbodyanimation_amd/csrc/device_once.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
  using namespace bodyfit;
  DeviceOnce once;
  CHECK(once.first(0));
  CHECK(!once.first(0));
  CHECK(once.first(3));        // another device: its own first time
  CHECK(!once.first(3));
  CHECK(once.first(63) && once.first(64) && once.first(255));   // word boundaries of the bit set
  CHECK(!once.first(64) && !once.first(255) && !once.first(0));
  DeviceMax grant;
  CHECK(!grant.raise(0, 48 * 1024, 48 * 1024));     // within what every device starts with
  CHECK(grant.raise(0, 72 * 1024, 48 * 1024));      // needs more: set the attribute
  CHECK(!grant.raise(0, 64 * 1024, 48 * 1024));     // covered by the earlier grant
  CHECK(grant.raise(1, 64 * 1024, 48 * 1024));      // device 1 has not been granted anything yet
  CHECK(grant.raise(0, 80 * 1024, 48 * 1024));
  // concurrent first use: exactly one thread per device wins
  DeviceOnce race;
  std::vector<int> wins(8, 0);
  std::vector<std::thread> th;
  std::atomic<int> total{0};
  for (int t = 0; t < 32; ++t)
    th.emplace_back([&, t] { if (race.first(t % 8)) total.fetch_add(1); });
  for (auto& x : th) x.join();
  CHECK(total.load() == 8);
  std::printf("device_once_test ok\n");
  return 0;
}
