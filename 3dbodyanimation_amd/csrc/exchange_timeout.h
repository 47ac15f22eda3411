// exchange_timeout.h — a bound on one exchange of a frame-sharded solve (bodyfit_set_exchange_timeout, include/bodyfit.h).
//
// The exchanges of bodyfit_solve_sharded are the CALLER's collectives (bodyfit_comm callbacks: torch.distributed "gloo" in the
// tests, MPI in a C++ host).  When one rank's transport fails, that rank returns at once — and its peers sit in the same
// collective until the transport's own timeout, if it has one.  With a bound set, the callback runs on a helper thread and the
// solve waits for it at most that long: every rank is back in the caller within the bound, whatever the transport does.  A
// callback that never returns keeps its helper thread (detached) and the buffers it was given (shared ownership); nothing
// the solve owns is referenced after the bound has passed.
//
// No HIP in this header: tests/cpp/exchange_timeout_test.cpp drives it on a CPU-only box.
#pragma once
#include <chrono>
#include <functional>
#include <future>
#include <memory>
#include <thread>

namespace bodyfit {

// Runs fn (which must own, by value or shared_ptr, everything it touches) and returns its result; seconds <= 0: on the calling
// thread, unbounded.  Otherwise on a helper thread, and after `seconds` without a result *timed_out = true and -1 is returned.
inline int call_with_timeout(std::function<int()> fn, double seconds, bool* timed_out) {
  *timed_out = false;
  if (!(seconds > 0.0)) return fn();
  auto prom = std::make_shared<std::promise<int>>();
  std::future<int> fut = prom->get_future();
  std::thread([prom, fn = std::move(fn)]() mutable {
    int rc = -1;
    try { rc = fn(); } catch (...) { rc = -1; }
    prom->set_value(rc);
  }).detach();
  if (fut.wait_for(std::chrono::duration<double>(seconds)) == std::future_status::ready) return fut.get();
  *timed_out = true;
  return -1;
}

}  // namespace bodyfit
