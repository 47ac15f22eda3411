// CPU-side test of the exchange bound of a sharded solve (3dbodyanimation_amd/csrc/exchange_timeout.h, bodyfit_set_exchange_timeout):
// two "ranks" (threads) run a loop of all-gathers through bodyfit_comm-shaped callbacks over a condition-variable rendezvous —
// what gloo / MPI are to the real solve.  Rank 1's transport fails in exchange 3 (its callback returns non-zero without entering
// the collective) and rank 1 leaves, as bodyfit_solve_sharded does.  Rank 0 is then alone in a collective that can never
// complete: without a bound it would sit there for ever; with one, BOTH ranks are back within it.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../3dbodyanimation_amd/csrc/exchange_timeout.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

namespace {
struct Rendezvous {          // an all-gather of n doubles between `size` callers; blocks until everybody has arrived
  std::mutex mu;
  std::condition_variable cv;
  int size = 2, arrived = 0, generation = 0;
  std::shared_ptr<std::vector<double>> cur;    // the slots of the generation being filled (its readers keep it alive)
};
struct Ctx { std::shared_ptr<Rendezvous> rv; int rank; };

int allgather_cb(void* vctx, const double* send, double* recv, int n) {
  Ctx* c = static_cast<Ctx*>(vctx);
  Rendezvous& r = *c->rv;
  std::unique_lock<std::mutex> lk(r.mu);
  if (r.arrived == 0) r.cur = std::make_shared<std::vector<double>>((size_t)n * r.size, 0.0);
  std::shared_ptr<std::vector<double>> mine = r.cur;
  for (int i = 0; i < n; ++i) (*mine)[(size_t)c->rank * n + i] = send[i];
  const int gen = r.generation;
  if (++r.arrived == r.size) { r.arrived = 0; ++r.generation; r.cv.notify_all(); }
  else r.cv.wait(lk, [&] { return r.generation != gen; });        // (no timeout of its own: the worst transport)
  for (size_t i = 0; i < mine->size(); ++i) recv[i] = (*mine)[i];
  return 0;
}

struct Result { int exchanges_done = 0; bool timed_out = false; int rc = 0; double seconds = 0.0; };

// the exchange loop of one rank, shaped like HostTransport::allgather: buffers shared with the helper thread, callback by value
Result rank_loop(std::shared_ptr<Ctx> ctx, int n_exchanges, int fail_at, double bound) {
  Result out;
  const auto t0 = std::chrono::steady_clock::now();
  for (int e = 0; e < n_exchanges; ++e) {
    auto send = std::make_shared<std::vector<double>>(4, (double)(10 * ctx->rank + e));
    auto recv = std::make_shared<std::vector<double>>(8, -1.0);
    bool to = false;
    int rc;
    if (e == fail_at) rc = 1;      // this rank's transport fails: it never enters the collective
    else rc = bodyfit::call_with_timeout([ctx, send, recv] { return allgather_cb(ctx.get(), send->data(), recv->data(), 4); }, bound, &to);
    if (to || rc) { out.timed_out = to; out.rc = rc; break; }
    if ((*recv)[0] != (double)e || (*recv)[4] != (double)(10 + e)) { out.rc = 99; break; }
    ++out.exchanges_done;
  }
  out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return out;
}
}  // namespace

int main() {
  // (1) healthy transport, with and without a bound: same results, every exchange completes
  for (double bound : {0.0, 5.0}) {
    auto rv = std::make_shared<Rendezvous>();
    Result r[2];
    std::thread t1([&] { r[1] = rank_loop(std::make_shared<Ctx>(Ctx{rv, 1}), 50, -1, bound); });
    r[0] = rank_loop(std::make_shared<Ctx>(Ctx{rv, 0}), 50, -1, bound);
    t1.join();
    CHECK(r[0].exchanges_done == 50 && r[1].exchanges_done == 50 && !r[0].timed_out && !r[1].timed_out && r[0].rc == 0 && r[1].rc == 0);
  }
  // (2) rank 1's transport fails in exchange 3: rank 1 returns at once, rank 0 within the bound (0.3 s), not never
  {
    auto rv = std::make_shared<Rendezvous>();
    Result r[2];
    std::thread t1([&] { r[1] = rank_loop(std::make_shared<Ctx>(Ctx{rv, 1}), 50, 3, 0.3); });
    r[0] = rank_loop(std::make_shared<Ctx>(Ctx{rv, 0}), 50, -1, 0.3);
    t1.join();
    CHECK(r[1].exchanges_done == 3 && r[1].rc == 1 && !r[1].timed_out && r[1].seconds < 0.25);
    CHECK(r[0].exchanges_done == 3 && r[0].timed_out && r[0].rc == -1);
    CHECK(r[0].seconds >= 0.3 && r[0].seconds < 2.0);
    // rank 0's helper thread is still inside the collective, holding its own references to the buffers and the context: let it
    // finish so that the process ends cleanly (a real transport would time out or be torn down by the caller)
    {
      std::lock_guard<std::mutex> lk(rv->mu);
      ++rv->generation; rv->arrived = 0;
    }
    rv->cv.notify_all();
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
  // (3) a callback that throws is a failed exchange, not a crash
  {
    bool to = false;
    const int rc = bodyfit::call_with_timeout([]() -> int { throw 1; }, 1.0, &to);
    CHECK(rc == -1 && !to);
  }
  std::printf("ok\n");
  return 0;
}
