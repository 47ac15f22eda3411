// collectives.h — the exchange step of a frame-sharded solve (bodyfit_solve_sharded*), host side.
//
// The sharded window LM needs ONE kind of collective: an all-gather of a few doubles (or of the 2 N interface blocks) from
// every rank's device buffer into every rank's device buffer, ordered on the solve's stream.  Two transports:
//   RcclTransport   RCCL (ncclAllGather) on the device buffers and the solve's stream: nothing touches the host, no stream
//                   synchronisation.  librccl is bound at run time (dlopen "librccl.so.1": the process usually has it
//                   already, through torch), so libbodyfit.so itself does not link against it.
//   HostTransport   the caller's bodyfit_comm callbacks on host buffers (D2H, synchronise, callback, H2D): the transport of
//                   the multi-process tests (torch.distributed "gloo", several ranks sharing the one GPU of a test box) and
//                   of hosts that bring MPI.  It also counts its calls (tests assert the number of exchanges per iteration).
// Sums are never taken by the transport: every rank receives every rank's partials and adds them in rank order, so all
// ranks hold bit-identical totals and take identical decisions without a broadcast.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bodyfit.h"
#include "exchange_timeout.h"

namespace bodyfit {

// hipStreamSynchronize with a bound (seconds <= 0: unbounded): a collective that a failed peer never enters would otherwise hold
// the status read of a sharded solve for ever.  1 = the bound passed (the stream is still busy), 0 = idle, -1 = a HIP error.
inline int wait_stream(hipStream_t st, double seconds, hipError_t* err) {
  *err = hipSuccess;
  if (!(seconds > 0.0)) { *err = hipStreamSynchronize(st); return *err == hipSuccess ? 0 : -1; }
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) { *err = q; return -1; }
    if (std::chrono::steady_clock::now() >= t_end) return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

// the few RCCL entry points used, resolved from librccl at run time (signatures: rccl/rccl.h of ROCm 7.2)
struct RcclApi {
  typedef void* comm_t;
  struct unique_id { char internal[128]; };
  int (*GetUniqueId)(unique_id*) = nullptr;
  int (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*CommCount)(comm_t, int*) = nullptr;
  int (*CommUserRank)(comm_t, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
  static constexpr int kDouble = 8, kSum = 0;   // ncclFloat64, ncclSum

  static RcclApi& get() {
    static RcclApi api = [] {
      RcclApi a;
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) { a.error = std::string("librccl not found: ") + dlerror(); return a; }
      auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p && a.error.empty()) a.error = std::string("librccl lacks ") + n; return p; };
      a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
      a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
      a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
      a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
      a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
      a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
      a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(sym("ncclCommUserRank"));
      return a;
    }();
    return api;
  }
  bool ok() const { return error.empty(); }
};

struct Transport {
  int rank = 0, size = 1;
  long n_calls = 0;
  double timeout_s = 0.0;     // bodyfit_set_exchange_timeout: bound of one exchange (0: none)
  bool timed_out = false;
  std::string error;
  virtual ~Transport() {}
  // every rank's n doubles at d_send -> d_recv [size][n] on every rank, ordered on `st`
  virtual int allgather(const double* d_send, double* d_recv, int n, hipStream_t st) = 0;
};

struct RcclTransport : Transport {
  RcclApi::comm_t comm = nullptr;
  int allgather(const double* d_send, double* d_recv, int n, hipStream_t st) override {
    ++n_calls;
    RcclApi& A = RcclApi::get();
    const int rc = A.AllGather(d_send, d_recv, (size_t)n, RcclApi::kDouble, comm, st);
    if (rc != 0) { error = std::string("ncclAllGather: ") + (A.GetErrorString ? A.GetErrorString(rc) : "error"); return 1; }
    return 0;
  }
};

struct HostTransport : Transport {
  bodyfit_comm cb{};
  int allgather(const double* d_send, double* d_recv, int n, hipStream_t st) override {
    ++n_calls;
    // (the buffers are shared with the helper thread of a bounded exchange: a callback that outlives the bound still owns them)
    auto send = std::make_shared<std::vector<double>>((size_t)n);
    auto recv = std::make_shared<std::vector<double>>((size_t)n * size);
    hipError_t he = hipMemcpyAsync(send->data(), d_send, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess && wait_stream(st, timeout_s, &he) == 1) { timed_out = true; error = "allgather: the stream did not drain within the exchange timeout"; return 1; }
    if (he != hipSuccess) { error = "allgather: device to host copy failed"; return 1; }
    const bodyfit_comm c = cb;
    const int rc = call_with_timeout([c, send, recv, n]() { return c.allgather(c.ctx, send->data(), recv->data(), n); }, timeout_s, &timed_out);
    if (timed_out) { error = "allgather callback did not return within the exchange timeout"; return 1; }
    if (rc) { error = "allgather callback failed"; return 1; }
    if (hipMemcpyAsync(d_recv, recv->data(), (size_t)n * size * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { error = "allgather: host to device copy failed"; return 1; }   // (recv lives until here)
    return 0;
  }
};

}  // namespace bodyfit
