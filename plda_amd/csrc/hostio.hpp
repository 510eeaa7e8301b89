// plda_amd/csrc/hostio.hpp -- pinned ring + host copy threads behind the host-pointer entry points (hostio.hip)
#pragma once

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace plda {

// a few threads that memcpy; jobs are counted so that a slot's consumer can wait for exactly its pieces
class CopyPool {
 public:
  struct Job { int pending = 0; };   // guarded by the pool's mutex
  explicit CopyPool(int nthreads);
  ~CopyPool();
  void submit(Job *job, char *dst, size_t dpitch, const char *src, size_t spitch, size_t row_bytes, size_t rows);
  void wait(Job *job);
  int threads() const { return (int)th.size(); }

 private:
  struct Task { Job *job; char *dst; const char *src; size_t dpitch, spitch, row_bytes, rows; };
  void run();
  std::vector<std::thread> th;
  std::deque<Task> q;
  std::mutex mu;
  std::condition_variable cv, done;
  bool stop = false;
};

struct HostPipe {
  static constexpr int NS = 3;
  static constexpr size_t SLOT_BYTES = (size_t)64 << 20;
  explicit HostPipe(int nthreads);
  ~HostPipe();
  hipError_t init();
  // pageable host -> device through the ring, ordered on `stream`; synchronises `stream` before returning
  hipError_t upload(hipStream_t stream, void *ddst, const void *hsrc, size_t bytes);
  // device -> pageable host, slab by slab (each <= SLOT_BYTES): begin_slab before producing slab i into a buffer that
  // alternates with i % 2, ship_slab after it, finish at the end (lands what is in flight, waits for the host copies)
  hipError_t begin_slab(hipStream_t compute, size_t i);
  hipError_t ship_slab(hipStream_t compute, size_t i, const void *dsrc, size_t bytes, char *hdst, size_t dpitch,
                       size_t row_bytes, size_t rows);
  hipError_t finish();

  CopyPool pool;

 private:
  struct Pending { char *dst; size_t dpitch, row_bytes, rows; bool live; };
  hipError_t land(int s);
  char *ring = nullptr;
  char *slot[NS] = {};
  CopyPool::Job job[NS], up_job;
  Pending pend[NS] = {};
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev[NS] = {}, ev_ready[2] = {}, ev_up[NS * 4] = {};
};

void advise_huge(void *p, size_t bytes);
int default_host_threads();

}  // namespace plda
