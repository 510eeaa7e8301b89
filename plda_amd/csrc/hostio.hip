// plda_amd/csrc/hostio.hip -- moving caller (pageable) memory to and from HBM at the speed of the link.
//
// The reference's API hands over NumPy arrays (host memory): pldamodule.cpp:64-74 copies them into Kaldi
// matrices.  Here they cross PCIe, and a plain hipMemcpy on pageable memory runs at 10-28 GB/s -- the
// runtime stages through a small pinned buffer on ONE thread, and a freshly allocated output array takes a
// page fault per 4 KiB on that thread ("PCIe").  So:
//   * a ring of three pinned 64 MiB slots per handle (allocated on first use),
//   * a small pool of host threads that memcpy between the caller's array and a slot -- page faults of a
//     fresh output and the copy itself are spread over the threads,
//   * copies between a slot and HBM on a dedicated copy stream, ordered against the compute stream by events,
//     so that producing slab i+1 (GEMM), shipping slab i (DMA) and landing slab i-1 (host threads) overlap.
// The host threads only move bytes; no arithmetic of the path runs on the CPU.
#include "hostio.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace plda {

CopyPool::CopyPool(int n) {
  for (int i = 0; i < n; ++i) th.emplace_back([this] { run(); });
}

CopyPool::~CopyPool() {
  {
    std::lock_guard<std::mutex> g(mu);
    stop = true;
  }
  cv.notify_all();
  for (auto &t : th) t.join();
}

void CopyPool::run() {
  for (;;) {
    Task t;
    {
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [this] { return stop || !q.empty(); });
      if (q.empty()) return;
      t = q.front();
      q.pop_front();
    }
    if (t.dpitch == t.row_bytes && t.spitch == t.row_bytes) {
      std::memcpy(t.dst, t.src, t.row_bytes * t.rows);
    } else {
      for (size_t r = 0; r < t.rows; ++r) std::memcpy(t.dst + r * t.dpitch, t.src + r * t.spitch, t.row_bytes);
    }
    {
      std::lock_guard<std::mutex> g(mu);
      --t.job->pending;
    }
    done.notify_all();
  }
}

// a 2-D copy cut into one piece per thread (contiguous copies are cut by bytes, not rows)
void CopyPool::submit(Job *job, char *dst, size_t dpitch, const char *src, size_t spitch, size_t row_bytes, size_t rows) {
  if (!rows || !row_bytes) return;
  const bool flat = dpitch == row_bytes && spitch == row_bytes;
  const size_t units = flat ? (row_bytes * rows + 4095) / 4096 : rows;   // flat: 4 KiB granules
  const size_t parts = std::min<size_t>(th.size(), units);
  std::lock_guard<std::mutex> g(mu);
  for (size_t p = 0; p < parts; ++p) {
    const size_t u0 = units * p / parts, u1 = units * (p + 1) / parts;
    Task t;
    t.job = job;
    if (flat) {
      const size_t total = row_bytes * rows, b0 = u0 * 4096, b1 = std::min(total, u1 * 4096);
      t.dst = dst + b0; t.src = src + b0; t.dpitch = t.spitch = t.row_bytes = b1 - b0; t.rows = 1;
    } else {
      t.dst = dst + u0 * dpitch; t.src = src + u0 * spitch; t.dpitch = dpitch; t.spitch = spitch;
      t.row_bytes = row_bytes; t.rows = u1 - u0;
    }
    ++job->pending;
    q.push_back(t);
  }
  cv.notify_all();
}

void CopyPool::wait(Job *job) {
  std::unique_lock<std::mutex> g(mu);
  done.wait(g, [job] { return job->pending == 0; });
}

HostPipe::HostPipe(int nthreads) : pool(nthreads) {}

HostPipe::~HostPipe() {
  for (auto &j : job) pool.wait(&j);
  for (auto &e : ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : ev_ready) if (e) (void)hipEventDestroy(e);
  for (auto &e : ev_up) if (e) (void)hipEventDestroy(e);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (ring) (void)hipHostFree(ring);
}

hipError_t HostPipe::init() {
  if (ring) return hipSuccess;
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ring), SLOT_BYTES * NS, hipHostMallocDefault);
  if (e != hipSuccess) { ring = nullptr; return e; }
  for (int s = 0; s < NS; ++s) slot[s] = ring + (size_t)s * SLOT_BYTES;
  if ((e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)) != hipSuccess) return e;
  for (auto &x : ev) if ((e = hipEventCreateWithFlags(&x, hipEventDisableTiming)) != hipSuccess) return e;
  for (auto &x : ev_ready) if ((e = hipEventCreateWithFlags(&x, hipEventDisableTiming)) != hipSuccess) return e;
  return hipSuccess;
}

// Transparent huge pages for a large caller array that is about to be written for the first time: with THP in
// "madvise" mode this turns 512 page faults into one.  Purely a hint; failures are ignored.
void advise_huge(void *p, size_t bytes) {
  const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + ((1u << 21) - 1)) & ~(uintptr_t)((1u << 21) - 1);
  const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(uintptr_t)((1u << 21) - 1);
  if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
}

// host [rows, row_bytes] (pitch spitch) -> device, contiguous rows of row_bytes; enqueued on `stream`
// (the caller's later work on that stream sees the data); returns after the last slot has been handed to the DMA
hipError_t HostPipe::upload(hipStream_t stream, void *ddst, const void *hsrc, size_t bytes) {
  hipError_t e = init();
  if (e != hipSuccess) return e;
  if (bytes < ((size_t)4 << 20))   // small: the runtime's own staging is as fast and has no thread hand-over
    return hipMemcpyAsync(ddst, hsrc, bytes, hipMemcpyHostToDevice, stream);
  const char *src = static_cast<const char *>(hsrc);
  char *dst = static_cast<char *>(ddst);
  // chunks of a quarter slot: the first DMA starts after 16 MiB of host copying, not after 64
  const size_t CH = SLOT_BYTES / 4;
  const int NC = NS * 4;
  size_t i = 0;
  for (size_t off = 0; off < bytes; off += CH, ++i) {
    const size_t n = std::min(CH, bytes - off);
    const int c = (int)(i % NC);
    char *stage = ring + (size_t)c * CH;
    if (i >= (size_t)NC && (e = hipEventSynchronize(ev_up[c])) != hipSuccess) return e;   // the DMA out of this chunk is done
    pool.submit(&up_job, stage, n, src + off, n, n, 1);
    pool.wait(&up_job);
    if ((e = hipMemcpyAsync(dst + off, stage, n, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if (!ev_up[c] && (e = hipEventCreateWithFlags(&ev_up[c], hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(ev_up[c], stream)) != hipSuccess) return e;
  }
  // the ring must not be reused (by a download, or the next upload) before the last DMAs have read it
  return hipStreamSynchronize(stream);
}

// ---- download pipeline: the caller produces slab i into dev_buf(i) on its compute stream, then ships it ----
hipError_t HostPipe::begin_slab(hipStream_t compute, size_t i) {
  // device buffer i % 2 was last read by the DMA of slab i - 2
  if (i >= 2) return hipStreamWaitEvent(compute, ev[(i - 2) % NS], 0);
  return hipSuccess;
}

hipError_t HostPipe::ship_slab(hipStream_t compute, size_t i, const void *dsrc, size_t bytes, char *hdst, size_t dpitch,
                               size_t row_bytes, size_t rows) {
  const int s = (int)(i % NS);
  hipError_t e;
  if ((e = hipEventRecord(ev_ready[i % 2], compute)) != hipSuccess) return e;
  pool.wait(&job[s]);                                           // the host copy out of this slot (slab i - NS) is done
  if ((e = hipStreamWaitEvent(copy_stream, ev_ready[i % 2], 0)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(slot[s], dsrc, bytes, hipMemcpyDeviceToHost, copy_stream)) != hipSuccess) return e;
  if ((e = hipEventRecord(ev[s], copy_stream)) != hipSuccess) return e;
  pend[s] = Pending{hdst, dpitch, row_bytes, rows, true};
  if (i >= 1) return land((i - 1) % NS);                        // slab i - 1 has had a whole slab's time to arrive
  return hipSuccess;
}

hipError_t HostPipe::land(int s) {
  if (!pend[s].live) return hipSuccess;
  hipError_t e = hipEventSynchronize(ev[s]);
  if (e != hipSuccess) return e;
  const Pending &p = pend[s];
  pool.submit(&job[s], p.dst, p.dpitch, slot[s], p.row_bytes, p.row_bytes, p.rows);
  pend[s].live = false;
  return hipSuccess;
}

hipError_t HostPipe::finish() {
  hipError_t first = hipSuccess;
  for (int s = 0; s < NS; ++s) {
    const hipError_t e = land(s);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  for (auto &j : job) pool.wait(&j);
  return first;
}

int default_host_threads() {
  if (const char *v = std::getenv("PLDA_HOST_THREADS")) {
    const int n = std::atoi(v);
    if (n > 0) return std::min(n, 64);
  }
  const long hw = sysconf(_SC_NPROCESSORS_ONLN);
  return (int)std::max<long>(2, std::min<long>(16, hw / 2));
}

}  // namespace plda
