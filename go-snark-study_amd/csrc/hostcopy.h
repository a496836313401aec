// Host -> device copies from PAGEABLE caller memory (the reference's signatures hand over plain Go slices).  hipMemcpyAsync from
// pageable memory stages through the runtime on the calling thread at ~12 GB/s; here the staging is ours: a few persistent host
// threads memcpy 8 MiB pieces into pinned buffers (each context owns three) while the previous pieces' DMAs are in flight.
// Round 5 (profiles/r05_ab_stage_pieces.txt): with 4 MiB pieces and two buffers a host-buffer ticket spent 2.8 + 5.0 ms of host time
// staging its 32 + 64 MiB of w and px.  The host copies themselves take 0.57 + 1.14 ms (56 GB/s with eight threads, 35 with one); the
// rest is waiting for DMAs that run at 10-30 GB/s beside the proofs' kernels (alone: 50-57).  8 MiB pieces over three buffers and ONE
// wait for w and px: the px-from-host stream 1.07x -> 1.02-1.03x the resident one.  Pieces of 16 MiB and more take a slower path
// in the runtime while the device is busy (in-place updates 10.0 -> 11.0 ms per proof; 32 MiB pieces: 7.5 ms per 32 MiB).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "runtime.h"

namespace gs {

class HostCopyPool {                       // process-wide; workers sleep on a condition variable between jobs
 public:
  static HostCopyPool& get() {
    static HostCopyPool* p = new HostCopyPool();       // leaked on purpose: worker threads must not be joined from a static destructor
    return *p;
  }
  // a train of jobs is coming (staged_h2d): the workers may watch for the next job instead of sleeping between two of them
  struct Train {
    Train() { get().trains_.fetch_add(1, std::memory_order_relaxed); }
    ~Train() { get().trains_.fetch_sub(1, std::memory_order_relaxed); }
    Train(const Train&) = delete;
    Train& operator=(const Train&) = delete;
  };
  // dst[0..n) = src[0..n), split over the workers and the calling thread; returns when every byte has been copied
  void copy(void* dst, const void* src, size_t n) {
    if (n < (256u << 10) || workers_.empty()) { memcpy(dst, src, n); return; }
    std::lock_guard<std::mutex> one_job(job_mu_);
    const size_t parts = workers_.size() + 1;
    Job j;
    j.dst = static_cast<char*>(dst); j.src = static_cast<const char*>(src); j.n = n;
    j.piece = ((n + parts - 1) / parts + 4095) & ~size_t(4095);
    {
      std::lock_guard<std::mutex> lk(mu_);
      j.gen = ++generation_;
      job_ = j;
      left_ = (n + j.piece - 1) / j.piece;
      next_.store(j.gen << 32);                       // the piece counter carries its job: a late worker of job g - 1 cannot take a piece of g
    }
    cv_.notify_all();
    run_pieces(j);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return left_ == 0; });
  }

 private:
  HostCopyPool() {
    unsigned hw = std::thread::hardware_concurrency();
    int want = (int)run_knob("GS_COPY_THREADS", 8, 1, 64);
    if (hw && (unsigned)want > hw) want = (int)hw;
    for (int i = 1; i < want; ++i) workers_.emplace_back([this] { loop(); });
    for (auto& t : workers_) t.detach();
  }
  struct Job { char* dst = nullptr; const char* src = nullptr; size_t n = 0, piece = 1; uint64_t gen = 0; };
  void run_pieces(const Job& j) {
    for (;;) {
      uint64_t ticket = next_.load();
      do {
        if ((ticket >> 32) != j.gen) return;           // another job owns the counter by now: do not touch it
      } while (!next_.compare_exchange_weak(ticket, ticket + 1));
      const size_t off = (size_t)(ticket & 0xffffffffu) * j.piece;
      if (off >= j.n) return;
      memcpy(j.dst + off, j.src + off, std::min(j.piece, j.n - off));
      std::lock_guard<std::mutex> lk(mu_);
      if (--left_ == 0) done_cv_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      // a staged upload is a train of jobs ~0.3 ms apart: after a job, watch the piece counter for the next one for a while before
      // sleeping (a futex wake-up costs 50-100 us -- as long as a thread's whole share of a 16 MiB piece)
      // (only while a staged upload is in progress -- `trains_` is raised by staged_h2d for its duration -- so that seven
      //  cores do not spin for 400 us after the LAST job of every upload, beside the host-side folds and the Go runtime)
      if (seen && trains_.load(std::memory_order_relaxed) > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
        while (trains_.load(std::memory_order_relaxed) > 0 && (next_.load(std::memory_order_relaxed) >> 32) == seen &&
               std::chrono::steady_clock::now() < until) cpu_relax();
      }
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        j = job_;
      }
      run_pieces(j);
    }
  }
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  std::vector<std::thread> workers_;
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  Job job_;
  size_t left_ = 0;
  std::atomic<uint64_t> next_{0};
  std::atomic<int> trains_{0};
  uint64_t generation_ = 0;
};

// Enqueue dst_dev[0..bytes) = src_host[0..bytes) on `stream`; returns when the LAST piece has been staged (its DMA may still be in
// flight: it is ordered on the stream like any other operation).  The caller holds the context lock.
inline void staged_h2d(Ctx& c, void* dst_dev, const void* src_host, size_t bytes, hipStream_t stream) {
  if (bytes < (1u << 20)) { GS_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream)); return; }
  // piece size and buffers in rotation: every piece costs a wake-up of the copy threads, an event wait and a DMA submission
  static const size_t piece = (size_t)run_knob("GS_STAGE_MIB", 8, 1, (long)(Ctx::kStageBytes >> 20)) << 20;
  static const int nbuf = (int)run_knob("GS_STAGE_BUFFERS", 3, 2, Ctx::kStageBuffers);
  for (int b = 0; b < nbuf; ++b) {
    if (!c.stage[b]) GS_HIP(hipHostMalloc(&c.stage[b], piece, hipHostMallocDefault));
    if (!c.stage_ev[b]) GS_HIP(hipEventCreateWithFlags(&c.stage_ev[b], hipEventDisableTiming));
  }
  HostCopyPool& pool = HostCopyPool::get();
  HostCopyPool::Train train;
  static const bool trace = run_flag("GS_HOST_TRACE");
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_wait = 0, t_copy = 0, t_submit = 0;
  size_t off = 0;
  for (int i = 0; off < bytes; ++i) {
    const int b = i % nbuf;
    const size_t len = std::min(piece, bytes - off);
    const double t0 = trace ? now() : 0;
    GS_HIP(hipEventSynchronize(c.stage_ev[b]));                     // the DMA that last read this buffer (no-op on a fresh event)
    const double t1 = trace ? now() : 0;
    pool.copy(c.stage[b], static_cast<const char*>(src_host) + off, len);
    const double t2 = trace ? now() : 0;
    GS_HIP(hipMemcpyAsync(static_cast<char*>(dst_dev) + off, c.stage[b], len, hipMemcpyHostToDevice, stream));
    GS_HIP(hipEventRecord(c.stage_ev[b], stream));
    if (trace) { t_wait += t1 - t0; t_copy += t2 - t1; t_submit += now() - t2; }
    off += len;
  }
  if (trace) fprintf(stderr, "[gs host] staged_h2d %zu MiB: buffer waits %.3f ms, host copies %.3f ms, DMA submissions %.3f ms\n", bytes >> 20, t_wait, t_copy, t_submit);
}

}  // namespace gs
