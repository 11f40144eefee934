// A few host worker threads for the small independent solves of a lockstep group check: the
// 16 Rayleigh-Ritz problems of a group (20-100 us each on one core) used to run one after the
// other on the calling thread -- 1.0-1.9 ms per group during which the group's stream has
// nothing queued but one speculative block (SC_GROUP_TRACE: "host Rayleigh-Ritz + analysis").
// run(count, fn): fn(0) .. fn(count - 1) on the workers AND the caller, returns when all are
// done.  One pool per lead handle (a handle is single-threaded), created on first use.
#ifndef SPECTRALCLUSTER_AMD_HOST_POOL_H_
#define SPECTRALCLUSTER_AMD_HOST_POOL_H_

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lock(m_);
      stop_ = true;
    }
    wake_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;

  void run(int count, const std::function<void(int)>& fn) {
    if (count <= 0) return;
    if (count == 1 || threads_.empty()) {
      for (int i = 0; i < count; ++i) fn(i);
      return;
    }
    {
      std::unique_lock<std::mutex> lock(m_);
      // (a worker that woke late for the previous job may still be on its way out of work())
      idle_.wait(lock, [this] { return active_ == 0; });
      fn_ = &fn;
      count_ = count;
      next_.store(0, std::memory_order_relaxed);
      pending_ = count;
      ++epoch_;
    }
    wake_.notify_all();
    work();  // the caller takes items too
    std::unique_lock<std::mutex> lock(m_);
    done_.wait(lock, [this] { return pending_ == 0; });
  }

 private:
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= count_) return;
      (*fn_)(i);
      std::lock_guard<std::mutex> lock(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(m_);
        wake_.wait(lock, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
        ++active_;  // (count_ / fn_ of this epoch were written under this lock)
      }
      work();
      std::lock_guard<std::mutex> lock(m_);
      if (--active_ == 0) idle_.notify_all();
    }
  }

  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable wake_, done_, idle_;
  const std::function<void(int)>* fn_ = nullptr;
  int count_ = 0, pending_ = 0, active_ = 0;
  std::atomic<int> next_{0};
  unsigned long long epoch_ = 0;
  bool stop_ = false;
};

#endif  // SPECTRALCLUSTER_AMD_HOST_POOL_H_
