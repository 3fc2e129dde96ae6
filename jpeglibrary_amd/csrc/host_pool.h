// jpeglibrary_amd/csrc/host_pool.h -- a crew of host threads for one jpgpu_batch_upload call.
//
// The reference is single-threaded by design ("one decoder per thread is the implied usage", SURVEY 8b Threading); a batch
// of files is the place where the host side may fan out: header parsing of the files and the copies into the pinned
// staging ring are independent per file.  The crew lives for one upload call (threads are started on first use and
// joined by the destructor); run() hands item indices out through an atomic counter and returns when all are done.
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace jpgpu {

class WorkCrew {
  public:
    explicit WorkCrew(int threads) : n_threads_(threads < 1 ? 1 : threads) {}
    ~WorkCrew() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    WorkCrew(const WorkCrew &) = delete;
    WorkCrew &operator=(const WorkCrew &) = delete;
    int threads() const { return n_threads_; }

    // fn(item, worker) for every item in [0, n); worker in [0, threads()).  The calling thread works too (as worker 0).
    // The first exception thrown by any fn is rethrown here after all items were handed out.
    void run(size_t n, const std::function<void(size_t, int)> &fn) {
        if (n == 0) return;
        if (n_threads_ == 1 || n == 1) {
            for (size_t i = 0; i < n; i++) fn(i, 0);
            return;
        }
        start_workers();
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_items_ = n;
            next_.store(0, std::memory_order_relaxed);
            busy_ = (int)workers_.size();
            error_ = nullptr;
            generation_++;
        }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return busy_ == 0; });
        fn_ = nullptr;
        if (error_) std::rethrow_exception(error_);
    }

  private:
    void start_workers() {
        if (!workers_.empty()) return;
        for (int w = 1; w < n_threads_; w++) workers_.emplace_back([this, w] { loop(w); });
    }
    void work(int worker) {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_items_) break;
            try {
                (*fn_)(i, worker);
            } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                if (!error_) error_ = std::current_exception();
            }
        }
    }
    void loop(int worker) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
            }
            work(worker);
            {
                std::lock_guard<std::mutex> lk(m_);
                busy_--;
            }
            done_cv_.notify_one();
        }
    }

    const int n_threads_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t, int)> *fn_ = nullptr;
    size_t n_items_ = 0;
    std::atomic<size_t> next_{0};
    int busy_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
    std::exception_ptr error_;
};

}  // namespace jpgpu
