// worker_pool.hpp -- the host threads of the batch file paths (capi.hip).  Header-only and free of HIP so that
// tests/cpp/worker_pool_test.cpp can run it under ThreadSanitizer on a machine without a GPU.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace jpeg_amd {

// The host threads of the batch file paths, kept in the context between calls and handed one parallel region after the other.
// (Starting 32 threads per chunk cost more than half of what a chunk of 32 1080p files takes them; starting and ending them
// per CALL still meant 32 stacks unmapped per call, and an munmap is what the GPU driver's MMU notifier answers by stopping
// the queues: every other batch of 512 files took 50 instead of 20 ms.)  Items are drawn from a counter (files differ in length); the calling thread works too.  Not re-entrant:
// one region at a time.
class WorkerPool {
public:
    explicit WorkerPool(int nthreads)
    {
        try {
            threads_.reserve((size_t)std::max(0, nthreads - 1));
            for (int t = 1; t < nthreads; ++t) threads_.emplace_back([this, t] { work(t - 1); });
        } catch (...) {      // fewer threads than asked for: the ones that did start (and the caller) do the work
        }
    }
    ~WorkerPool()
    {
        finish();
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        go_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    int size() const { return (int)threads_.size() + 1; }      // the calling thread included
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    // fn(i) for i in [0, count); fn does not throw.  begin() hands the region to the workers and returns; finish() has the
    // calling thread take its share and waits for the rest.
    // `threads`: how many threads may work on the region, the calling one (in finish()) included; the pool may be larger
    void begin(int count, std::function<void(int)> fn, int threads = 1 << 30)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = std::move(fn); count_ = std::max(0, count); next_.store(0);
            limit_ = std::max(0, threads - 1);
            busy_ = (int)threads_.size();
            ++generation_;
            open_ = true;
        }
        go_.notify_all();
    }
    void finish()
    {
        if (!open_) return;
        for (int i; (i = next_.fetch_add(1)) < count_;) job_(i);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return busy_ == 0; });
        open_ = false;
    }
    void run(int count, std::function<void(int)> fn, int threads = 1 << 30)
    {
        begin(count, std::move(fn), threads);
        finish();
    }

private:
    void work(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            int count;
            {
                std::unique_lock<std::mutex> g(m_);
                go_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                count = id < limit_ ? count_ : -1;             // (a thread beyond the region's limit only reports back: it must not draw an item)
            }
            if (count >= 0)
                for (int i; (i = next_.fetch_add(1)) < count;) job_(i);   // (job_ is not touched until every worker has reported back)
            std::lock_guard<std::mutex> g(m_);
            if (--busy_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable go_, done_;
    std::function<void(int)> job_;
    std::atomic<int> next_{0};
    int count_ = 0, busy_ = 0, limit_ = 0;
    bool open_ = false;
    unsigned long generation_ = 0;
    bool stop_ = false;
};

}  // namespace jpeg_amd
