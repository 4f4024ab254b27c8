// WorkerPool (jpeg_amd/csrc/worker_pool.hpp) under ThreadSanitizer: every item of every region runs exactly once, a region
// uses no more threads than it was given, begin() returns before the work is done and finish() joins it, regions of every
// size follow each other on one pool, and the queue pattern of jpeg_amd_decompress_batch (threads that block on a condition
// inside their item until the directing thread lets them go on) comes to an end.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <vector>
#include "worker_pool.hpp"

using jpeg_amd::WorkerPool;

int main()
{
    std::mt19937 rng(7);
    int failures = 0;
    for (int pool_threads : {1, 2, 5, 16}) {
        WorkerPool pool(pool_threads);
        for (int round = 0; round < 300; ++round) {
            const int count = (int)(rng() % 70), limit = 1 + (int)(rng() % (unsigned)(pool_threads + 2));
            std::vector<std::atomic<int>> hits((size_t)std::max(count, 1));
            for (auto &h : hits) h.store(0);
            std::mutex m;
            std::set<std::thread::id> who;
            const bool split = rng() & 1;
            auto job = [&](int i) {
                hits[(size_t)i].fetch_add(1);
                std::lock_guard<std::mutex> g(m);
                who.insert(std::this_thread::get_id());
            };
            if (split) { pool.begin(count, job, limit); pool.finish(); }
            else pool.run(count, job, limit);
            for (int i = 0; i < count; ++i) if (hits[(size_t)i].load() != 1) { std::printf("pool %d round %d: item %d ran %d times\n", pool_threads, round, i, hits[(size_t)i].load()); ++failures; }
            if ((int)who.size() > std::min(limit, pool_threads)) { std::printf("pool %d round %d: %zu threads worked, limit %d\n", pool_threads, round, who.size(), limit); ++failures; }
        }
        // the directing-thread pattern: items block until they are let through, chunk by chunk
        for (int round = 0; round < 20; ++round) {
            const int files = 1 + (int)(rng() % 200), chunk = 1 + (int)(rng() % 32), nchunks = (files + chunk - 1) / chunk;
            const int t_n = std::min(pool_threads, files);
            std::mutex m; std::condition_variable cv;
            int open_chunks = std::min(2, nchunks);
            std::vector<int> left((size_t)nchunks);
            for (int k = 0; k < nchunks; ++k) left[(size_t)k] = std::min(chunk, files - k * chunk);
            std::atomic<int> next{0}, done{0};
            auto worker = [&](int) {
                for (;;) {
                    const int f = next.fetch_add(1);
                    if (f >= files) return;
                    const int k = f / chunk;
                    { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return open_chunks > k; }); }
                    done.fetch_add(1);
                    std::lock_guard<std::mutex> g(m);
                    if (--left[(size_t)k] == 0) cv.notify_all();
                }
            };
            if (pool.size() < 2) continue;   // (one thread: nobody to direct)
            pool.begin(t_n, worker, t_n + 1);
            for (int k = 0; k < nchunks; ++k) {
                if (k + 1 < nchunks) { { std::lock_guard<std::mutex> g(m); open_chunks = std::max(open_chunks, k + 2); } cv.notify_all(); }
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return left[(size_t)k] == 0; });
            }
            pool.finish();
            if (done.load() != files) { std::printf("pool %d: %d of %d files\n", pool_threads, done.load(), files); ++failures; }
        }
    }
    std::printf("%s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
