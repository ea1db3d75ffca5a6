// A handful of host threads for the library's host-only set-up work (rp_corridor_coeffs): created on first use, asleep on a
// condition variable between jobs (a job of ~1 ms is too short to pay thread creation per call, ~15 us a thread), the calling
// thread works along.  Items of a job are handed out one by one through an atomic counter.  Plain C++11; no OpenMP runtime beside
// the one a host application (PyTorch) may bring.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>

namespace rppool {

class Pool {
public:
    static Pool &get() {
        static Pool *p = new Pool();   // (never destroyed: worker threads may outlive static destruction at process exit)
        return *p;
    }
    int threads() const { return (int)workers_.size() + 1; }

    // f(i) for every i in [0, n), on the pool's threads and the caller; returns when all are done.  One job at a time.
    // An item that throws (std::bad_alloc from a growing vector) is counted as done and reported: false = some item threw.  Nothing
    // leaves a worker thread (std::terminate) or unwinds through the extern "C" caller with the other items still running.
    bool parallel_for(int n, const std::function<void(int)> &f) {
        if (n <= 0) return true;
        if (workers_.empty() || n == 1) {
            bool ok = true;
            for (int i = 0; i < n; ++i) { try { f(i); } catch (...) { ok = false; } }
            return ok;
        }
        std::lock_guard<std::mutex> job(job_mutex_);
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &f; n_ = n; next_.store(0); done_.store(0); failed_.store(false); ++generation_;
        }
        cv_.notify_all();
        run_items();
        // (a worker joins a job and leaves it under the mutex: none is inside run_items once active_ is back to 0, and one that wakes
        //  up after that finds no job)
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [&] { return done_.load() >= n_ && active_ == 0; });
        fn_ = nullptr;
        return !failed_.load();
    }

private:
    Pool() {
        int want = 8;
        if (const char *e = std::getenv("RP_AMD_HOST_THREADS")) want = std::atoi(e);
        // the CPUs this process may run on (affinity mask: what a container or taskset leaves of the machine), not the machine's:
        // threads that spin-wait on each other's results (rp_corridor_coeffs) must not outnumber them
        int hw = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0 && (hw <= 0 || n < hw)) hw = n; }
        if (hw > 0 && want > hw) want = hw;
        for (int i = 1; i < want; ++i) workers_.emplace_back([this] { worker(); });
        for (auto &t : workers_) t.detach();
    }
    void run_items() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            try { (*fn_)(i); } catch (...) { failed_.store(true); }
            done_.fetch_add(1);
        }
    }
    void worker() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (!fn_) continue;   // (the job is over already)
                ++active_;
            }
            run_items();
            {
                std::lock_guard<std::mutex> g(m_);
                --active_;
            }
            done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, job_mutex_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0, active_ = 0;
    std::atomic<int> next_{0}, done_{0};
    std::atomic<bool> failed_{false};
    unsigned long long generation_ = 0;
};

}  // namespace rppool
