// fcpp_parallel.h -- a small persistent worker pool for the host-side setup of a batch (fcpp_host.cpp: per-field plan, fcpp_tiler.cpp:
// per-field tiling).  Fields are independent, so the setup of a batch is cut into BLOCKS of consecutive fields; the blocks are handed
// out through an atomic counter (dynamic balance), every block writes only its own results, and the merge walks the blocks in order --
// what comes out does not depend on the number of threads or on which thread took which block.
//
// The pool is created on first use and never destroyed (its threads sleep on a condition variable between calls; process exit ends
// them).  A fork()ed child has none of the parent's threads: the atfork handler drops the pool and the child creates its own on demand.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace fcpp {

class WorkerPool {
public:
    // threads that take part in a parallel_for, the caller included: FCPP_THREADS, else the hardware's, at most 16 (a one-GPU box's share)
    static int width()
    {
        static const int w = [] {
            const char *e = getenv("FCPP_THREADS");
            int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
            if (n < 1) n = 1;
            if (n > 16 && !e) n = 16;
            if (n > 256) n = 256;
            return n;
        }();
        return w;
    }

    // fn(k) for k in [0, n_items), every k exactly once; returns when all have finished.  One parallel_for at a time: calls from
    // several host threads take turns.  An exception thrown by fn (std::bad_alloc from a growing vector) is caught where it is thrown,
    // the remaining items are skipped, every thread is waited for, and the FIRST exception is rethrown in the calling thread.
    static void parallel_for(int64_t n_items, const std::function<void(int64_t)> &fn)
    {
        if (n_items <= 0) return;
        const int w = width();
        if (w == 1 || n_items == 1) {
            for (int64_t k = 0; k < n_items; ++k) fn(k);
            return;
        }
        WorkerPool &p = instance();
        std::lock_guard<std::mutex> turn(p.turn_);
        p.error_ = nullptr;
        p.failed_.store(false, std::memory_order_relaxed);
        while ((int)p.n_workers_ < w - 1) {
            std::thread([&p] { p.worker(); }).detach();
            ++p.n_workers_;
        }
        {
            std::lock_guard<std::mutex> lk(p.m_);
            p.fn_ = &fn; p.n_items_ = n_items; p.next_.store(0, std::memory_order_relaxed);
            p.busy_ = p.n_workers_;
            ++p.generation_;
        }
        p.cv_.notify_all();
        p.drain(fn, n_items);
        std::unique_lock<std::mutex> lk(p.m_);
        p.done_.wait(lk, [&] { return p.busy_ == 0; });
        p.fn_ = nullptr;
        if (p.error_) {
            std::exception_ptr e = p.error_;
            p.error_ = nullptr;
            lk.unlock();
            std::rethrow_exception(e);
        }
    }

private:
    static std::atomic<WorkerPool *> &slot()
    {
        static std::atomic<WorkerPool *> s{ nullptr };
        return s;
    }
    static WorkerPool &instance()
    {
        // (no mutex here: a fork()ed child whose parent held one at the time of the fork would wait for it for ever; the pool is published
        // by one compare-and-swap, a thread that loses the race deletes its copy)
        static std::once_flag once;
        std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { slot().store(nullptr); }); });     // (the parent's object leaks in the child)
        WorkerPool *p = slot().load(std::memory_order_acquire);
        if (!p) {
            WorkerPool *mine = new WorkerPool();
            if (slot().compare_exchange_strong(p, mine, std::memory_order_acq_rel)) p = mine;
            else delete mine;
        }
        return *p;
    }
    void drain(const std::function<void(int64_t)> &fn, int64_t n_items)
    {
        for (;;) {
            const int64_t k = next_.fetch_add(1, std::memory_order_relaxed);
            if (k >= n_items) break;
            if (failed_.load(std::memory_order_relaxed)) continue;       // (an item has thrown: the rest is skipped, the counter runs out)
            try { fn(k); }
            catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                if (!error_) error_ = std::current_exception();
                failed_.store(true, std::memory_order_relaxed);
            }
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int64_t)> *fn;
            int64_t n;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                fn = fn_; n = n_items_;
            }
            drain(*fn, n);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--busy_ == 0) done_.notify_all();
            }
        }
    }

    std::mutex turn_, m_;
    std::condition_variable cv_, done_;
    const std::function<void(int64_t)> *fn_ = nullptr;
    int64_t n_items_ = 0;
    std::atomic<int64_t> next_{ 0 };
    int n_workers_ = 0, busy_ = 0;
    uint64_t generation_ = 0;
    std::exception_ptr error_;           // the first exception an item threw (guarded by m_)
    std::atomic<bool> failed_{ false };
};

}  // namespace fcpp
