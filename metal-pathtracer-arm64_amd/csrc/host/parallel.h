// Host worker threads that cannot take the process down: every worker body runs inside a try block, every started thread is
// joined whatever happens (also when starting a later thread fails), and the first exception is rethrown on the calling thread
// afterwards - so a std::bad_alloc inside a builder pass reaches the C boundary's catch-all as an error message instead of
// std::terminate.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ptr {

// fn(k) for k in [0, n): k = 0 on the calling thread, the others on threads of their own.
template <typename Fn>
void runOnThreads(uint32_t n, Fn&& fn) {
    if (n <= 1u) {
        if (n == 1u) fn(0u);
        return;
    }
    std::exception_ptr first;
    std::mutex guard;
    auto body = [&](uint32_t k) {
        try {
            fn(k);
        } catch (...) {
            std::lock_guard<std::mutex> lock(guard);
            if (!first) first = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve(n - 1u);
        for (uint32_t k = 1; k < n; ++k) pool.emplace_back(body, k);
    } catch (...) {   // std::system_error: no more threads - the started ones still finish their share, then the error goes up
        std::lock_guard<std::mutex> lock(guard);
        if (!first) first = std::current_exception();
    }
    const uint32_t started = static_cast<uint32_t>(pool.size()) + 1u;
    body(0u);
    for (std::thread& th : pool) th.join();
    if (first) std::rethrow_exception(first);
    // (shares k >= started were never run: only reachable together with the exception rethrown above)
    (void)started;
}

// The same contract on threads that are kept: a BVH build over tens of millions of primitives runs a few hundred passes of a few
// milliseconds each, and starting 16-64 fresh threads per pass cost as much as the passes did.  One process-wide set of workers, started
// at first use and joined when the library is unloaded; a caller that finds it busy (another build at the same time) falls back to
// threads of its own.
class WorkerPool {
public:
    static WorkerPool& instance() {
        static WorkerPool pool;
        return pool;
    }
    template <typename Fn>
    void run(uint32_t n, Fn&& fn) {
        if (n <= 1u) {
            if (n == 1u) fn(0u);
            return;
        }
        std::unique_lock<std::mutex> owner(busy_, std::try_to_lock);
        if (!owner.owns_lock() || !ensure(n - 1u)) {
            runOnThreads(n, fn);
            return;
        }
        std::exception_ptr first;
        std::mutex guard;
        std::function<void(uint32_t)> body = [&](uint32_t k) {
            try {
                fn(k);
            } catch (...) {
                std::lock_guard<std::mutex> lock(guard);
                if (!first) first = std::current_exception();
            }
        };
        {
            std::lock_guard<std::mutex> lock(m_);
            job_ = &body;
            next_ = 1u;
            count_ = n;
            pending_ = n - 1u;
            ++generation_;
        }
        wake_.notify_all();
        body(0u);
        {
            std::unique_lock<std::mutex> lock(m_);
            done_.wait(lock, [&] { return pending_ == 0u; });
            job_ = nullptr;
        }
        if (first) std::rethrow_exception(first);
    }

private:
    WorkerPool() = default;
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    bool ensure(uint32_t wanted) {
        try {
            while (workers_.size() < wanted) workers_.emplace_back([this] { loop(); });
        } catch (...) {
            return false;   // no more threads: the caller runs the pass on threads of its own (or fails there)
        }
        return true;
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(m_);
        while (true) {
            wake_.wait(lock, [&] { return stop_ || (generation_ != seen && job_ != nullptr && next_ < count_); });
            if (stop_) return;
            while (job_ != nullptr && next_ < count_) {
                const uint32_t k = next_++;
                const std::function<void(uint32_t)>* job = job_;
                lock.unlock();
                (*job)(k);
                lock.lock();
                if (--pending_ == 0u) done_.notify_all();
            }
            seen = generation_;
        }
    }
    std::mutex busy_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(uint32_t)>* job_ = nullptr;
    uint32_t next_ = 0, count_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

}  // namespace ptr
