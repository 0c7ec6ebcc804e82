// Host worker threads that cannot take the process down: every worker body runs inside a try block, every started thread is
// joined whatever happens (also when starting a later thread fails), and the first exception is rethrown on the calling thread
// afterwards - so a std::bad_alloc inside a builder pass reaches the C boundary's catch-all as an error message instead of
// std::terminate.
#pragma once

#include <cstdint>
#include <exception>
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

}  // namespace ptr
