// pool.hpp -- the fork-join pool of the batched prover (host only; header-only so that tests/pool_check.cpp can
// run it under ThreadSanitizer).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace zk {

// Minimal fork-join pool: run(n, grain, fn) calls fn(i) for i < n on the workers and the calling thread.
// Between the rounds of a proof (a few hundred microseconds apart) the workers spin instead of sleeping, so a
// round's per-proof transcript steps start within a microsecond; after spin_us without work they block.
class Pool {
  public:
    // spin_us: how long an idle worker spins before it blocks (waking a blocked worker costs tens of microseconds)
    explicit Pool(unsigned workers, double spin_us = 600.0) : spin_us_(spin_us) {
        for (unsigned w = 0; w < workers; ++w) th_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        stop_.store(true, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> g(m_); }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    unsigned workers() const { return (unsigned)th_.size(); }
    template <class F>
    void run(size_t n, size_t grain, F&& fn) {
        if (th_.empty() || n <= grain) { for (size_t i = 0; i < n; ++i) fn(i); return; }
        std::function<void(size_t)> f = fn;
        fn_ = &f; n_ = n; grain_ = grain;
        next_.store(0, std::memory_order_relaxed);
        busy_.store((unsigned)th_.size(), std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);          // publishes the job
        { std::lock_guard<std::mutex> g(m_); }                  // a worker about to block has either seen it or is waiting
        cv_.notify_all();
        work();
        while (busy_.load(std::memory_order_acquire) != 0) cpu_relax();
        fn_ = nullptr;
    }

  private:
    const double spin_us_;
    static double clock_us() {
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    static void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
        __builtin_ia32_pause();
#endif
    }
    void work() {
        for (;;) {
            size_t b = next_.fetch_add(grain_, std::memory_order_relaxed);
            if (b >= n_) break;
            size_t e = b + grain_ < n_ ? b + grain_ : n_;
            for (size_t i = b; i < e; ++i) (*fn_)(i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            double t0 = clock_us();
            unsigned spins = 0;
            while (gen_.load(std::memory_order_acquire) == seen) {
                cpu_relax();
                if ((++spins & 255) == 0 && clock_us() - t0 > spin_us_) {
                    std::unique_lock<std::mutex> g(m_);
                    cv_.wait(g, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                }
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load(std::memory_order_acquire)) return;
            work();
            busy_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, grain_ = 1;
    std::atomic<size_t> next_{0};
    std::atomic<unsigned> busy_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace zk
