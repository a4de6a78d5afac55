// pool_check.cpp -- csrc/pool.hpp under ThreadSanitizer (tests/test_host_sanitizers.py).
// Exercises what the batched prover does with it: many back-to-back run() calls with varying sizes and grains
// (workers spinning), pauses longer than the spin window (workers blocked on the condition variable), zero
// workers, destruction while idle and right after a job.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../zkstark_amd/csrc/pool.hpp"

static int check(unsigned workers, int rounds) {
    zk::Pool pool(workers);
    std::vector<uint64_t> out(4096);
    for (int r = 0; r < rounds; ++r) {
        const size_t n = (size_t)1 + (size_t)((r * 131) % 4096);
        const size_t grain = (size_t)1 << (r % 7);
        std::atomic<size_t> calls{0};
        pool.run(n, grain, [&](size_t i) { out[i] = (uint64_t)i * 2654435761u + (uint64_t)r; calls.fetch_add(1, std::memory_order_relaxed); });
        if (calls.load() != n) { fprintf(stderr, "round %d: %zu calls for n = %zu\n", r, calls.load(), n); return 1; }
        for (size_t i = 0; i < n; ++i)
            if (out[i] != (uint64_t)i * 2654435761u + (uint64_t)r) { fprintf(stderr, "round %d: slot %zu not written\n", r, i); return 1; }
        if (r % 97 == 96) std::this_thread::sleep_for(std::chrono::milliseconds(3));   // past the spin window: workers block
    }
    return 0;
}

int main() {
    for (unsigned w : {0u, 1u, 3u, 7u})
        if (check(w, w ? 600 : 50)) return 1;
    { zk::Pool idle(4); std::this_thread::sleep_for(std::chrono::milliseconds(2)); }   // destroyed while the workers sleep
    { zk::Pool busy(4); std::atomic<int> s{0}; busy.run(1000, 1, [&](size_t) { s.fetch_add(1); }); }   // destroyed while they spin
    printf("pool ok\n");
    return 0;
}
