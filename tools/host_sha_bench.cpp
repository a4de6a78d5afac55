// Host-side Merkle-top hashing rate (csrc/host_sha.cpp):  g++ -O3 -std=c++17 -Izkstark_amd/csrc tools/host_sha_bench.cpp zkstark_amd/csrc/host_sha.cpp
#include <chrono>
#include <cstdio>
#include <vector>

#include "host_sha.hpp"
using namespace zk;

int main() {
    printf("sha extensions: %d, sixteen at a time (AVX-512F): %d\n", host_sha_available(), host_sha_wide_available());
    const bool have_wide = host_sha_wide_available();
    for (int wide = have_wide ? 1 : 0; wide >= 0; --wide) {
    host_sha_use_wide(wide != 0);
    printf("-- levels of >= 16 nodes %s\n", wide ? "sixteen at a time on 512-bit registers" : "on the SHA unit");
    for (int depth = 5; depth <= 10; ++depth) {
        std::vector<uint32_t> nodes(8 * ((2u << depth) - 1));
        for (size_t i = 0; i < nodes.size(); ++i) nodes[i] = (uint32_t)(i * 2654435761u);
        const int reps = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < reps; ++k) { nodes[8 * ((1u << depth) - 1)] = k; host_sha_reduce(nodes.data(), depth); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // one node at a time, for comparison with the two-at-a-time loop inside host_sha_reduce
        t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < reps; ++k) {
            nodes[8 * ((1u << depth) - 1)] = k;
            for (uint32_t d = depth; d-- > 0;) {
                size_t base = ((size_t)1 << d) - 1, child = ((size_t)2 << d) - 1;
                for (size_t i = 0; i < ((size_t)1 << d); ++i)
                    host_sha_inner(&nodes[8 * (child + 2 * i)], &nodes[8 * (child + 2 * i + 1)], &nodes[8 * (base + i)]);
            }
        }
        double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("top of 2^%d nodes: %.2f us (%.1f ns/node); one at a time %.2f us (%.1f ns/node)\n", depth, us,
               us * 1000 / ((1 << depth) - 1), us1, us1 * 1000 / ((1 << depth) - 1));
    }
    std::vector<uint32_t> vals(512), out(8 * 512);
    for (size_t i = 0; i < vals.size(); ++i) vals[i] = (uint32_t)(i * 2654435761u);
    auto t1 = std::chrono::steady_clock::now();
    for (int k = 0; k < 2000; ++k) { vals[0] = k; host_sha_leaves(vals.data(), vals.size(), out.data()); }
    printf("512 leaves: %.2f us (%.1f ns per leaf)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / 2000,
           std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t1).count() / 2000 / 512);
    }
    uint32_t o[8];
    auto t0 = std::chrono::steady_clock::now();
    uint32_t acc = 0;
    for (uint32_t v = 0; v < 100000; ++v) { host_sha_leaf(v, o); acc ^= o[0]; }
    printf("leaf: %.1f ns (%08x)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 1e5, acc);
}
