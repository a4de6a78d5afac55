// board_check.cpp -- csrc/board.hpp (the shared-memory exchange of subtree roots between the ranks of the
// sharded prover) under ThreadSanitizer / ASan (tests/test_host_sanitizers.py).  G threads stand in for G
// processes: each maps the same POSIX shared-memory object and runs thousands of exchanges with skewed timing;
// every rank must read, for every exchange, exactly what every other rank posted for THAT exchange.
#include <cstdio>
#include <thread>
#include <vector>

#include "../zkstark_amd/csrc/board.hpp"

using zk::impl::RootBoard;

static int run(int G, int rounds) {
    char name[64];
    snprintf(name, sizeof name, "/zkstark_amd_boardcheck_%d_%d", (int)getpid(), G);
    std::vector<RootBoard> boards(G);
    if (!boards[0].open_or_create(name, 0, G, true)) { fprintf(stderr, "create failed\n"); return 1; }
    for (int r = 1; r < G; ++r)
        if (!boards[r].open_or_create(name, r, G, false)) { fprintf(stderr, "open failed\n"); return 1; }
    shm_unlink(name);
    std::vector<int> bad(G, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r)
        th.emplace_back([&, r] {
            std::vector<uint32_t> all((size_t)G * 8);
            for (int s = 1; s <= rounds; ++s) {
                uint32_t mine[8];
                for (int i = 0; i < 8; ++i) mine[i] = (uint32_t)s * 2654435761u + (uint32_t)(r * 8 + i);
                if ((s + r) % 37 == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));   // a slow rank
                if (boards[r].exchange((uint64_t)s, mine, all.data(), 30.0) != RootBoard::kOk) { bad[r] = 1; return; }
                for (int q = 0; q < G; ++q)
                    for (int i = 0; i < 8; ++i)
                        if (all[(size_t)q * 8 + i] != (uint32_t)s * 2654435761u + (uint32_t)(q * 8 + i)) { bad[r] = 2; return; }
            }
        });
    for (auto& t : th) t.join();
    for (auto& b : boards) b.close();
    for (int r = 0; r < G; ++r)
        if (bad[r]) { fprintf(stderr, "G = %d: rank %d failed (%d)\n", G, r, bad[r]); return 1; }
    // a second user cannot open an object of another size
    RootBoard a, b;
    if (!a.open_or_create(name, 0, 2, true)) return 1;
    if (b.open_or_create(name, 1, 4, false)) { fprintf(stderr, "size mismatch accepted\n"); return 1; }
    shm_unlink(name);
    a.close();
    return 0;
}

// The blob area (the decommitment's contributions, shard.hip): every rank posts a blob of its own length per exchange and
// reads every peer's blob of THAT exchange in place; a ring of two suffices because a rank posts exchange k + 1 only after
// it has read all blobs of exchange k.
static int run_blobs(int G, int rounds, size_t cap) {
    char name[64];
    snprintf(name, sizeof name, "/zkstark_amd_boardcheck_%d_b%d", (int)getpid(), G);
    std::vector<RootBoard> boards(G);
    if (!boards[0].open_or_create(name, 0, G, true, cap)) { fprintf(stderr, "create (blobs) failed\n"); return 1; }
    for (int r = 1; r < G; ++r)
        if (!boards[r].open_or_create(name, r, G, false, cap)) { fprintf(stderr, "open (blobs) failed\n"); return 1; }
    {   // an opener that expects another blob capacity must be refused
        RootBoard other;
        if (other.open_or_create(name, 0, G, false, cap + 64)) { fprintf(stderr, "blob capacity mismatch accepted\n"); return 1; }
    }
    shm_unlink(name);
    std::vector<int> bad(G, 0);
    std::vector<std::thread> th;
    auto word = [](int s, int r, size_t i) { return (uint32_t)s * 2246822519u + (uint32_t)r * 40503u + (uint32_t)i; };
    for (int r = 0; r < G; ++r)
        th.emplace_back([&, r] {
            std::vector<uint32_t> mine(cap);
            uint32_t roots[8], all[8 * 16];
            for (int s = 1; s <= rounds; ++s) {
                const size_t len = (size_t)((s * 7 + r * 13) % (int)cap);           // lengths differ per rank and exchange, 0 included
                for (size_t i = 0; i < len; ++i) mine[i] = word(s, r, i);
                if ((s + 3 * r) % 41 == 0) std::this_thread::sleep_for(std::chrono::microseconds(40));
                boards[r].blob_post((uint64_t)s, mine.data(), len);
                for (int q = 0; q < G; ++q) {
                    const uint32_t* got = nullptr;
                    if (boards[r].blob_wait((uint64_t)s, q, &got, 30.0) != RootBoard::kOk) { bad[r] = 1; return; }
                    const size_t lq = (size_t)((s * 7 + q * 13) % (int)cap);
                    for (size_t i = 0; i < lq; ++i)
                        if (got[i] != word(s, q, i)) { bad[r] = 2; return; }
                }
                if (s % 5 == 0) {                                                   // root exchanges interleave with blob exchanges in a proof
                    for (int i = 0; i < 8; ++i) roots[i] = word(s, r, 1000 + i);
                    if (boards[r].exchange((uint64_t)(s / 5), roots, all, 30.0) != RootBoard::kOk) { bad[r] = 3; return; }
                }
            }
        });
    for (auto& t : th) t.join();
    for (auto& b : boards) b.close();
    for (int r = 0; r < G; ++r)
        if (bad[r]) { fprintf(stderr, "blobs, G = %d: rank %d failed (%d)\n", G, r, bad[r]); return 1; }
    return 0;
}

int main() {
    for (int G : {1, 2, 4, 8})
        if (run(G, G == 8 ? 1500 : 4000)) return 1;
    for (int G : {1, 2, 8})
        if (run_blobs(G, G == 8 ? 600 : 2000, 96)) return 1;
    // timeout: a rank that never posts
    RootBoard lone;
    char name[64];
    snprintf(name, sizeof name, "/zkstark_amd_boardcheck_%d_t", (int)getpid());
    if (!lone.open_or_create(name, 0, 2, true)) return 1;
    shm_unlink(name);
    uint32_t mine[8] = {0}, all[16];
    if (lone.exchange(1, mine, all, 0.2) != RootBoard::kTimeout || lone.bad_peer != 1) { fprintf(stderr, "exchange with a dead rank: wrong status\n"); return 1; }
    lone.close();
    // abort: a rank that leaves with an error releases the waiting rank at once, with its rank and code
    {
        RootBoard a, b;
        snprintf(name, sizeof name, "/zkstark_amd_boardcheck_%d_a", (int)getpid());
        if (!a.open_or_create(name, 0, 2, true) || !b.open_or_create(name, 1, 2, false)) return 1;
        shm_unlink(name);
        std::thread quitter([&] { std::this_thread::sleep_for(std::chrono::milliseconds(20)); b.post_abort(7); });
        auto t0 = std::chrono::steady_clock::now();
        RootBoard::Status st = a.exchange(1, mine, all, 30.0);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        quitter.join();
        if (st != RootBoard::kPeerAborted || a.bad_peer != 1 || a.bad_code != 7 || dt > 5.0) { fprintf(stderr, "abort not seen (%d, peer %d, code %u, %.2f s)\n", (int)st, a.bad_peer, a.bad_code, dt); return 1; }
        // sequence numbers beyond 32 bits work (no wrap into the zero-filled state)
        uint32_t m2[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        RootBoard c;
        snprintf(name, sizeof name, "/zkstark_amd_boardcheck_%d_w", (int)getpid());
        if (!c.open_or_create(name, 0, 1, true)) return 1;
        shm_unlink(name);
        for (uint64_t sq : {0xFFFFFFFFull, 0x100000000ull, 0x100000001ull})
            if (c.exchange(sq, m2, all, 1.0) != RootBoard::kOk || all[7] != 8) { fprintf(stderr, "wide sequence failed\n"); return 1; }
        a.close(); b.close(); c.close();
    }
    printf("board ok\n");
    return 0;
}
