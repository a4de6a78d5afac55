// peer_check.cpp -- the protocol of csrc/peer.hpp (the peer-copy transport of the sharded prover: staging buffers announced on a
// shared-memory page, pulls, the two meetings per collective, abort words, bounded waits) under ThreadSanitizer / ASan
// (tests/test_host_sanitizers.py).  Host memory stands in for the GPU (-DZK_PEER_NO_HIP: the device runtime is a policy of the
// transport), G threads stand in for G ranks: thousands of all-to-alls and all-gathers of changing sizes with skewed timing, every
// word checked; then a rank that leaves (its peers must return at once, naming it) and a rank that never comes (a bounded wait).
#define ZK_PEER_NO_HIP 1
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../zkstark_amd/csrc/peer.hpp"

struct HostOps {
    using stream_t = void*;
    static bool alloc(void** p, size_t bytes) { *p = malloc(bytes); return *p != nullptr; }
    static void release(void* p) { free(p); }
    static bool export_handle(void*, uint8_t out[64], std::string&) { memset(out, 0x5a, 64); return true; }
    static void* import_handle(const uint8_t*, std::string& err) { err = "another process in a one-process test"; return nullptr; }
    static void unimport(void*) {}
    static bool copy(void* dst, const void* src, size_t bytes, stream_t) { memcpy(dst, src, bytes); return true; }
    static bool sync(stream_t) { return true; }
    static const char* last_error() { return "none"; }
};
using Transport = zk::impl::PeerTransportT<HostOps>;

static uint32_t pat(int from, int to, size_t j, uint32_t round) { return (uint32_t)(from * 1000003u + to * 7919u + (uint32_t)j * 2654435761u + round * 97u); }

static int run(int G, int rounds) {
    char name[64];
    snprintf(name, sizeof name, "/zkstark_amd_peercheck_%d_%d", (int)getpid(), G);
    const size_t max_words = 1 << 10;
    std::vector<Transport> tp(G);
    std::vector<int> bad(G, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r)
        th.emplace_back([&, r] {
            if (!tp[r].open(name, r, G, 30.0, (size_t)G * max_words * 4)) { fprintf(stderr, "rank %d: open: %s\n", r, tp[r].error.c_str()); bad[r] = 1; return; }
            std::vector<std::vector<uint32_t>> send(G, std::vector<uint32_t>(max_words)), recv(G, std::vector<uint32_t>(max_words));
            std::vector<uint32_t> gsend(max_words), grecv((size_t)G * max_words);
            for (int s = 1; s <= rounds; ++s) {
                const size_t words = 1 + (size_t)(s * 37 % (int)max_words);
                if ((s + r) % 41 == 0) std::this_thread::sleep_for(std::chrono::microseconds(40));   // a slow rank
                if (s % 3) {
                    std::vector<const uint32_t*> sp(G);
                    std::vector<uint32_t*> rp(G);
                    for (int p = 0; p < G; ++p) {
                        for (size_t j = 0; j < words; ++j) send[p][j] = pat(r, p, j, (uint32_t)s);
                        sp[p] = send[p].data(); rp[p] = recv[p].data();
                    }
                    if (tp[r].exchange(sp.data(), nullptr, rp.data(), nullptr, words, nullptr)) { fprintf(stderr, "rank %d: %s\n", r, tp[r].error.c_str()); bad[r] = 2; return; }
                    for (int q = 0; q < G; ++q)
                        for (size_t j = 0; j < words; ++j)
                            if (recv[q][j] != pat(q, r, j, (uint32_t)s)) { bad[r] = 3; return; }
                } else {
                    for (size_t j = 0; j < words; ++j) gsend[j] = pat(r, 99, j, (uint32_t)s);
                    if (tp[r].exchange(nullptr, gsend.data(), nullptr, grecv.data(), words, nullptr)) { fprintf(stderr, "rank %d: %s\n", r, tp[r].error.c_str()); bad[r] = 4; return; }
                    for (int q = 0; q < G; ++q)
                        for (size_t j = 0; j < words; ++j)
                            if (grecv[(size_t)q * words + j] != pat(q, 99, j, (uint32_t)s)) { bad[r] = 5; return; }
                }
            }
            // a collective that does not fit the staging buffer is refused, not truncated (every rank: nobody is left waiting)
            std::vector<uint32_t> big((size_t)G * max_words + 64);
            if (!tp[r].exchange(nullptr, big.data(), nullptr, big.data(), (size_t)G * max_words + 1, nullptr)) { bad[r] = 6; return; }
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < G; ++r)
        if (bad[r]) { fprintf(stderr, "G = %d: rank %d failed (%d)\n", G, r, bad[r]); return 1; }
    // rank G-1 leaves with an error while the others are inside a collective: they return at once and name it
    th.clear();
    std::vector<int> saw(G, -1);
    std::vector<double> took(G, 0.0);
    for (int r = 0; r < G; ++r)
        th.emplace_back([&, r] {
            std::vector<uint32_t> a(4, (uint32_t)r), b((size_t)4 * G);
            if (r == G - 1) { std::this_thread::sleep_for(std::chrono::milliseconds(30)); tp[r].post_abort(7); return; }
            const double t0 = Transport::now_s();
            const int rc = tp[r].exchange(nullptr, a.data(), nullptr, b.data(), 4, nullptr);
            took[r] = Transport::now_s() - t0;
            saw[r] = rc ? tp[r].bad_peer : -2;
        });
    for (auto& t : th) t.join();
    for (int r = 0; r + 1 < G; ++r)
        if (saw[r] < 0 || took[r] > 5.0) { fprintf(stderr, "G = %d: rank %d did not see the abort (bad_peer %d, %.2f s)\n", G, r, saw[r], took[r]); return 1; }
    for (auto& t : tp) t.close();
    // a rank whose peers never come: open() gives up after its bound and leaves nothing in /dev/shm
    Transport lone;
    const double t0 = Transport::now_s();
    if (lone.open(name, 0, 2, 0.3, 4096)) { fprintf(stderr, "open succeeded without a peer\n"); return 1; }
    if (Transport::now_s() - t0 > 5.0) { fprintf(stderr, "open did not give up in time\n"); return 1; }
    lone.close();
    if (shm_open(name, O_RDWR, 0600) >= 0) { fprintf(stderr, "the shared page was left behind\n"); shm_unlink(name); return 1; }
    return 0;
}

int main() {
    for (int G : {1, 2, 4, 8})
        if (run(G, G == 8 ? 600 : 1500)) return 1;
    printf("peer transport protocol ok\n");
    return 0;
}
