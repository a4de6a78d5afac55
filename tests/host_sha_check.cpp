// Prints digests from csrc/host_sha.cpp for tests/test_host_sha.py (compared there with hashlib).
//   host_sha_check <ext: 0 portable | 1 SHA extensions, one or two nodes at a time | 2 + sixteen at a time (AVX-512F)> <depth> <seed>
// prints "<sha extensions in use> <wide path in use>", the heap, the chain, the byte-stream chain
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../zkstark_amd/csrc/host_sha.hpp"

static void hex(const uint32_t w[8]) {
    for (int i = 0; i < 8; ++i) printf("%08x", w[i]);
    printf("\n");
}
int main(int argc, char** argv) {
    if (argc != 4) return 2;
    zk::host_sha_use_extensions(atoi(argv[1]) != 0);
    zk::host_sha_use_wide(atoi(argv[1]) == 2);
    printf("%d %d\n", zk::host_sha_available() ? 1 : 0, zk::host_sha_wide_available() ? 1 : 0);
    const uint32_t depth = (uint32_t)atoi(argv[2]);
    uint32_t x = (uint32_t)strtoul(argv[3], nullptr, 10);
    const size_t m = (size_t)1 << depth;
    std::vector<uint32_t> nodes(8 * (2 * m - 1));
    std::vector<uint32_t> vals(m);
    for (size_t i = 0; i < m; ++i) {                     // leaves from an LCG; the first few values are edge cases
        x = x * 1664525u + 1013904223u;
        vals[i] = i == 0 ? 0u : i == 1 ? 0xffffffffu : i == 2 ? 3221225472u : x;
    }
    zk::host_sha_leaves(vals.data(), m, &nodes[8 * (m - 1)]);
    if (m > 2) zk::host_sha_leaf(vals[2], &nodes[8 * (m - 1 + 2)]);      // the one-leaf entry point too
    zk::host_sha_reduce(nodes.data(), depth);
    for (size_t i = 0; i < 2 * m - 1; ++i) hex(&nodes[8 * i]);
    // transcript-style compression chain: state <- compress(state, block) over three blocks
    uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t blk[16];
    for (int b = 0; b < 3; ++b) {
        for (int i = 0; i < 16; ++i) { x = x * 1664525u + 1013904223u; blk[i] = x; }
        if (b == 2) { blk[8] = 0x80000000u; for (int i = 9; i < 15; ++i) blk[i] = 0; blk[15] = (2 * 64 + 32) * 8; }
        zk::host_sha_compress(st, blk);
    }
    hex(st);
    // the same through the byte-stream entry point (host_sha_blocks: several blocks per call, big-endian decoding inside):
    // five blocks of LCG bytes, at an odd address, fed as 2 + 3
    std::vector<uint8_t> bytes(5 * 64 + 1);
    for (size_t i = 1; i < bytes.size(); ++i) { x = x * 1664525u + 1013904223u; bytes[i] = (uint8_t)(x >> 24); }
    uint32_t s2[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    zk::host_sha_blocks(s2, bytes.data() + 1, 2);
    zk::host_sha_blocks(s2, bytes.data() + 1 + 128, 3);
    hex(s2);
    return 0;
}
