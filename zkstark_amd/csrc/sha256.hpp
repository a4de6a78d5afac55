// sha256.hpp -- SHA-256 (FIPS 180-4) for the Merkle commitment and the transcript.
//
// Replaces the `sha2` crate as used by merkle.rs:31-33, :42-45, :87-104 and
// channel.rs:21-24.  Two block shapes exist on the device path
// (SURVEY.md A.6):
//   leaf : one block, W0 = value (u32.to_be_bytes() read as a big-endian word is the
//          word itself, merkle.rs:32), W1 = 0x80000000, W15 = 32
//   inner: block 1 = left || right (16 words), block 2 = constant padding
//          (0x80000000, 0, ..., 512) whose whole message schedule folds at compile time.
// Digests live in HBM as the eight big-endian STATE WORDS (native u32), so no byte
// swaps occur between tree levels; bytes are produced only when a digest is
// exported to the host (root, auth paths).
#pragma once
#include <stdint.h>
#include <string.h>

#include "host_sha.hpp"

#if defined(__HIPCC__)
#define ZK_SHA_HD __host__ __device__ __forceinline__
#else
#define ZK_SHA_HD inline
#endif

namespace zk {

struct Digest {
    uint32_t w[8];
};

constexpr uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

ZK_SHA_HD uint32_t sha_rotr(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x, x, n);   // v_alignbit_b32
#else
    return (x >> n) | (x << (32 - n));
#endif
}

constexpr uint32_t SHA_IV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                                0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

// Three-input boolean functions.  gfx950 has v_bitop3_b32 (any 3-input truth table in one
// VALU op; the table is the function evaluated on a=0xF0, b=0xCC, c=0xAA); hipcc does not form
// it for the rotate-xor sums on its own, which costs ~20 % more instructions per compression.
// Compile-time-constant operands stay on plain operators so that they still fold.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_BITOP3(a, b, c, tt, expr) \
    ((__builtin_constant_p(a) && __builtin_constant_p(b) && __builtin_constant_p(c)) ? (expr) : __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt)))
#else
#define ZK_BITOP3(a, b, c, tt, expr) (expr)
#endif
ZK_SHA_HD uint32_t sha_xor3(uint32_t a, uint32_t b, uint32_t c) { return ZK_BITOP3(a, b, c, 0x96, a ^ b ^ c); }
ZK_SHA_HD uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return ZK_BITOP3(e, f, g, 0xCA, (e & f) | (~e & g)); }
ZK_SHA_HD uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return ZK_BITOP3(a, b, c, 0xE8, (a & b) | (a & c) | (b & c)); }

// One compression, fully unrolled with a rolling 16-word schedule.  Every call
// site is inlined, so constant message words / constant chaining values fold away.
// Per round: 6 v_alignbit + 2 xor3 + ch + maj + 4 adds (add3 where possible) = 14 VALU ops;
// per schedule step: 4 v_alignbit/shift + 2 xor3 + 2 adds = 10.
ZK_SHA_HD void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = sha_xor3(sha_rotr(w15, 7), sha_rotr(w15, 18), w15 >> 3);
            uint32_t s1 = sha_xor3(sha_rotr(w2, 17), sha_rotr(w2, 19), w2 >> 10);
            wi = (w[i & 15] + s0 + w[(i - 7) & 15]) + s1;
            w[i & 15] = wi;
        }
        uint32_t S1 = sha_xor3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25));
        uint32_t t1 = (h + S1 + sha_ch(e, f, g)) + (SHA_K[i] + wi);
        uint32_t S0 = sha_xor3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22));
        uint32_t mj = sha_maj(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// The same compression in 16-round slices, for the split main/helper schedule of the latency-bound
// tree levels (kernels.hip, merkle_wg_kernel): rounds I0 .. I0+15 on already-scheduled words, and
// the in-place step W[t+16] = s1(W[t+14]) + W[t+9] + s0(W[t+1]) + W[t] for 16 consecutive t.
template <int I0>
ZK_SHA_HD void sha256_rounds16(uint32_t (&st)[8], const uint32_t (&w)[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint32_t S1 = sha_xor3(sha_rotr(e, 6), sha_rotr(e, 11), sha_rotr(e, 25));
        uint32_t t1 = (h + S1 + sha_ch(e, f, g)) + (SHA_K[I0 + i] + w[i]);
        uint32_t S0 = sha_xor3(sha_rotr(a, 2), sha_rotr(a, 13), sha_rotr(a, 22));
        uint32_t mj = sha_maj(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
    }
    st[0] = a; st[1] = b; st[2] = c; st[3] = d; st[4] = e; st[5] = f; st[6] = g; st[7] = h;
}
ZK_SHA_HD void sha256_schedule16(uint32_t (&w)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        uint32_t w1 = w[(t + 1) & 15], w14 = w[(t + 14) & 15];
        uint32_t s0 = sha_xor3(sha_rotr(w1, 7), sha_rotr(w1, 18), w1 >> 3);
        uint32_t s1 = sha_xor3(sha_rotr(w14, 17), sha_rotr(w14, 19), w14 >> 10);
        w[t] = (w[t] + s0 + w[(t + 9) & 15]) + s1;
    }
}

// merkle.rs:30-34: SHA256(v.to_be_bytes())
ZK_SHA_HD Digest sha256_leaf(uint32_t v) {
    uint32_t w[16] = {v, 0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 32u};
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.w[i] = SHA_IV[i];
    sha256_compress(d.w, w);
    return d;
}

// merkle.rs:42-45: SHA256(left || right)
ZK_SHA_HD Digest sha256_inner(const Digest& l, const Digest& r) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[i] = l.w[i]; w[8 + i] = r.w[i]; }
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.w[i] = SHA_IV[i];
    sha256_compress(d.w, w);
    uint32_t pad[16] = {0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 512u};
    sha256_compress(d.w, pad);
    return d;
}

// ---- host-only generic hashing (transcript, verifier) ---------------------
struct Sha256 {
    uint32_t st[8];
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    Sha256() { for (int i = 0; i < 8; ++i) st[i] = SHA_IV[i]; }
    void update(const void* data, size_t n) {
        const uint8_t* m = (const uint8_t*)data;
        total += n;
        if (fill) {
            const size_t take = 64 - fill < n ? 64 - fill : n;
            memcpy(buf + fill, m, take);
            fill += take; m += take; n -= take;
            if (fill < 64) return;
            host_sha_blocks(st, buf, 1);               // SHA extensions when the CPU has them (host_sha.cpp)
            fill = 0;
        }
        if (n >= 64) {                                 // whole blocks straight from the caller's bytes
            host_sha_blocks(st, m, n / 64);
            m += n & ~(size_t)63; n &= 63;
        }
        memcpy(buf, m, n);
        fill = n;
    }
    void finalize(uint8_t out[32]) {
        uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (fill < 56 ? 56 - fill : 120 - fill);
        update(pad, padlen);
        uint8_t len[8];
        for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; ++i) {
            out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
            out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
        }
    }
};
// digest state words -> the 32 bytes the reference's Hash = [u8; 32] holds (merkle.rs:9)
inline void digest_words_to_bytes(const uint32_t w[8], uint8_t out[32]) {
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(w[i] >> 24); out[4 * i + 1] = (uint8_t)(w[i] >> 16);
        out[4 * i + 2] = (uint8_t)(w[i] >> 8); out[4 * i + 3] = (uint8_t)w[i];
    }
}

}  // namespace zk
