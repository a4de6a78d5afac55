// host_sha.cpp -- see host_sha.hpp.  Host-only translation unit (nothing here is device code).
#include "host_sha.hpp"

#include <atomic>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

namespace zk {
namespace {

#define ZK_SHA_TARGET __attribute__((target("sha,sse4.1,ssse3")))

alignas(16) const uint32_t K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
const uint32_t IV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

// K[t] + W[t] of the second block of a 64-byte message (0x80, zeros, length 512): its schedule is constant
alignas(16) uint32_t PADKW[64];
// What the CPU can do (set once at load) and what is in use.  The setters (zk_host_set_hash_mode) may run while pool
// threads hash, so the flags in use are atomics (relaxed: either setting gives the same digests).
bool g_cpu_has_sha = false, g_cpu_has_x16 = false;    // x16 = AVX-512F: sixteen nodes per instruction stream (below)
std::atomic<bool> g_have_sha{false}, g_have_x16{false};
std::atomic<bool> g_want_x16{true};                   // host_sha_use_wide(false) survives a later host_sha_use_extensions(true)

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

struct Init {
    Init() {
        __builtin_cpu_init();
        g_cpu_has_sha = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
        g_cpu_has_x16 = g_cpu_has_sha && __builtin_cpu_supports("avx512f");
        g_have_sha = g_cpu_has_sha;
        g_have_x16 = g_cpu_has_x16;
        uint32_t w[64] = {0x80000000u};
        w[15] = 512u;
        for (int t = 16; t < 64; ++t) {
            uint32_t s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3);
            uint32_t s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10);
            w[t] = w[t - 16] + s0 + w[t - 7] + s1;
        }
        for (int t = 0; t < 64; ++t) PADKW[t] = K[t] + w[t];
    }
} g_init;

// state words (a..h) <-> the (ABEF, CDGH) register pair the SHA instructions work on
ZK_SHA_TARGET inline void load_state(const uint32_t st[8], __m128i& s0, __m128i& s1) {
    __m128i t = _mm_loadu_si128((const __m128i*)st);          // a b c d
    s1 = _mm_loadu_si128((const __m128i*)(st + 4));           // e f g h
    t = _mm_shuffle_epi32(t, 0xB1);
    s1 = _mm_shuffle_epi32(s1, 0x1B);
    s0 = _mm_alignr_epi8(t, s1, 8);
    s1 = _mm_blend_epi16(s1, t, 0xF0);
}
ZK_SHA_TARGET inline void store_state(uint32_t st[8], __m128i s0, __m128i s1) {
    __m128i t = _mm_shuffle_epi32(s0, 0x1B);
    s1 = _mm_shuffle_epi32(s1, 0xB1);
    s0 = _mm_blend_epi16(t, s1, 0xF0);
    s1 = _mm_alignr_epi8(s1, t, 8);
    _mm_storeu_si128((__m128i*)st, s0);
    _mm_storeu_si128((__m128i*)(st + 4), s1);
}

// one block whose sixteen message words are m[0..3] (numeric words, word 0 in lane 0)
ZK_SHA_TARGET inline void rounds_msg(__m128i& s0, __m128i& s1, __m128i m[4]) {
    const __m128i a0 = s0, a1 = s1;
#pragma GCC unroll 16
    for (int g = 0; g < 16; ++g) {
        __m128i cur = m[g & 3];
        __m128i msg = _mm_add_epi32(cur, _mm_load_si128((const __m128i*)(K + 4 * g)));
        s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
        if (g >= 3 && g <= 14) {
            __m128i t = _mm_alignr_epi8(cur, m[(g + 3) & 3], 4);
            m[(g + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(m[(g + 1) & 3], t), cur);
        }
        msg = _mm_shuffle_epi32(msg, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
        if (g >= 1 && g <= 12) m[(g + 3) & 3] = _mm_sha256msg1_epu32(m[(g + 3) & 3], cur);
    }
    s0 = _mm_add_epi32(s0, a0);
    s1 = _mm_add_epi32(s1, a1);
}
// the constant padding block of a 64-byte message
ZK_SHA_TARGET inline void rounds_pad64(__m128i& s0, __m128i& s1) {
    const __m128i a0 = s0, a1 = s1;
#pragma GCC unroll 16
    for (int g = 0; g < 16; ++g) {
        __m128i msg = _mm_load_si128((const __m128i*)(PADKW + 4 * g));
        s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
        msg = _mm_shuffle_epi32(msg, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
    }
    s0 = _mm_add_epi32(s0, a0);
    s1 = _mm_add_epi32(s1, a1);
}

ZK_SHA_TARGET void inner_ni(const uint32_t* l, const uint32_t* r, uint32_t* out) {
    __m128i s0, s1, m[4];
    load_state(IV, s0, s1);
    m[0] = _mm_loadu_si128((const __m128i*)l);
    m[1] = _mm_loadu_si128((const __m128i*)(l + 4));
    m[2] = _mm_loadu_si128((const __m128i*)r);
    m[3] = _mm_loadu_si128((const __m128i*)(r + 4));
    rounds_msg(s0, s1, m);
    rounds_pad64(s0, s1);
    store_state(out, s0, s1);
}
// two independent nodes at once: the rnds2 chain of one hash is latency-bound, two chains fill the unit
ZK_SHA_TARGET void inner_ni_x2(const uint32_t* in, uint32_t* out) {      // in: l0 r0 l1 r1 (8 words each); out: 2 digests
    __m128i s0, s1, u0, u1, m[4], n[4];
    load_state(IV, s0, s1);
    u0 = s0; u1 = s1;
    const __m128i a0 = s0, a1 = s1;
    for (int i = 0; i < 4; ++i) {
        m[i] = _mm_loadu_si128((const __m128i*)(in + 4 * i));
        n[i] = _mm_loadu_si128((const __m128i*)(in + 16 + 4 * i));
    }
#pragma GCC unroll 16
    for (int g = 0; g < 16; ++g) {
        const __m128i k = _mm_load_si128((const __m128i*)(K + 4 * g));
        __m128i cm = m[g & 3], cn = n[g & 3];
        __m128i xm = _mm_add_epi32(cm, k), xn = _mm_add_epi32(cn, k);
        s1 = _mm_sha256rnds2_epu32(s1, s0, xm);
        u1 = _mm_sha256rnds2_epu32(u1, u0, xn);
        if (g >= 3 && g <= 14) {
            __m128i tm = _mm_alignr_epi8(cm, m[(g + 3) & 3], 4), tn = _mm_alignr_epi8(cn, n[(g + 3) & 3], 4);
            m[(g + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(m[(g + 1) & 3], tm), cm);
            n[(g + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(n[(g + 1) & 3], tn), cn);
        }
        xm = _mm_shuffle_epi32(xm, 0x0E);
        xn = _mm_shuffle_epi32(xn, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, xm);
        u0 = _mm_sha256rnds2_epu32(u0, u1, xn);
        if (g >= 1 && g <= 12) {
            m[(g + 3) & 3] = _mm_sha256msg1_epu32(m[(g + 3) & 3], cm);
            n[(g + 3) & 3] = _mm_sha256msg1_epu32(n[(g + 3) & 3], cn);
        }
    }
    s0 = _mm_add_epi32(s0, a0); s1 = _mm_add_epi32(s1, a1);
    u0 = _mm_add_epi32(u0, a0); u1 = _mm_add_epi32(u1, a1);
    const __m128i b0 = s0, b1 = s1, c0 = u0, c1 = u1;
#pragma GCC unroll 16
    for (int g = 0; g < 16; ++g) {
        __m128i msg = _mm_load_si128((const __m128i*)(PADKW + 4 * g));
        s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
        u1 = _mm_sha256rnds2_epu32(u1, u0, msg);
        msg = _mm_shuffle_epi32(msg, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
        u0 = _mm_sha256rnds2_epu32(u0, u1, msg);
    }
    s0 = _mm_add_epi32(s0, b0); s1 = _mm_add_epi32(s1, b1);
    u0 = _mm_add_epi32(u0, c0); u1 = _mm_add_epi32(u1, c1);
    store_state(out, s0, s1);
    store_state(out + 8, u0, u1);
}
ZK_SHA_TARGET void compress_ni(uint32_t* st, const uint32_t* blk) {
    __m128i s0, s1, m[4];
    load_state(st, s0, s1);
    for (int i = 0; i < 4; ++i) m[i] = _mm_loadu_si128((const __m128i*)(blk + 4 * i));
    rounds_msg(s0, s1, m);
    store_state(st, s0, s1);
}
ZK_SHA_TARGET void blocks_ni(uint32_t* st, const uint8_t* data, size_t blocks) {
    const __m128i be = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);   // bytes of each 32-bit word reversed
    __m128i s0, s1, m[4];
    load_state(st, s0, s1);
    for (; blocks; --blocks, data += 64) {
        for (int i = 0; i < 4; ++i) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * i)), be);
        rounds_msg(s0, s1, m);
    }
    store_state(st, s0, s1);
}
ZK_SHA_TARGET void leaf_ni(uint32_t v, uint32_t* out) {
    __m128i s0, s1, m[4];
    load_state(IV, s0, s1);
    m[0] = _mm_set_epi32(0, 0, (int)0x80000000u, (int)v);
    m[1] = _mm_setzero_si128();
    m[2] = _mm_setzero_si128();
    m[3] = _mm_set_epi32(32, 0, 0, 0);
    rounds_msg(s0, s1, m);
    store_state(out, s0, s1);
}
// ---- sixteen nodes at a time (AVX-512F) ------------------------------------------------------------------------
// One SHA unit finishes a node in ~31 ns however many chains it is fed (two sha256rnds2 per four rounds, back to back).
// A level of >= 16 nodes is wide enough for the other way round: the textbook compression on 512-bit registers, one
// node per 32-bit lane -- vprord for the rotations, vpternlogd for the three-input functions (Ch, Maj, the xor of three
// rotations), sixteen message words transposed in and eight state words transposed out.  ~2 800 vector instructions
// per sixteen nodes instead of sixteen times ~140 SHA-unit steps: measured ~12 ns per node (tools/host_sha_bench.cpp).
#define ZK_X16_TARGET __attribute__((target("avx512f")))
#define ZK_ROR(x, n) _mm512_ror_epi32((x), (n))
#define ZK_XOR3(a, b, c) _mm512_ternarylogic_epi32((a), (b), (c), 0x96)

// r[i] lane j  <->  r[j] lane i
ZK_X16_TARGET inline void transpose16(__m512i r[16]) {
    __m512i t[16], u[16];
    for (int i = 0; i < 8; ++i) {
        t[2 * i] = _mm512_unpacklo_epi32(r[2 * i], r[2 * i + 1]);
        t[2 * i + 1] = _mm512_unpackhi_epi32(r[2 * i], r[2 * i + 1]);
    }
    for (int i = 0; i < 16; i += 4) {
        u[i] = _mm512_unpacklo_epi64(t[i], t[i + 2]);
        u[i + 1] = _mm512_unpackhi_epi64(t[i], t[i + 2]);
        u[i + 2] = _mm512_unpacklo_epi64(t[i + 1], t[i + 3]);
        u[i + 3] = _mm512_unpackhi_epi64(t[i + 1], t[i + 3]);
    }
    // u[4 g + k], 128-bit lane q: column 4 q + k of rows 4 g .. 4 g + 3
    for (int k = 0; k < 4; ++k) {
        t[k] = _mm512_shuffle_i32x4(u[k], u[4 + k], 0x88);            // lanes q = 0, 2 of row groups 0, 1
        t[4 + k] = _mm512_shuffle_i32x4(u[k], u[4 + k], 0xdd);        // lanes q = 1, 3
        t[8 + k] = _mm512_shuffle_i32x4(u[8 + k], u[12 + k], 0x88);   // row groups 2, 3
        t[12 + k] = _mm512_shuffle_i32x4(u[8 + k], u[12 + k], 0xdd);
    }
    for (int k = 0; k < 4; ++k) {
        r[k] = _mm512_shuffle_i32x4(t[k], t[8 + k], 0x88);            // q = 0: columns 0..3
        r[8 + k] = _mm512_shuffle_i32x4(t[k], t[8 + k], 0xdd);        // q = 2: columns 8..11
        r[4 + k] = _mm512_shuffle_i32x4(t[4 + k], t[12 + k], 0x88);   // q = 1: columns 4..7
        r[12 + k] = _mm512_shuffle_i32x4(t[4 + k], t[12 + k], 0xdd);  // q = 3: columns 12..15
    }
}

// 64 rounds on sixteen independent states; MSG: w[16] is the message (rolling schedule), otherwise the constant padding
// block of a 64-byte message (K + W from PADKW)
template <bool MSG>
ZK_X16_TARGET inline void rounds_x16(__m512i st[8], __m512i w[16]) {
    __m512i a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma GCC unroll 64
    for (int t = 0; t < 64; ++t) {
        __m512i wk;
        if (MSG) {
            if (t >= 16) {
                const __m512i w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
                const __m512i s0 = ZK_XOR3(ZK_ROR(w15, 7), ZK_ROR(w15, 18), _mm512_srli_epi32(w15, 3));
                const __m512i s1 = ZK_XOR3(ZK_ROR(w2, 17), ZK_ROR(w2, 19), _mm512_srli_epi32(w2, 10));
                w[t & 15] = _mm512_add_epi32(_mm512_add_epi32(w[t & 15], s0), _mm512_add_epi32(w[(t + 9) & 15], s1));
            }
            wk = _mm512_add_epi32(w[t & 15], _mm512_set1_epi32((int)K[t]));
        } else {
            wk = _mm512_set1_epi32((int)PADKW[t]);
        }
        const __m512i S1 = ZK_XOR3(ZK_ROR(e, 6), ZK_ROR(e, 11), ZK_ROR(e, 25));
        const __m512i t1 = _mm512_add_epi32(_mm512_add_epi32(h, S1), _mm512_add_epi32(_mm512_ternarylogic_epi32(e, f, g, 0xCA), wk));
        const __m512i S0 = ZK_XOR3(ZK_ROR(a, 2), ZK_ROR(a, 13), ZK_ROR(a, 22));
        const __m512i t2 = _mm512_add_epi32(S0, _mm512_ternarylogic_epi32(a, b, c, 0xE8));
        h = g; g = f; f = e; e = _mm512_add_epi32(d, t1); d = c; c = b; b = a; a = _mm512_add_epi32(t1, t2);
    }
    st[0] = _mm512_add_epi32(st[0], a); st[1] = _mm512_add_epi32(st[1], b); st[2] = _mm512_add_epi32(st[2], c); st[3] = _mm512_add_epi32(st[3], d);
    st[4] = _mm512_add_epi32(st[4], e); st[5] = _mm512_add_epi32(st[5], f); st[6] = _mm512_add_epi32(st[6], g); st[7] = _mm512_add_epi32(st[7], h);
}
ZK_X16_TARGET inline void store_x16(const __m512i st[8], uint32_t* out) {       // sixteen digests of eight words
    __m512i r[16];
    for (int i = 0; i < 8; ++i) { r[i] = st[i]; r[8 + i] = _mm512_setzero_si512(); }
    transpose16(r);
    for (int n = 0; n < 16; ++n) _mm256_storeu_si256((__m256i*)(out + 8 * n), _mm512_castsi512_si256(r[n]));
}
// in: 32 digests (the children of sixteen consecutive nodes, contiguous in the heap); out: the sixteen nodes
ZK_X16_TARGET void inner_x16(const uint32_t* in, uint32_t* out) {
    __m512i w[16], st[8];
    for (int n = 0; n < 16; ++n) w[n] = _mm512_loadu_si512((const void*)(in + 16 * n));   // row n = message of node n
    transpose16(w);                                                                        // row t = word t of every node
    for (int i = 0; i < 8; ++i) st[i] = _mm512_set1_epi32((int)IV[i]);
    rounds_x16<true>(st, w);
    rounds_x16<false>(st, w);
    store_x16(st, out);
}
// sixteen leaves: SHA256(be32(v)) for v = vals[0..15] (merkle.rs:30-34)
ZK_X16_TARGET void leaves_x16(const uint32_t* vals, uint32_t* out) {
    __m512i w[16], st[8];
    w[0] = _mm512_loadu_si512((const void*)vals);
    w[1] = _mm512_set1_epi32((int)0x80000000u);
    for (int i = 2; i < 15; ++i) w[i] = _mm512_setzero_si512();
    w[15] = _mm512_set1_epi32(32);
    for (int i = 0; i < 8; ++i) st[i] = _mm512_set1_epi32((int)IV[i]);
    rounds_x16<true>(st, w);
    store_x16(st, out);
}

// `cnt` consecutive nodes of one level from their 2 cnt contiguous children
void inner_run(const uint32_t* child, uint32_t* out, size_t cnt) {
    size_t i = 0;
    if (g_have_x16)
        for (; i + 16 <= cnt; i += 16) inner_x16(child + 16 * i, out + 8 * i);
    for (; i + 2 <= cnt; i += 2) inner_ni_x2(child + 16 * i, out + 8 * i);
    for (; i < cnt; ++i) inner_ni(child + 16 * i, child + 16 * i + 8, out + 8 * i);
}

ZK_SHA_TARGET void reduce_ni(uint32_t* nodes, uint32_t depth) {
    for (uint32_t d = depth; d-- > 0;) {
        const size_t base = ((size_t)1 << d) - 1, child = ((size_t)2 << d) - 1, cnt = (size_t)1 << d;
        inner_run(nodes + 8 * child, nodes + 8 * base, cnt);       // the children of consecutive nodes are contiguous
    }
}

// levels depth-1 .. top of the sub-tree below node `sub` of level `top` (sub < 2^top): the part of reduce_ni that one
// thread of a team takes; the nodes of a sub-tree are contiguous within every level
ZK_SHA_TARGET void reduce_sub_ni(uint32_t* nodes, uint32_t depth, uint32_t top, size_t sub) {
    for (uint32_t d = depth; d-- > top;) {
        const size_t cnt = (size_t)1 << (d - top);
        const size_t base = ((size_t)1 << d) - 1 + sub * cnt, child = ((size_t)2 << d) - 1 + 2 * sub * cnt;
        inner_run(nodes + 8 * child, nodes + 8 * base, cnt);
    }
}

// portable compression for CPUs without the SHA extensions
void compress_generic(uint32_t st[8], const uint32_t blk[16]) {
    uint32_t w[64];
    memcpy(w, blk, 64);
    for (int t = 16; t < 64; ++t) {
        uint32_t s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3);
        uint32_t s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10);
        w[t] = w[t - 16] + s0 + w[t - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int t = 0; t < 64; ++t) {
        uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[t] + w[t];
        uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

}  // namespace

bool host_sha_available() { return g_have_sha; }
void host_sha_use_extensions(bool on) {
    g_have_sha = on && g_cpu_has_sha;
    g_have_x16 = on && g_cpu_has_x16 && g_want_x16.load(std::memory_order_relaxed);
}
bool host_sha_wide_available() { return g_have_x16; }
void host_sha_use_wide(bool on) {
    g_want_x16 = on;
    g_have_x16 = on && g_have_sha.load(std::memory_order_relaxed) && g_cpu_has_x16;
}

void host_sha_compress(uint32_t state[8], const uint32_t block[16]) {
    if (g_have_sha) compress_ni(state, block);
    else compress_generic(state, block);
}

void host_sha_blocks(uint32_t state[8], const uint8_t* data, size_t blocks) {
    if (g_have_sha) { blocks_ni(state, data, blocks); return; }
    for (; blocks; --blocks, data += 64) {
        uint32_t w[16];
        for (int i = 0; i < 16; ++i)
            w[i] = ((uint32_t)data[4 * i] << 24) | ((uint32_t)data[4 * i + 1] << 16) | ((uint32_t)data[4 * i + 2] << 8) | data[4 * i + 3];
        compress_generic(state, w);
    }
}

void host_sha_leaf(uint32_t v, uint32_t out[8]) {
    if (g_have_sha) { leaf_ni(v, out); return; }
    uint32_t blk[16] = {v, 0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 32u};
    uint32_t st[8];
    memcpy(st, IV, 32);
    compress_generic(st, blk);
    memcpy(out, st, 32);
}

void host_sha_leaves(const uint32_t* vals, size_t n, uint32_t* out) {
    size_t i = 0;
    if (g_have_x16)
        for (; i + 16 <= n; i += 16) leaves_x16(vals + i, out + 8 * i);
    for (; i < n; ++i) host_sha_leaf(vals[i], out + 8 * i);
}

void host_sha_inner(const uint32_t left[8], const uint32_t right[8], uint32_t out[8]) {
    if (g_have_sha) { inner_ni(left, right, out); return; }
    uint32_t blk[16], st[8];
    memcpy(blk, left, 32);
    memcpy(blk + 8, right, 32);
    memcpy(st, IV, 32);
    compress_generic(st, blk);
    uint32_t pad[16] = {0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 512u};
    compress_generic(st, pad);
    memcpy(out, st, 32);
}

void host_sha_inner_run(const uint32_t* children, uint32_t* out, size_t cnt) {
    if (g_have_sha) { inner_run(children, out, cnt); return; }
    for (size_t i = 0; i < cnt; ++i) host_sha_inner(children + 16 * i, children + 16 * i + 8, out + 8 * i);
}

void host_sha_reduce_sub(uint32_t* nodes, uint32_t depth, uint32_t top, size_t sub) {
    if (g_have_sha) { reduce_sub_ni(nodes, depth, top, sub); return; }
    for (uint32_t d = depth; d-- > top;) {
        const size_t cnt = (size_t)1 << (d - top);
        const size_t base = ((size_t)1 << d) - 1 + sub * cnt, child = ((size_t)2 << d) - 1 + 2 * sub * cnt;
        for (size_t i = 0; i < cnt; ++i)
            host_sha_inner(nodes + 8 * (child + 2 * i), nodes + 8 * (child + 2 * i + 1), nodes + 8 * (base + i));
    }
}

void host_sha_reduce(uint32_t* nodes, uint32_t depth) {
    if (g_have_sha) { reduce_ni(nodes, depth); return; }
    for (uint32_t d = depth; d-- > 0;) {
        const size_t base = ((size_t)1 << d) - 1, child = ((size_t)2 << d) - 1, cnt = (size_t)1 << d;
        for (size_t i = 0; i < cnt; ++i)
            host_sha_inner(nodes + 8 * (child + 2 * i), nodes + 8 * (child + 2 * i + 1), nodes + 8 * (base + i));
    }
}

}  // namespace zk
#endif
