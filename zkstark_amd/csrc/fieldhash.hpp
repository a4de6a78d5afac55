// fieldhash.hpp -- field-native Merkle hash over GF(P) (BASELINE.json configs[4]).
//
// The reference has only SHA-256 (merkle.rs:1-2); this Poseidon2-style permutation is the build's
// own definition, so its parity is "self-defined": the test suite holds an independent
// implementation in plain residues, this one works in the Montgomery domain.  It exists to compare
// an arithmetic-bound commitment with the SHA-256 one on the same pipeline; it is a performance
// stand-in, NOT a vetted instance (round numbers and matrices were not analysed).
//
// Spec.  State: 16 field elements.  S-box x^5 (gcd(5, P-1) = 1).  perm(s):
//     s <- E(s)
//     4 x { s_i <- (s_i + c_full[r][i])^5 for all i;  s <- E(s) }
//    22 x { s_0 <- (s_0 + c_part[r])^5;               s <- I(s) }
//     4 x { s_i <- (s_i + c_full[4+r][i])^5;          s <- E(s) }
//   E: each 4-block (x0..x3) -> M4 block map (the add/double sequence in fh_m4), then every element
//      gets the sum of the elements in its column position added (circ(2 M4, M4, M4, M4)).
//   I: s_i <- d_i * s_i + sum_j s_j,  d = (-2, 1, 2, 4, ..., 2^14).
//   Constants: c = (first 8 bytes, big-endian, of SHA-256("zkstark_amd.fieldhash.v1" || LE32(k))) mod P,
//      k = 16 r + i for the full rounds, k = 128 + r for the partial rounds.
//   leaf(v)      = trunc8(perm(s) + s),  s = (v, 0, ..., 0, 1)
//   node(l, r)   = trunc8(perm(s) + s),  s = l || r (8 elements each)
//   A digest is 8 canonical residues; as bytes each is 4 bytes big-endian (32 bytes in all).
#pragma once
#include <stdint.h>
#include <string.h>

#include "field.hpp"
#include "sha256.hpp"

namespace zk {

constexpr int kFhT = 16, kFhRF = 8, kFhRP = 22;

struct FieldHashConsts {   // Montgomery form
    uint32_t rc_full[kFhRF][kFhT];
    uint32_t rc_part[kFhRP];
    uint32_t diag[kFhT];
};

constexpr uint32_t R2 = (uint32_t)((((uint64_t)R1) * R1) % P);   // 2^64 mod P: canonical -> Montgomery

inline void fieldhash_make_consts(FieldHashConsts& c) {
    const char* tag = "zkstark_amd.fieldhash.v1";
    const size_t tl = strlen(tag);
    for (uint32_t k = 0; k < (uint32_t)(kFhRF * kFhT + kFhRP); ++k) {
        uint8_t msg[64], dig[32];
        memcpy(msg, tag, tl);
        msg[tl] = (uint8_t)k; msg[tl + 1] = (uint8_t)(k >> 8); msg[tl + 2] = (uint8_t)(k >> 16); msg[tl + 3] = (uint8_t)(k >> 24);
        Sha256 h; h.update(msg, tl + 4); h.finalize(dig);
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | dig[i];
        uint32_t m = to_mont((uint32_t)(v % P));
        if (k < (uint32_t)(kFhRF * kFhT)) c.rc_full[k / kFhT][k % kFhT] = m; else c.rc_part[k - kFhRF * kFhT] = m;
    }
    c.diag[0] = to_mont(P - 2);
    for (int i = 1; i < kFhT; ++i) c.diag[i] = to_mont(1u << (i - 1));
}

ZK_HD uint32_t fh_sbox(uint32_t x) {
    uint32_t x2 = mont_mul(x, x);
    return mont_mul(mont_mul(x2, x2), x);
}
ZK_HD uint32_t fh_dbl(uint32_t x) { return add(x, x); }
ZK_HD void fh_m4(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    uint32_t t0 = add(x0, x1), t1 = add(x2, x3);
    uint32_t t2 = add(fh_dbl(x1), t1), t3 = add(fh_dbl(x3), t0);
    uint32_t t4 = add(fh_dbl(fh_dbl(t1)), t3), t5 = add(fh_dbl(fh_dbl(t0)), t2);
    uint32_t t6 = add(t3, t5), t7 = add(t2, t4);
    x0 = t6; x1 = t5; x2 = t7; x3 = t4;
}
ZK_HD void fh_external(uint32_t (&s)[kFhT]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fh_m4(s[4 * b], s[4 * b + 1], s[4 * b + 2], s[4 * b + 3]);
    uint32_t col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) col[j] = add(add(s[j], s[4 + j]), add(s[8 + j], s[12 + j]));
#pragma unroll
    for (int i = 0; i < kFhT; ++i) s[i] = add(s[i], col[i & 3]);
}
// (a * b + c) * R^-1 mod P for a 64-bit addend c with a * b + c < P * 2^32: the addend rides in v_mad_u64_u32 for free
ZK_HD uint32_t mont_mul_add(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t t = (uint64_t)a * b + c;
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo + (lo << 30);
    return mont_tail(hi, mul_hi_u32(m, P));
}
// v < 2^36 (a sum of 16 residues) -> v mod P.  q' = floor((v >> 30) / 3) is floor(v / P) or one more (P = 3 * 2^30 + 1),
// so v - q' P lies in [-P, P): one conditional correction on the 64-bit borrow.
ZK_HD uint32_t fh_reduce36(uint64_t v) {
    uint32_t x = (uint32_t)(v >> 30);                 // < 64
    uint32_t q = (x * 43u) >> 7;                      // floor(x / 3) for x < 128
    uint64_t t = (uint64_t)q * P;
    uint64_t r = v - t;                               // as a signed 64-bit value in [-P, P)
    uint32_t lo = (uint32_t)r;
    return (int64_t)r < 0 ? lo + P : lo;
}
// I: s_i <- d_i s_i + sum_j s_j.  The sum is accumulated in 64 bits (no per-add correction) and reduced once; each
// d_i s_i + sum is ONE Montgomery reduction of s_i * (d_i R) + (sum R), the addend riding in the multiply-add.
ZK_HD void fh_internal(uint32_t (&s)[kFhT], const FieldHashConsts& c) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < kFhT; ++i) acc += s[i];
    const uint32_t sum = fh_reduce36(acc);
    const uint64_t sum_r = mont_mul(sum, R2);          // sum * R mod P
#pragma unroll
    for (int i = 0; i < kFhT; ++i) s[i] = i == 1 ? add(s[i], sum) : mont_mul_add(s[i], c.diag[i], sum_r);
}
// s in Montgomery form
ZK_HD void fh_permute(uint32_t (&s)[kFhT], const FieldHashConsts& c) {
    fh_external(s);
#pragma unroll 1
    for (int r = 0; r < kFhRF / 2; ++r) {
#pragma unroll
        for (int i = 0; i < kFhT; ++i) s[i] = fh_sbox(add(s[i], c.rc_full[r][i]));
        fh_external(s);
    }
#pragma unroll 1
    for (int r = 0; r < kFhRP; ++r) {
        s[0] = fh_sbox(add(s[0], c.rc_part[r]));
        fh_internal(s, c);
    }
#pragma unroll 1
    for (int r = kFhRF / 2; r < kFhRF; ++r) {
#pragma unroll
        for (int i = 0; i < kFhT; ++i) s[i] = fh_sbox(add(s[i], c.rc_full[r][i]));
        fh_external(s);
    }
}
// in: canonical residues; out: 8 canonical residues
ZK_HD void fh_compress(const uint32_t (&in)[kFhT], uint32_t (&out)[8], const FieldHashConsts& c) {
    uint32_t s[kFhT], keep[8];
#pragma unroll
    for (int i = 0; i < kFhT; ++i) s[i] = mont_mul(in[i], R2);   // any u32 in, reduced (field.rs:20-24)
#pragma unroll
    for (int i = 0; i < 8; ++i) keep[i] = s[i];
    fh_permute(s, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = mont_mul(add(s[i], keep[i]), 1u);
}
ZK_HD Digest fieldhash_leaf(uint32_t v, const FieldHashConsts& c) {
    uint32_t in[kFhT] = {v, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1u};
    Digest d;
    fh_compress(in, d.w, c);
    return d;
}
ZK_HD Digest fieldhash_inner(const Digest& l, const Digest& r, const FieldHashConsts& c) {
    uint32_t in[kFhT];
#pragma unroll
    for (int i = 0; i < 8; ++i) { in[i] = l.w[i]; in[8 + i] = r.w[i]; }
    Digest d;
    fh_compress(in, d.w, c);
    return d;
}


#if defined(__HIPCC__)
// ---- one hash on SIXTEEN lanes -------------------------------------------------------------------------------------
// The latency-bound levels of a tree (fewer nodes than the chip has lanes) cost one hash LATENCY each, and the serial
// permutation above is ~9 000 dependent-ish instructions on one lane.  Its state has 16 elements and a DPP row is 16
// lanes: here lane g of a row holds state element g, the S-boxes of a full round run on 16 lanes at once, the 4x4 block
// map and the column sums of the external layer are quad broadcasts and row rotations (v_mov_b32 dpp), the sum of the
// internal layer is four rotate-and-add steps.  ~1 700 instructions per hash: five times less latency, sixteen times
// the lanes -- exactly what a level with <= 4 096 nodes has to spare.  Same function as fieldhash_inner, bit for bit.
template <int CTRL>
__device__ __forceinline__ uint32_t fh_dpp(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
#else
    return v;                              // host pass of the compiler only parses this
#endif
}
constexpr int kDppRowRor = 0x120;       // row_ror:n = 0x120 + n: lane g reads lane (g + n) mod 16 of its row... (rotate right)
// E on a row: y = M4 * (own quad), then s = y + (sum of the four quads' y at the same position)
__device__ __forceinline__ uint32_t fh_external_row(uint32_t s, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    const uint32_t x0 = fh_dpp<0x00>(s), x1 = fh_dpp<0x55>(s), x2 = fh_dpp<0xAA>(s), x3 = fh_dpp<0xFF>(s);   // quad_perm broadcasts
    const uint64_t acc = (uint64_t)c0 * x0 + (uint64_t)c1 * x1 + (uint64_t)c2 * x2 + (uint64_t)c3 * x3;        // < 16 P
    const uint32_t y = fh_reduce36(acc);
    const uint64_t t = (uint64_t)y + y + fh_dpp<kDppRowRor + 4>(y) + fh_dpp<kDppRowRor + 8>(y) + fh_dpp<kDppRowRor + 12>(y);   // < 5 P
    return fh_reduce36(t);
}
// in_word: word g of left || right (canonical), g = lane & 15; returns word g of the digest in lanes g < 8.
__device__ __forceinline__ uint32_t fieldhash_inner_row16(uint32_t in_word, uint32_t g, const FieldHashConsts& c) {
    // row (g & 3) of M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] (the add/double sequence of fh_m4 written out)
    const uint32_t q = g & 3u;
    const uint32_t c0 = q == 0 ? 5u : q == 1 ? 4u : 1u;
    const uint32_t c1 = q == 0 ? 7u : q == 1 ? 6u : q == 2 ? 3u : 1u;
    const uint32_t c2 = q < 2 ? 1u : q == 2 ? 5u : 4u;
    const uint32_t c3 = q == 0 ? 3u : q == 1 ? 1u : q == 2 ? 7u : 6u;
    uint32_t s = mont_mul(in_word, R2);
    const uint32_t keep = s;
    s = fh_external_row(s, c0, c1, c2, c3);
#pragma unroll 1
    for (int r = 0; r < kFhRF / 2; ++r) {
        s = fh_sbox(add(s, c.rc_full[r][g]));
        s = fh_external_row(s, c0, c1, c2, c3);
    }
    const uint32_t dg = c.diag[g];
#pragma unroll 1
    for (int r = 0; r < kFhRP; ++r) {
        const uint32_t t = fh_sbox(add(s, c.rc_part[r]));
        s = g == 0 ? t : s;
        uint32_t sum = add(s, fh_dpp<kDppRowRor + 8>(s));
        sum = add(sum, fh_dpp<kDppRowRor + 4>(sum));
        sum = add(sum, fh_dpp<kDppRowRor + 2>(sum));
        sum = add(sum, fh_dpp<kDppRowRor + 1>(sum));
        s = mont_mul_add(s, dg, (uint64_t)mont_mul(sum, R2));     // d_g s + sum (d_1 = 1 in Montgomery form: plain s + sum)
    }
#pragma unroll 1
    for (int r = kFhRF / 2; r < kFhRF; ++r) {
        s = fh_sbox(add(s, c.rc_full[r][g]));
        s = fh_external_row(s, c0, c1, c2, c3);
    }
    return mont_mul(add(s, keep), 1u);
}
#endif

}  // namespace zk
