// fieldhash_f64.hpp -- the field-native Merkle hash of fieldhash.hpp (same function, bit for bit) computed in
// double precision on the device.
//
// Why.  P = 3 * 2^30 + 1 > 2^31 leaves 32-bit arithmetic no headroom: every modular addition carries its own
// correction (4 instructions) and the permutation is ~1 400 additions around ~800 products, 9 092 VALU instructions per
// hash.  A double holds every integer below 2^53 exactly, v_add_f64 / v_fma_f64 issue at the rate of the 4-cycle integer
// ops, and an addition is ONE instruction with no correction at all: sums are left to grow (the whole external layer and
// the round constants run without a single reduction) and only products are reduced, because a product has to be.
//
//   mulmod(a, b):  h = a * b (rounded), l = fma(a, b, -h)        a * b = h + l exactly, whatever the magnitudes
//                  q = rint(h / P), r = fma(-q, P, h) + l         r = a b (mod P), |r| <= 0.75 P + 2^28
//     exact while |h| < 2^50 P (q is then an integer within 3/4 of h / P, so h - q P is an integer below 2^32 and the
//     fused multiply-add returns it exactly); every product but one kind has |a b| < 2^76.
//     The exception is the FIRST product of the S-box of s_0 in the partial rounds: s_0 <- sum - 2 s_0 is never
//     reduced by the internal layer, so that S-box squares an input of up to 2^41.9 (sum <= 2^41.8, below): h < 2^83.8,
//     h / P < 2^52.2.  There the quotient is no longer within 3/4: h * (1 / P) carries a relative error of 2^-52 (the rounded
//     reciprocal and the rounded product), an absolute one of <= 1.15, so q = rint(..) is an integer within 1.65 of h / P; it is
//     still exact as a double (< 2^53), h (a multiple of 2^31) - q P is an integer below 1.65 P < 2^32.4, the fused multiply-add
//     returns it exactly, and l = fma(a, a, -h) is an integer below 2^30.8: x2 = x^2 (mod P) EXACTLY, with |x2| < 2^32.6 instead
//     of < 0.84 P.  The two products that follow are in the ordinary regime (x2^2 < 2^65.2; x4 x < 2^31.3 * 2^41.9 = 2^73.2), so
//     the S-box output is <= 0.75 P + 2^19.2 like every other.  zk_probe_fieldhash_forms feeds the S-box directed inputs of
//     +-2^37.8 ... +-2^42 and multiples of P next to them, against x^5 in plain integer arithmetic (tests/test_fieldhash.py).
//   The state is kept as SIGNED representatives; magnitudes (bounds, not estimates):
//     S-box output            <= 0.75 P + 2^28 < 0.84 P
//     external layer output   <= 5 * 16 * 0.84 P < 2^37.7           (M4 row sums <= 16, then + the four blocks' sum)
//     S-box input             <  2^37.8 (full rounds), < 2^41.9 (s_0 in the partial rounds: see above)
//     internal layer          d_i s_i + sum <= 2^14 * 2^37.7 + 2^41.7 < 2^51.7 on entry (exact); s_1 .. s_15 are reduced to
//                             [-P/2, P/2] every second round, s_12 .. s_15 every round (fh64_internal); s_0 by its S-box
//   Digest words are canonical residues again: reduce, add P if negative, convert.
//
// Used by the throughput kernel (merkle_subtree_kernel<.., HASH = 1>), the leaf / wide levels of the latency kernel, the
// chain probe, and -- as a 16-lane row form, at the end of this file -- the narrow levels.  Device only; the host (verifier, tree
// tops of the sharded prover) keeps the Montgomery code; the tests compare both with an independent plain-residue implementation.
#pragma once
#include "fieldhash.hpp"

#if defined(__HIPCC__)
namespace zk {

struct FieldHashConsts64 {   // canonical residues as doubles
    double rc_full[kFhRF][kFhT];
    double rc_part[kFhRP];
};

inline void fieldhash_make_consts64(const FieldHashConsts& c, FieldHashConsts64& d) {
    // Montgomery form -> canonical: x R * 1 * R^-1
    for (int r = 0; r < kFhRF; ++r)
        for (int i = 0; i < kFhT; ++i) d.rc_full[r][i] = (double)mont_mul(c.rc_full[r][i], 1u);
    for (int r = 0; r < kFhRP; ++r) d.rc_part[r] = (double)mont_mul(c.rc_part[r], 1u);
}

constexpr double kPd = 3221225473.0;
constexpr double kPinvd = 1.0 / 3221225473.0;

// x (an integer, |x| < 2^50 P) -> the representative of x mod P in [-P/2, P/2] (up to the rounding of x / P: |r| < 0.75 P)
__device__ __forceinline__ double fh64_reduce(double x) {
    const double q = __builtin_rint(x * kPinvd);
    return __builtin_fma(-q, kPd, x);
}
__device__ __forceinline__ double fh64_mulmod(double a, double b) {
    const double h = __dmul_rn(a, b);
    const double l = __builtin_fma(a, b, -h);
    return fh64_reduce(h) + l;
}
__device__ __forceinline__ double fh64_sbox(double x) {
    const double x2 = fh64_mulmod(x, x);
    return fh64_mulmod(fh64_mulmod(x2, x2), x);
}
// the add / double sequence of fh_m4 with the doublings folded into multiply-adds: 8 instructions per block
__device__ __forceinline__ void fh64_m4(double& x0, double& x1, double& x2, double& x3) {
    const double t0 = x0 + x1, t1 = x2 + x3;
    const double t2 = __builtin_fma(x1, 2.0, t1), t3 = __builtin_fma(x3, 2.0, t0);
    const double t4 = __builtin_fma(t1, 4.0, t3), t5 = __builtin_fma(t0, 4.0, t2);
    x0 = t3 + t5; x1 = t5; x2 = t2 + t4; x3 = t4;
}
__device__ __forceinline__ void fh64_external(double (&s)[kFhT]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fh64_m4(s[4 * b], s[4 * b + 1], s[4 * b + 2], s[4 * b + 3]);
    double col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) col[j] = (s[j] + s[4 + j]) + (s[8 + j] + s[12 + j]);
#pragma unroll
    for (int i = 0; i < kFhT; ++i) s[i] = s[i] + col[i & 3];
}
// I: s_i <- d_i s_i + sum_j s_j, d = (-2, 1, 2, 4, ..., 2^14): one multiply-add per element, and a reduction where the next
// round needs one.  ALL = true reduces s_1 .. s_15 to [-P/2, P/2] (rho = 0.5 P); ALL = false only s_12 .. s_15, the elements a
// second unreduced round would push past 2^53:
//   after a partial round on a reduced state:   |s_i| <= 2^(i-1) rho + 8.4 P  <= 2^40.7   (i <= 11),  |s_0| <= 10 P
//   the round after it (always ALL = true):      sum <= 2^41.8,  |d_i s_i + sum| <= 2^10 * 2^40.7 + 2^41.8 < 2^50.8, exact
// s_0 is never reduced here: its S-box in the next round (or the full round's) does it.
template <bool ALL>
__device__ __forceinline__ void fh64_internal(double (&s)[kFhT]) {
    double sum = s[0];
#pragma unroll
    for (int i = 1; i < kFhT; ++i) sum += s[i];
    s[0] = __builtin_fma(s[0], -2.0, sum);
#pragma unroll
    for (int i = 1; i < kFhT; ++i) {
        const double t = __builtin_fma(s[i], (double)(1u << (i - 1)), sum);
        s[i] = (ALL || i >= 12) ? fh64_reduce(t) : t;
    }
}
template <bool ALL>
__device__ __forceinline__ void fh64_partial_round(double (&s)[kFhT], double rc) {
    s[0] = fh64_sbox(s[0] + rc);
    fh64_internal<ALL>(s);
}
__device__ __forceinline__ void fh64_permute(double (&s)[kFhT], const FieldHashConsts64& c) {
    fh64_external(s);
#pragma unroll 1
    for (int r = 0; r < kFhRF / 2; ++r) {
#pragma unroll
        for (int i = 0; i < kFhT; ++i) s[i] = fh64_sbox(s[i] + c.rc_full[r][i]);
        fh64_external(s);
    }
    // 22 partial rounds.  The first one takes the external layer's output (up to 2^37.7) and the last one feeds the full
    // rounds: both reduce everything; in between the rounds alternate (partly reduced, fully reduced).
    static_assert(kFhRP % 2 == 0 && kFhRP >= 4, "partial-round schedule");
    fh64_partial_round<true>(s, c.rc_part[0]);
#pragma unroll 1
    for (int r = 1; r < kFhRP - 1; r += 2) {
        fh64_partial_round<false>(s, c.rc_part[r]);
        fh64_partial_round<true>(s, c.rc_part[r + 1]);
    }
    fh64_partial_round<true>(s, c.rc_part[kFhRP - 1]);
#pragma unroll 1
    for (int r = kFhRF / 2; r < kFhRF; ++r) {
#pragma unroll
        for (int i = 0; i < kFhT; ++i) s[i] = fh64_sbox(s[i] + c.rc_full[r][i]);
        fh64_external(s);
    }
}
// signed representative -> canonical residue
__device__ __forceinline__ uint32_t fh64_canonical(double x) {
    double r = fh64_reduce(x);
    r = r < 0.0 ? r + kPd : r;
    return (uint32_t)r;                                    // an integer in [0, P): v_cvt_u32_f64 is exact
}
// in: any u32 words (a raw word >= P is the same field element as its residue: field.rs:20-24); out: 8 canonical residues
__device__ __forceinline__ void fh64_compress(const uint32_t (&in)[kFhT], uint32_t (&out)[8], const FieldHashConsts64& c) {
    double s[kFhT];
#pragma unroll
    for (int i = 0; i < kFhT; ++i) s[i] = (double)in[i];
    fh64_permute(s, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = fh64_canonical(s[i] + (double)in[i]);
}
__device__ __forceinline__ Digest fieldhash_leaf64(uint32_t v, const FieldHashConsts64& c) {
    const uint32_t in[kFhT] = {v, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1u};
    Digest d;
    fh64_compress(in, d.w, c);
    return d;
}
__device__ __forceinline__ Digest fieldhash_inner64(const Digest& l, const Digest& r, const FieldHashConsts64& c) {
    uint32_t in[kFhT];
#pragma unroll
    for (int i = 0; i < 8; ++i) { in[i] = l.w[i]; in[8 + i] = r.w[i]; }
    Digest d;
    fh64_compress(in, d.w, c);
    return d;
}

// ---- one hash on a ROW of sixteen lanes, in double precision ------------------------------------------------------------
// The narrow levels of a tree (fewer nodes than the workgroup has rows) cost one hash latency each.  fieldhash_inner_row16
// (fieldhash.hpp) spreads the 32-bit permutation over a DPP row: ~1 700 instructions, 5.2 us per pass on a lone wave (64-bit
// multiply-adds and their reductions, DPP hazards).  The same layout with the arithmetic of this file: lane g of a row holds
// state element g as a signed double; the M4 block map is four quad broadcasts and a multiply-add chain with the lane's matrix
// row, the block sum three row rotations, the internal layer's sum four rotate-and-add steps; every lane reduces its element in
// every partial round (uniform code: the row form cannot skip per element).  ~1 170 instructions per hash.
template <int CTRL>
__device__ __forceinline__ double fh64_dpp(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
#else
    return v;
#endif
}
// E on a row: y = M4 * (own quad), then s = y + (sum of the four quads' y at the same position)
__device__ __forceinline__ double fh64_external_row(double s, double c0, double c1, double c2, double c3) {
    const double x0 = fh64_dpp<0x00>(s), x1 = fh64_dpp<0x55>(s), x2 = fh64_dpp<0xAA>(s), x3 = fh64_dpp<0xFF>(s);   // quad_perm broadcasts
    const double y = __builtin_fma(c3, x3, __builtin_fma(c2, x2, __builtin_fma(c1, x1, c0 * x0)));
    return (__builtin_fma(y, 2.0, fh64_dpp<kDppRowRor + 4>(y)) + fh64_dpp<kDppRowRor + 8>(y)) + fh64_dpp<kDppRowRor + 12>(y);
}
// in_word: word g of left || right (any u32), g = lane & 15; returns word g of the digest (canonical) in lanes g < 8.
__device__ __forceinline__ uint32_t fieldhash_inner_row16_f64(uint32_t in_word, uint32_t g, const FieldHashConsts64& c) {
    // row (g & 3) of M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
    const uint32_t q = g & 3u;
    const double c0 = q == 0 ? 5.0 : q == 1 ? 4.0 : 1.0;
    const double c1 = q == 0 ? 7.0 : q == 1 ? 6.0 : q == 2 ? 3.0 : 1.0;
    const double c2 = q < 2 ? 1.0 : q == 2 ? 5.0 : 4.0;
    const double c3 = q == 0 ? 3.0 : q == 1 ? 1.0 : q == 2 ? 7.0 : 6.0;
    const double dg = g == 0 ? -2.0 : (double)(1u << (g == 0 ? 0u : g - 1u));
    double rcf[kFhRF];                                      // this lane's constants of the full rounds: read before they are needed
#pragma unroll
    for (int r = 0; r < kFhRF; ++r) rcf[r] = c.rc_full[r][g];
    const double keep = (double)in_word;
    double s = fh64_external_row(keep, c0, c1, c2, c3);
#pragma unroll
    for (int r = 0; r < kFhRF / 2; ++r) s = fh64_external_row(fh64_sbox(s + rcf[r]), c0, c1, c2, c3);
#pragma unroll 1
    for (int r = 0; r < kFhRP; ++r) {
        const double t = fh64_sbox(s + c.rc_part[r]);
        s = g == 0 ? t : s;
        double sum = s + fh64_dpp<kDppRowRor + 8>(s);
        sum = sum + fh64_dpp<kDppRowRor + 4>(sum);
        sum = sum + fh64_dpp<kDppRowRor + 2>(sum);
        sum = sum + fh64_dpp<kDppRowRor + 1>(sum);
        s = fh64_reduce(__builtin_fma(s, dg, sum));         // d_g s + sum: <= 2^14 * 2^37.7 + 2^41.7 on the first round, 2^45 after
    }
#pragma unroll
    for (int r = kFhRF / 2; r < kFhRF; ++r) s = fh64_external_row(fh64_sbox(s + rcf[r]), c0, c1, c2, c3);
    return fh64_canonical(s + keep);
}

// ---- one hash on a QUAD of four lanes, in double precision ----------------------------------------------------------------
// Between the one-lane hash (10.6 us on a lone wave, 256 nodes per workgroup and pass) and the 16-lane row form (3.9 us, 16
// nodes per pass): lane q of a quad holds 4-block q of the state.  The M4 block map is local (8 instructions), the block sums and
// the internal layer's sum are two quad exchanges per value (quad_perm [1,0,3,2] and [2,3,0,1]), the S-boxes of a full round run
// four per lane.  ~1 950 instructions per hash, ~6 us on a lone wave, 64 nodes per pass: levels of 33 .. 64 nodes per workgroup.
__device__ __forceinline__ double fh64_quad_sum(double v) {
    v = v + fh64_dpp<0xB1>(v);                             // quad_perm [1,0,3,2]
    return v + fh64_dpp<0x4E>(v);                          // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void fh64_external_quad(double (&x)[4]) {
    fh64_m4(x[0], x[1], x[2], x[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = x[j] + fh64_quad_sum(x[j]);
}
// in_words: words 4q .. 4q+3 of left || right (any u32), q = lane & 3; lanes q < 2 return words 4q .. 4q+3 of the digest (canonical)
__device__ __forceinline__ void fieldhash_inner_quad_f64(const uint32_t (&in_words)[4], uint32_t q, uint32_t (&out_words)[4], const FieldHashConsts64& c) {
    double rcf[kFhRF][4];                                   // this lane's constants of the full rounds, all read up front
#pragma unroll
    for (int r = 0; r < kFhRF; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) rcf[r][j] = c.rc_full[r][4 * q + j];
    double dg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dg[j] = (q == 0 && j == 0) ? -2.0 : (double)(1u << ((4 * q + j) == 0 ? 0u : 4 * q + j - 1u));
    double x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = (double)in_words[j];
    fh64_external_quad(x);
#pragma unroll
    for (int r = 0; r < kFhRF / 2; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = fh64_sbox(x[j] + rcf[r][j]);
        fh64_external_quad(x);
    }
#pragma unroll 1
    for (int r = 0; r < kFhRP; ++r) {
        const double t = fh64_sbox(x[0] + c.rc_part[r]);
        x[0] = q == 0 ? t : x[0];
        const double sum = fh64_quad_sum((x[0] + x[1]) + (x[2] + x[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = fh64_reduce(__builtin_fma(x[j], dg[j], sum));
    }
#pragma unroll
    for (int r = kFhRF / 2; r < kFhRF; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = fh64_sbox(x[j] + rcf[r][j]);
        fh64_external_quad(x);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out_words[j] = fh64_canonical(x[j] + (double)in_words[j]);
}

}  // namespace zk
#endif
