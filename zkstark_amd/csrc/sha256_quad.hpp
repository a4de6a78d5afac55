// sha256_quad.hpp -- one SHA256(left || right) (merkle.rs:42-45) spread over FOUR lanes, for the latency-bound levels
// of a Merkle tree (kernels.hip, merkle_wg_kernel).
//
// A level narrower than the chip costs one hash latency, and a wave that has its SIMD to itself issues one VALU
// instruction every ~4.4 cycles whatever the instruction does: the only way to shorten the level is to issue FEWER
// instructions per hash.  Lanes are free there (most of the chip idles), so the work of one compression is laid across
// the four banks of a 16-lane DPP row and every instruction does a different job in each bank:
//
//   bank 0 "E"  holds e, f, g, h:  Sigma1(e), Ch(e, f, g), T1 = h + Sigma1 + Ch + K + W
//   bank 1 "A"  holds a, b, c, d:  Sigma0(a), Maj(a, b, c), T2 = Sigma0 + Maj
//   bank 2 "S0" holds the message window: sigma0(W[t+1]) + W[t] + W[t+9]
//   bank 3 "S1" holds W[t+14 ..]:         sigma1(W[t+14])
//
// The three rotations take their amounts from a per-lane register (6/11/25, 2/13/22, 7/18/3, 17/19/10; the third term of
// the small sigmas is a shift: its high operand is masked to zero in banks 2-3), Ch and Maj are one bit-select on
// differently prepared operands (Maj(a, b, c) = (a ^ c) ? b : c), and the places where a value crosses banks --
// W[t] to E, d to E, T1 to A, the two halves of W[t+16] -- are v_add_u32_dpp with row_ror and a bank mask, so the
// exchange is fused into additions the round needs anyway.  All sixteen "registers" of a lane are one ring Z[t mod 16]:
// in banks 0-1 Z[t] is e_t / a_t (f = Z[t-1], g = Z[t-2], h = Z[t-3]), in bank 2 Z[j] = W[j+1], in bank 3
// Z[j] = W[j+14], which makes every operand of round t the same register name in all four banks.
// 16 instructions per round with the schedule, 12 without, 11 in the constant padding block: ~1 700 per hash instead
// of 2 293 on one lane (1 850 on the critical path of the main/helper split this replaces on levels of <= 64 nodes per
// workgroup), and no LDS exchange or barrier inside the hash.  A wave holds 16 hashes (row r, lane-in-bank i).
#pragma once
#include "sha256.hpp"

namespace zk {

struct PadWK {
    uint32_t v[64];
};
// K[t] + W[t] of the constant second block (0x80000000, 0, ..., 512) of a 64-byte message
constexpr PadWK make_pad_wk() {
    uint32_t w[64] = {0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 512u};
    for (int i = 16; i < 64; ++i) {
        const uint32_t a = w[i - 15], b = w[i - 2];
        const uint32_t s0 = ((a >> 7) | (a << 25)) ^ ((a >> 18) | (a << 14)) ^ (a >> 3);
        const uint32_t s1 = ((b >> 17) | (b << 15)) ^ ((b >> 19) | (b << 13)) ^ (b >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    PadWK r{};
    for (int i = 0; i < 64; ++i) r.v[i] = w[i] + SHA_K[i];
    return r;
}
constexpr PadWK kPadWK = make_pad_wk();

#if defined(__HIPCC__)   // the host pass of hipcc parses the kernels too: it sees an empty body where the device has DPP

// what a lane is, computed once per kernel
struct QuadLane {
    uint32_t s1, s2, s3;     // shift amounts of the three terms
    uint32_t hmask;          // ~0 where the third term is a rotation (banks 0-1), 0 where it is a shift
    uint32_t amask;          // ~0 in bank 1 (Maj wants a ^ c), 0 elsewhere
    uint32_t iv[4];          // initial e,f,g,h (bank 0) / a,b,c,d (bank 1)
    bool ea, s1lane;
};
__device__ __forceinline__ QuadLane quad_lane(uint32_t lane) {
    const uint32_t role = (lane >> 2) & 3u;
    QuadLane q;
    q.s1 = role == 0 ? 6u : role == 1 ? 2u : role == 2 ? 7u : 17u;
    q.s2 = role == 0 ? 11u : role == 1 ? 13u : role == 2 ? 18u : 19u;
    q.s3 = role == 0 ? 25u : role == 1 ? 22u : role == 2 ? 3u : 10u;
    q.hmask = role < 2 ? 0xffffffffu : 0u;
    q.amask = role == 1 ? 0xffffffffu : 0u;
    q.ea = role < 2;
    q.s1lane = role == 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) q.iv[i] = role == 0 ? SHA_IV[4 + i] : SHA_IV[i];
    return q;
}

// (Tried and dropped, tools/sha_quad_probe.hip: the 128 round constants resident in VGPRs -- 7 555 instead of 7 697 cycles per
// hash, but 260 registers for every kernel that contains this code; preparing h + K + W one round ahead so that no
// instruction needs the result of the one before it -- slower, the compiler separates split assembly blocks with s_nop.)
// One round as ONE block of assembly.  The order is the point: a DPP operand must have been written at least two
// instructions earlier (the hardware does not interlock that, and neither the compiler's hazard recogniser nor its
// scheduler looks into inline assembly), and on a wave that has its SIMD to itself an s_nop costs about as much as an
// instruction.  Only T1 is both produced and DPP-read inside a round: the schedule additions (or T2) sit in between.
//   SCHED: also produce W[T+16];  MSG: W[T] comes from bank 2 and KC is K[T], otherwise KC is K[T] + W[T] of the
//   padding block (handed over in an SGPR where the instruction is v_add3_u32, which takes no literal).
// row_ror:n -- lane L of a row reads lane (L - n) mod 16: bank + 1 is row_ror:12, bank + 2 row_ror:8, bank - 1 row_ror:4
// (checked on the device by tools/sha_quad_probe.hip).
// x = Z[T], y = Z[T-1] (bank 2 overwrites it with W[T+16]), z = Z[T-2], hd = Z[T-3], w9 = Z[T+8], n1 = Z[T+1], n2 = Z[T+2].
template <bool SCHED, bool MSG, uint32_t KC>
__device__ __forceinline__ void quad_round(uint32_t x, uint32_t& y, uint32_t z, uint32_t hd, uint32_t w9, uint32_t& n1, uint32_t& n2,
                                           const QuadLane& q) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t0, t1, t2, t3;
    if constexpr (SCHED) {
        uint32_t t4;
        asm("v_and_b32 %0, %8, %15\n\t"
            "v_alignbit_b32 %1, %8, %8, %12\n\t"
            "v_alignbit_b32 %2, %8, %8, %13\n\t"
            "v_alignbit_b32 %0, %0, %8, %14\n\t"
            "v_bitop3_b32 %1, %1, %2, %0 bitop3:0x96\n\t"                                   // s
            "v_bitop3_b32 %2, %8, %9, %16 bitop3:0x78\n\t"                                  // e | a ^ c
            "v_bitop3_b32 %2, %2, %5, %9 bitop3:0xca\n\t"                                   // Ch | Maj
            "v_add_u32 %2, %1, %2\n\t"                                                      // T2
            "v_add_u32 %3, %17, %2\n\t"                                                     // + K
            "v_add_u32_dpp %3, %10, %3 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1\n\t"  // E: + h
            "v_add3_u32 %4, %5, %1, %11\n\t"                                                // S0: W[T] + sigma0 + W[T+9]
            "v_add_u32_dpp %3, %5, %3 row_ror:8 row_mask:0xf bank_mask:0x1\n\t"             // E: + W[T] (S0) = T1
            "v_add_u32_dpp %5, %1, %4 row_ror:12 row_mask:0xf bank_mask:0x4\n\t"            // S0: W[T+16] = sigma1 (S1) + ...
            "v_add_u32_dpp %6, %10, %3 row_ror:12 row_mask:0xf bank_mask:0x1\n\t"           // E: e' = d (A) + T1
            "v_add_u32_dpp %7, %4, %1 row_ror:4 row_mask:0xf bank_mask:0x8\n\t"             // S1: W[T+16] = ... (S0) + sigma1
            "v_add_u32_dpp %6, %3, %2 row_ror:4 row_mask:0xf bank_mask:0x2"                 // A: a' = T1 (E) + T2
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "+v"(y), "+v"(n1), "+v"(n2)
            : "v"(x), "v"(z), "v"(hd), "v"(w9), "v"(q.s1), "v"(q.s2), "v"(q.s3), "v"(q.hmask), "v"(q.amask), "n"(KC));
    } else if constexpr (MSG) {
        const uint32_t k = KC;
        asm("v_alignbit_b32 %1, %6, %6, %9\n\t"
            "v_alignbit_b32 %2, %6, %6, %10\n\t"
            "v_alignbit_b32 %0, %6, %6, %11\n\t"
            "v_bitop3_b32 %1, %1, %2, %0 bitop3:0x96\n\t"
            "v_bitop3_b32 %2, %6, %7, %12 bitop3:0x78\n\t"
            "v_bitop3_b32 %2, %2, %5, %7 bitop3:0xca\n\t"
            "v_add3_u32 %3, %1, %2, %13\n\t"                                                // Sigma + Ch + K
            "v_add_u32_dpp %3, %8, %3 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1\n\t"   // E: + h
            "v_add_u32_dpp %3, %5, %3 row_ror:8 row_mask:0xf bank_mask:0x1\n\t"             // E: + W[T] (S0) = T1
            "v_add_u32_dpp %4, %8, %3 row_ror:12 row_mask:0xf bank_mask:0x1\n\t"            // E: e' = d (A) + T1
            "v_add_u32 %0, %1, %2\n\t"                                                      // T2
            "v_add_u32_dpp %4, %3, %0 row_ror:4 row_mask:0xf bank_mask:0x2"                 // A: a' = T1 (E) + T2
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(n1)
            : "v"(y), "v"(x), "v"(z), "v"(hd), "v"(q.s1), "v"(q.s2), "v"(q.s3), "v"(q.amask), "s"(k));
    } else {
        const uint32_t k = KC;
        asm("v_alignbit_b32 %1, %6, %6, %9\n\t"
            "v_alignbit_b32 %2, %6, %6, %10\n\t"
            "v_alignbit_b32 %0, %6, %6, %11\n\t"
            "v_bitop3_b32 %1, %1, %2, %0 bitop3:0x96\n\t"
            "v_bitop3_b32 %2, %6, %7, %12 bitop3:0x78\n\t"
            "v_bitop3_b32 %2, %2, %5, %7 bitop3:0xca\n\t"
            "v_add3_u32 %3, %1, %2, %13\n\t"                                                // Sigma + Ch + (K + W)
            "v_add_u32_dpp %3, %8, %3 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1\n\t"   // E: + h = T1
            "v_add_u32_dpp %4, %8, %3 row_ror:12 row_mask:0xf bank_mask:0x1\n\t"            // E: e' = d (A) + T1
            "v_add_u32 %0, %1, %2\n\t"                                                      // T2
            "v_add_u32_dpp %4, %3, %0 row_ror:4 row_mask:0xf bank_mask:0x2"                 // A: a' = T1 (E) + T2
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(n1)
            : "v"(y), "v"(x), "v"(z), "v"(hd), "v"(q.s1), "v"(q.s2), "v"(q.s3), "v"(q.amask), "s"(k));
    }
    (void)w9; (void)n2;
#else
    (void)x; (void)y; (void)z; (void)hd; (void)w9; (void)n1; (void)n2; (void)q;
#endif
}

template <int T0, int N, bool SCHED, bool MSG>
__device__ __forceinline__ void quad_rounds(uint32_t (&Z)[16], const QuadLane& q) {
    if constexpr (N > 0) {
        quad_round<SCHED, MSG, MSG ? SHA_K[T0] : kPadWK.v[T0]>(Z[T0 & 15], Z[(T0 + 15) & 15], Z[(T0 + 14) & 15], Z[(T0 + 13) & 15], Z[(T0 + 8) & 15],
                                                               Z[(T0 + 1) & 15], Z[(T0 + 2) & 15], q);
        quad_rounds<T0 + 1, N - 1, SCHED, MSG>(Z, q);
    }
}

// msg: the 16 message words of this lane's hash (left || right), 16-byte aligned, usually LDS; all four lanes of a
// hash pass the same pointer.  Result: bank 1 lanes hold digest words 0..3 in out[0..3], bank 0 lanes words 4..7;
// banks 2-3 hold nothing of use.  Every lane of the wave must execute the call (DPP reads its neighbours).
__device__ __forceinline__ void sha256_inner_quad(const uint32_t* msg, const QuadLane& q, uint32_t (&out)[4]) {
    const uint4* m4 = reinterpret_cast<const uint4*>(msg);
    const uint4 m0 = m4[0], m1 = m4[1], m2 = m4[2], m3 = m4[3];
    const uint32_t M[16] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w, m3.x, m3.y, m3.z, m3.w};
    uint32_t Z[16];
#pragma unroll
    for (int j = 2; j < 13; ++j) Z[j] = M[j + 1];                            // bank 2: Z[j] = W[j+1]
    Z[13] = q.ea ? q.iv[3] : M[14];
    Z[14] = q.ea ? q.iv[2] : M[15];
    Z[15] = q.ea ? q.iv[1] : M[0];
    Z[0] = q.ea ? q.iv[0] : (q.s1lane ? M[14] : M[1]);                       // bank 3: Z[j] = W[j+14]
    Z[1] = q.s1lane ? M[15] : M[2];
    quad_rounds<0, 48, true, true>(Z, q);
    quad_rounds<48, 16, false, true>(Z, q);
    // feed-forward; the chaining value is the initial state of the padding block (ring position 64 = 0)
    const uint32_t h0 = Z[0] + q.iv[0], h1 = Z[15] + q.iv[1], h2 = Z[14] + q.iv[2], h3 = Z[13] + q.iv[3];
    Z[0] = h0; Z[15] = h1; Z[14] = h2; Z[13] = h3;
    quad_rounds<0, 64, false, false>(Z, q);
    out[0] = Z[0] + h0; out[1] = Z[15] + h1; out[2] = Z[14] + h2; out[3] = Z[13] + h3;
}

#endif  // __HIPCC__

}  // namespace zk
