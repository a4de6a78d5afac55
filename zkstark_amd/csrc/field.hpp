// field.hpp -- arithmetic in GF(P), P = 3*2^30 + 1 = 3221225473, shared by host and device code.
//
// Replaces the reference's Gf<P> newtype over num_modular::MontgomeryInt<u32>
// (field.rs:8-211, modulus at main.rs:13).  As in the reference the internal
// Montgomery form is not observable: only canonical residues in [0, P) are
// ever stored in HBM or cross the C ABI (Merkle leaves hash residue(),
// prover.rs:81).
//
// Representation choice for gfx950.  P > 2^31, so a + b overflows u32 and there
// is no lazy-reduction headroom; every add/sub is fully reduced with a
// carry/borrow select.  Multiplication is Montgomery with R = 2^32, and the
// shape of P makes the reduction cheap: P^-1 mod 2^32 = 2^30 + 1, so
// m = lo * P^-1 is one shift-add, leaving three 32x32 multiplies per product
// (lo, hi of a*b and hi of m*P).
//
// Convention used by every kernel: bulk data (trace, evaluations, FRI layers)
// is stored as CANONICAL residues; constants and twiddles are stored in
// MONTGOMERY form (x*R mod P).  mont_mul(canonical, montgomery) = canonical, so
// butterflies and scalings never convert the data.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

constexpr uint32_t P = 3221225473u;        // main.rs:13
constexpr uint32_t P_INV = 0x40000001u;    // P * P_INV == 1 (mod 2^32)
constexpr uint32_t R1 = 1073741823u;       // 2^32 mod P   (Montgomery form of 1)
constexpr uint32_t GEN_W = 5u;             // smallest primitive root (field.rs:52-86, prover.rs:44)

// Build-time variants of the reduction tails (tools/ab_field_variants.sh measures them in the real kernels):
//   ZK_MONT_VARIANT 2 (default): borrow of the subtract (v_sub_co_u32) selects the correction     -- 6 VALU
//   ZK_MONT_VARIANT 4: v_cmp_lt_u32 + v_cndmask(0, P) + v_add, simple ops only after the multiplies -- 7 VALU
//   ZK_MONT_VARIANT 0: round 2's form (the compiler compares the 64-bit halves)                     -- 8 VALU
//   ZK_ADDSUB_CMP 1: add / sub select the correction with v_cmp_lt_u32 instead of the subtract's borrow
#ifndef ZK_MONT_VARIANT
#define ZK_MONT_VARIANT 2
#endif
#ifndef ZK_ADDSUB_CMP
#define ZK_ADDSUB_CMP 0
#endif

// P if a < b else 0, from a 32-bit compare (device: one v_cmp_lt_u32 + one v_cndmask_b32, no carry-out instruction)
ZK_HD uint32_t p_if_less(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t c;
    asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(c) : "v"(a), "v"(b), "v"(P) : "vcc");
    return c;
#else
    return a < b ? P : 0u;
#endif
}

// field.rs:99-111
ZK_HD uint32_t add(uint32_t a, uint32_t b) {
    uint32_t nb = P - b;            // in (0, P]
    uint32_t d = a - nb;            // a + b - P (mod 2^32)
#if ZK_ADDSUB_CMP
    return d + p_if_less(a, nb);
#else
    return a < nb ? d + P : d;      // a + b < P  ->  a + b
#endif
}
// field.rs:113-132
ZK_HD uint32_t sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
#if ZK_ADDSUB_CMP
    return d + p_if_less(a, b);
#else
    return a < b ? d + P : d;
#endif
}
// field.rs:198-203
ZK_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }

// Montgomery product a*b*R^-1 mod P, result canonical in [0, P).
// Requires a*b < P*2^32, i.e. at least one operand < P (the other may be any u32,
// which is how raw u32 challenges >= P are absorbed: field.rs:20-24).
// Default form: hipcc emits six VALU instructions: v_mad_u64_u32 (lo and hi of a*b in one four-cycle op),
// v_lshl_add_u32, v_mul_hi_u32, v_sub_co_u32 (the borrow IS the comparison), v_add_u32, v_cndmask_b32.
ZK_HD uint32_t mul_hi_u32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// hi - mp_hi (mod P) for hi, mp_hi < P: the tail of every Montgomery reduction here
ZK_HD uint32_t mont_tail(uint32_t hi, uint32_t mp_hi) {
#if ZK_MONT_VARIANT == 4
    return hi - mp_hi + p_if_less(hi, mp_hi);
#else
    uint32_t r;
    return __builtin_sub_overflow(hi, mp_hi, &r) ? r + P : r;
#endif
}
ZK_HD uint32_t mont_mul(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo + (lo << 30);                           // lo * P_INV mod 2^32
#if ZK_MONT_VARIANT == 0
    uint32_t mp_hi = (uint32_t)(((uint64_t)m * P) >> 32);   // low words of t and m*P are equal
    uint32_t r = hi - mp_hi;
    return hi < mp_hi ? r + P : r;
#else
    return mont_tail(hi, mul_hi_u32(m, P));
#endif
}

// Host-side helpers (setup, verifier, scalar API).  Plain residues in and out.
inline uint32_t mulmod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
inline uint32_t powmod(uint32_t a, uint64_t e) {    // field.rs:26-38
    uint32_t r = 1, b = a % P;
    while (e) {
        if (e & 1) r = mulmod(r, b);
        b = mulmod(b, b);
        e >>= 1;
    }
    return r;
}
inline uint32_t invmod(uint32_t a) { return powmod(a, P - 2); }            // field.rs:205-210
inline uint32_t to_mont(uint32_t a) { return (uint32_t)((((uint64_t)(a % P)) << 32) % P); }
// generator of the multiplicative subgroup of order 2^log_order: w^((P-1)/2^log_order)
// (prover.rs:48-49: exponents 3145728 and 393216 for orders 1024 and 8192)
inline uint32_t root_of_unity(uint32_t log_order) { return powmod(GEN_W, (uint64_t)(P - 1) >> log_order); }

}  // namespace zk
