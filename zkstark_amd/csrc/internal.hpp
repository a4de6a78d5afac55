// internal.hpp -- shared between the translation units of the host side (domain.hip, zkstark.hip,
// batch.hip).  Not part of the C ABI (include/zkstark_amd.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/zkstark_amd.h"
#include "field.hpp"
#include "kernels.hpp"
#include "transcript.hpp"

// Channel (channel.rs:6-37) behind the C ABI
struct zk_channel {
    zk::Channel ch;
};

namespace zk {
namespace impl {

// error reporting of the C ABI: every entry point returns fail(code, ...) and zk_last_error() hands the text out
int fail(int code, const char* fmt, ...);
const char* last_error();

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::zk::impl::fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ---- ABI rule of include/zkstark_amd.h ("ABI version and caller-allocated structs") --------------------------------------
// Size of each caller-allocated struct in ABI version 6, the first version that carries struct_size: the smallest size the
// library accepts.  A struct only ever grows at its end, so these numbers never change.
template <class T> struct AbiMin;
template <> struct AbiMin<zk_transcript_info> { static constexpr uint32_t v = 1244; static constexpr const char* name = "zk_transcript_info"; };
template <> struct AbiMin<zk_kernel_stat>     { static constexpr uint32_t v = 40;   static constexpr const char* name = "zk_kernel_stat"; };
template <> struct AbiMin<zk_shard_options>   { static constexpr uint32_t v = 48;   static constexpr const char* name = "zk_shard_options"; };
template <> struct AbiMin<zk_shard_stats>     { static constexpr uint32_t v = 112;  static constexpr const char* name = "zk_shard_stats"; };
template <> struct AbiMin<zk_shard_plan_info> { static constexpr uint32_t v = 200;  static constexpr const char* name = "zk_shard_plan_info"; };
template <> struct AbiMin<zk_chain_probe>     { static constexpr uint32_t v = 48;   static constexpr const char* name = "zk_chain_probe"; };
// Bytes of *p the library may touch (min of the caller's and the library's size), or 0 after fail(): struct_size is 0, below
// the version-6 size (a caller compiled against an older header, or one that never set it), or absurd.
template <class T> inline size_t abi_bytes(const T* p, const char* who) {
    static_assert(sizeof(T) >= AbiMin<T>::v && offsetof(T, struct_size) == 0, "a struct of the C ABI shrank or lost its size field");
    const uint32_t n = p->struct_size;
    if (n < AbiMin<T>::v || n > (1u << 16)) {
        fail(ZK_ERR_INVALID, "%s: %s.struct_size is %u; the caller sets it to sizeof(%s) before the call (ZK_STRUCT_INIT): at least %u bytes "
                             "(ABI version 6), %zu in this library (ABI version %u) -- was the caller compiled against another zkstark_amd.h?",
             who, AbiMin<T>::name, n, AbiMin<T>::name, AbiMin<T>::v, sizeof(T), (unsigned)ZK_ABI_VERSION);
        return 0;
    }
    return n < sizeof(T) ? n : sizeof(T);
}
// An output struct: the fields of `src` the caller's layout has room for; struct_size itself stays the caller's.
template <class T> inline int abi_put(T* out, const T& src, const char* who) {
    const size_t k = abi_bytes(out, who);
    if (!k) return ZK_ERR_INVALID;
    memcpy(reinterpret_cast<char*>(out) + sizeof(uint32_t), reinterpret_cast<const char*>(&src) + sizeof(uint32_t), k - sizeof(uint32_t));
    return ZK_OK;
}
// An input struct: a full-size copy with the fields the caller does not know at 0 (= default).
template <class T> inline int abi_get(const T* in, T* local, const char* who) {
    const size_t k = abi_bytes(in, who);
    if (!k) return ZK_ERR_INVALID;
    memset(static_cast<void*>(local), 0, sizeof(T));
    memcpy(static_cast<void*>(local), in, k);
    local->struct_size = (uint32_t)sizeof(T);
    return ZK_OK;
}

constexpr uint32_t kMaxQueries = 64;
constexpr uint32_t kHostTopSingle = 8;                       // hand-over depth one thread reduces alone (255 nodes, ~8 us)
constexpr uint32_t kMaxHostLog = 10;                         // host_top, host_tail, log_batch <= 10
constexpr size_t kMailValsOff = kMailDigests + ((size_t)8 << kMaxHostLog);   // after the digests of depth host_top
constexpr size_t kMailWords = kMailValsOff + ((size_t)2 << kMaxHostLog);     // values of the layer that feeds the host tail
constexpr uint32_t kMaxRadixLog = 8;

// Sizes a proof can have: shared by zk_ctx_create and zk_batch_create so that everything the provers accept is something
// the verifier (transcript.hpp: log_n >= 2) can check.  n = 8 is degenerate: g^4 = -1 cancels the leading terms of
// f(gx)^2 + f(x)^2, so deg c2 < n-1 and the generalised asserts of prover.rs:156/:169 would fail; n = 2 makes the boundary
// constraints collide (g^(n-2) = g^0).  Returns 0 or an error already recorded with fail().
inline int check_proof_size(const char* who, uint32_t log_n, uint32_t log_b, uint32_t log_extra = 0) {
    if (log_n < 2 || log_b < 1 || log_b > 5 || log_n + log_b + log_extra > 30)
        return fail(ZK_ERR_INVALID, "%s: need 2 <= log_n, 1 <= log_blowup <= 5, log_n + log_blowup%s <= 30 (got %u, %u)", who,
                    log_extra ? " + log_batch" : "", log_n, log_b);
    if (log_n == 3)
        return fail(ZK_ERR_INVALID, "%s: n = 8 is degenerate for the Fibonacci-square constraints (the degree asserts of prover.rs:156/:169 would fail)", who);
    return ZK_OK;
}

// ---- transform plan: the size-2^log_m transform as radix-2^bits[d] passes, d = 0 the slowest storage digit
struct Plan {
    uint32_t nd = 0;
    uint32_t bits[kMaxDigits] = {0};
};
Plan make_plan(uint32_t log_m);
uint32_t pick_logC(uint32_t log_total, uint32_t logR);

// ---- two-level power tables on the device
struct DevTable {
    uint32_t* lo = nullptr;
    uint32_t* hi = nullptr;
    uint32_t lo_bits = 0;
    PowTable view() const { return PowTable{lo, hi, lo_bits}; }
};
int build_table(uint32_t root, uint32_t log_order, DevTable* t);
void free_table(DevTable* t);

// Inverse transform, natural order in, digit-reversed out, optionally scaled by scale_mont.
int run_dif(const uint32_t* src, uint32_t* data, uint32_t log_m, const Plan& pl, PowTable tw_inv, uint32_t L, uint32_t scale_mont,
            hipStream_t s, Profiler* prof = nullptr, uint32_t batch = 1, size_t src_stride = 0, size_t data_stride = 0);
// Forward transform, digit-reversed in, natural out.
int run_dit(uint32_t* data, uint32_t log_m, const Plan& pl, PowTable tw, uint32_t L, hipStream_t s);

}  // namespace impl
}  // namespace zk

// ===========================================================================
// Coset evaluation domain: {shift * h^i, i < N}, h = root_of_unity(log_n + log_b)
// ===========================================================================
// Everything the kernels need that depends only on (log_n, log_b, shift): power tables,
// the 1/(x-1) table, the transform plan and Montgomery constants.  The reference's
// domain is shift = w = 5 (prover.rs:69).  A shard of a multi-GPU proof is the same
// structure with shift = w * h_global^rank and a smaller blow-up.
struct zk_dom {
    int device = 0;
    uint32_t log_n = 0, log_b = 0, L = 0;
    size_t n = 0, N = 0, B = 0;
    uint32_t shift = 0, g = 0, h = 0;
    zk::impl::DevTable H, Hinv, W;
    uint32_t* d_inv_xm1 = nullptr;   // null for fold-only domains
    zk::impl::Plan plan;
    uint32_t shift_mont = 0, gm1_mont = 0, gm2_mont = 0, gm3_mont = 0, ninv_mont = 0, inv2_mont = 0;
    // challenge-independent factors of the per-round constants, computed once (they sat on the commit -> challenge -> launch
    // path of every round: two modular exponentiations per fold, B inversions per composition)
    uint32_t fold_k[32] = {0};       // w^(-2^r) / 2, r < L                    (FoldArgs.c_mont = beta * fold_k[r])
    uint32_t inv_den[32] = {0};      // 1 / (x^n - 1) for the B values x^n takes (ComposeArgs.zz[r] = alpha2 * inv_den[r]); 0: not invertible
    size_t device_bytes = 0;
};

namespace zk {
namespace impl {

void dom_free(zk_dom* d);
// fold_only: tables for fri_fold only (no w-power table, no 1/(x-1) table)
int dom_make(int device, uint32_t log_n, uint32_t log_b, uint32_t shift, bool fold_only, hipStream_t stream, zk_dom** out);
// lagrange + solve over the coset (polynomial.rs:337, :49; prover.rs:60-70); batch > 1: proof b at
// d_trace + b*n, d_coef + b*2n, d_out + b*N
int dom_lde(const zk_dom* d, const uint32_t* d_trace, uint32_t* d_coef, uint32_t* d_out, hipStream_t s, Profiler* prof, uint32_t batch = 1);
int compose_args(const zk_dom* d, const uint32_t* d_f, uint32_t* d_cp, uint32_t first, uint32_t last, const uint32_t alpha_raw[3], ComposeArgs& a);
int dom_compose(const zk_dom* d, const uint32_t* d_f, uint32_t* d_cp, uint32_t first, uint32_t last, const uint32_t alpha_raw[3],
                hipStream_t s, Profiler* prof);
int fold_args(const zk_dom* d, const uint32_t* d_in, uint32_t* d_out, uint32_t log_m, uint32_t round, uint32_t beta_raw, FoldArgs& a);
int dom_fold(const zk_dom* d, const uint32_t* d_in, uint32_t* d_out, uint32_t log_m, uint32_t round, uint32_t beta_raw, hipStream_t s,
             Profiler* prof);

// Waits until *flag (host-mapped memory written by a commit launch on `stream`) equals `want`.  poll (optional) is
// called every few thousand spins; a non-zero return ends the wait with that code (the sharded prover looks for a
// peer that has left the proof: the launch may sit behind a collective that will never complete).
// timeout_s: how long to wait for the flag (the sharded prover passes its own bound: the launch may sit behind an exchange)
int wait_flag(const uint32_t* flag, uint32_t want, hipStream_t stream, int (*poll)(void*) = nullptr, void* poll_user = nullptr, double timeout_s = 30.0);
double now_us();
// zk_tail_open in two steps (zkstark.hip): begin enqueues at most one launch on the tail's stream, end waits and copies out
int tail_open_begin(zk_ctx* tail, size_t x);
int tail_open_end(zk_ctx* tail, uint32_t* vals_out, uint8_t* paths_out);
// merkle.rs:54-71: node indices of the authentication path of `leaf` in a tree of m leaves
void path_nodes(size_t m, size_t leaf, std::vector<size_t>& out);

}  // namespace impl
}  // namespace zk
