// kernels.hpp -- launch interface of the gfx950 kernels (kernels.hip).
// Everything here is stream-ordered and takes raw device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace zk {

// ---- optional per-kernel timing (HIP events on the launch stream) ---------------
// Kernel classes; a context times only the classes whose bit is set in Profiler::mask,
// so a benchmark can bracket just the dominant kernel inside its timed region.
enum KernelClass : int {
    K_NTT = 0,
    K_MERKLE_LEAF = 1,    // merkle_subtree_kernel<true>: leaf hashes + k inner levels
    K_MERKLE_INNER = 2,   // merkle_subtree_kernel<false>
    K_MERKLE_TOP = 3,     // merkle_wg_kernel (latency phase: workgroup-local levels)
    K_COMPOSE = 4,
    K_FOLD = 5,
    K_GATHER = 6,
    K_COUNT = 7
};

constexpr double kShaLeafOps = 1259.0;    // VALU instructions of one leaf hash (sha256.hpp, measured from the ISA)
constexpr double kShaInnerOps = 2293.0;
constexpr double kNttOpsPerElement = 57.0;  // VALU instructions per element of a radix-128 pass (SQ_INSTS_VALU, round 4: 904.5 per wave of 16 elements/lane; round 3: 78)
// field-native hash: one permutation per hash, in double precision since round 5 (fieldhash_f64.hpp).  VALU instructions by
// ISA loop count (tools/kernel_descriptors.py --loops: straight-line part + 8 trips of the full-round loops at 364 + 10 trips
// of the two-partial-round loop at 179): leaf / inner hash of the subtree kernel, 5 046 for the chain probe's loop body.
// (32-bit Montgomery form, rounds 1-4: 8 895 / 9 235 / 9 092.)
constexpr double kFieldLeafOps = 5015.0;
constexpr double kFieldInnerOps = 5072.0;

struct Profiler {
    uint32_t mask = 0;
    struct Rec { int cls; hipEvent_t a, b; double bytes, ops; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    bool on(int cls) const { return (mask >> cls) & 1u; }
};

// Brackets one kernel launch; bytes = algorithmic bytes of that launch, ops = its compulsory
// 32-bit VALU lane-ops (SHA-256 kernels; 0 elsewhere) -- DESIGN.md section 4.
struct ScopedKernelTimer {
    Profiler* p; int cls; double bytes, ops; hipStream_t s; hipEvent_t a = nullptr;
    ScopedKernelTimer(Profiler* p_, int cls_, double bytes_, hipStream_t s_, double ops_ = 0.0) : p(p_), cls(cls_), bytes(bytes_), ops(ops_), s(s_) {
        if (p && p->on(cls)) { a = p->get(); (void)hipEventRecord(a, s); }
    }
    ~ScopedKernelTimer() {
        if (a) { hipEvent_t b = p->get(); (void)hipEventRecord(b, s); p->recs.push_back({cls, a, b, bytes, ops}); }
    }
};

// Two-level table of powers of a fixed root, entries in Montgomery form:
//   root^e = hi[e >> lo_bits] * lo[e & ((1 << lo_bits) - 1)]
// Both halves are a few KiB, so twiddles are served from L1/L2 instead of
// streaming a full N-entry table from HBM.
struct PowTable {
    const uint32_t* lo;
    const uint32_t* hi;
    uint32_t lo_bits;
};

constexpr int kMaxDigits = 6;

// One LDS-tiled radix-2^logR pass over an array viewed as [A][R][S] (S fastest).
enum NttMode : uint32_t {
    NTT_DIF = 0,        // natural in -> digit-reversed out; post-twiddle (inverse-transform passes)
    NTT_DIT = 1,        // digit-reversed in -> natural out; pre-twiddle (forward passes)
    NTT_DIT_LDE = 2,    // first forward pass of the LDE: reads n prepared coefficients, writes N = n*B values
};

// Workgroup tile of an NTT pass, in words: 4096 for passes over more than 2^20 words, 2048 below.  A pass is a chain of
// latencies per workgroup (table look-up, loads, two register DFTs, stores) and, in this field, VALU-heavy (P > 2^31: every
// add / sub carries its correction), so it wants many light workgroups per compute unit: eight 4096-word workgroups
// (16.5 KiB of LDS, <= 64 VGPRs) hold 8 waves per SIMD, where the 8192-word tile of rounds 1-3 held 4.  Measured on the
// 2^24-word passes of a 2^24 proof (rocprofv3, profiles/r04_ab_ntt_tile.txt): 31.1 us against 32.7 us in place, 37.3 against
// 40.1 us for the first LDE pass; a radix-128 row segment is still a whole 128-byte line (HBM traffic 1.00 x algorithmic).
// The 2048-word tile has 64-byte segments at radix 128 (2 x the fetched bytes at 2^21, round 3), fine for the latency-bound
// passes over <= 2^20 words whose data sits in L2.  Build-time constant (ZK_BUILD_DEFS="-DZK_NTT_SMALL_MAX_LOG=..." to A/B it).
#ifndef ZK_NTT_SMALL_MAX_LOG
#define ZK_NTT_SMALL_MAX_LOG 20
#endif
// (Radix-512 passes -- 2^17 words in two passes instead of three -- were built and measured in round 5: parity-green and SLOWER,
// configs[1] 203 against 190 us, profiles/r05_ab_ntt_radix512.txt.  The branch was removed in round 6; the record stays.)
constexpr uint32_t kMidTileLog = 12, kSmallTileLog = 11;
constexpr uint32_t kNttSmallTileMaxLog = ZK_NTT_SMALL_MAX_LOG;
inline uint32_t ntt_tile_log(uint32_t log_total) { return log_total <= kNttSmallTileMaxLog ? kSmallTileLog : kMidTileLog; }
// ... of one PASS: a radix-512 pass (transforms of 2^17 / 2^18 words) keeps >= 8 columns per tile, i.e. the 4096-word tile
inline uint32_t ntt_pass_tile_log(uint32_t log_total, uint32_t logR) {
    const uint32_t t = ntt_tile_log(log_total);
    return logR + 3 > t ? (logR + 3 < kMidTileLog ? logR + 3 : kMidTileLog) : t;
}

struct NttPassArgs {
    const uint32_t* src;
    uint32_t* dst;
    uint32_t log_total;   // log2 of the number of elements of dst
    uint32_t logR, logS;  // this pass: radix and inner stride
    uint32_t logC;        // columns per workgroup tile (tile = R * C elements)
    uint32_t tile_log;    // log2 of the tile (ntt_tile_log of the pass; the register-radix kernel needs logC = tile_log - logR)
    uint32_t L;           // the table root has order 2^L
    PowTable tw;          // h (forward) or h^-1 (inverse)
    uint32_t scale_mont;  // NTT_DIF only: multiply outputs by this Montgomery constant when S == 1 (n^-1); 0 = none
    // batch of independent transforms (zk_batch_*): grid.y = batch, transform b at src + b*src_stride / dst + b*dst_stride
    uint32_t batch;       // 0 or 1 = a single transform
    size_t src_stride, dst_stride;
    // NTT_DIT_LDE with prep != 0: src is the raw DIF output U and the coefficient preparation of
    // coef_prepare_kernel (CoefPrepArgs below) happens inside the pass (once per coefficient, staged in LDS, when the
    // blow-up is >= 2; at the load for B = 1); register-radix kernel only (ntt_fast_ok)
    uint32_t prep;
    uint32_t prep_log_n, prep_log_b, prep_ninv_mont, prep_nd;
    uint32_t prep_bits[kMaxDigits];
    PowTable prep_wtab;   // powers of the coset shift
};
bool ntt_fast_ok(const NttPassArgs& a, NttMode mode);

struct CoefPrepArgs {
    uint32_t log_n, log_b;
    PowTable tw;          // h (order 2^(log_n+log_b)): g^j = h^(j << log_b)
    PowTable wtab;        // powers of the coset shift
    uint32_t ninv_mont;   // n^-1 in Montgomery form
    uint32_t nd;          // storage digits of the coefficient array, slowest first
    uint32_t dig_bits[kMaxDigits];
};
// batch > 1: transform b reads U + b*u_stride and writes out + b*out_stride
hipError_t launch_coef_prepare(const uint32_t* U, uint32_t* out, const CoefPrepArgs& a, hipStream_t s, Profiler* prof = nullptr,
                               uint32_t batch = 1, size_t u_stride = 0, size_t out_stride = 0);

hipError_t launch_ntt_pass(const NttPassArgs& a, NttMode mode, hipStream_t s, Profiler* prof = nullptr);
// out[pos] = in[true_index(pos)] for the mixed-radix digit reversal (standalone NTT API only)
hipError_t launch_digit_reverse(const uint32_t* in, uint32_t* out, uint32_t log_m, uint32_t nd,
                                const uint32_t* dig_bits, int to_natural, hipStream_t s);

// self-checks: coef = unscaled DIF output of a layer; out[0] += non-zero coefficients of true index >= bound, out[1] = coefficient bound - 1
hipError_t launch_degree_check(const uint32_t* coef, uint32_t log_m, uint32_t nd, const uint32_t* dig_bits, uint32_t bound, uint32_t* out, hipStream_t s);

// inv_xm1[i] = 1 / (shift h^i - 1) in Montgomery form, i < N (domain setup)
hipError_t launch_build_inv_xm1(uint32_t* out, uint32_t logN, PowTable htab, uint32_t shift_mont, hipStream_t s);
// the same table over a RANGE of positions: out[t] = 1 / (shift h^((e0 + t) mod 2^log_order) - 1), t < count (the block of a
// sharded proof: csrc/shard.hip, ComposeBlockArgs)
hipError_t launch_build_inv_xm1_range(uint32_t* out, size_t count, uint32_t e0, uint32_t log_order, PowTable htab, uint32_t shift_mont, hipStream_t s);
// Sharded proof, rank r of G = 2^lg: out[q * h + u] = loc[((q + 1) * per + u) mod (G * per)], q < G, u < h -- the values of
// this rank's cyclic shard that lie in the first G * h positions after the block of rank q (the halo of ComposeBlockArgs)
hipError_t launch_halo_pack(const uint32_t* loc, uint32_t* out, uint32_t log_per, uint32_t lg, uint32_t h, hipStream_t s);
hipError_t launch_interleave(const uint32_t* in, uint32_t* out, uint32_t log_parts, uint32_t log_cnt, hipStream_t s);

// Known-pattern exchange of the sharded prover's self-test (shard.hip): word j of the piece rank `from` sends to rank `to`.
inline __host__ __device__ uint32_t shard_pattern(uint32_t from, uint32_t to, uint32_t j) {
    uint32_t x = ((from + 1u) * 0x9E3779B1u) ^ ((to + 1u) * 0x85EBCA6Bu) ^ (j * 0xC2B2AE35u + 0x27D4EB2Fu);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}
// dst[p << log_per | j] = shard_pattern(rank, p, j), p < 2^log_parts
hipError_t launch_pattern_fill(uint32_t* dst, uint32_t rank, uint32_t log_per, uint32_t log_parts, hipStream_t s);
// Receive buffer of an all-to-all of those pieces: plain (log_chunks = 0): src[q << log_per | j] from rank q; in 2^log_chunks
// chunks of cc = 2^(log_per - log_chunks) words: src[(c * parts + q) * cc + u] = word c * cc + u from rank q.
// out[0] += words that are not shard_pattern(q, rank, j); out[1] = min over their indices (preset to 0xffffffff).
hipError_t launch_pattern_check(const uint32_t* src, uint32_t rank, uint32_t log_per, uint32_t log_parts, uint32_t log_chunks, uint32_t* out, hipStream_t s);

struct ComposeArgs {
    const uint32_t* f;        // N canonical evaluations of the trace polynomial
    const uint32_t* inv_xm1;  // N, Montgomery
    uint32_t* cp;             // N out
    uint32_t logN, log_b;
    PowTable htab;
    uint32_t w_mont;          // w
    uint32_t gm1_mont, gm2_mont, gm3_mont;  // g^-1, g^-2, g^-3
    uint32_t first;           // a[0] canonical
    uint32_t last;            // a[n-2] canonical
    uint32_t alpha0_mont;     // alpha0
    uint32_t alpha1g2_mont;   // alpha1 * g^2
    uint32_t zz[32];          // B entries: alpha2 / (x^n - 1) * R^2  (per i mod B)
};
hipError_t launch_compose(const ComposeArgs& a, hipStream_t s, Profiler* prof = nullptr);

// cp over ONE RANK'S BLOCK of a sharded proof, computed from the block of f this rank received for the commitment of f
// (DESIGN.md section 6): position t < 2^log_m of the block is x = w h^(e0 + t); f at t, t + B, t + 2B is read through the
// all-to-all order of the receive buffer (leaf t of chunk c = t >> (lg + log_cnt): piece (t mod G) of that chunk, word
// (t mod 2^(lg + log_cnt)) >> lg), the 2B positions after the block from `halo` (an all-gather of what launch_halo_pack
// wrote: value v at halo[(v mod G) * halo_stride + (v >> lg)]).  a.f = the receive buffer, a.inv_xm1 = the range table
// (2^log_m + 2B entries from e0), a.htab / a.w_mont / a.log_b / a.zz = the GLOBAL domain's; a.cp is not written.
struct ComposeBlockArgs {
    ComposeArgs a;
    const uint32_t* halo;
    uint32_t lg, log_cnt, log_m, halo_stride, e0;
};

struct FoldArgs {
    const uint32_t* in;   // m values
    uint32_t* out;        // m/2 values
    uint32_t log_m;       // log2 m
    uint32_t round;       // r: x_i = (w h^i)^(2^r)
    PowTable hinv;        // h^-1 table, order 2^L
    uint32_t L;
    uint32_t inv2_mont;   // 1/2
    uint32_t c_mont;      // beta * w^(-2^r) / 2
    // Early launch (zk_ctx_set_early_launch): when non-null the launch was enqueued BEFORE its challenge existed, behind a
    // command-processor wait; c_mont is then read from here (pinned host memory, an ordinary cached load like a kernel argument:
    // the scalar cache is invalidated at every dispatch) when a workgroup starts -- kernels.hip: FoldSrc::prepare
    const uint32_t* dyn;
};
hipError_t launch_fri_fold(const FoldArgs& a, hipStream_t s, Profiler* prof = nullptr);

// Merkle tree over m = 2^log_m u32 leaves.  nodes: (2m-1) * 8 words, heap order
// (merkle.rs:14-51), each node the eight SHA-256 state words.
// mail (optional): where the result of the build is posted for the host, see MailArgs.
// hash: 0 = SHA-256 (the reference, merkle.rs:1-2), 1 = field-native hash (fieldhash.hpp, configs[4]).
// The commit -> challenge hand-off.  mailbox is host-mapped memory: word 0 <- seq (last), words
// kMailDigests.. <- the 2^top digests of depth `top` (top = 0: the root; the build stops at that depth and
// the host finishes the tree, host_sha.hpp), then, if dump_src is set, 2^dump_log leaf values.  The
// workgroup that finishes last copies everything in one burst of wide stores (counter: one zeroed device word).
constexpr uint32_t kMailDigests = 16;
struct MailArgs {
    uint32_t* mailbox = nullptr;
    uint32_t seq = 0, top = 0;
    uint32_t* counter = nullptr;
    const uint32_t* dump_src = nullptr;   // device pointer to the layer the tree is built over
    uint32_t dump_log = 0;
    uint32_t vals_off = 0;                // word offset of the value area inside the mailbox
};
hipError_t launch_merkle_build(const uint32_t* vals, uint32_t log_m, uint32_t* nodes, hipStream_t s, Profiler* prof = nullptr,
                               const MailArgs& mail = MailArgs{}, int hash = 0);
// level size (log2) at which a build switches from throughput launches to the workgroup-local latency phase;
// 0 restores the build's default (zk_dev_set_merkle_latency_log); false: out of range (12 .. 24)
bool set_merkle_latency_log(uint32_t v);
struct ScatterSeg { uint64_t src, dst; uint32_t words, kind; };   // kind 0: into the trees array, 1: into the layers array
hipError_t launch_scatter(const uint32_t* stage, const ScatterSeg* segs, uint32_t count, double words, uint32_t* trees, uint32_t* layers,
                          hipStream_t s, Profiler* prof = nullptr);

hipError_t launch_merkle_build_interleaved(const uint32_t* recv, uint32_t log_parts, uint32_t log_cnt, uint32_t* nodes, hipStream_t s,
                                           Profiler* prof = nullptr, int hash = 0, const MailArgs& mail = MailArgs{});
hipError_t launch_merkle_build_chunk(const uint32_t* recv, uint32_t log_parts, uint32_t log_cnt, uint32_t* nodes, uint32_t log_m,
                                     uint32_t chunk, hipStream_t s, Profiler* prof = nullptr, int hash = 0);
// depth at which the chunk builds of a 2^log_m-leaf tree stop and launch_merkle_finish takes over
uint32_t merkle_finish_start_depth(uint32_t log_m, uint32_t log_chunks);
hipError_t launch_merkle_finish(uint32_t* nodes, uint32_t log_m, uint32_t log_chunks, hipStream_t s, Profiler* prof = nullptr, int hash = 0,
                                const MailArgs& mail = MailArgs{});
// Fused producer + commitment: the layer is computed, stored and leaf-hashed in one pass.
hipError_t launch_fold_merkle(const FoldArgs& a, uint32_t* nodes, hipStream_t s, Profiler* prof = nullptr,
                              const MailArgs& mail = MailArgs{}, int hash = 0);
hipError_t launch_compose_merkle(const ComposeArgs& a, uint32_t* nodes, hipStream_t s, Profiler* prof = nullptr,
                                 const MailArgs& mail = MailArgs{}, int hash = 0);
// the same for a block of a sharded proof (ComposeBlockArgs): the subtree over this rank's 2^log_m leaves of cp
hipError_t launch_compose_block_merkle(const ComposeBlockArgs& b, uint32_t* nodes, hipStream_t s, Profiler* prof = nullptr,
                                       const MailArgs& mail = MailArgs{}, int hash = 0);

// ---- batched proving: 2^log_batch proofs of one size in lockstep ------------------
// Layer l of the batch is stored proof-major, [batch][m_l]; the 2^log_batch trees over it are the bottom of
// ONE tree over batch*m_l leaves, whose nodes of depth log_batch are the per-proof roots (MailArgs.top).
struct BatchChal {              // per proof, derived from its own transcript (device array)
    uint32_t first, last;       // a[0], a[n-2] canonical
    uint32_t alpha0_mont, alpha1g2_mont, alpha2_mont;
    uint32_t c_mont;            // this round's beta * w^(-2^r) / 2
    uint32_t pad[2];
};
struct ComposeBatchArgs { ComposeArgs a; const BatchChal* chal; };   // a.f, a.cp: [batch][N]; a.zz: R^2 / (x^n - 1), no alpha2
struct FoldBatchArgs { FoldArgs a; const BatchChal* chal; };         // a.in: [batch][m], a.out: [batch][m/2]
hipError_t launch_compose_merkle_batch(const ComposeBatchArgs& a, uint32_t log_batch, uint32_t* nodes, hipStream_t s, Profiler* prof,
                                       const MailArgs& mail, int hash);
hipError_t launch_fold_merkle_batch(const FoldBatchArgs& a, uint32_t log_batch, uint32_t* nodes, hipStream_t s, Profiler* prof,
                                    const MailArgs& mail, int hash);

// batch traces of prover.rs:32-39, one lane per trace: out[t*stride + i], i < count (stride 0 = count)
hipError_t launch_trace_fibsq_batch(const uint32_t* a0, const uint32_t* a1, uint32_t batch, uint32_t count, uint32_t* out, hipStream_t s,
                                    uint32_t stride = 0);

// measurement only: blocks x 256 lanes, each a chain of `hashes` inner hashes; rec (optional): per wave {shader clocks, 100 MHz ticks}
hipError_t launch_hash_chain_probe(int hash, uint32_t blocks, uint32_t* out, uint32_t seed, uint32_t hashes, unsigned long long* rec, hipStream_t s);
// d_res[0] <- number of threads whose hashes differ between the forms of the field hash, d_res[1] <- the first such thread (d_res preset to {0, ~0})
hipError_t launch_fieldhash_forms(uint32_t blocks, uint32_t seed, uint32_t* d_res, hipStream_t s);

// out[i*words .. ] = src[offsets[i] .. +words]   (decommit gather)
hipError_t launch_gather(const uint32_t* src, const uint64_t* offsets, uint32_t count, uint32_t words,
                         uint32_t* out, hipStream_t s, Profiler* prof = nullptr);
// Decommitment read-out without copy commands (the one-call prover): `nv` values of `layers` and `ndg` 8-word nodes of
// `trees`, word offsets in `items` (values first), which lies in host-mapped memory like `out`: out[8 j ..] = node j,
// out[8 ndg + i] = value i.  The workgroup that finishes last raises mailbox[0] = seq behind the results (counter: one
// zeroed device word, as in MailArgs); the host polls instead of synchronising the stream.
hipError_t launch_fetch(const uint32_t* layers, const uint32_t* trees, const uint64_t* items, uint32_t nv, uint32_t ndg, uint32_t* out,
                        uint32_t* mailbox, uint32_t seq, uint32_t* counter, hipStream_t s, Profiler* prof = nullptr);

}  // namespace zk
