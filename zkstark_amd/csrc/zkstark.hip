// zkstark.hip -- context, pipeline stages, host prover and the C ABI (include/zkstark_amd.h).
//
// Host orchestration of prover.rs:9-293 re-written around device-resident data:
// everything between "trace on device" and "proof bytes on host" stays in HBM;
// per commitment the digests of one tree level (or the root) come back and one 4-byte challenge goes out.
// Domains and transforms live in domain.hip, the batched prover in batch.hip (shared: internal.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "../../include/zkstark_amd.h"
#include "field.hpp"
#include "kernels.hpp"
#include "sha256.hpp"
#include "transcript.hpp"
#include "host_sha.hpp"
#include "internal.hpp"
#include "pool.hpp"

using namespace zk;
using namespace zk::impl;

// ===========================================================================
// Context
// ===========================================================================
struct zk_ctx {
    int device = 0;
    uint32_t log_n = 0, log_b = 0, L = 0;
    size_t n = 0, N = 0, B = 0;
    uint32_t R = 0;   // FRI rounds = log_n (prover.rs:198)
    hipStream_t stream = nullptr;
    zk_dom* dom = nullptr;
    uint32_t* d_trace = nullptr;    // n words: a[0..n-2], 0   (stays resident across proofs)
    uint32_t* d_coef = nullptr;     // 2n words: iNTT output, then prepared coefficients (digit-reversed order)
    uint32_t* d_layers = nullptr;   // layer 0 (N) | layer 1 (N) | layer 2 (N/2) | ... | layer R+1 (B)
    std::vector<size_t> layer_off, layer_len;
    uint32_t* d_trees = nullptr;    // tree t over layer t, (2m-1)*8 words each
    std::vector<size_t> tree_off;
    uint64_t* d_gather_off = nullptr;
    uint32_t* d_gather_out = nullptr;
    uint64_t* h_gather_off = nullptr;   // pinned, device-mapped (dm_*: the device's view): the one-call prover's decommitment
    uint32_t* h_gather_out = nullptr;   // launch reads its work list and writes its results there (launch_fetch)
    uint64_t* dm_gather_off = nullptr;
    uint32_t* dm_gather_out = nullptr;
    // what this thread built during the current proof and still holds in the staging buffer: nodes [0, host_node_cnt) of
    // a tree (its top, or all of it), a whole small layer.  The decommitment reads those from there.
    const uint32_t* host_nodes[40] = {};
    size_t host_node_cnt[40] = {};
    const uint32_t* host_vals[40] = {};
    std::vector<const uint32_t*> fetch_vals, fetch_nodes;
    std::vector<uint32_t> fetch_vdev;
    std::vector<uint64_t> fetch_ditems;
    std::vector<uint8_t> commit_buf;
    // zk_merkle_commit leaves the device copy of the host-built tree top PENDING (the segments stay in the staging buffer):
    // it is ordered on the stream by the first call that reads trees or layers from the device, dropped by a later
    // commitment of the same tree or a proof (both rebuild those nodes), flushed before any other reuse of the staging buffer
    bool pending_top = false;
    uint32_t pending_tree = 0;
    size_t open_nv = 0, open_ndg = 0, open_vi = 0;   // the decommitment being assembled (open_begin ..)
    bool open_pending = false;
    uint32_t* h_small = nullptr;        // pinned: root words + last layer
    uint32_t* h_mailbox = nullptr;      // pinned, host-coherent, device-mapped: layout in kernels.hpp (MailArgs)
    uint32_t* d_mailbox = nullptr;      // the device view of h_mailbox
    uint32_t* d_counter = nullptr;      // one zeroed word: workgroups of a commit launch that are done
    // A commit launch that was waited for in vain (wait_flag: device error, time-out) may have left the last-arriver counters of
    // merkle_wg_kernel non-zero: only the workgroup that arrives last resets them.  Every later tree of this context would then
    // pick the wrong "last" workgroup and post wrong digests silently, so the next launch zeroes them first.
    bool counters_dirty = false;
    uint32_t mail_seq = 0;
    uint32_t tree_seq[40] = {};         // the mailbox sequence number the commit launch of tree t posts with (mail_of)
    // Early launch (zk_ctx_set_early_launch; docs/LOG.md round 6): the fold + commit launches of the NEXT FRI round are enqueued
    // before the current commitment's digests are waited for, behind a command-processor wait on a host word (hipStreamWaitValue32);
    // when the challenge is known the host stores the round's constant into a pinned parameter slot and releases the word.
    bool early = false, early_ok = false, gate_pending = false;
    uint32_t* h_gate = nullptr;         // pinned, coherent, device-mapped: word 0 = the gate
    uint32_t* d_gate = nullptr;
    uint32_t* h_dyn = nullptr;          // pinned: two parameter slots of 16 words (a slot is reused every second round)
    uint32_t* d_dyn = nullptr;
    uint32_t gate_seq = 0;
    // Host-finished pieces of the one-call prover (host_sha.hpp): the top `host_top` levels of every tree
    // with more than 2^host_top leaves, and whole FRI layers of <= 2^host_tail values (fold + tree).
    uint32_t host_top = 0, host_tail = 0;
    // hand-over depths above kHostTopSingle are reduced by a small team: 2^(host_top - 8) sub-trees of 256 digests, one
    // per thread (workers spin between the commitments of a proof), then the calling thread hashes the levels above
    Pool* pool = nullptr;
    uint32_t* h_stage = nullptr;        // pinned, device-mapped: host-built nodes / values waiting for scatter_kernel
    uint32_t* d_stage = nullptr;
    size_t stage_words = 0, stage_used = 0;
    ScatterSeg* h_segs = nullptr;       // at the end of h_stage
    ScatterSeg* d_segs = nullptr;
    uint32_t n_segs = 0;
    double seg_words = 0;
    std::vector<uint32_t> tail_vals;    // current host-side FRI layer (valid when tail_log != 0)
    uint32_t tail_log = 0;              // log2 size of tail_vals
    bool tail_have = false;             // false: the current FRI layer lives on the device only
    double t_wait = 0, t_host_hash = 0, t_launch = 0;   // ZK_HOST_TIMING: where the host thread spends a proof
    bool tail = false;                  // FRI-tail context (zk_tail_*): no trace / LDE / composition
    uint32_t hinv_host = 0;             // 1 / h (the host-side FRI rounds step through its powers)
    uint32_t queries = 1;               // decommitment queries (1 = the reference, prover.rs:263)
    int hash = 0;                       // Merkle hash: 0 = SHA-256 (reference), 1 = field-native (configs[4])
    // opt-in reference self-checks (zk_ctx_set_checks; prover.rs:64-66, :148-159, :169, :228-251)
    bool checks = false;
    uint32_t* d_check = nullptr;        // N words of scratch + 2 result words
    DevTable ones;                      // powers of 1: the coefficient preparation without the coset shift
    size_t gather_cap = 0;
    size_t device_bytes = 0;
    double setup_ms = 0;
    // per-proof state
    bool have_trace = false, have_lde = false;
    uint32_t first = 0, last = 0;
    zk_transcript_info info{};
    Profiler prof;
    zk_kernel_stat kstat[K_COUNT] = {};
};

namespace {

template <typename T>
int dmalloc(zk_ctx* c, T** p, size_t bytes) {
    hipError_t e = hipMalloc((void**)p, bytes ? bytes : 4);
    if (e != hipSuccess) return fail(ZK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    c->device_bytes += bytes;
    return ZK_OK;
}

size_t layer_size(const zk_ctx* c, uint32_t layer) { return layer == 0 ? c->N : (c->N >> (layer - 1)); }
uint32_t layer_log(const zk_ctx* c, uint32_t layer) { return layer == 0 ? c->L : c->L - (layer - 1); }

Profiler* prof_of(zk_ctx* c) { return c->prof.mask ? &c->prof : nullptr; }

// Resolves pending event pairs into the per-class accumulators (synchronises on them).
void collect_kernel_stats(zk_ctx* c) {
    for (auto& r : c->prof.recs) {
        float ms = 0;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        c->kstat[r.cls].launches += 1;
        c->kstat[r.cls].ms += ms;
        c->kstat[r.cls].bytes += r.bytes;
        c->kstat[r.cls].ops += r.ops;
        c->prof.pool.push_back(r.a);
        c->prof.pool.push_back(r.b);
    }
    c->prof.recs.clear();
}

int do_lde(zk_ctx* c) {
    int rc = dom_lde(c->dom, c->d_trace, c->d_coef, c->d_layers + c->layer_off[0], c->stream, prof_of(c));
    if (!rc) c->have_lde = true;
    return rc;
}

// Sub-tree of the handed-over digests one host thread reduces alone: 2^8 digests (255 nodes, ~5 us).  Build-time constant
// (ZK_BUILD_DEFS="-DZK_HOST_SUB_LOG=7"; profiles/r03_ab_host_team.txt swept it).
#ifndef ZK_HOST_SUB_LOG
#define ZK_HOST_SUB_LOG 8
#endif
static_assert(ZK_HOST_SUB_LOG >= 4 && ZK_HOST_SUB_LOG <= (int)kMaxHostLog, "host sub-tree size");
constexpr uint32_t host_sub_log() { return ZK_HOST_SUB_LOG; }
// How much of tree `tree` the host finishes: the top `host_top` levels of SHA-256 trees larger than that.
uint32_t top_of(const zk_ctx* c, uint32_t tree) {
    if (c->hash != 0 || !c->host_top) return 0;
    const uint32_t lg = layer_log(c, tree);
    return lg > c->host_top ? c->host_top : lg - 1;                 // a smaller tree hands over one level below its leaves
}
// What the commit launch of `tree` posts to the host.  host = true (one-call flows): the digests of depth
// host_top instead of the root, and for the layer with 2^(host_tail+1) values the values too -- the host
// folds on from there.
MailArgs mail_of(zk_ctx* c, uint32_t tree, bool host, bool feed_tail = true) {
    MailArgs m;
    if (c->counters_dirty) {                              // stream-ordered behind whatever is left of the failed launch
        (void)hipMemsetAsync(c->d_counter, 0, 64, c->stream);
        c->counters_dirty = false;
    }
    m.mailbox = c->d_mailbox;
    m.seq = ++c->mail_seq;
    c->tree_seq[tree] = m.seq;
    m.counter = c->d_counter;
    m.top = host ? top_of(c, tree) : 0;
    if (feed_tail && m.top && c->host_tail && tree >= 1 && layer_log(c, tree) == c->host_tail + 1) {
        m.dump_src = c->d_layers + c->layer_off[tree];
        m.dump_log = c->host_tail + 1;
        m.vals_off = (uint32_t)kMailValsOff;
        c->tail_log = m.dump_log;                         // tail_vals is filled by read_commit
        c->tail_have = true;
    }
    return m;
}

// Builds tree `layer`; the launch that reaches the hand-over depth posts its digests to the host mailbox.
// host = false: the whole tree on the device; feed_tail = false: a stand-alone commitment (stage-by-stage API).
int do_merkle(zk_ctx* c, uint32_t layer, bool host = false, bool feed_tail = true) {
    HIPCHK(launch_merkle_build(c->d_layers + c->layer_off[layer], layer_log(c, layer), c->d_trees + c->tree_off[layer], c->stream,
                               prof_of(c), mail_of(c, layer, host, feed_tail), c->hash));
    return ZK_OK;
}

// Fused stages of the one-call prover: the producer of a layer runs inside the leaf hashing of its tree.
int do_compose_commit(zk_ctx* c, const uint32_t alpha_raw[3]) {
    ComposeArgs a;
    int rc = compose_args(c->dom, c->d_layers + c->layer_off[0], c->d_layers + c->layer_off[1], c->first, c->last, alpha_raw, a);
    if (rc) return rc;
    HIPCHK(launch_compose_merkle(a, c->d_trees + c->tree_off[1], c->stream, prof_of(c), mail_of(c, 1, true), c->hash));
    return ZK_OK;
}
int do_fold_commit(zk_ctx* c, uint32_t round, uint32_t beta_raw) {
    FoldArgs a;
    int rc = fold_args(c->dom, c->d_layers + c->layer_off[1 + round], c->d_layers + c->layer_off[2 + round], c->L - round, round, beta_raw, a);
    if (rc) return rc;
    HIPCHK(launch_fold_merkle(a, c->d_trees + c->tree_off[2 + round], c->stream, prof_of(c), mail_of(c, 2 + round, true), c->hash));
    return ZK_OK;
}

int wait_mail(zk_ctx* c, uint32_t seq) {
    double t0 = now_us();
    int rc = wait_flag(c->h_mailbox, seq, c->stream);
    c->t_wait += now_us() - t0;
    if (rc) c->counters_dirty = true;
    return rc;
}

// Root posted by a whole-tree build.
int read_root(zk_ctx* c, uint32_t tree, uint8_t out[32]) {
    int rc = wait_mail(c, c->tree_seq[tree]);
    if (rc) return rc;
    digest_words_to_bytes(c->h_mailbox + kMailDigests, out);
    return ZK_OK;
}

uint32_t* stage_alloc(zk_ctx* c, size_t words) {
    if (c->stage_used + words > c->stage_words) return nullptr;
    uint32_t* p = c->h_stage + c->stage_used;
    c->stage_used += (words + 7) & ~(size_t)7;
    return p;
}
int stage_seg(zk_ctx* c, const uint32_t* src, size_t dst, size_t words, uint32_t kind) {
    if (c->n_segs >= 4 * 34) return fail(ZK_ERR_STATE, "scatter segment table full");
    c->h_segs[c->n_segs++] = ScatterSeg{(uint64_t)(src - c->h_stage), (uint64_t)dst, (uint32_t)words, kind};
    c->seg_words += (double)words;
    return ZK_OK;
}

// Result of the commit launch of `tree`: either the root itself, or the 2^host_top digests of depth
// host_top, which this thread reduces to the root (merkle.rs:40-46) and keeps for the scatter.
int read_commit(zk_ctx* c, uint32_t tree, uint8_t root[32]) {
    const uint32_t H = top_of(c, tree);
    if (!H) return read_root(c, tree, root);
    const size_t cnt = (size_t)1 << H;
    uint32_t* nodes = stage_alloc(c, (2 * cnt - 1) * 8);
    if (!nodes) return fail(ZK_ERR_STATE, "host staging buffer exhausted");
    int rc = wait_mail(c, c->tree_seq[tree]);
    if (rc) return rc;
    double t0 = now_us();
    memcpy(nodes + 8 * (cnt - 1), c->h_mailbox + kMailDigests, cnt * 32);
    const uint32_t sub_log = host_sub_log();
    if (H > sub_log && c->pool) {
        const uint32_t top = H - sub_log;                      // 2^top sub-trees of 2^sub_log digests (2^8: ~8 us each), in parallel
        c->pool->run((size_t)1 << top, 1, [&](size_t sub) { host_sha_reduce_sub(nodes, H, top, sub); });
        host_sha_reduce(nodes, top);
    } else {
        host_sha_reduce(nodes, H);
    }
    digest_words_to_bytes(nodes, root);
    c->t_host_hash += now_us() - t0;
    if (c->tail_have && c->tail_log == layer_log(c, tree) && tree >= 1) {     // this launch dumped its leaves (mail_of)
        c->tail_vals.assign(c->h_mailbox + kMailValsOff, c->h_mailbox + kMailValsOff + ((size_t)1 << c->tail_log));
    }
    c->host_nodes[tree] = nodes;
    c->host_node_cnt[tree] = 2 * cnt - 1;                 // the posted digests of depth H included
    return stage_seg(c, nodes, c->tree_off[tree], (cnt - 1) * 8, 0);
}

// FRI round `round` on the host: fold tail_vals with beta (the formula of fold_at, kernels.hip), hash the
// new layer's tree (merkle.rs:14-51); values and nodes are staged for the device copy.
int host_fold_commit(zk_ctx* c, uint32_t round, uint32_t beta_raw, uint8_t root[32]) {
    const uint32_t log_out = c->L - round - 1;
    const size_t half = (size_t)1 << log_out;
    const double t_begin = now_us();
    uint32_t* vals = stage_alloc(c, half);
    uint32_t* nodes = stage_alloc(c, (2 * half - 1) * 8);
    if (!vals || !nodes) return fail(ZK_ERR_STATE, "host staging buffer exhausted");
    const zk_dom* d = c->dom;
    const uint32_t cc_m = to_mont(mulmod(beta_raw % P, d->fold_k[round]));                  // beta / (2 w^(2^r))
    const uint32_t step_m = to_mont(powmod(c->hinv_host, (uint64_t)1 << round));            // h^(-2^r)
    const uint32_t* in = c->tail_vals.data();
    uint32_t xinv_m = to_mont(1);                         // Montgomery form, like the device tables: canonical * Montgomery = canonical
    for (size_t i = 0; i < half; ++i) {
        uint32_t u = in[i], v = in[i + half];
        vals[i] = add(mont_mul(add(u, v), d->inv2_mont), mont_mul(mont_mul(sub(u, v), xinv_m), cc_m));
        xinv_m = mont_mul(xinv_m, step_m);
    }
    host_sha_leaves(vals, half, nodes + 8 * (half - 1));
    host_sha_reduce(nodes, log_out);
    digest_words_to_bytes(nodes, root);
    c->tail_vals.assign(vals, vals + half);
    c->tail_log = log_out;
    c->t_host_hash += now_us() - t_begin;
    c->host_vals[2 + round] = vals;
    c->host_nodes[2 + round] = nodes;
    c->host_node_cnt[2 + round] = 2 * half - 1;
    int rc = stage_seg(c, vals, c->layer_off[2 + round], half, 1);
    if (!rc) rc = stage_seg(c, nodes, c->tree_off[2 + round], (2 * half - 1) * 8, 0);
    return rc;
}

// One FRI round of the one-call flows: on the host once the layers are small and present there.
int fri_round_commit(zk_ctx* c, uint32_t round, uint32_t beta_raw, uint8_t root[32]) {
    if (c->tail_have && c->tail_log == c->L - round && c->L - round - 1 <= c->host_tail) return host_fold_commit(c, round, beta_raw, root);
    int rc = do_fold_commit(c, round, beta_raw);
    return rc ? rc : read_commit(c, 2 + round, root);
}
// ---- early launch of the next FRI round (zk_ctx_set_early_launch) --------------------------------------------------------------
// Round r is folded and committed by the host thread once the layers are small (host_fold_commit): exactly the rounds whose INPUT
// layer has at most 2^(host_tail + 1) values, provided some committed layer has that size to feed the host (mail_of).
bool round_on_host(const zk_ctx* c, uint32_t round) {
    return c->hash == 0 && c->host_top && c->host_tail && c->L >= c->host_tail + 1 && c->L - round <= c->host_tail + 1;
}
bool can_gate(const zk_ctx* c, uint32_t round) { return c->early && c->early_ok && !c->checks && round < c->R && !round_on_host(c, round); }
// The launches of do_fold_commit(round), enqueued behind a wait on the gate word; their one challenge-dependent constant is read
// from a parameter slot the host fills in release_gated_fold.
int enqueue_gated_fold(zk_ctx* c, uint32_t round) {
    FoldArgs a;
    int rc = fold_args(c->dom, c->d_layers + c->layer_off[1 + round], c->d_layers + c->layer_off[2 + round], c->L - round, round, 0u, a);
    if (rc) return rc;
    ++c->gate_seq;
    a.dyn = c->d_dyn + 16 * (c->gate_seq & 1u);
    HIPCHK(hipStreamWaitValue32(c->stream, c->d_gate, c->gate_seq, hipStreamWaitValueEq, 0xFFFFFFFFu));
    c->gate_pending = true;                                  // from here on the stream is blocked until the word is stored
    HIPCHK(launch_fold_merkle(a, c->d_trees + c->tree_off[2 + round], c->stream, prof_of(c), mail_of(c, 2 + round, true), c->hash));
    return ZK_OK;
}
void release_gate(zk_ctx* c, uint32_t c_mont) {
    c->h_dyn[16 * (c->gate_seq & 1u)] = c_mont;
    __atomic_store_n(c->h_gate, c->gate_seq, __ATOMIC_RELEASE);   // after the parameter (x86 stores are ordered; the device reads both over PCIe)
    c->gate_pending = false;
}
void release_gated_fold(zk_ctx* c, uint32_t round, uint32_t beta_raw) {
    release_gate(c, to_mont(mulmod(beta_raw % P, c->dom->fold_k[round])));   // beta * w^(-2^r) / 2, as fold_args
}
// A proof that ends early (an error between the enqueue and the release) must not leave the stream blocked: the pending launches
// run with a meaningless constant into this context's own buffers, which the next proof rebuilds.
struct GateGuard {
    zk_ctx* c;
    ~GateGuard() { if (c->gate_pending) release_gate(c, 0u); }
};

void begin_proof(zk_ctx* c) {
    c->stage_used = 0; c->n_segs = 0; c->seg_words = 0; c->tail_log = 0; c->tail_have = false;
    c->t_wait = c->t_host_hash = c->t_launch = 0;
    memset(c->host_nodes, 0, sizeof c->host_nodes);
    memset(c->host_node_cnt, 0, sizeof c->host_node_cnt);
    memset(c->host_vals, 0, sizeof c->host_vals);
}
// Device copies of everything the host built, stream-ordered before any later read of trees / layers.
int flush_host_parts(zk_ctx* c) {
    HIPCHK(launch_scatter(c->d_stage, c->d_segs, c->n_segs, c->seg_words, c->d_trees, c->d_layers, c->stream, prof_of(c)));
    c->n_segs = 0;
    c->pending_top = false;
    return ZK_OK;
}
// Before anything reads trees / layers from the device, or the stream is handed out: the tree top a stand-alone commitment
// left pending is copied now.
int settle_pending(zk_ctx* c) { return c->pending_top ? flush_host_parts(c) : (int)ZK_OK; }
// B evaluations of a degree-0 polynomial (prover.rs:238, :251) -> the free term (prover.rs:254)
int last_layer_value(zk_ctx* c, uint32_t* out) {
    const uint32_t* v;
    if (c->tail_have && c->tail_log == c->log_b) v = c->tail_vals.data();
    else {
        HIPCHK(hipMemcpyAsync(c->h_small, c->d_layers + c->layer_off[1 + c->R], c->B * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        v = c->h_small;
    }
    for (size_t i = 1; i < c->B; ++i)
        if (v[i] != v[0]) return fail(ZK_ERR_CHECK, "last FRI layer is not constant (prover.rs:238): trace does not satisfy the constraints");
    *out = v[0];
    return ZK_OK;
}

int do_compose(zk_ctx* c, const uint32_t alpha_raw[3]) {
    return dom_compose(c->dom, c->d_layers + c->layer_off[0], c->d_layers + c->layer_off[1], c->first, c->last,
                       alpha_raw, c->stream, prof_of(c));
}

int do_fold(zk_ctx* c, uint32_t round, uint32_t beta_raw) {
    return dom_fold(c->dom, c->d_layers + c->layer_off[1 + round], c->d_layers + c->layer_off[2 + round],
                    c->L - round, round, beta_raw, c->stream, prof_of(c));
}


// ---- opt-in reference self-checks ------------------------------------------------------------------------------
// The reference asserts its way through generate_proof; with zk_ctx_set_checks(ctx, 1) the same checkpoints run here,
// each on the layer as it sits in HBM, and the first one that fails is named (ZK_ERR_CHECK).
int checks_alloc(zk_ctx* c) {
    if (c->d_check) return ZK_OK;
    int rc = dmalloc(c, &c->d_check, (c->N + 2) * 4);
    if (rc) return rc;
    return build_table(1, c->log_n, &c->ones);
}
// prover.rs:64-66: the interpolant passes through every trace point.  The coefficients (virtual-point correction and
// 1/n as in coef_prepare, but no coset shift) are transformed forward over the TRACE group and compared with the trace.
int check_interpolant(zk_ctx* c) {
    const zk_dom* d = c->dom;
    CoefPrepArgs pa{};
    pa.log_n = d->log_n; pa.log_b = d->log_b;
    pa.tw = d->H.view(); pa.wtab = c->ones.view(); pa.ninv_mont = d->ninv_mont;
    pa.nd = d->plan.nd;
    for (uint32_t t = 0; t < d->plan.nd; ++t) pa.dig_bits[t] = d->plan.bits[t];
    HIPCHK(launch_coef_prepare(c->d_coef, c->d_check, pa, c->stream));          // d_coef[0..n) still holds the DIF output
    int rc = run_dit(c->d_check, d->log_n, d->plan, d->H.view(), d->L, c->stream);
    if (rc) return rc;
    std::vector<uint32_t> got(c->n), want(c->n);
    HIPCHK(hipMemcpyAsync(got.data(), c->d_check, c->n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(want.data(), c->d_trace, c->n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i + 1 < c->n; ++i)
        if (got[i] != want[i])
            return fail(ZK_ERR_CHECK, "self-check prover.rs:64-66 failed: f(g^%zu) = %u, trace value %u", i, got[i], want[i]);
    return ZK_OK;
}
// Degree of the polynomial whose evaluations are `layer` (1 = cp_0, 2 + r = FRI layer r + 1) is exactly want_deg:
// prover.rs:148-159 + :169 for cp (every division exact <=> deg cp = n - 1), :228-251 for the FRI layers.
int check_degree(zk_ctx* c, uint32_t layer, uint32_t want_deg, const char* cite) {
    const uint32_t lg = layer_log(c, layer);
    const size_t m = (size_t)1 << lg;
    const uint32_t* src = c->d_layers + c->layer_off[layer];
    if (lg <= 10) {
        // small layer: values to the host (or already there), plain inverse DFT of the coset-evaluations' sequence
        std::vector<uint32_t> v(m);
        if (c->tail_have && c->tail_log == lg && layer >= 2) v.assign(c->tail_vals.begin(), c->tail_vals.begin() + m);
        else {
            HIPCHK(hipMemcpyAsync(v.data(), src, m * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        const uint32_t winv = invmod(root_of_unity(lg));
        for (size_t k = m; k-- > want_deg; ) {                    // coefficients of p(shift x), up to the factor m
            uint32_t acc = 0, step = powmod(winv, k), x = 1;
            for (size_t i = 0; i < m; ++i) { acc = add(acc, mulmod(v[i], x)); x = mulmod(x, step); }
            if (k > want_deg && acc != 0)
                return fail(ZK_ERR_CHECK, "self-check %s failed: layer %u has a non-zero coefficient of degree %zu > %u", cite, layer, k, want_deg);
            if (k == want_deg && acc == 0)
                return fail(ZK_ERR_CHECK, "self-check %s failed: layer %u has degree below %u", cite, layer, want_deg);
        }
        return ZK_OK;
    }
    const Plan pl = make_plan(lg);
    int rc = run_dif(src, c->d_check, lg, pl, c->dom->Hinv.view(), c->dom->L, 0, c->stream);
    if (rc) return rc;
    uint32_t* res = c->d_check + c->N;
    HIPCHK(hipMemsetAsync(res, 0, 8, c->stream));
    HIPCHK(launch_degree_check(c->d_check, lg, pl.nd, pl.bits, want_deg + 1, res, c->stream));
    uint32_t h[2];
    HIPCHK(hipMemcpyAsync(h, res, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h[0]) return fail(ZK_ERR_CHECK, "self-check %s failed: layer %u has %u non-zero coefficients of degree > %u (a division left a remainder)", cite, layer, h[0], want_deg);
    if (!h[1]) return fail(ZK_ERR_CHECK, "self-check %s failed: layer %u has degree below %u", cite, layer, want_deg);
    return ZK_OK;
}

// ---- openings (prover.rs:266-289): where a value or a path node comes from -----------------------------------------
// Values and path nodes this thread built itself during the proof (tree tops, the small FRI layers and their trees) are
// still in the staging buffer and are read from there; everything else comes back through ONE launch (fetch_kernel) that
// takes its work list from host-mapped memory, writes its results there and raises the mailbox flag behind them: no copy
// commands, no stream synchronisation.  open_begin .. open_val / open_path .. open_launch .. open_wait; afterwards
// c->fetch_vals[i] / c->fetch_nodes[j] point at value i / path node j in the order they were asked for.
void open_begin(zk_ctx* c, size_t nvals) {
    c->fetch_vals.assign(nvals, nullptr); c->fetch_vdev.assign(nvals, 0); c->fetch_nodes.clear(); c->fetch_ditems.clear();
    c->open_nv = c->open_ndg = c->open_vi = 0;
}
void open_val(zk_ctx* c, uint32_t layer, size_t x) {
    if (c->host_vals[layer]) c->fetch_vals[c->open_vi++] = c->host_vals[layer] + x;
    else { c->fetch_vdev[c->open_vi++] = (uint32_t)c->open_nv; c->h_gather_off[c->open_nv++] = (uint64_t)c->layer_off[layer] + x; }
}
void open_path(zk_ctx* c, uint32_t tree, size_t m, size_t leaf) {          // merkle.rs:54-71: the sibling at every depth
    for (size_t i = leaf + m - 1; i != 0; i = (i - 1) >> 1) {
        const size_t nd = (i & 1) ? i + 1 : i - 1;
        if (nd < c->host_node_cnt[tree]) c->fetch_nodes.push_back(c->host_nodes[tree] + 8 * nd);
        else { c->fetch_nodes.push_back(c->h_gather_out + 8 * c->open_ndg); ++c->open_ndg; c->fetch_ditems.push_back((uint64_t)c->tree_off[tree] + (uint64_t)nd * 8); }
    }
}
int open_launch(zk_ctx* c) {
    const size_t nv = c->open_nv, ndg = c->open_ndg;
    if (nv + ndg > c->gather_cap) return fail(ZK_ERR_STATE, "gather capacity exceeded");
    if (ndg) memcpy(c->h_gather_off + nv, c->fetch_ditems.data(), ndg * 8);
    for (size_t i = 0; i < c->fetch_vals.size(); ++i)
        if (!c->fetch_vals[i]) c->fetch_vals[i] = c->h_gather_out + 8 * ndg + c->fetch_vdev[i];
    c->open_pending = nv + ndg != 0;
    if (c->open_pending)
        HIPCHK(launch_fetch(c->d_layers, c->d_trees, c->dm_gather_off, (uint32_t)nv, (uint32_t)ndg, c->dm_gather_out, c->d_mailbox,
                            ++c->mail_seq, c->d_counter, c->stream, prof_of(c)));
    return ZK_OK;
}
int open_wait(zk_ctx* c) {
    if (!c->open_pending) return ZK_OK;
    c->open_pending = false;
    return wait_mail(c, c->mail_seq);
}

// generate_proof(channel) (prover.rs:9): everything is committed to, and every challenge drawn from, the
// caller's channel `ch`, which may already hold a transcript prefix (main.rs:19 starts from a fresh one).
int prove_resident(zk_ctx* c, Channel& ch) {
    if (!c->have_trace) return fail(ZK_ERR_STATE, "zk_prove_resident: no trace uploaded");
    static const bool timing = getenv("ZK_HOST_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[zk timing] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - T0).count());
        T0 = t;
    };
    const uint32_t R = c->R;
    const size_t B = c->B, N = c->N;
    ch.data.reserve(ch.data.size() + proof_data_len(c->log_n, c->log_b, c->queries));
    uint8_t root[32];
    int rc;
    memset(&c->info, 0, sizeof c->info);
    c->info.public_last = c->last;
    c->pending_top = false;                               // a pending stand-alone tree top: every tree is rebuilt below
    begin_proof(c);
    GateGuard gate_guard{c};                              // whatever happens below, the stream is not left behind a closed gate
    if ((rc = do_lde(c))) return rc;                      // prover.rs:60-70
    if (c->checks && ((rc = checks_alloc(c)) || (rc = check_interpolant(c)))) return rc;   // prover.rs:64-66
    if ((rc = do_merkle(c, 0, true))) return rc;          // prover.rs:81
    if ((rc = read_commit(c, 0, root))) return rc;
    ch.commit_hash(root);                                 // prover.rs:85
    memcpy(c->info.roots[0], root, 32);
    uint32_t alpha[3];
    for (int i = 0; i < 3; ++i) alpha[i] = c->info.alpha_raw[i] = ch.get_u32();   // prover.rs:163-165
    if ((rc = do_compose_commit(c, alpha))) return rc;    // prover.rs:166-176 (composition fused into the leaf hashing)
    if (can_gate(c, 0) && (rc = enqueue_gated_fold(c, 0))) return rc;   // early launch: round 0 queued before cp's digests are waited for
    if ((rc = read_commit(c, 1, root))) return rc;
    if (c->checks && (rc = check_degree(c, 1, (uint32_t)c->n - 1, "prover.rs:148-159/:169 (exact divisions, deg cp = n - 1)"))) return rc;
    ch.commit_hash(root);                                 // prover.rs:180
    memcpy(c->info.roots[1], root, 32);
    for (uint32_t r = 0; r < R; ++r) {                    // prover.rs:198-225
        uint32_t beta = c->info.beta_raw[r] = ch.get_u32();   // prover.rs:200
        // prover.rs:201-214 (fold fused into the leaf hashing); on the host once the layers are small and present there
        if (c->gate_pending) {
            release_gated_fold(c, r, beta);               // this round's launches were enqueued before the challenge existed
            if (can_gate(c, r + 1) && (rc = enqueue_gated_fold(c, r + 1))) return rc;   // ... and so are the next round's, now
            if ((rc = read_commit(c, 2 + r, root))) return rc;
        } else if (c->tail_have && c->tail_log == c->L - r && c->L - r - 1 <= c->host_tail) {
            if ((rc = host_fold_commit(c, r, beta, root))) return rc;
        } else {
            if ((rc = do_fold_commit(c, r, beta))) return rc;
            if (can_gate(c, r + 1) && (rc = enqueue_gated_fold(c, r + 1))) return rc;
            if ((rc = read_commit(c, 2 + r, root))) return rc;
        }
        if (c->checks && (rc = check_degree(c, 2 + r, (uint32_t)(c->n >> (r + 1)) - ((c->n >> (r + 1)) ? 1 : 0), "prover.rs:228-251 (FRI layer degree)"))) return rc;
        ch.commit_hash(root);                             // prover.rs:224
        memcpy(c->info.roots[2 + r], root, 32);
    }
    lap("lde .. last root");
    if (timing) fprintf(stderr, "[zk timing]   of which: waiting for the device %.1f us, host hashing (tops + tail) %.1f us\n", c->t_wait, c->t_host_hash);
    // last layer: B evaluations of a degree-0 polynomial (prover.rs:238, :251); free term prover.rs:254
    uint32_t free_term = 0;
    if ((rc = last_layer_value(c, &free_term))) return rc;
    c->info.free_term = free_term;
    ch.commit_u32(free_term);                             // prover.rs:254
    const uint32_t Q = c->queries;
    uint32_t qraws[64];
    for (uint32_t k = 0; k < Q; ++k) qraws[k] = ch.get_u32();   // prover.rs:263 (x Q)
    c->info.query_raw = qraws[0];

    // decommit (prover.rs:266-289).  Values and path nodes this thread built itself during the proof (tree tops, the small
    // FRI layers and their trees) are still in the staging buffer and are read from there; everything else comes back
    // through ONE launch that takes its work list from host-mapped memory, writes its results there and raises the
    // mailbox flag behind them: no copy commands, no stream synchronisation.  The scatter that completes the device
    // arrays with the host-built parts is enqueued behind it, off the proof's critical path.
    open_begin(c, (size_t)Q * (4 + 2 * R));
    for (uint32_t k = 0; k < Q; ++k) {
        const size_t x = (size_t)qraws[k] % (N - 2 * B);
        open_val(c, 0, x);         open_path(c, 0, N, x);
        open_val(c, 0, x + B);     open_path(c, 0, N, x + B);
        open_val(c, 0, x + 2 * B); open_path(c, 0, N, x + 2 * B);
        open_val(c, 1, x);         open_path(c, 1, N, x);
        for (uint32_t i = 0; i < R; ++i) {
            size_t len = N >> i, xi = x % len, nx = (xi + len / 2) % len;
            open_val(c, 1 + i, xi); open_path(c, 1 + i, len, xi);
            open_val(c, 1 + i, nx); open_path(c, 1 + i, len, nx);
        }
    }
    if ((rc = open_launch(c))) return rc;
    if ((rc = flush_host_parts(c))) return rc;            // completes the device arrays, behind the fetch: off the critical path
    lap("free term + fetch enqueue");
    if ((rc = open_wait(c))) return rc;
    lap("fetch wait");
    std::vector<const uint32_t*>& vsrc = c->fetch_vals;
    std::vector<const uint32_t*>& dsrc = c->fetch_nodes;
    size_t vi = 0;
    const size_t Lp = c->L;
    std::vector<uint8_t>& buf = c->commit_buf;
    buf.resize(8 + 2 * (8 + 32 * Lp));
    auto put32 = [](uint8_t* p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i)); };
    auto put_path = [&](uint8_t* p, size_t first, size_t plen) {       // Box<[Hash]>: u64 count + items
        for (int i = 0; i < 8; ++i) p[i] = (uint8_t)((uint64_t)plen >> (8 * i));
        for (size_t j = 0; j < plen; ++j) digest_words_to_bytes(dsrc[first + j], p + 8 + 32 * j);
        return 8 + 32 * plen;
    };
    size_t dpos = 0;
    for (uint32_t q = 0; q < Q; ++q) {
        for (int k = 0; k < 4; ++k) {                         // (u32, AuthPath): prover.rs:274-277
            put32(buf.data(), *vsrc[vi++]);
            const size_t len = 4 + put_path(buf.data() + 4, dpos, Lp);
            ch.commit_bytes(buf.data(), len);
            dpos += Lp;
        }
        for (uint32_t i = 0; i < R; ++i) {                    // (u32, u32, AuthPath, AuthPath): prover.rs:280-289
            const size_t pl = Lp - i;
            put32(buf.data(), *vsrc[vi]); put32(buf.data() + 4, *vsrc[vi + 1]);
            vi += 2;
            size_t len = 8 + put_path(buf.data() + 8, dpos, pl);
            len += put_path(buf.data() + len, dpos + pl, pl);
            ch.commit_bytes(buf.data(), len);
            dpos += 2 * pl;
        }
    }
    lap("decommit host hashing");
    return ZK_OK;                                         // the proof is the channel: channel.rs:34-36
}

}  // namespace

// The openings of the replicated tail in two steps, so that the sharded prover can have its own fetch in flight beside
// this one: begin enqueues at most one launch on the tail's stream (what this thread hashed itself during zk_tail_run --
// tree tops, the small layers and their trees -- is read from the staging buffer), end waits for its flag and copies out.
namespace zk {
namespace impl {
int tail_open_begin(zk_ctx* c, size_t x) {
    if (!c || !c->tail) return fail(ZK_ERR_INVALID, "tail_open_begin: not a tail context");
    HIPCHK(hipSetDevice(c->device));
    open_begin(c, 2 * (size_t)c->R);
    for (uint32_t i = 0; i < c->R; ++i) {                     // prover.rs:280-289 for the tail layers
        const size_t len = c->N >> i, xi = x % len, nx = (xi + len / 2) % len;
        open_val(c, 1 + i, xi); open_path(c, 1 + i, len, xi);
        open_val(c, 1 + i, nx); open_path(c, 1 + i, len, nx);
    }
    return open_launch(c);
}
int tail_open_end(zk_ctx* c, uint32_t* vals_out, uint8_t* paths_out) {
    int rc = open_wait(c);
    if (rc) return rc;
    for (size_t i = 0; i < c->fetch_vals.size(); ++i) vals_out[i] = *c->fetch_vals[i];
    for (size_t i = 0; i < c->fetch_nodes.size(); ++i) digest_words_to_bytes(c->fetch_nodes[i], paths_out + 32 * i);
    return ZK_OK;
}
}  // namespace impl
}  // namespace zk

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char* zk_last_error(void) { return last_error(); }
int zk_host_hash_mode(void) { return host_sha_wide_available() ? 2 : host_sha_available() ? 1 : 0; }
int zk_host_set_hash_mode(int mode) {
    if (mode < 0 || mode > 2) return fail(ZK_ERR_INVALID, "zk_host_set_hash_mode: mode %d out of range (0 portable, 1 SHA extensions, 2 + AVX-512)", mode);
    host_sha_use_extensions(mode >= 1);
    host_sha_use_wide(mode >= 2);
    return ZK_OK;
}

uint32_t zk_field_add(uint32_t a, uint32_t b) { return add(a % P, b % P); }
uint32_t zk_field_sub(uint32_t a, uint32_t b) { return sub(a % P, b % P); }
uint32_t zk_field_mul(uint32_t a, uint32_t b) { return mulmod(a % P, b % P); }
uint32_t zk_field_neg(uint32_t a) { return neg(a % P); }
uint32_t zk_field_inv(uint32_t a) { return invmod(a % P); }
uint32_t zk_field_pow(uint32_t a, uint32_t e) { return powmod(a, e); }
uint32_t zk_field_from_u32(uint32_t v) { return v % P; }
// field.rs:10-18 From<i32>: a negative value is the negation of |f| (i32::MIN: |f| = 2^31, what the release build's wrapping abs() gives)
uint32_t zk_field_from_i32(int32_t v) {
    if (v < 0) return neg(uint32_t(-int64_t(v)) % P);
    return uint32_t(v) % P;
}
// field.rs:165-177 Div: a * b^-1.  The reference panics on a zero divisor (MontgomeryInt's inverse does not exist); here 0 and an error message.
uint32_t zk_field_div(uint32_t a, uint32_t b) {
    if (b % P == 0) { fail(ZK_ERR_INVALID, "zk_field_div: division by zero (field.rs:165-177 panics)"); return 0; }
    return mulmod(a % P, invmod(b % P));
}
// field.rs:89-94 Rem<u32>: the RESIDUE reduced by an integer modulus, back in the field.  rhs = 0 panics there; here 0 and an error message.
uint32_t zk_field_rem(uint32_t a, uint32_t rhs) {
    if (rhs == 0) { fail(ZK_ERR_INVALID, "zk_field_rem: remainder by zero (field.rs:89-94 panics)"); return 0; }
    return (a % P) % rhs;
}
// field.rs:52-86 Gf::generator(): the first x >= 2 with x^((P-1)/q) != 1 for every prime factor q of P - 1.  Searched as the
// reference searches it (unique prime factors by trial division, then candidates in order), once; the kernels use the
// constant GEN_W, which this search must -- and does -- return (5; pinned by tests/test_cabi.py).
uint32_t zk_field_generator(void) {
    static const uint32_t g = [] {
        uint32_t factors[32], nf = 0, p = P - 1;
        for (uint32_t it = 2; p != 1; ++it) {              // field.rs:56-67
            if (p % it == 0) factors[nf++] = it;
            while (p % it == 0) p /= it;
        }
        for (uint32_t x = 2; x < P; ++x) {                 // field.rs:78-86
            bool primitive = true;
            for (uint32_t i = 0; i < nf && primitive; ++i) primitive = powmod(x, (P - 1) / factors[i]) != 1;   // field.rs:70-76: (P-1) * factor^-1 = (P-1) / factor
            if (primitive) return x;
        }
        return 0u;
    }();
    return g;
}
// field.rs:45-49 order(): the reference brute-forces it; P - 1 = 3 * 2^30 gives it in 32 squarings
uint32_t zk_field_order(uint32_t a) {
    a %= P;
    if (a == 0) return 0;
    uint32_t ord = P - 1;
    if (powmod(a, ord / 3) == 1) ord /= 3;
    while (ord % 2 == 0 && powmod(a, ord / 2) == 1) ord /= 2;
    return ord;
}
uint32_t zk_field_root_of_unity(uint32_t log_order) { return log_order > 30 ? 0 : root_of_unity(log_order); }

static int ctx_make(int device, uint32_t log_n, uint32_t log_b, uint32_t shift, bool tail, zk_ctx** out);
// the team that reduces hand-over depths above the sub-tree size one thread takes (none needed at or below it)
static int ctx_team(zk_ctx* c) {
    const uint32_t sub_log = host_sub_log();
    const unsigned want = c->host_top > sub_log ? (1u << (c->host_top - sub_log)) - 1u : 0u;
    if (c->pool && c->pool->workers() == want) return ZK_OK;
    delete c->pool;
    c->pool = nullptr;
    if (!want) return ZK_OK;
    // the commitments of one big proof are up to 1.6 ms apart (a 2^24-leaf launch): the team spins across them
    // (5 ms), or every tree top would pay a futex wake-up
    const double spin_us = 5000.0;
    c->pool = new (std::nothrow) Pool(want, spin_us);
    return c->pool ? (int)ZK_OK : fail(ZK_ERR_NOMEM, "out of host memory");
}

int zk_ctx_create(int device, uint32_t log_n, uint32_t log_b, zk_ctx** out) {
    if (!out) return fail(ZK_ERR_INVALID, "zk_ctx_create: out is null");
    *out = nullptr;
    if (int rc = check_proof_size("zk_ctx_create", log_n, log_b)) return rc;
    return ctx_make(device, log_n, log_b, GEN_W, false, out);
}

// tail = true: a context that only runs the FRI phase (layers 1.., fold-only domain with an arbitrary shift)
static int ctx_make(int device, uint32_t log_n, uint32_t log_b, uint32_t shift, bool tail, zk_ctx** out) {
    auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(device));
    zk_ctx* c = new (std::nothrow) zk_ctx();
    if (!c) return fail(ZK_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->log_n = log_n; c->log_b = log_b; c->L = log_n + log_b;
    c->n = (size_t)1 << log_n; c->B = (size_t)1 << log_b; c->N = c->n << log_b;
    c->R = log_n;
    int rc = ZK_OK;
    auto bail = [&](int code) { zk_ctx_destroy(c); return code; };
#define HIPCHK_C(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            zk_ctx_destroy(c);                                                                \
            return fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                     \
    } while (0)
    HIPCHK_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if ((rc = dom_make(device, log_n, log_b, shift, tail, c->stream, &c->dom))) return bail(rc);
    c->tail = tail;
    c->hinv_host = invmod(c->dom->h);
    c->device_bytes += c->dom->device_bytes;
    // layers: 0 = f_eval (N), 1 + r = FRI layer r (N >> r), r = 0 .. R
    size_t off = 0;
    for (uint32_t l = 0; l <= c->R + 1; ++l) {
        c->layer_off.push_back(off);
        c->layer_len.push_back(layer_size(c, l));
        off += layer_size(c, l);
    }
    size_t layer_words = off;
    off = 0;
    for (uint32_t l = 0; l <= c->R + 1; ++l) {
        c->tree_off.push_back(off);
        off += (2 * layer_size(c, l) - 1) * 8;
    }
    size_t tree_words = off;
    if ((rc = dmalloc(c, &c->d_trace, c->n * 4))) return bail(rc);
    if ((rc = dmalloc(c, &c->d_coef, 2 * c->n * 4))) return bail(rc);
    if ((rc = dmalloc(c, &c->d_layers, layer_words * 4))) return bail(rc);
    if ((rc = dmalloc(c, &c->d_trees, tree_words * 4))) return bail(rc);
    c->gather_cap = (size_t)kMaxQueries * (4 + 2 * c->R) * (c->L + 1) + 64;
    if ((rc = dmalloc(c, &c->d_gather_off, c->gather_cap * 8))) return bail(rc);
    if ((rc = dmalloc(c, &c->d_gather_out, c->gather_cap * 32))) return bail(rc);
    HIPCHK_C(hipHostMalloc((void**)&c->h_gather_off, c->gather_cap * 8, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_C(hipHostMalloc((void**)&c->h_gather_out, c->gather_cap * 32, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_C(hipHostGetDevicePointer((void**)&c->dm_gather_off, c->h_gather_off, 0));
    HIPCHK_C(hipHostGetDevicePointer((void**)&c->dm_gather_out, c->h_gather_out, 0));
    HIPCHK_C(hipHostMalloc((void**)&c->h_small, 4096));
    const size_t mail_bytes = kMailWords * 4;
    if ((rc = dmalloc(c, &c->d_counter, 64))) return bail(rc);
    HIPCHK_C(hipMemsetAsync(c->d_counter, 0, 64, c->stream));
    HIPCHK_C(hipHostMalloc((void**)&c->h_mailbox, mail_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_mailbox, 0, mail_bytes);
    HIPCHK_C(hipHostGetDevicePointer((void**)&c->d_mailbox, c->h_mailbox, 0));
    // early launch (zk_ctx_set_early_launch): the gate word and the parameter slots; usable when the device has stream memory operations
    HIPCHK_C(hipHostMalloc((void**)&c->h_gate, 64, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_C(hipHostMalloc((void**)&c->h_dyn, 2 * 16 * 4, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_gate, 0, 64);
    memset(c->h_dyn, 0, 2 * 16 * 4);
    HIPCHK_C(hipHostGetDevicePointer((void**)&c->d_gate, c->h_gate, 0));
    HIPCHK_C(hipHostGetDevicePointer((void**)&c->d_dyn, c->h_dyn, 0));
    {
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) { can = 0; (void)hipGetLastError(); }
        c->early_ok = can != 0 && !tail;
    }
    // staging for host-built tree tops and tail layers: one top per tree, the tail layers and their trees, the segment table
    c->stage_words = (size_t)(c->R + 2) * ((size_t)16 << kMaxHostLog) + ((size_t)64 << kMaxHostLog);
    const size_t seg_bytes = 4 * 34 * sizeof(ScatterSeg);
    HIPCHK_C(hipHostMalloc((void**)&c->h_stage, c->stage_words * 4 + seg_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_C(hipHostGetDevicePointer((void**)&c->d_stage, c->h_stage, 0));
    c->h_segs = reinterpret_cast<ScatterSeg*>(c->h_stage + c->stage_words);
    c->d_segs = reinterpret_cast<ScatterSeg*>(c->d_stage + c->stage_words);
    if (host_sha_available()) {            // without the SHA extensions the device builds every tree to the root
        c->host_top = 8;
        c->host_tail = 9;
    }
    if (int prc = ctx_team(c)) { zk_ctx_destroy(c); return prc; }
#undef HIPCHK_C
    c->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = c;
    return ZK_OK;
}

#ifdef ZK_WG_TRACE
extern "C++" { namespace zk { void dump_wg_trace(); } }   // kernels.hip: diagnostic build only (tools/wg_trace.py)
#endif

int zk_ctx_destroy(zk_ctx* c) {
    if (!c) return ZK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
#ifdef ZK_WG_TRACE
    zk::dump_wg_trace();
#endif
    dom_free(c->dom);
    if (c->d_trace) (void)hipFree(c->d_trace);
    if (c->d_coef) (void)hipFree(c->d_coef);
    if (c->d_layers) (void)hipFree(c->d_layers);
    if (c->d_trees) (void)hipFree(c->d_trees);
    if (c->d_counter) (void)hipFree(c->d_counter);
    if (c->d_check) (void)hipFree(c->d_check);
    free_table(&c->ones);
    if (c->d_gather_off) (void)hipFree(c->d_gather_off);
    if (c->d_gather_out) (void)hipFree(c->d_gather_out);
    if (c->h_gather_off) (void)hipHostFree(c->h_gather_off);
    if (c->h_gather_out) (void)hipHostFree(c->h_gather_out);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_mailbox) (void)hipHostFree(c->h_mailbox);
    if (c->h_gate) (void)hipHostFree(c->h_gate);
    if (c->h_dyn) (void)hipHostFree(c->h_dyn);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    collect_kernel_stats(c);
    delete c->pool;
    for (hipEvent_t e : c->prof.pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ZK_OK;
}

double zk_ctx_setup_ms(const zk_ctx* c) { return c ? c->setup_ms : 0.0; }
size_t zk_ctx_device_bytes(const zk_ctx* c) { return c ? c->device_bytes : 0; }
void* zk_ctx_stream(zk_ctx* c) {
    if (!c) return nullptr;
    if (c->pending_top && hipSetDevice(c->device) == hipSuccess) (void)settle_pending(c);   // work the caller orders behind the stream sees whole trees
    return (void*)c->stream;
}
int zk_ctx_sync(zk_ctx* c) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    if (int rc = settle_pending(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
// Early launch of the FRI rounds' commit launches (include/zkstark_amd.h).  on != 0 where the device has no stream memory
// operations is accepted and has no effect (results never depend on it).
int zk_ctx_set_early_launch(zk_ctx* c, int on) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    c->early = on != 0;
    return ZK_OK;
}
int zk_ctx_get_early_launch(const zk_ctx* c) { return c && c->early && c->early_ok ? 1 : 0; }
int zk_ctx_set_queries(zk_ctx* c, uint32_t n_queries) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    if (n_queries < 1 || n_queries > kMaxQueries) return fail(ZK_ERR_INVALID, "zk_ctx_set_queries: need 1 <= n_queries <= %u", kMaxQueries);
    c->queries = n_queries;
    return ZK_OK;
}

int zk_ctx_set_host_levels(zk_ctx* c, uint32_t top_log, uint32_t tail_log) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    if (top_log > kMaxHostLog || tail_log > kMaxHostLog || (tail_log && !top_log))
        return fail(ZK_ERR_INVALID, "zk_ctx_set_host_levels: need top_log, tail_log <= %u, and top_log > 0 when tail_log > 0", kMaxHostLog);
    c->host_top = top_log;
    c->host_tail = tail_log;
    return ctx_team(c);
}
int zk_ctx_get_host_levels(const zk_ctx* c, uint32_t* top_log, uint32_t* tail_log) {
    if (!c || !top_log || !tail_log) return fail(ZK_ERR_INVALID, "zk_ctx_get_host_levels: null argument");
    *top_log = c->host_top;
    *tail_log = c->host_tail;
    return ZK_OK;
}
// Opt-in: the reference's in-prover assertions (prover.rs:64-66 interpolant hits the trace, :148-159/:169 exact
// divisions and deg cp = n - 1, :228-251 degree of every FRI layer) run inside zk_prove*; the first that fails is
// named in a ZK_ERR_CHECK.  Costs a forward transform of the interpolant and one inverse transform per layer.
int zk_ctx_set_checks(zk_ctx* c, int on) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    if (c->tail) return fail(ZK_ERR_STATE, "zk_ctx_set_checks: FRI-tail context");
    c->checks = on != 0;
    return ZK_OK;
}
int zk_ctx_set_hash(zk_ctx* c, int hash_kind) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    if (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) return fail(ZK_ERR_INVALID, "zk_ctx_set_hash: unknown hash %d", hash_kind);
    c->hash = hash_kind;
    return ZK_OK;
}

int zk_ctx_set_profiling(zk_ctx* c, uint32_t class_mask) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    c->prof.mask = class_mask & ((1u << K_COUNT) - 1u);
    return ZK_OK;
}

// out[0].struct_size is the stride of the caller's array (include/zkstark_amd.h: zk_kernel_stat)
static int put_kernel_stats(zk_kernel_stat* out, size_t count, const zk_kernel_stat* src, const char* who) {
    if (!count) return ZK_OK;
    const uint32_t stride = out[0].struct_size;
    if (!abi_bytes(out, who)) return ZK_ERR_INVALID;
    for (size_t i = 0; i < count && i < (size_t)K_COUNT; ++i) {
        zk_kernel_stat* o = reinterpret_cast<zk_kernel_stat*>(reinterpret_cast<char*>(out) + i * stride);
        o->struct_size = stride;
        if (int rc = abi_put(o, src[i], who)) return rc;
    }
    return ZK_OK;
}

int zk_kernel_stats(zk_ctx* c, zk_kernel_stat* out, size_t count, int reset) {
    if (!c || (!out && count)) return fail(ZK_ERR_INVALID, "zk_kernel_stats: null argument");
    collect_kernel_stats(c);
    if (int rc = put_kernel_stats(out, count, c->kstat, "zk_kernel_stats")) return rc;
    if (reset) for (int i = 0; i < K_COUNT; ++i) c->kstat[i] = zk_kernel_stat{};
    return ZK_OK;
}

int zk_trace_fibsq(uint32_t a0, uint32_t a1, size_t count, uint32_t* out) {
    if (!out && count) return fail(ZK_ERR_INVALID, "zk_trace_fibsq: out is null");
    if (count > 0) out[0] = a0 % P;                       // prover.rs:33
    if (count > 1) out[1] = a1 % P;                       // prover.rs:34
    for (size_t i = 2; i < count; ++i)                    // prover.rs:35-39
        out[i] = add(mulmod(out[i - 2], out[i - 2]), mulmod(out[i - 1], out[i - 1]));
    return ZK_OK;
}

int zk_trace_upload(zk_ctx* c, const uint32_t* trace, size_t count) {
    if (!c || !trace) return fail(ZK_ERR_INVALID, "zk_trace_upload: null argument");
    if (count != c->n - 1) return fail(ZK_ERR_INVALID, "zk_trace_upload: expected n-1 = %zu values, got %zu", c->n - 1, count);
    for (size_t i = 0; i < count; ++i)
        if (trace[i] >= P) return fail(ZK_ERR_INVALID, "zk_trace_upload: trace[%zu] = %u is not a canonical residue", i, trace[i]);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_trace, trace, count * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_trace + count, 0, 4, c->stream));   // slot n-1: see DESIGN.md "LDE"
    HIPCHK(hipStreamSynchronize(c->stream));
    c->first = trace[0];
    c->last = trace[count - 1];                           // a[n-2], the public output (prover.rs:42)
    c->have_trace = true;
    c->have_lde = false;
    return ZK_OK;
}

int zk_lde(zk_ctx* c) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    if (!c->have_trace) return fail(ZK_ERR_STATE, "zk_lde: no trace uploaded");
    HIPCHK(hipSetDevice(c->device));
    static const bool timing = getenv("ZK_HOST_TIMING") != nullptr;
    const double t0 = now_us();
    int rc = do_lde(c);
    if (timing) fprintf(stderr, "[zk timing] lde: enqueue of %u passes %.1f us\n", 2 * c->dom->plan.nd, now_us() - t0);
    return rc;
}

int zk_merkle_commit(zk_ctx* c, uint32_t layer, uint8_t root_out[32]) {
    if (!c || !root_out) return fail(ZK_ERR_INVALID, "zk_merkle_commit: null argument");
    if (layer > c->R + 1) return fail(ZK_ERR_INVALID, "zk_merkle_commit: layer %u out of range", layer);
    HIPCHK(hipSetDevice(c->device));
    static const bool timing = getenv("ZK_HOST_TIMING") != nullptr;
    const double t0 = now_us();
    // the top of an earlier stand-alone commitment still waiting in the staging buffer: the same tree is rebuilt now (drop
    // it), another tree's top goes to the device before its staging space is reused
    if (c->pending_top && c->pending_tree == layer) c->pending_top = false;
    int rc = settle_pending(c);
    if (rc) return rc;
    begin_proof(c);                                      // a stand-alone commitment: nothing staged, no host-side FRI tail
    rc = do_merkle(c, layer, true, false);               // Merkle::new (merkle.rs:14); the last levels on this thread
    const double t1 = now_us();
    if (!rc) rc = read_commit(c, layer, root_out);
    const double t2 = now_us();
    // The root is known.  The device copy of the host-built top (255 nodes) is NOT ordered on the stream here: it would sit
    // in front of whatever the caller enqueues next (7 us per commitment in a commit loop); settle_pending() orders it
    // before the first read of the device arrays.
    if (!rc && c->n_segs) { c->pending_top = true; c->pending_tree = layer; }
    if (timing)
        fprintf(stderr, "[zk timing] merkle_commit(layer %u): enqueue %.1f us, wait for the device %.1f us, host top %.1f us, scatter enqueue %.1f us\n",
                layer, t1 - t0, c->t_wait, c->t_host_hash, now_us() - t2);
    return rc;
}

int zk_compose(zk_ctx* c, const uint32_t alpha_raw[3]) {
    if (!c || !alpha_raw) return fail(ZK_ERR_INVALID, "zk_compose: null argument");
    if (!c->have_lde) return fail(ZK_ERR_STATE, "zk_compose: run zk_lde first");
    HIPCHK(hipSetDevice(c->device));
    return do_compose(c, alpha_raw);
}

int zk_fri_fold(zk_ctx* c, uint32_t round, uint32_t beta_raw) {
    if (!c) return fail(ZK_ERR_INVALID, "null context");
    if (round >= c->R) return fail(ZK_ERR_INVALID, "zk_fri_fold: round %u out of range (%u rounds)", round, c->R);
    HIPCHK(hipSetDevice(c->device));
    return do_fold(c, round, beta_raw);
}

int zk_layer_read(zk_ctx* c, uint32_t layer, size_t offset, size_t count, uint32_t* out) {
    if (!c || (!out && count)) return fail(ZK_ERR_INVALID, "zk_layer_read: null argument");
    if (layer > c->R + 1 || offset + count > layer_size(c, layer)) return fail(ZK_ERR_INVALID, "zk_layer_read: out of range");
    HIPCHK(hipSetDevice(c->device));
    if (int prc = settle_pending(c)) return prc;
    HIPCHK(hipMemcpyAsync(out, c->d_layers + c->layer_off[layer] + offset, count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_layer_write(zk_ctx* c, uint32_t layer, size_t offset, size_t count, const uint32_t* in) {
    if (!c || (!in && count)) return fail(ZK_ERR_INVALID, "zk_layer_write: null argument");
    if (layer > c->R + 1 || offset + count > layer_size(c, layer)) return fail(ZK_ERR_INVALID, "zk_layer_write: out of range");
    for (size_t i = 0; i < count; ++i)
        if (in[i] >= P) return fail(ZK_ERR_INVALID, "zk_layer_write: value %zu is not a canonical residue", i);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_layers + c->layer_off[layer] + offset, in, count * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (layer == 0) c->have_lde = true;
    return ZK_OK;
}

int zk_merkle_node(zk_ctx* c, uint32_t tree, size_t index, uint8_t out[32]) {
    if (!c || !out) return fail(ZK_ERR_INVALID, "zk_merkle_node: null argument");
    if (tree > c->R + 1 || index >= 2 * layer_size(c, tree) - 1) return fail(ZK_ERR_INVALID, "zk_merkle_node: out of range");
    HIPCHK(hipSetDevice(c->device));
    if (int prc = settle_pending(c)) return prc;
    HIPCHK(hipMemcpyAsync(c->h_small, c->d_trees + c->tree_off[tree] + index * 8, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    digest_words_to_bytes(c->h_small, out);
    return ZK_OK;
}

int zk_merkle_path(zk_ctx* c, uint32_t tree, size_t leaf, uint8_t* out, size_t* path_len) {
    if (!c || !out) return fail(ZK_ERR_INVALID, "zk_merkle_path: null argument");
    if (tree > c->R + 1 || leaf >= layer_size(c, tree)) return fail(ZK_ERR_INVALID, "zk_merkle_path: out of range");
    HIPCHK(hipSetDevice(c->device));
    if (int prc = settle_pending(c)) return prc;
    std::vector<size_t> nodes;
    path_nodes(layer_size(c, tree), leaf, nodes);
    for (size_t i = 0; i < nodes.size(); ++i) c->h_gather_off[i] = (uint64_t)c->tree_off[tree] + (uint64_t)nodes[i] * 8;
    HIPCHK(hipMemcpyAsync(c->d_gather_off, c->h_gather_off, nodes.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_gather(c->d_trees, c->d_gather_off, (uint32_t)nodes.size(), 8, c->d_gather_out, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_gather_out, c->d_gather_out, nodes.size() * 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < nodes.size(); ++i) digest_words_to_bytes(c->h_gather_out + 8 * i, out + 32 * i);
    if (path_len) *path_len = nodes.size();
    return ZK_OK;
}

int zk_prove_resident(zk_ctx* c, uint8_t* proof_out, size_t cap, size_t* proof_len, uint8_t state_out[32]) {
    if (!c || !proof_out || !state_out) return fail(ZK_ERR_INVALID, "zk_prove_resident: null argument");
    HIPCHK(hipSetDevice(c->device));
    Channel ch;                                           // main.rs:19
    int rc = prove_resident(c, ch);
    if (rc) return rc;
    const std::vector<uint8_t>& proof = ch.data;          // channel.rs:34-36
    if (proof_len) *proof_len = proof.size();
    if (proof.size() > cap) return fail(ZK_ERR_BUFFER, "zk_prove: proof needs %zu bytes, buffer has %zu", proof.size(), cap);
    memcpy(proof_out, proof.data(), proof.size());
    memcpy(state_out, ch.state, 32);
    return ZK_OK;
}

// generate_proof(channel: Channel) -> Proof (prover.rs:9): the literal drop-in.  The resident trace is proved on the
// caller's channel: every commitment is appended to it and every challenge drawn from it, whatever it already holds.
int zk_prove_channel(zk_ctx* c, zk_channel* chan) {
    if (!c || !chan) return fail(ZK_ERR_INVALID, "zk_prove_channel: null argument");
    if (c->tail) return fail(ZK_ERR_STATE, "zk_prove_channel: FRI-tail context");
    HIPCHK(hipSetDevice(c->device));
    return prove_resident(c, chan->ch);
}

int zk_prove(zk_ctx* c, const uint32_t* trace, size_t count, uint8_t* proof_out, size_t cap, size_t* proof_len,
             uint8_t state_out[32]) {
    int rc = zk_trace_upload(c, trace, count);
    if (rc) return rc;
    return zk_prove_resident(c, proof_out, cap, proof_len, state_out);
}

// Several independent proofs in flight on one GPU: one host thread per context, so the latency-bound tree
// tops of one proof overlap the hashing of the others (the device is VALU-saturated with three).
int zk_prove_many(zk_ctx* const* ctxs, size_t count, uint8_t* proofs_out, size_t stride, size_t* lens_out, uint8_t* states_out) {
    if (!ctxs || !proofs_out || !lens_out || !states_out || count == 0 || count > 16)
        return fail(ZK_ERR_INVALID, "zk_prove_many: need 1..16 contexts and non-null outputs");
    for (size_t i = 0; i < count; ++i) {
        if (!ctxs[i]) return fail(ZK_ERR_INVALID, "zk_prove_many: context %zu is null", i);
        for (size_t j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ZK_ERR_INVALID, "zk_prove_many: context %zu appears twice", i);
    }
    std::vector<int> rcs(count, ZK_OK);
    std::vector<std::string> errs(count);
    auto one = [&](size_t i) {
        rcs[i] = zk_prove_resident(ctxs[i], proofs_out + i * stride, stride, &lens_out[i], states_out + 32 * i);
        if (rcs[i]) errs[i] = last_error();              // the message lives in the worker's thread-local slot
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < count; ++i) th.emplace_back(one, i);
    one(0);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < count; ++i)
        if (rcs[i]) return fail(rcs[i], "zk_prove_many: context %zu: %s", i, errs[i].c_str());
    return ZK_OK;
}

int zk_last_transcript(const zk_ctx* c, zk_transcript_info* out) {
    if (!c || !out) return fail(ZK_ERR_INVALID, "zk_last_transcript: null argument");
    return abi_put(out, c->info, "zk_last_transcript");
}

int zk_verify_ex(const uint8_t* proof, size_t len, uint32_t log_n, uint32_t log_b, uint32_t public_last, int hash_kind) {
    if (!proof) return fail(ZK_ERR_INVALID, "zk_verify: null proof");
    if (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) return fail(ZK_ERR_INVALID, "zk_verify: unknown hash %d", hash_kind);
    int rc = verify_proof(proof, len, log_n, log_b, public_last, hash_kind);
    if (rc) return fail(ZK_ERR_VERIFY, "proof rejected at check %d (proof.rs:15-149)", rc);
    return ZK_OK;
}
int zk_verify(const uint8_t* proof, size_t len, uint32_t log_n, uint32_t log_b, uint32_t public_last) {
    return zk_verify_ex(proof, len, log_n, log_b, public_last, ZK_HASH_SHA256);
}
int zk_verify_queries(const uint8_t* proof, size_t len, const uint8_t* state, uint32_t log_n, uint32_t log_b, uint32_t public_last,
                      int hash_kind, uint32_t n_queries) {
    if (!proof) return fail(ZK_ERR_INVALID, "zk_verify_queries: null proof");
    if (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) return fail(ZK_ERR_INVALID, "zk_verify_queries: unknown hash %d", hash_kind);
    if (state) {
        int rc = verify_transcript(proof, len, state, log_n, log_b, n_queries);
        if (rc) return fail(ZK_ERR_VERIFY, "transcript replay failed at check %d", rc);
    }
    int rc = verify_proof(proof, len, log_n, log_b, public_last, hash_kind, n_queries);
    if (rc) return fail(ZK_ERR_VERIFY, "proof rejected at check %d (proof.rs:15-149)", rc);
    return ZK_OK;
}

int zk_verify_strict(const uint8_t* proof, size_t len, const uint8_t state[32], uint32_t log_n, uint32_t log_b, uint32_t public_last) {
    if (!proof || !state) return fail(ZK_ERR_INVALID, "zk_verify_strict: null argument");
    int rc = verify_transcript(proof, len, state, log_n, log_b);
    if (rc) return fail(ZK_ERR_VERIFY, "transcript replay failed at check %d (challenge not derived from the transcript, or final state mismatch)", rc);
    return zk_verify(proof, len, log_n, log_b, public_last);
}

size_t zk_proof_size(size_t data_len) { return 48 + data_len; }   // proof.rs:151-154: size_of::<Proof>() = 32 + 16
size_t zk_proof_data_len(uint32_t log_n, uint32_t log_b) { return proof_data_len(log_n, log_b); }
size_t zk_proof_data_len_queries(uint32_t log_n, uint32_t log_b, uint32_t n_queries) { return proof_data_len(log_n, log_b, n_queries); }

int zk_compute_root_from_path_ex(uint32_t element, size_t index, const uint8_t* path, size_t path_len, uint8_t out[32], int hash_kind) {
    if ((!path && path_len) || !out || path_len > 62 || (hash_kind != 0 && hash_kind != 1))
        return fail(ZK_ERR_INVALID, "zk_compute_root_from_path: bad argument");
    compute_root_from_path(element, index, path, path_len, out, hash_kind);
    return ZK_OK;
}
int zk_compute_root_from_path(uint32_t element, size_t index, const uint8_t* path, size_t path_len, uint8_t out[32]) {
    return zk_compute_root_from_path_ex(element, index, path, path_len, out, ZK_HASH_SHA256);
}

// ---- Channel -------------------------------------------------------------------

// ---- FRI tail -----------------------------------------------------------------------
// The last FRI layers of a proof whose earlier layers live elsewhere (the sharded prover hands
// over once a layer is small enough to be replicated, shard.hip).  Layer rho0 of a
// (log_n, log_b) proof is layer 0 of a domain with n' = n >> rho0 and shift w^(2^rho0), so the tail
// is the ordinary fused fold + commit loop of prove_resident on that domain, driven by the caller's
// channel.
int zk_tail_create(int device, uint32_t log_n_tail, uint32_t log_b, uint32_t shift, zk_ctx** out) {
    if (!out) return fail(ZK_ERR_INVALID, "zk_tail_create: out is null");
    *out = nullptr;
    if (log_n_tail < 1 || log_b < 1 || log_b > 5 || log_n_tail + log_b > 30)
        return fail(ZK_ERR_INVALID, "zk_tail_create: bad sizes (%u, %u)", log_n_tail, log_b);
    return ctx_make(device, log_n_tail, log_b, shift, true, out);
}

int zk_tail_run(zk_ctx* c, const uint32_t* d_layer0, void* src_stream, zk_channel* chan, int hash_kind,
                uint32_t* betas_out, uint8_t* roots_out, uint32_t* free_term_out) {
    if (!c || !c->tail || !d_layer0 || !chan || !betas_out || !roots_out || !free_term_out)
        return fail(ZK_ERR_INVALID, "zk_tail_run: bad argument");
    if (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) return fail(ZK_ERR_INVALID, "zk_tail_run: unknown hash %d", hash_kind);
    HIPCHK(hipSetDevice(c->device));
    c->hash = hash_kind;
    HIPCHK(hipStreamSynchronize((hipStream_t)src_stream));           // the producer of d_layer0 ran on another stream
    HIPCHK(hipMemcpyAsync(c->d_layers + c->layer_off[1], d_layer0, c->N * 4, hipMemcpyDeviceToDevice, c->stream));
    Channel& ch = chan->ch;
    uint8_t root[32];
    int rc;
    begin_proof(c);
    if ((rc = do_merkle(c, 1, true))) return rc;                       // prover.rs:214 for the handed-over layer
    if ((rc = read_commit(c, 1, root))) return rc;
    ch.commit_hash(root);                                              // prover.rs:224
    memcpy(roots_out, root, 32);
    for (uint32_t r = 0; r < c->R; ++r) {                              // prover.rs:198-225
        uint32_t beta = betas_out[r] = ch.get_u32();
        if ((rc = fri_round_commit(c, r, beta, root))) return rc;
        ch.commit_hash(root);
        memcpy(roots_out + 32 * (r + 1), root, 32);
    }
    if ((rc = last_layer_value(c, free_term_out))) return rc;
    return flush_host_parts(c);                                        // zk_tail_open reads the device arrays
}

// Openings of the tail layers for global query index x (prover.rs:280-289): for tail layer i < R,
// vals_out[2i], vals_out[2i+1] = layer[xi], layer[nx]; the two paths (L - i digests each) follow each
// other in paths_out.  Total digests: sum 2 (L - i).
int zk_tail_open(zk_ctx* c, size_t x, uint32_t* vals_out, uint8_t* paths_out) {
    if (!c || !c->tail || !vals_out || !paths_out) return fail(ZK_ERR_INVALID, "zk_tail_open: bad argument");
    int rc = tail_open_begin(c, x);
    return rc ? rc : tail_open_end(c, vals_out, paths_out);
}
int zk_channel_new(zk_channel** out) {
    if (!out) return fail(ZK_ERR_INVALID, "null argument");
    *out = new (std::nothrow) zk_channel();
    return *out ? ZK_OK : fail(ZK_ERR_NOMEM, "out of host memory");
}
int zk_channel_free(zk_channel* ch) { delete ch; return ZK_OK; }
// Adopts the two fields of a Channel kept elsewhere (channel.rs:6-9): the caller's Rust Channel crosses the FFI as
// (state, data) and comes back through zk_channel_state / zk_channel_data.
int zk_channel_import(zk_channel* ch, const uint8_t state[32], const uint8_t* data, size_t n) {
    if (!ch || !state || (!data && n)) return fail(ZK_ERR_INVALID, "null argument");
    memcpy(ch->ch.state, state, 32);
    ch->ch.data.assign(data, data + n);
    return ZK_OK;
}
int zk_channel_commit(zk_channel* ch, const uint8_t* bytes, size_t n) {
    if (!ch || (!bytes && n)) return fail(ZK_ERR_INVALID, "null argument");
    ch->ch.commit_bytes(bytes, n);
    return ZK_OK;
}
int zk_channel_get_u32(zk_channel* ch, uint32_t* out) {
    if (!ch || !out) return fail(ZK_ERR_INVALID, "null argument");
    *out = ch->ch.get_u32();
    return ZK_OK;
}
int zk_channel_state(const zk_channel* ch, uint8_t out[32]) {
    if (!ch || !out) return fail(ZK_ERR_INVALID, "null argument");
    memcpy(out, ch->ch.state, 32);
    return ZK_OK;
}
size_t zk_channel_data_len(const zk_channel* ch) { return ch ? ch->ch.data.size() : 0; }
int zk_channel_data(const zk_channel* ch, uint8_t* out, size_t cap) {
    if (!ch || !out) return fail(ZK_ERR_INVALID, "null argument");
    if (ch->ch.data.size() > cap) return fail(ZK_ERR_BUFFER, "buffer too small");
    memcpy(out, ch->ch.data.data(), ch->ch.data.size());
    return ZK_OK;
}

// ---- domains and device-pointer primitives ------------------------------------------
static Profiler g_dev_prof;                    // optional timing of the zk_dev_* launches
static zk_kernel_stat g_dev_kstat[K_COUNT] = {};
static Profiler* dev_prof() { return g_dev_prof.mask ? &g_dev_prof : nullptr; }

int zk_dev_set_profiling(uint32_t class_mask) {
    g_dev_prof.mask = class_mask & ((1u << K_COUNT) - 1u);
    return ZK_OK;
}
int zk_dev_kernel_stats(zk_kernel_stat* out, size_t count, int reset) {
    if (!out && count) return fail(ZK_ERR_INVALID, "zk_dev_kernel_stats: null argument");
    for (auto& r : g_dev_prof.recs) {
        float ms = 0;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        g_dev_kstat[r.cls].launches += 1; g_dev_kstat[r.cls].ms += ms; g_dev_kstat[r.cls].bytes += r.bytes; g_dev_kstat[r.cls].ops += r.ops;
        g_dev_prof.pool.push_back(r.a); g_dev_prof.pool.push_back(r.b);
    }
    g_dev_prof.recs.clear();
    if (int rc = put_kernel_stats(out, count, g_dev_kstat, "zk_dev_kernel_stats")) return rc;
    if (reset) for (int i = 0; i < K_COUNT; ++i) g_dev_kstat[i] = zk_kernel_stat{};
    return ZK_OK;
}

int zk_dom_create(int device, uint32_t log_n, uint32_t log_b, uint32_t shift, int fold_only, zk_dom** out) {
    if (!out) return fail(ZK_ERR_INVALID, "zk_dom_create: out is null");
    return dom_make(device, log_n, log_b, shift, fold_only != 0, nullptr, out);
}
int zk_dom_destroy(zk_dom* d) {
    if (d) { (void)hipSetDevice(d->device); dom_free(d); }
    return ZK_OK;
}
int zk_dev_lde(const zk_dom* d, const uint32_t* d_trace, uint32_t* d_coef, uint32_t* d_out, void* stream) {
    if (!d || !d_trace || !d_coef || !d_out) return fail(ZK_ERR_INVALID, "zk_dev_lde: null argument");
    if (!d->d_inv_xm1) return fail(ZK_ERR_STATE, "zk_dev_lde: fold-only domain");
    HIPCHK(hipSetDevice(d->device));
    return dom_lde(d, d_trace, d_coef, d_out, (hipStream_t)stream, dev_prof());
}
int zk_dev_compose(const zk_dom* d, const uint32_t* d_f, uint32_t* d_cp, uint32_t first, uint32_t last,
                   const uint32_t alpha_raw[3], void* stream) {
    if (!d || !d_f || !d_cp || !alpha_raw) return fail(ZK_ERR_INVALID, "zk_dev_compose: null argument");
    if (!d->d_inv_xm1) return fail(ZK_ERR_STATE, "zk_dev_compose: fold-only domain");
    HIPCHK(hipSetDevice(d->device));
    return dom_compose(d, d_f, d_cp, first, last, alpha_raw, (hipStream_t)stream, dev_prof());
}
int zk_dev_fri_fold(const zk_dom* d, const uint32_t* d_in, uint32_t* d_out, uint32_t log_m, uint32_t round,
                    uint32_t beta_raw, void* stream) {
    if (!d || !d_in || !d_out) return fail(ZK_ERR_INVALID, "zk_dev_fri_fold: null argument");
    HIPCHK(hipSetDevice(d->device));
    return dom_fold(d, d_in, d_out, log_m, round, beta_raw, (hipStream_t)stream, dev_prof());
}
int zk_dev_trace_fibsq_batch(const uint32_t* d_a0, const uint32_t* d_a1, uint32_t batch, uint32_t count, uint32_t* d_out, void* stream) {
    if ((batch && (!d_a0 || !d_a1 || !d_out))) return fail(ZK_ERR_INVALID, "zk_dev_trace_fibsq_batch: null argument");
    HIPCHK(launch_trace_fibsq_batch(d_a0, d_a1, batch, count, d_out, (hipStream_t)stream));
    return ZK_OK;
}
int zk_trace_fibsq_batch_host(int device, const uint32_t* a0, const uint32_t* a1, uint32_t batch, uint32_t count, uint32_t* out) {
    if (batch && (!a0 || !a1 || !out)) return fail(ZK_ERR_INVALID, "zk_trace_fibsq_batch_host: null argument");
    if (!batch || !count) return ZK_OK;
    HIPCHK(hipSetDevice(device));
    uint32_t *d0 = nullptr, *d1 = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(&d0, batch * 4)); HIPCHK(hipMalloc(&d1, batch * 4));
    hipError_t e = hipMalloc(&dout, (size_t)batch * count * 4);
    int rc = ZK_OK;
    if (e != hipSuccess) rc = fail(ZK_ERR_NOMEM, "hipMalloc failed");
    if (!rc && (hipMemcpy(d0, a0, batch * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d1, a1, batch * 4, hipMemcpyHostToDevice) != hipSuccess)) rc = fail(ZK_ERR_HIP, "H2D failed");
    if (!rc && launch_trace_fibsq_batch(d0, d1, batch, count, dout, nullptr) != hipSuccess) rc = fail(ZK_ERR_HIP, "launch failed");
    if (!rc && hipMemcpy(out, dout, (size_t)batch * count * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ZK_ERR_HIP, "D2H failed");
    (void)hipFree(d0); (void)hipFree(d1); if (dout) (void)hipFree(dout);
    return rc;
}
int zk_dev_interleave(const uint32_t* d_in, uint32_t* d_out, uint32_t log_parts, uint32_t log_cnt, void* stream) {
    if (!d_in || !d_out || log_parts + log_cnt > 31) return fail(ZK_ERR_INVALID, "zk_dev_interleave: bad argument");
    HIPCHK(launch_interleave(d_in, d_out, log_parts, log_cnt, (hipStream_t)stream));
    return ZK_OK;
}
int zk_dev_gather(const uint32_t* d_src, const uint64_t* d_offsets, uint32_t count, uint32_t words, uint32_t* d_out, void* stream) {
    if (!d_src || (!d_offsets && count) || (!d_out && count)) return fail(ZK_ERR_INVALID, "zk_dev_gather: null argument");
    HIPCHK(launch_gather(d_src, d_offsets, count, words, d_out, (hipStream_t)stream, dev_prof()));
    return ZK_OK;
}

// ---- stand-alone primitives --------------------------------------------------------
int zk_dev_merkle_build_ex(const uint32_t* d_vals, uint32_t log_m, uint32_t* d_nodes, void* stream, int hash_kind) {
    if (!d_vals || !d_nodes || log_m > 30 || (hash_kind != 0 && hash_kind != 1)) return fail(ZK_ERR_INVALID, "zk_dev_merkle_build: bad argument");
    HIPCHK(launch_merkle_build(d_vals, log_m, d_nodes, (hipStream_t)stream, dev_prof(), MailArgs{}, hash_kind));
    return ZK_OK;
}
int zk_dev_merkle_build_interleaved(const uint32_t* d_recv, uint32_t log_parts, uint32_t log_cnt, uint32_t* d_nodes, void* stream, int hash_kind) {
    if (!d_recv || !d_nodes || log_parts + log_cnt > 30 || (hash_kind != 0 && hash_kind != 1))
        return fail(ZK_ERR_INVALID, "zk_dev_merkle_build_interleaved: bad argument");
    HIPCHK(launch_merkle_build_interleaved(d_recv, log_parts, log_cnt, d_nodes, (hipStream_t)stream, dev_prof(), hash_kind));
    return ZK_OK;
}
// ---- committer: zk_dev_merkle_build* + root on the host, with the tree top finished by the calling thread ----
struct zk_committer {
    int device = 0;
    uint32_t* h_mail = nullptr;     // pinned, mapped: MailArgs layout
    uint32_t* d_mail = nullptr;
    uint32_t* d_counter = nullptr;
    bool counters_dirty = false;    // a commit launch was waited for in vain: zero the counters before the next one (as zk_ctx)
    uint32_t* h_stage = nullptr;    // pinned, mapped: host-built top nodes of up to kSlots commits, then kSlots ScatterSegs
    uint32_t* d_stage = nullptr;
    uint32_t seq = 0, top = 0;
    // lazy mode (committer_set_lazy; the sharded prover): the host-built tops of successive commits collect in the staging
    // buffer and ONE scatter launch copies them into the tree array when somebody needs whole trees on the device
    // (committer_flush) -- not one launch per commit in front of the caller's next kernel.  base: the array every d_nodes of
    // a lazy commit points into.
    static constexpr uint32_t kSlots = 40;
    // a slot holds a commit's whole host-side scratch: the 2^top posted digests BEHIND the 2^top - 1 nodes built above them
    // (committer_collect: 8 (2 cnt - 1) words).  Round 5 sized it for the nodes alone, so a commit spilled into the next,
    // still empty, slot and the 40th would have run over the segment table (never reached: <= 32 commits per flush)
    static constexpr size_t kSlotWords = (size_t)16 << kHostTopSingle;
    bool lazy = false;
    uint32_t* base = nullptr;
    uint32_t n_pending = 0;
    double pending_words = 0;
    // h_stage is read by the scatter launch of the previous commit, which may sit on ANY stream the caller passed:
    // its completion is awaited (an event, normally long signalled) before the next commit overwrites the buffer
    hipEvent_t stage_free = nullptr;
    bool stage_busy = false;
    // optional: consulted while waiting for the posted digests (shard.hip: has a peer left the proof?)
    int (*poll)(void*) = nullptr;
    void* poll_user = nullptr;
    double timeout_s = 30.0;        // bound of that wait (shard.hip passes the prover's own: the launch may sit behind an exchange)
};
extern "C++" {
namespace zk { namespace impl {
void committer_set_poll(zk_committer* k, int (*poll)(void*), void* user, double timeout_s) {
    if (k) { k->poll = poll; k->poll_user = user; if (timeout_s > 0) k->timeout_s = timeout_s; }
}
} }
}
int zk_committer_destroy(zk_committer* k) {
    if (!k) return ZK_OK;
    (void)hipSetDevice(k->device);
    (void)hipDeviceSynchronize();                      // a scatter launch may still read the staging buffer
    if (k->stage_free) (void)hipEventDestroy(k->stage_free);
    if (k->h_mail) (void)hipHostFree(k->h_mail);
    if (k->h_stage) (void)hipHostFree(k->h_stage);
    if (k->d_counter) (void)hipFree(k->d_counter);
    delete k;
    return ZK_OK;
}
int zk_committer_create(int device, zk_committer** out) {
    if (!out) return fail(ZK_ERR_INVALID, "zk_committer_create: out is null");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    zk_committer* k = new (std::nothrow) zk_committer();
    if (!k) return fail(ZK_ERR_NOMEM, "out of host memory");
    k->device = device;
    static_assert(zk_committer::kSlots * zk_committer::kSlotWords >= ((size_t)16 << kMaxHostLog), "staging: one eager commit fits");
    static_assert(zk_committer::kSlotWords >= 8 * ((size_t)2 << kHostTopSingle) - 8, "staging: a lazy commit's nodes AND posted digests fit its own slot");
    const size_t stage_bytes = zk_committer::kSlots * zk_committer::kSlotWords * 4 + zk_committer::kSlots * sizeof(ScatterSeg);
    hipError_t e = hipHostMalloc((void**)&k->h_mail, kMailValsOff * 4, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&k->d_mail, k->h_mail, 0);
    if (e == hipSuccess) e = hipHostMalloc((void**)&k->h_stage, stage_bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&k->d_stage, k->h_stage, 0);
    if (e == hipSuccess) e = hipMalloc((void**)&k->d_counter, 64);
    if (e == hipSuccess) e = hipMemset(k->d_counter, 0, 64);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&k->stage_free, hipEventDisableTiming);
    if (e != hipSuccess) { zk_committer_destroy(k); return fail(ZK_ERR_HIP, "zk_committer_create: %s", hipGetErrorString(e)); }
    memset(k->h_mail, 0, kMailValsOff * 4);
    k->top = host_sha_available() ? 8 : 0;
    *out = k;
    return ZK_OK;
}
// Hand-over depth of later commits, as zk_ctx_set_host_levels' top_log (0: the device builds every tree to the root).
int zk_committer_set_top(zk_committer* k, uint32_t top_log) {
    if (!k) return fail(ZK_ERR_INVALID, "null committer");
    if (top_log > kHostTopSingle) return fail(ZK_ERR_INVALID, "zk_committer_set_top: need top_log <= %u (one thread reduces the top)", kHostTopSingle);
    k->top = top_log;
    return ZK_OK;
}
// The last-arriver counters of the next commit launch; zeroed first when an earlier launch was waited for in vain.
static uint32_t* committer_counter(zk_committer* k, hipStream_t s) {
    if (k->counters_dirty) { (void)hipMemsetAsync(k->d_counter, 0, 64, s); k->counters_dirty = false; }
    return k->d_counter;
}
// Waits for the digests a commit launch posted; with a hand-over depth, hashes the levels above on this thread
// and queues the copy of those nodes into d_nodes.
static ScatterSeg* committer_segs(zk_committer* k, bool device) {
    return reinterpret_cast<ScatterSeg*>((device ? k->d_stage : k->h_stage) + zk_committer::kSlots * zk_committer::kSlotWords);
}
// lazy mode: ONE scatter launch for the host-built tops collected since the last flush
static int committer_flush_impl(zk_committer* k, hipStream_t s) {
    if (!k->n_pending) return ZK_OK;
    HIPCHK(launch_scatter(k->d_stage, committer_segs(k, true), k->n_pending, k->pending_words, k->base, nullptr, s, dev_prof()));
    HIPCHK(hipEventRecord(k->stage_free, s));
    k->stage_busy = true;
    k->n_pending = 0;
    k->pending_words = 0;
    return ZK_OK;
}
static int committer_collect(zk_committer* k, const MailArgs& m, uint32_t* d_nodes, hipStream_t s, uint8_t root_out[32]) {
    int rc = wait_flag(k->h_mail, m.seq, s, k->poll, k->poll_user, k->timeout_s);
    if (rc) { k->counters_dirty = true; return rc; }
    if (!m.top) { digest_words_to_bytes(k->h_mail + kMailDigests, root_out); return ZK_OK; }
    const size_t cnt = (size_t)1 << m.top;
    const bool lazy = k->lazy && k->base && d_nodes >= k->base;
    if (lazy && k->n_pending == zk_committer::kSlots)
        if ((rc = committer_flush_impl(k, s))) return rc;
    if (k->stage_busy && (!lazy || k->n_pending == 0)) {      // the previous scatter launch is done with the staging buffer
        HIPCHK(hipEventSynchronize(k->stage_free));
        k->stage_busy = false;
    }
    const uint32_t slot = lazy ? k->n_pending : 0u;
    uint32_t* nodes = k->h_stage + (size_t)slot * zk_committer::kSlotWords;
    memcpy(nodes + 8 * (cnt - 1), k->h_mail + kMailDigests, cnt * 32);
    host_sha_reduce(nodes, m.top);
    digest_words_to_bytes(nodes, root_out);
    ScatterSeg* seg = committer_segs(k, false) + slot;
    *seg = ScatterSeg{(uint64_t)slot * zk_committer::kSlotWords, lazy ? (uint64_t)(d_nodes - k->base) : 0u, (uint32_t)((cnt - 1) * 8), 0};
    if (lazy) {
        k->n_pending += 1;
        k->pending_words += (double)seg->words;
        return ZK_OK;
    }
    HIPCHK(launch_scatter(k->d_stage, committer_segs(k, true), 1, (double)seg->words, d_nodes, nullptr, s, dev_prof()));
    HIPCHK(hipEventRecord(k->stage_free, s));
    k->stage_busy = true;
    return ZK_OK;
}
extern "C++" {
namespace zk { namespace impl {
void committer_set_lazy(zk_committer* k, uint32_t* trees_base) {
    if (k) { k->lazy = trees_base != nullptr; k->base = trees_base; }
}
int committer_flush(zk_committer* k, hipStream_t s) {
    if (!k) return ZK_OK;
    HIPCHK(hipSetDevice(k->device));
    return committer_flush_impl(k, s);
}
void committer_drop_pending(zk_committer* k) {
    if (k) { k->n_pending = 0; k->pending_words = 0; }
}
} }
}

// Tree over 2^(log_parts + log_cnt) leaves (log_parts = 0: d_src in natural order; else in all-to-all order as
// for zk_dev_merkle_build_interleaved), root returned to the host.  The device stops at depth `top`, the calling
// thread hashes the levels above and a stream-ordered copy completes d_nodes (merkle.rs:14-51 either way).
int zk_dev_merkle_commit(zk_committer* k, const uint32_t* d_src, uint32_t log_parts, uint32_t log_cnt, uint32_t* d_nodes, void* stream,
                         int hash_kind, uint8_t root_out[32]) {
    if (!k || !d_src || !d_nodes || !root_out || log_parts + log_cnt > 30 || (hash_kind != 0 && hash_kind != 1))
        return fail(ZK_ERR_INVALID, "zk_dev_merkle_commit: bad argument");
    HIPCHK(hipSetDevice(k->device));
    const uint32_t log_m = log_parts + log_cnt;
    hipStream_t s = (hipStream_t)stream;
    MailArgs m;
    m.mailbox = k->d_mail;
    m.seq = ++k->seq;
    m.counter = committer_counter(k, s);
    m.top = (hash_kind == 0 && k->top && log_m > k->top) ? k->top : 0;
    if (log_parts) HIPCHK(launch_merkle_build_interleaved(d_src, log_parts, log_cnt, d_nodes, s, dev_prof(), hash_kind, m));
    else HIPCHK(launch_merkle_build(d_src, log_m, d_nodes, s, dev_prof(), m, hash_kind));
    return committer_collect(k, m, d_nodes, s, root_out);
}

// cp over this rank's block of a sharded proof, computed from the block of f it holds in all-to-all order, and the
// subtree over it (kernels.hpp: ComposeBlockArgs; prover.rs:101-176 on the positions of the block).  glob: the GLOBAL
// domain (fold-only is enough: power table, x^n - 1 values, g^-k); geom: halo, lg, log_cnt, log_m, halo_stride, e0 and
// geom.a.f / geom.a.inv_xm1 (receive buffer, range table) filled in by the caller; everything else of geom.a is set here.
// enqueued (optional) is called between the launches and the wait for the posted digests.
extern "C++" {
namespace zk {
namespace impl {
int dev_compose_block_commit(zk_committer* k, const zk_dom* glob, ComposeBlockArgs geom, uint32_t first, uint32_t last, const uint32_t alpha_raw[3],
                             uint32_t* d_nodes, hipStream_t s, int hash_kind, uint8_t root_out[32], int (*enqueued)(void*), void* user) {
    if (!k || !glob || !geom.a.f || !geom.a.inv_xm1 || !geom.halo || !d_nodes || !root_out || (hash_kind != 0 && hash_kind != 1) ||
        geom.log_m > 30 || geom.lg + geom.log_cnt > geom.log_m)
        return fail(ZK_ERR_INVALID, "dev_compose_block_commit: bad argument");
    if (((size_t)1 << geom.log_m) < 2 * glob->B || geom.halo_stride < (2 * glob->B) >> geom.lg)
        return fail(ZK_ERR_INVALID, "dev_compose_block_commit: a block of 2^%u leaves is shorter than the 2B = %zu taps", geom.log_m, 2 * glob->B);
    HIPCHK(hipSetDevice(k->device));
    const uint32_t* f = geom.a.f;
    const uint32_t* inv = geom.a.inv_xm1;
    int rc = compose_args(glob, f, nullptr, first, last, alpha_raw, geom.a);
    if (rc) return rc;
    geom.a.inv_xm1 = inv;
    MailArgs m;
    m.mailbox = k->d_mail;
    m.seq = ++k->seq;
    m.counter = committer_counter(k, s);
    m.top = (hash_kind == 0 && k->top && geom.log_m > k->top) ? k->top : 0;
    HIPCHK(launch_compose_block_merkle(geom, d_nodes, s, dev_prof(), m, hash_kind));
    if (enqueued)                                   // the caller's next launches go behind the hashing before this thread waits for the digests
        if ((rc = enqueued(user))) return rc;
    return committer_collect(k, m, d_nodes, s, root_out);
}
}  // namespace impl
}  // namespace zk
}  // extern "C++"

// The same hand-over for a tree built in chunks (zk_dev_merkle_build_chunk): the latency-bound top of the
// whole tree down to depth `top` on the device, the rest on the calling thread, root returned.
int zk_dev_merkle_commit_finish(zk_committer* k, uint32_t* d_nodes, uint32_t log_m, uint32_t log_chunks, void* stream, int hash_kind,
                                uint8_t root_out[32]) {
    if (!k || !d_nodes || !root_out || log_m > 30 || log_chunks > 10 || log_chunks > log_m || (hash_kind != 0 && hash_kind != 1))
        return fail(ZK_ERR_INVALID, "zk_dev_merkle_commit_finish: bad argument");
    HIPCHK(hipSetDevice(k->device));
    hipStream_t s = (hipStream_t)stream;
    MailArgs m;
    m.mailbox = k->d_mail;
    m.seq = ++k->seq;
    m.counter = committer_counter(k, s);
    // the finish pass starts at the hand-over depth of the chunk builds; the host takes over only below that
    m.top = (hash_kind == 0 && k->top && merkle_finish_start_depth(log_m, log_chunks) > k->top) ? k->top : 0;
    HIPCHK(launch_merkle_finish(d_nodes, log_m, log_chunks, s, dev_prof(), hash_kind, m));
    return committer_collect(k, m, d_nodes, s, root_out);
}

int zk_dev_merkle_build_chunk(const uint32_t* d_recv, uint32_t log_parts, uint32_t log_cnt, uint32_t* d_nodes, uint32_t log_m,
                              uint32_t chunk, void* stream, int hash_kind) {
    if (!d_recv || !d_nodes || log_m > 30 || log_parts + log_cnt > log_m || (hash_kind != 0 && hash_kind != 1) ||
        chunk >= (1u << (log_m - log_parts - log_cnt)))
        return fail(ZK_ERR_INVALID, "zk_dev_merkle_build_chunk: bad argument");
    HIPCHK(launch_merkle_build_chunk(d_recv, log_parts, log_cnt, d_nodes, log_m, chunk, (hipStream_t)stream, dev_prof(), hash_kind));
    return ZK_OK;
}
int zk_dev_merkle_finish(uint32_t* d_nodes, uint32_t log_m, uint32_t log_chunks, void* stream, int hash_kind) {
    if (!d_nodes || log_m > 30 || log_chunks > 10 || log_chunks > log_m || (hash_kind != 0 && hash_kind != 1))
        return fail(ZK_ERR_INVALID, "zk_dev_merkle_finish: bad argument");
    HIPCHK(launch_merkle_finish(d_nodes, log_m, log_chunks, (hipStream_t)stream, dev_prof(), hash_kind));
    return ZK_OK;
}
int zk_dev_set_merkle_latency_log(uint32_t log_nodes) {
    if (!set_merkle_latency_log(log_nodes)) return fail(ZK_ERR_INVALID, "zk_dev_set_merkle_latency_log: need 12 <= log_nodes <= 24 (0 = default)");
    return ZK_OK;
}
int zk_dev_merkle_build(const uint32_t* d_vals, uint32_t log_m, uint32_t* d_nodes, void* stream) {
    return zk_dev_merkle_build_ex(d_vals, log_m, d_nodes, stream, ZK_HASH_SHA256);
}

int zk_dev_merkle_node(const uint32_t* d_nodes, size_t index, uint8_t out[32], void* stream) {
    if (!d_nodes || !out) return fail(ZK_ERR_INVALID, "zk_dev_merkle_node: null argument");
    uint32_t w[8];
    HIPCHK(hipMemcpyAsync(w, d_nodes + index * 8, 32, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    digest_words_to_bytes(w, out);
    return ZK_OK;
}

int zk_merkle_build_host(int device, const uint32_t* vals, size_t m, uint8_t* nodes_out) {
    return zk_merkle_build_host_ex(device, vals, m, nodes_out, ZK_HASH_SHA256);
}

int zk_merkle_build_host_ex(int device, const uint32_t* vals, size_t m, uint8_t* nodes_out, int hash_kind) {
    if (!vals || !nodes_out) return fail(ZK_ERR_INVALID, "zk_merkle_build_host: null argument");
    if (hash_kind != 0 && hash_kind != 1) return fail(ZK_ERR_INVALID, "zk_merkle_build_host: unknown hash %d", hash_kind);
    if (m == 0 || (m & (m - 1)) || m > ((size_t)1 << 30))      // merkle.rs:16-21 asserts a power of two
        return fail(ZK_ERR_INVALID, "zk_merkle_build_host: size %zu is not a power of two (merkle.rs:18)", m);
    uint32_t log_m = 0;
    while (((size_t)1 << log_m) < m) ++log_m;
    HIPCHK(hipSetDevice(device));
    uint32_t *d_vals = nullptr, *d_nodes = nullptr;
    size_t words = (2 * m - 1) * 8;
    HIPCHK(hipMalloc(&d_vals, m * 4));
    hipError_t e = hipMalloc(&d_nodes, words * 4);
    if (e != hipSuccess) { (void)hipFree(d_vals); return fail(ZK_ERR_NOMEM, "hipMalloc failed"); }
    int rc = ZK_OK;
    std::vector<uint32_t> host(words);
    do {
        if (hipMemcpy(d_vals, vals, m * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "H2D failed"); break; }
        if (launch_merkle_build(d_vals, log_m, d_nodes, nullptr, nullptr, MailArgs{}, hash_kind) != hipSuccess) { rc = fail(ZK_ERR_HIP, "merkle launch failed"); break; }
        if (hipMemcpy(host.data(), d_nodes, words * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "D2H failed: %s", hipGetErrorString(hipGetLastError())); break; }
    } while (0);
    (void)hipFree(d_vals); (void)hipFree(d_nodes);
    if (rc) return rc;
    for (size_t i = 0; i < 2 * m - 1; ++i) digest_words_to_bytes(host.data() + 8 * i, nodes_out + 32 * i);
    return ZK_OK;
}

// Roofline probe: the compiled inner hash in a dependent chain, `launches` launches back to back (steady state,
// no residency tail), waves_per_simd resident waves on every SIMD.  Measurement only.
int zk_probe_hash_chain(int device, int hash_kind, uint32_t waves_per_simd, uint32_t hashes, uint32_t launches, zk_chain_probe* out) {
    if (!out || (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) || waves_per_simd < 1 || waves_per_simd > 8 || !hashes || !launches)
        return fail(ZK_ERR_INVALID, "zk_probe_hash_chain: bad argument");
    if (!abi_bytes(out, "zk_probe_hash_chain")) return ZK_ERR_INVALID;
    zk_chain_probe* const user_out = out;
    zk_chain_probe res{};
    out = &res;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount, blocks = cus * waves_per_simd;   // one 256-thread workgroup = one wave per SIMD
    uint32_t* d_out = nullptr;
    unsigned long long* d_rec = nullptr;
    HIPCHK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    hipError_t e = hipMalloc(&d_rec, (size_t)blocks * 4 * 16);
    if (e != hipSuccess) { (void)hipFree(d_out); return fail(ZK_ERR_NOMEM, "hipMalloc failed"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = ZK_OK;
    std::vector<unsigned long long> rec((size_t)blocks * 8);
    do {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = fail(ZK_ERR_HIP, "hipEventCreate failed"); break; }
        if (launch_hash_chain_probe(hash_kind, blocks, d_out, 3u, hashes, d_rec, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            rc = fail(ZK_ERR_HIP, "probe launch failed"); break;
        }
        (void)hipEventRecord(e0, nullptr);
        for (uint32_t r = 0; r < launches; ++r) (void)launch_hash_chain_probe(hash_kind, blocks, d_out, 12345u + r, hashes, d_rec, nullptr);
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(ZK_ERR_HIP, "probe failed: %s", hipGetErrorString(hipGetLastError())); break; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (hipMemcpy(rec.data(), d_rec, rec.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "D2H failed"); break; }
        std::vector<double> ghz;
        for (size_t i = 0; i + 1 < rec.size(); i += 2)
            if (rec[i + 1]) ghz.push_back((double)rec[i] / (double)rec[i + 1] * 0.1);     // s_memrealtime ticks at 100 MHz
        std::sort(ghz.begin(), ghz.end());
        out->ms = ms;
        out->ns_per_hash_per_simd = (double)ms * 1e6 / ((double)launches * waves_per_simd * hashes);
        out->clock_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
        out->waves_per_simd = waves_per_simd; out->launches = launches; out->hashes = hashes; out->cus = cus;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_out); (void)hipFree(d_rec);
    return rc ? rc : abi_put(user_out, res, "zk_probe_hash_chain");
}

// Test hook: the device's forms of the field hash against each other on `count` (rounded up to 256) pseudo-random and edge inputs.
int zk_probe_fieldhash_forms(int device, uint32_t count, uint32_t seed, uint32_t* mismatches, uint32_t* first_bad) {
    if (!mismatches || count == 0 || count > (1u << 26)) return fail(ZK_ERR_INVALID, "zk_probe_fieldhash_forms: bad argument");
    HIPCHK(hipSetDevice(device));
    uint32_t* d_res = nullptr;
    HIPCHK(hipMalloc(&d_res, 8));
    const uint32_t init[2] = {0u, 0xFFFFFFFFu};
    int rc = ZK_OK;
    uint32_t res[2] = {0, 0};
    if (hipMemcpy(d_res, init, 8, hipMemcpyHostToDevice) != hipSuccess || launch_fieldhash_forms((count + 255) / 256, seed, d_res, nullptr) != hipSuccess ||
        hipMemcpy(res, d_res, 8, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(ZK_ERR_HIP, "zk_probe_fieldhash_forms: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_res);
    if (rc) return rc;
    *mismatches = res[0];
    if (first_bad) *first_bad = res[1];
    return ZK_OK;
}

int zk_ntt_host(int device, uint32_t* data, uint32_t log_m, int inverse) {
    if (!data || log_m < 1 || log_m > 30) return fail(ZK_ERR_INVALID, "zk_ntt_host: bad argument");
    size_t m = (size_t)1 << log_m;
    for (size_t i = 0; i < m; ++i)
        if (data[i] >= P) return fail(ZK_ERR_INVALID, "zk_ntt_host: data[%zu] is not a canonical residue", i);
    HIPCHK(hipSetDevice(device));
    uint32_t root = root_of_unity(log_m);
    DevTable T;
    int rc = build_table(inverse ? invmod(root) : root, log_m, &T);
    if (rc) return rc;
    Plan pl = make_plan(log_m);
    uint32_t *d_a = nullptr, *d_b = nullptr;
    HIPCHK(hipMalloc(&d_a, m * 4));
    HIPCHK(hipMalloc(&d_b, m * 4));
    do {
        if (hipMemcpy(d_a, data, m * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ZK_ERR_HIP, "H2D failed"); break; }
        uint32_t* res = d_a;
        if (inverse) {
            if ((rc = run_dif(d_a, d_a, log_m, pl, T.view(), log_m, to_mont(invmod((uint32_t)(m % P))), nullptr))) break;
            if (launch_digit_reverse(d_a, d_b, log_m, pl.nd, pl.bits, 1, nullptr) != hipSuccess) { rc = fail(ZK_ERR_HIP, "launch failed"); break; }
            res = d_b;
        } else {
            if (launch_digit_reverse(d_a, d_b, log_m, pl.nd, pl.bits, 0, nullptr) != hipSuccess) { rc = fail(ZK_ERR_HIP, "launch failed"); break; }
            if ((rc = run_dit(d_b, log_m, pl, T.view(), log_m, nullptr))) break;
            res = d_b;
        }
        if (hipMemcpy(data, res, m * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(ZK_ERR_HIP, "D2H failed: %s", hipGetErrorString(hipGetLastError())); break; }
    } while (0);
    (void)hipFree(d_a); (void)hipFree(d_b);
    free_table(&T);
    return rc;
}

int zk_lde_host(int device, const uint32_t* trace, uint32_t log_n, uint32_t log_b, uint32_t* out) {
    if (!trace || !out) return fail(ZK_ERR_INVALID, "zk_lde_host: null argument");
    zk_ctx* c = nullptr;
    int rc = zk_ctx_create(device, log_n, log_b, &c);
    if (rc) return rc;
    rc = zk_trace_upload(c, trace, c->n - 1);
    if (!rc) rc = zk_lde(c);
    if (!rc) rc = zk_layer_read(c, 0, 0, c->N, out);
    zk_ctx_destroy(c);
    return rc;
}

}  // extern "C"
