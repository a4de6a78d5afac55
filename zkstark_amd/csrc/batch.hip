// batch.hip -- batched proving: 2^log_batch proofs of one size in lockstep (zk_batch_*).
//
// prover.rs:9-293 is run for every proof of the batch with the SAME kernels as one proof on a domain
// batch times larger: layer l is stored proof-major ([batch][m_l]), so the trees of the batch are the
// bottom of one heap over batch*m_l leaves whose nodes of depth log_batch are the per-proof roots.
// The device posts those (MailArgs.top = log_batch), each proof's own channel absorbs its root and
// draws its own challenges (host threads), and the next launch reads them from a per-proof table.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "host_sha.hpp"
#include "internal.hpp"
#include "pool.hpp"
#include "sha256.hpp"
#include "transcript.hpp"

using namespace zk;
using namespace zk::impl;

struct zk_batch {
    int device = 0;
    std::atomic<bool> busy{false};    // a zk_batch_prove is running: the setters refuse to change the batch under it
    uint32_t log_n = 0, log_b = 0, lb = 0, L = 0, R = 0;
    size_t n = 0, N = 0, B = 0, batch = 0;
    hipStream_t stream = nullptr;
    zk_dom* dom = nullptr;
    uint32_t* d_trace = nullptr;      // [batch][n]: a[0..n-2], 0
    uint32_t* d_coef = nullptr;       // [batch][2n]
    uint32_t* d_layers = nullptr;     // layer l: [batch][m_l]
    uint32_t* d_trees = nullptr;      // tree l: heap over batch*m_l leaves
    uint32_t* d_seed = nullptr;       // [2][batch] seeds of zk_batch_gen_fibsq
    std::vector<size_t> layer_off, tree_off;
    BatchChal* h_chal = nullptr;      // pinned
    BatchChal* d_chal = nullptr;
    uint32_t* h_mail = nullptr;       // pinned, mapped (MailArgs layout)
    uint32_t* d_mail = nullptr;
    uint32_t* d_counter = nullptr;
    bool counters_dirty = false;
    uint32_t mail_seq = 0;
    uint64_t *d_goff = nullptr, *h_goff = nullptr;
    uint32_t *d_gout = nullptr, *h_gout = nullptr;
    uint32_t* h_last = nullptr;       // pinned: [batch][B] last layer / [batch] last trace values
    size_t per_proof_vals = 0, per_proof_digs = 0;   // per query
    uint32_t queries = 1;             // decommitment queries per proof (1 = the reference, prover.rs:263)
    int hash = 0;                     // Merkle hash: 0 = SHA-256 (reference), 1 = field-native
    std::vector<uint32_t> first, last;
    bool have_traces = false;
    size_t device_bytes = 0;
    Pool* pool = nullptr;
    // small batches: the device stops each tree `extra` levels above the per-proof roots and the host threads
    // hash those levels (host_sha.hpp); staged level-major like the batch heap, copied back by scatter_kernel
    bool host_levels = false;
    uint32_t* h_stage = nullptr;      // pinned, mapped
    uint32_t* d_stage = nullptr;
    size_t stage_words = 0, stage_used = 0;
    ScatterSeg* h_segs = nullptr;
    ScatterSeg* d_segs = nullptr;
    uint32_t n_segs = 0;
    double seg_words = 0;
    // A batch of ONE proof is a single proof: it runs on the one-call prover (zk_prove_resident: fused host tail,
    // no per-round challenge table), so the smallest batch is never slower than zk_prove.
    zk_ctx* single = nullptr;
    std::vector<uint32_t> single_trace;
};

namespace {

size_t blayer_size(const zk_batch* b, uint32_t layer) { return layer == 0 ? b->N : (b->N >> (layer - 1)); }
uint32_t blayer_log(const zk_batch* b, uint32_t layer) { return layer == 0 ? b->L : b->L - (layer - 1); }

// Levels of each proof's tree (over 2^log_m leaves) that the host hashes: as many as the mailbox holds
// (2^kMaxHostLog digests for the whole batch), none for the field hash or without SHA extensions.
uint32_t bextra(const zk_batch* b, uint32_t log_m) {
    if (!b->host_levels || b->hash != 0 || b->lb >= kMaxHostLog || log_m < 2) return 0;
    uint32_t h = kMaxHostLog - b->lb;
    if (h > 8) h = 8;                                    // <= 255 nodes per proof on one thread (~8 us), as the single prover
    if (h > log_m - 1) h = log_m - 1;
    return h >= 3 ? h : 0;                               // two levels are not worth a hand-over
}
MailArgs bmail(zk_batch* b, uint32_t log_m) {
    MailArgs m;
    if (b->counters_dirty) { (void)hipMemsetAsync(b->d_counter, 0, 64, b->stream); b->counters_dirty = false; }
    m.mailbox = b->d_mail; m.seq = ++b->mail_seq; m.counter = b->d_counter; m.top = b->lb + bextra(b, log_m);
    return m;
}
// the batch's roots of the last commit launch: [batch][8] state words in the mailbox
int bwait_roots(zk_batch* b) {
    const int rc = wait_flag(b->h_mail, b->mail_seq, b->stream);
    if (rc) b->counters_dirty = true;                     // see zk_ctx::counters_dirty (zkstark.hip)
    return rc;
}
// After bwait_roots: hashes the `extra` host levels of every proof's tree `tree` and returns where the roots
// are ([batch][8] state words): the mailbox itself when the device went all the way.
const uint32_t* bfinish_roots(zk_batch* b, uint32_t tree, uint32_t log_m) {
    const uint32_t h = bextra(b, log_m);
    const uint32_t* posted = b->h_mail + kMailDigests;
    if (!h) return posted;
    const size_t nb = b->batch;
    // stage[d], d < h: [proof][2^d] digests = level lb + d of the batch heap
    std::vector<uint32_t*> lvl(h + 1);
    for (uint32_t dd = 0; dd < h; ++dd) {
        lvl[dd] = b->h_stage + b->stage_used;
        b->stage_used += (nb << dd) * 8;
    }
    lvl[h] = const_cast<uint32_t*>(posted);
    b->pool->run(nb, 1, [&](size_t p) {
        for (uint32_t dd = h; dd-- > 0;) {
            const uint32_t* child = lvl[dd + 1] + ((p << (dd + 1)) * 8);
            uint32_t* out = lvl[dd] + ((p << dd) * 8);
            host_sha_inner_run(child, out, (size_t)1 << dd);
        }
    });
    for (uint32_t dd = 0; dd < h; ++dd) {
        b->h_segs[b->n_segs++] = ScatterSeg{(uint64_t)(lvl[dd] - b->h_stage),
                                            (uint64_t)b->tree_off[tree] + ((((uint64_t)1 << (b->lb + dd)) - 1) * 8),
                                            (uint32_t)((nb << dd) * 8), 0};
        b->seg_words += (double)((nb << dd) * 8);
    }
    return lvl[0];
}

// gather buffers for q queries per proof (offsets in, values + digests out; device and pinned host copies)
int balloc_gather(zk_batch* b, uint32_t q) {
    for (void* p : {(void*)b->d_goff, (void*)b->d_gout}) if (p) (void)hipFree(p);
    for (void* p : {(void*)b->h_goff, (void*)b->h_gout}) if (p) (void)hipHostFree(p);
    b->d_goff = nullptr; b->h_goff = nullptr; b->d_gout = nullptr; b->h_gout = nullptr;
    const size_t slots = b->batch * q * (b->per_proof_vals + b->per_proof_digs);
    const size_t out_words = b->batch * q * (b->per_proof_vals + 8 * b->per_proof_digs);
    HIPCHK(hipMalloc((void**)&b->d_goff, slots * 8));
    HIPCHK(hipMalloc((void**)&b->d_gout, out_words * 4));
    HIPCHK(hipHostMalloc((void**)&b->h_goff, slots * 8));
    HIPCHK(hipHostMalloc((void**)&b->h_gout, out_words * 4));
    b->queries = q;
    return ZK_OK;
}

int bchal_upload(zk_batch* b) {
    HIPCHK(hipMemcpyAsync(b->d_chal, b->h_chal, b->batch * sizeof(BatchChal), hipMemcpyHostToDevice, b->stream));
    return ZK_OK;
}

}  // namespace

extern "C" {

int zk_batch_destroy(zk_batch* b) {
    if (!b) return ZK_OK;
    (void)hipSetDevice(b->device);
    if (b->single) zk_ctx_destroy(b->single);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    dom_free(b->dom);
    for (void* p : {(void*)b->d_trace, (void*)b->d_coef, (void*)b->d_layers, (void*)b->d_trees, (void*)b->d_seed, (void*)b->d_chal,
                    (void*)b->d_counter, (void*)b->d_goff, (void*)b->d_gout})
        if (p) (void)hipFree(p);
    for (void* p : {(void*)b->h_chal, (void*)b->h_mail, (void*)b->h_goff, (void*)b->h_gout, (void*)b->h_last, (void*)b->h_stage})
        if (p) (void)hipHostFree(p);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b->pool;
    delete b;
    return ZK_OK;
}

int zk_batch_create(int device, uint32_t log_n, uint32_t log_b, uint32_t log_batch, zk_batch** out) {
    if (!out) return fail(ZK_ERR_INVALID, "zk_batch_create: out is null");
    *out = nullptr;
    if (log_batch > kMaxHostLog) return fail(ZK_ERR_INVALID, "zk_batch_create: need log_batch <= %u", kMaxHostLog);
    if (int rc0 = check_proof_size("zk_batch_create", log_n, log_b, log_batch)) return rc0;   // the same sizes zk_ctx_create accepts
    HIPCHK(hipSetDevice(device));
    zk_batch* b = new (std::nothrow) zk_batch();
    if (!b) return fail(ZK_ERR_NOMEM, "out of host memory");
    b->device = device;
    b->log_n = log_n; b->log_b = log_b; b->lb = log_batch; b->L = log_n + log_b; b->R = log_n;
    b->n = (size_t)1 << log_n; b->B = (size_t)1 << log_b; b->N = b->n << log_b; b->batch = (size_t)1 << log_batch;
    int rc = ZK_OK;
    auto bail = [&](int code) { zk_batch_destroy(b); return code; };
    if (log_batch == 0) {                                  // one proof: the single prover
        if ((rc = zk_ctx_create(device, log_n, log_b, &b->single))) return bail(rc);
        b->first.assign(1, 0);
        b->last.assign(1, 0);
        b->device_bytes = zk_ctx_device_bytes(b->single);
        *out = b;
        return ZK_OK;
    }
#define HIPCHK_B(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            zk_batch_destroy(b);                                                              \
            return fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                     \
    } while (0)
    HIPCHK_B(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    if ((rc = dom_make(device, log_n, log_b, GEN_W, false, b->stream, &b->dom))) return bail(rc);
    b->device_bytes += b->dom->device_bytes;
    size_t off = 0;
    for (uint32_t l = 0; l <= b->R + 1; ++l) { b->layer_off.push_back(off); off += blayer_size(b, l) * b->batch; }
    const size_t layer_words = off;
    off = 0;
    for (uint32_t l = 0; l <= b->R + 1; ++l) { b->tree_off.push_back(off); off += (2 * blayer_size(b, l) * b->batch - 1) * 8; }
    const size_t tree_words = off;
    auto dm = [&](auto** p, size_t bytes) {
        hipError_t e = hipMalloc((void**)p, bytes ? bytes : 4);
        if (e != hipSuccess) return fail(ZK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        b->device_bytes += bytes;
        return (int)ZK_OK;
    };
    if ((rc = dm(&b->d_trace, b->batch * b->n * 4)) || (rc = dm(&b->d_coef, b->batch * 2 * b->n * 4)) ||
        (rc = dm(&b->d_layers, layer_words * 4)) || (rc = dm(&b->d_trees, tree_words * 4)) || (rc = dm(&b->d_seed, 2 * b->batch * 4)) ||
        (rc = dm(&b->d_chal, b->batch * sizeof(BatchChal))) || (rc = dm(&b->d_counter, 64)))
        return bail(rc);
    HIPCHK_B(hipMemsetAsync(b->d_counter, 0, 64, b->stream));
    HIPCHK_B(hipMemsetAsync(b->d_trace, 0, b->batch * b->n * 4, b->stream));
    // openings of one proof (prover.rs:266-289): 4 + 2R values, 4 L + sum 2 (L - i) digests
    b->per_proof_vals = 4 + 2 * (size_t)b->R;
    b->per_proof_digs = 4 * (size_t)b->L;
    for (uint32_t i = 0; i < b->R; ++i) b->per_proof_digs += 2 * (size_t)(b->L - i);
    if ((rc = balloc_gather(b, 1))) return bail(rc);
    HIPCHK_B(hipHostMalloc((void**)&b->h_chal, b->batch * sizeof(BatchChal)));
    HIPCHK_B(hipHostMalloc((void**)&b->h_last, b->batch * (b->B > 2 ? b->B : 2) * 4));
    HIPCHK_B(hipHostMalloc((void**)&b->h_mail, kMailWords * 4, hipHostMallocMapped | hipHostMallocCoherent));
    memset(b->h_mail, 0, kMailWords * 4);
    memset(b->h_chal, 0, b->batch * sizeof(BatchChal));
    HIPCHK_B(hipHostGetDevicePointer((void**)&b->d_mail, b->h_mail, 0));
    b->stage_words = (size_t)(b->R + 2) * ((size_t)8 << kMaxHostLog);
    const size_t seg_bytes = (size_t)(b->R + 2) * kMaxHostLog * sizeof(ScatterSeg);
    HIPCHK_B(hipHostMalloc((void**)&b->h_stage, b->stage_words * 4 + seg_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_B(hipHostGetDevicePointer((void**)&b->d_stage, b->h_stage, 0));
    b->h_segs = reinterpret_cast<ScatterSeg*>(b->h_stage + b->stage_words);
    b->d_segs = reinterpret_cast<ScatterSeg*>(b->d_stage + b->stage_words);
    b->host_levels = host_sha_available();               // zk_batch_set_host_levels(b, 0) keeps every level on the device
    HIPCHK_B(hipStreamSynchronize(b->stream));
#undef HIPCHK_B
    b->first.assign(b->batch, 0);
    b->last.assign(b->batch, 0);
    unsigned hw = std::thread::hardware_concurrency();
    unsigned want = hw > 1 ? (hw - 1 < 15 ? hw - 1 : 15) : 0;
    if (b->batch < 2) want = 0;
    b->pool = new (std::nothrow) Pool(want);
    if (!b->pool) return bail(fail(ZK_ERR_NOMEM, "out of host memory"));
    *out = b;
    return ZK_OK;
}

size_t zk_batch_size(const zk_batch* b) { return b ? b->batch : 0; }
// One batch is used from one host thread at a time (include/zkstark_amd.h); a setter that arrives while zk_batch_prove runs on
// another thread would pull the pool, the gather buffers or the traces from under it, so it is refused instead.
struct BusyScope {
    zk_batch* b; bool mine;
    explicit BusyScope(zk_batch* b_) : b(b_) { bool expected = false; mine = b->busy.compare_exchange_strong(expected, true, std::memory_order_acq_rel); }
    ~BusyScope() { if (mine) b->busy.store(false, std::memory_order_release); }
};
// (the setters hold the same flag for their own duration, so a prove that starts while a setter replaces the pool is refused too)
#define ZK_BATCH_EXCLUSIVE(b, who)                                                                       \
    BusyScope _excl(b);                                                                                  \
    if (!_excl.mine) return fail(ZK_ERR_STATE, "%s: another call (zk_batch_prove or a setter) is running on this batch", who)
int zk_batch_set_host_levels(zk_batch* b, int on) {
    if (!b) return fail(ZK_ERR_INVALID, "null batch");
    ZK_BATCH_EXCLUSIVE(b, "zk_batch_set_host_levels");
    if (b->single) return zk_ctx_set_host_levels(b->single, on && host_sha_available() ? 8 : 0, on && host_sha_available() ? 9 : 0);
    b->host_levels = on != 0 && host_sha_available();
    return ZK_OK;
}
int zk_batch_set_threads(zk_batch* b, uint32_t threads) {
    if (!b) return fail(ZK_ERR_INVALID, "null batch");
    ZK_BATCH_EXCLUSIVE(b, "zk_batch_set_threads");
    if (threads < 1 || threads > 64) return fail(ZK_ERR_INVALID, "zk_batch_set_threads: need 1 <= threads <= 64");
    if (b->single || !b->pool || b->pool->workers() == threads - 1) return ZK_OK;
    Pool* np = new (std::nothrow) Pool(threads - 1);
    if (!np) return fail(ZK_ERR_NOMEM, "out of host memory");
    delete b->pool;
    b->pool = np;
    return ZK_OK;
}
int zk_batch_set_queries(zk_batch* b, uint32_t n_queries) {
    if (!b) return fail(ZK_ERR_INVALID, "null batch");
    ZK_BATCH_EXCLUSIVE(b, "zk_batch_set_queries");
    if (n_queries < 1 || n_queries > 16) return fail(ZK_ERR_INVALID, "zk_batch_set_queries: need 1 <= n_queries <= 16");
    if (b->single) { b->queries = n_queries; return zk_ctx_set_queries(b->single, n_queries); }
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    return n_queries == b->queries ? (int)ZK_OK : balloc_gather(b, n_queries);
}
int zk_batch_set_hash(zk_batch* b, int hash_kind) {
    if (!b) return fail(ZK_ERR_INVALID, "null batch");
    ZK_BATCH_EXCLUSIVE(b, "zk_batch_set_hash");
    if (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) return fail(ZK_ERR_INVALID, "zk_batch_set_hash: unknown hash %d", hash_kind);
    b->hash = hash_kind;
    if (b->single) return zk_ctx_set_hash(b->single, hash_kind);
    return ZK_OK;
}
size_t zk_batch_device_bytes(const zk_batch* b) { return b ? b->device_bytes : 0; }

// traces: [batch][n-1] canonical residues on the host (prover.rs:32-39 per proof)
static int set_traces_exclusive(zk_batch* b, const uint32_t* traces);
int zk_batch_set_traces(zk_batch* b, const uint32_t* traces) {
    if (!b || !traces) return fail(ZK_ERR_INVALID, "zk_batch_set_traces: null argument");
    ZK_BATCH_EXCLUSIVE(b, "zk_batch_set_traces");
    return set_traces_exclusive(b, traces);
}
static int set_traces_exclusive(zk_batch* b, const uint32_t* traces) {      // the caller holds the batch
    if (b->single) {
        int rc = zk_trace_upload(b->single, traces, b->n - 1);
        if (rc) return rc;
        b->first[0] = traces[0]; b->last[0] = traces[b->n - 2];
        b->have_traces = true;
        return ZK_OK;
    }
    HIPCHK(hipSetDevice(b->device));
    for (size_t p = 0; p < b->batch * (b->n - 1); ++p)
        if (traces[p] >= P) return fail(ZK_ERR_INVALID, "zk_batch_set_traces: value %zu is not a canonical residue", p);
    HIPCHK(hipMemcpy2DAsync(b->d_trace, b->n * 4, traces, (b->n - 1) * 4, (b->n - 1) * 4, b->batch, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (size_t p = 0; p < b->batch; ++p) { b->first[p] = traces[p * (b->n - 1)]; b->last[p] = traces[p * (b->n - 1) + b->n - 2]; }
    b->have_traces = true;
    return ZK_OK;
}

// Fibonacci-square traces generated on the device from per-proof seeds (one lane per trace).
int zk_batch_gen_fibsq(zk_batch* b, const uint32_t* a0, const uint32_t* a1) {
    if (!b || !a0 || !a1) return fail(ZK_ERR_INVALID, "zk_batch_gen_fibsq: null argument");
    ZK_BATCH_EXCLUSIVE(b, "zk_batch_gen_fibsq");
    if (b->single) {                                       // prover.rs:32-39 is serial: one trace gains nothing from the device
        b->single_trace.resize(b->n - 1);
        int rc = zk_trace_fibsq(a0[0], a1[0], b->n - 1, b->single_trace.data());
        if (!rc) rc = set_traces_exclusive(b, b->single_trace.data());
        return rc;
    }
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipMemcpyAsync(b->d_seed, a0, b->batch * 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_seed + b->batch, a1, b->batch * 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(launch_trace_fibsq_batch(b->d_seed, b->d_seed + b->batch, (uint32_t)b->batch, (uint32_t)(b->n - 1), b->d_trace, b->stream, (uint32_t)b->n));
    // a[n-2] of every proof is a public input (prover.rs:42) and a constant of the second constraint
    HIPCHK(hipMemcpy2DAsync(b->h_last, 4, b->d_trace + (b->n - 2), b->n * 4, 4, b->batch, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (size_t p = 0; p < b->batch; ++p) { b->first[p] = a0[p] % P; b->last[p] = b->h_last[p]; }
    b->have_traces = true;
    return ZK_OK;
}

int zk_batch_public_last(const zk_batch* b, uint32_t* out) {
    if (!b || !out) return fail(ZK_ERR_INVALID, "zk_batch_public_last: null argument");
    if (!b->have_traces) return fail(ZK_ERR_STATE, "zk_batch_public_last: no traces");
    memcpy(out, b->last.data(), b->batch * 4);
    return ZK_OK;
}

// generate_proof (prover.rs:9-293) for every resident trace.  proofs_out: [batch][stride] bytes, stride >=
// zk_proof_data_len(log_n, log_b); states_out: [batch][32] (Channel.state of each proof, proof.rs:6).
int zk_batch_prove(zk_batch* b, uint8_t* proofs_out, size_t stride, uint8_t* states_out) {
    if (!b || !proofs_out || !states_out) return fail(ZK_ERR_INVALID, "zk_batch_prove: null argument");
    if (!b->have_traces) return fail(ZK_ERR_STATE, "zk_batch_prove: no traces");
    BusyScope busy(b);
    if (!busy.mine) return fail(ZK_ERR_STATE, "zk_batch_prove: another zk_batch_prove is running on this batch");
    const uint32_t Q = b->queries;
    const int hash = b->hash;
    const size_t plen = proof_data_len(b->log_n, b->log_b, Q);
    if (stride < plen) return fail(ZK_ERR_BUFFER, "zk_batch_prove: stride %zu < proof length %zu", stride, plen);
    if (b->single) {
        size_t len = 0;
        int rc1 = zk_prove_resident(b->single, proofs_out, stride, &len, states_out);
        if (rc1 == ZK_ERR_CHECK) return fail(ZK_ERR_CHECK, "proof 0 of the batch: %s", last_error());
        return rc1;
    }
    HIPCHK(hipSetDevice(b->device));
    const size_t nb = b->batch, N = b->N, B = b->B;
    const uint32_t R = b->R, L = b->L, lb = b->lb;
    const zk_dom* d = b->dom;
    std::vector<Channel> ch(nb);
    for (auto& c : ch) c.data.reserve(plen);
    int rc;
    static const bool timing = getenv("ZK_HOST_TIMING") != nullptr;
    double T0 = now_us(), t_wait = 0;
    auto lap = [&](const char* what) {
        if (!timing) return;
        double t = now_us();
        fprintf(stderr, "[zk batch timing] %-28s %8.1f us\n", what, t - T0);
        T0 = t;
    };
    auto wait_roots = [&]() { double t = now_us(); int r = bwait_roots(b); t_wait += now_us() - t; return r; };
    // f = LDE of every trace, committed (prover.rs:60-85)
    if ((rc = dom_lde(d, b->d_trace, b->d_coef, b->d_layers + b->layer_off[0], b->stream, nullptr, (uint32_t)nb))) return rc;
    b->stage_used = 0; b->n_segs = 0; b->seg_words = 0;
    HIPCHK(launch_merkle_build(b->d_layers + b->layer_off[0], L + lb, b->d_trees + b->tree_off[0], b->stream, nullptr, bmail(b, L), hash));
    // proof-independent part of the composition constants (compose_args with alpha = 1)
    ComposeBatchArgs ca;
    {
        const uint32_t one[3] = {1, 1, 1};
        if ((rc = compose_args(d, b->d_layers + b->layer_off[0], b->d_layers + b->layer_off[1], 0, 0, one, ca.a))) return rc;
        ca.chal = b->d_chal;
    }
    const uint32_t g2 = mulmod(d->g, d->g);
    if ((rc = wait_roots())) return rc;
    const uint32_t* roots = bfinish_roots(b, 0, L);
    b->pool->run(nb, 16, [&](size_t p) {
        uint8_t root[32];
        digest_words_to_bytes(roots + 8 * p, root);
        ch[p].commit_hash(root);                                          // prover.rs:85
        uint32_t a0 = ch[p].get_u32() % P, a1 = ch[p].get_u32() % P, a2 = ch[p].get_u32() % P;   // prover.rs:163-165
        BatchChal& c = b->h_chal[p];
        c.first = b->first[p]; c.last = b->last[p];
        c.alpha0_mont = to_mont(a0); c.alpha1g2_mont = to_mont(mulmod(a1, g2)); c.alpha2_mont = to_mont(a2);
    });
    if ((rc = bchal_upload(b))) return rc;
    HIPCHK(launch_compose_merkle_batch(ca, lb, b->d_trees + b->tree_off[1], b->stream, nullptr, bmail(b, L), hash));   // prover.rs:166-176
    for (uint32_t r = 0; r <= R; ++r) {
        if ((rc = wait_roots())) return rc;
        roots = bfinish_roots(b, 1 + r, L - r);                           // tree 1 + r: 2^(L - r) leaves per proof
        if (r == R) {
            b->pool->run(nb, 32, [&](size_t p) { uint8_t root[32]; digest_words_to_bytes(roots + 8 * p, root); ch[p].commit_hash(root); });
            break;
        }
        const uint32_t winv_half = mulmod(invmod(powmod(d->shift, (uint64_t)1 << r)), invmod(2));
        b->pool->run(nb, 32, [&](size_t p) {
            uint8_t root[32];
            digest_words_to_bytes(roots + 8 * p, root);
            ch[p].commit_hash(root);                                      // prover.rs:180 / :224
            uint32_t beta = ch[p].get_u32() % P;                          // prover.rs:200
            b->h_chal[p].c_mont = to_mont(mulmod(beta, winv_half));
        });
        if ((rc = bchal_upload(b))) return rc;
        FoldBatchArgs fa;
        if ((rc = fold_args(d, b->d_layers + b->layer_off[1 + r], b->d_layers + b->layer_off[2 + r], L - r, r, 0, fa.a))) return rc;
        fa.chal = b->d_chal;
        HIPCHK(launch_fold_merkle_batch(fa, lb, b->d_trees + b->tree_off[2 + r], b->stream, nullptr, bmail(b, L - r - 1), hash));   // prover.rs:201-214
    }
    lap("lde .. last roots");
    if (timing) fprintf(stderr, "[zk batch timing]   of which waiting for the device %.1f us\n", t_wait);
    // last layers: B equal values per proof (prover.rs:238, :251), free term (prover.rs:254)
    HIPCHK(hipMemcpyAsync(b->h_last, b->d_layers + b->layer_off[1 + R], nb * B * 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (size_t p = 0; p < nb; ++p)
        for (size_t i = 1; i < B; ++i)
            if (b->h_last[p * B + i] != b->h_last[p * B])
                return fail(ZK_ERR_CHECK, "proof %zu of the batch: last FRI layer is not constant (prover.rs:238): its trace does not satisfy the constraints", p);
    // queries and the offsets of every opening (prover.rs:263-289); node j of proof p's tree over m leaves
    // (depth dd, index i) is node 2^(lb+dd) - 1 + p 2^dd + i of the batch heap
    const size_t nv = Q * b->per_proof_vals, ndg = Q * b->per_proof_digs;     // per proof, all its queries
    uint64_t* voff = b->h_goff;
    uint64_t* doff = b->h_goff + nb * nv;
    b->pool->run(nb, 16, [&](size_t p) {
        ch[p].commit_u32(b->h_last[p * B]);                               // prover.rs:254
        uint32_t qraw[kMaxQueries];
        for (uint32_t k = 0; k < Q; ++k) qraw[k] = ch[p].get_u32();       // prover.rs:263 (x Q)
        uint64_t* vo = voff + p * nv;
        uint64_t* dofs = doff + p * ndg;
        std::vector<size_t> nodes;
        auto add_path = [&](uint32_t tree, uint32_t log_m, size_t leaf) {
            nodes.clear();
            path_nodes((size_t)1 << log_m, leaf, nodes);
            uint32_t dd = log_m;                                          // path_nodes walks from the leaf level up
            for (size_t nd : nodes) {
                size_t i = nd - (((size_t)1 << dd) - 1);
                *dofs++ = (uint64_t)b->tree_off[tree] + (uint64_t)((((size_t)1 << (lb + dd)) - 1) + (p << dd) + i) * 8;
                --dd;
            }
        };
        for (uint32_t k = 0; k < Q; ++k) {
            const size_t x = (size_t)qraw[k] % (N - 2 * B);
            *vo++ = b->layer_off[0] + p * N + x;         add_path(0, L, x);
            *vo++ = b->layer_off[0] + p * N + x + B;     add_path(0, L, x + B);
            *vo++ = b->layer_off[0] + p * N + x + 2 * B; add_path(0, L, x + 2 * B);
            *vo++ = b->layer_off[1] + p * N + x;         add_path(1, L, x);
            for (uint32_t i = 0; i < R; ++i) {
                size_t len = N >> i, xi = x % len, nx = (xi + len / 2) % len;
                *vo++ = b->layer_off[1 + i] + p * len + xi; add_path(1 + i, L - i, xi);
                *vo++ = b->layer_off[1 + i] + p * len + nx; add_path(1 + i, L - i, nx);
            }
        }
    });
    lap("queries + opening offsets");
    const size_t tv = nb * nv, td = nb * ndg;
    HIPCHK(hipMemcpyAsync(b->d_goff, b->h_goff, (tv + td) * 8, hipMemcpyHostToDevice, b->stream));
    HIPCHK(launch_scatter(b->d_stage, b->d_segs, b->n_segs, b->seg_words, b->d_trees, nullptr, b->stream, nullptr));   // host-built levels
    HIPCHK(launch_gather(b->d_layers, b->d_goff, (uint32_t)tv, 1, b->d_gout, b->stream, nullptr));
    HIPCHK(launch_gather(b->d_trees, b->d_goff + tv, (uint32_t)td, 8, b->d_gout + tv, b->stream, nullptr));
    HIPCHK(hipMemcpyAsync(b->h_gout, b->d_gout, (tv + td * 8) * 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    lap("gather");
    std::atomic<int> bad{0};
    b->pool->run(nb, 4, [&](size_t p) {
        const uint32_t* vals = b->h_gout + p * nv;
        const uint32_t* dw = b->h_gout + tv + p * ndg * 8;
        std::vector<uint8_t> dig(ndg * 32);
        for (size_t i = 0; i < ndg; ++i) digest_words_to_bytes(dw + 8 * i, dig.data() + 32 * i);
        size_t dpos = 0;
        for (uint32_t q = 0; q < Q; ++q, vals += 4 + 2 * R) {
            for (int k = 0; k < 4; ++k) { ch[p].commit_val_path(vals[k], dig.data() + 32 * dpos, L); dpos += L; }   // prover.rs:274-277
            for (uint32_t i = 0; i < R; ++i) {                                                                       // prover.rs:280-289
                size_t pl = L - i;
                ch[p].commit_pair_paths(vals[4 + 2 * i], vals[5 + 2 * i], dig.data() + 32 * dpos, dig.data() + 32 * (dpos + pl), pl);
                dpos += 2 * pl;
            }
        }
        if (ch[p].data.size() != plen) { bad.store(1); return; }
        memcpy(proofs_out + p * stride, ch[p].data.data(), plen);          // channel.rs:34-36
        memcpy(states_out + 32 * p, ch[p].state, 32);
    });
    lap("decommit hashing + copy out");
    if (bad.load()) return fail(ZK_ERR_STATE, "zk_batch_prove: unexpected proof length");
    return ZK_OK;
}

}  // extern "C"
