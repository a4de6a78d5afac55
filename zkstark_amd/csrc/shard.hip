// shard.hip -- one proof sharded over the GPUs of one node (zk_shard_*): the host orchestration of
// prover.rs:9-293 for a cyclically distributed evaluation domain.
//
// Rank r of G holds the elements i = r (mod G) of every layer.  Its shard of the coset {w h^i} is the coset
// {(w h^r) (h^G)^j}, a domain with blow-up B/G, so LDE, composition and every fold are the single-GPU kernels
// (zk_dev_*) on that domain, with no communication.  A commitment needs the leaves in natural order: one
// all-to-all turns the cyclic layout into contiguous blocks of m/G leaves (chunk q of the receive buffer comes
// from rank q; leaf u*G + q of the block = recv[q][u], hashed straight from the receive buffer), each rank builds
// its subtree, the G subtree roots are exchanged and the top log2(G) levels are hashed on the host by every rank.
// cp is the exception (round 5): over a rank's block it is a function of f over that block and the 2B positions after it,
// and the block of f is still in the receive buffer when cp is due, so it is recomputed there inside the leaf hashing
// (commit_cp_from_f) and only a 2B-word all-gather travels: 8N instead of 12N bytes per proof on the links.
// Small layers are replicated and finished in one call (zk_tail_*).  The transcript runs identically on every rank.
//
// The collectives go through a two-function transport: RCCL (grouped ncclSend/ncclRecv over xGMI, loaded at run
// time) or the caller's own (tests: several ranks on one GPU).
#include "shard.hpp"
#include "board.hpp"
#include "peer.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "host_sha.hpp"
#include "sha256.hpp"

using namespace zk;
using namespace zk::impl;

// ===========================================================================
// RCCL at run time
// ===========================================================================
namespace zk {
namespace impl {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const RcclApi* rccl_api() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static RcclApi api;
    static int state = 0;                   // 0 = not tried, 1 = loaded, -1 = failed
    static std::string why;
    if (state == 1) return &api;
    if (state == -1) { fail(ZK_ERR_STATE, "%s", why.c_str()); return nullptr; }
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) {
        state = -1;
        why = std::string("RCCL is not available (dlopen librccl.so.1: ") + dlerror() + "): pass a zk_shard_transport";
        fail(ZK_ERR_STATE, "%s", why.c_str());
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) { ok = false; why = std::string("librccl lacks ") + n; } return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { state = -1; fail(ZK_ERR_STATE, "%s", why.c_str()); return nullptr; }
    state = 1;
    return &api;
}

}  // namespace impl
}  // namespace zk

// ===========================================================================
// The sharded prover
// ===========================================================================
struct zk_shard {
    int device = 0, rank = 0, G = 1;
    uint32_t lg = 0, log_n = 0, log_b = 0, L = 0, R = 0;
    size_t n = 0, N = 0, B = 0;
    hipStream_t stream = nullptr, xstream = nullptr;   // xstream: chunked exchanges run beside the hashing
    hipStream_t bstream = nullptr;                     // odd chunks are hashed here, so that one chunk's launch fills the CUs the previous one drains
    // transport
    zk_shard_transport tp{};
    const RcclApi* rccl = nullptr;                     // non-null: tp is the built-in RCCL transport
    ncclComm_t comm = nullptr;                         // collectives on `stream`
    ncclComm_t xcomm = nullptr;                        // chunked exchanges on `xstream`: their own communicator (null: comm serves both)
    PeerTransport* peer = nullptr;                     // non-null: tp is the built-in peer-copy transport (zk_shard_options.peer_copy; peer.hpp)
    bool force = false;                                // collectives even with G = 1
    bool plain = false, single_build_stream = false, single_comm = false;   // zk_shard_options
    // every collective waits for the previous collective of the OTHER stream (explicit HIP event, whatever the transport)
    hipEvent_t ev_coll = nullptr;
    hipStream_t last_coll_stream = nullptr;
    // zk_shard_set_profiling: event pairs around the exchanges (kind 0: an exchange on xstream, 1: an exchange on a hashing
    // stream = exposed in full, 2: a hashing stream stalled on a chunk's exchange)
    bool profiling = false;
    struct Timed { hipEvent_t a, b; int kind; };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> ev_pool;
    double tail_ms_acc = 0;
    // layout
    uint32_t min_layer_log = 21, min_chunk_log = 14, overlap_min_log = 21;   // zk_shard_plan has the defaults in force
    static constexpr uint32_t kLogChunks = 2;          // chunked layers: 4 chunks
    uint32_t n_sharded = 1, tail_rounds = 0, chunked_mask = 0;   // zk_shard_plan
    zk_dom* dom_loc = nullptr;
    // cp is committed WITHOUT an exchange (zk_shard_plan_info.cp_from_f): the block of f this rank received for the commitment
    // of f is still in the receive buffer, so cp over the block is recomputed from it inside the leaf hashing; the 2B positions
    // after the block come from a 2B-word all-gather (the halo), 1/(x - 1) over the block from a table built at creation
    bool cp_from_f = false;
    zk_dom* dom_glob = nullptr;                        // fold-only: the global domain's power table and constants
    uint32_t *d_inv_blk = nullptr, *d_halo_send = nullptr, *d_halo = nullptr;
    hipEvent_t ev_halo = nullptr;
    zk_ctx* tail = nullptr;
    zk_committer* committer = nullptr;
    uint32_t *d_trace = nullptr, *d_coef = nullptr, *d_layers = nullptr, *d_trees = nullptr;
    uint32_t *d_recv = nullptr, *d_gbuf = nullptr, *d_repl = nullptr, *d_small = nullptr;
    std::vector<size_t> layer_off, layer_len, tree_off, tree_leaves;
    hipEvent_t ev_layer = nullptr, ev_chunk[1u << kLogChunks] = {}, ev_built[1u << kLogChunks] = {};
    // decommit
    uint64_t *d_goff = nullptr, *h_goff = nullptr;
    uint32_t *d_gout = nullptr, *d_gall = nullptr, *h_gall = nullptr;
    size_t gather_slots = 0, gather_words = 0;
    uint32_t* h_small = nullptr;                       // pinned scratch (subtree roots, flags)
    // one-launch decommitment (fetch_kernel): work list and results in host-mapped memory, flag behind the results
    uint64_t *h_fitems = nullptr, *dm_fitems = nullptr;
    uint32_t *h_fout = nullptr, *dm_fout = nullptr;
    uint32_t *h_fmail = nullptr, *dm_fmail = nullptr, *d_fcounter = nullptr;
    uint32_t fetch_seq = 0;
    uint64_t blob_seq = 0;                             // decommitment contributions through the shared page (board.blob_*)
    std::vector<uint32_t> blob_buf;
    double decommit_ms_acc = 0;
    // root board
    RootBoard board;
    bool have_board = false;                           // the shared page is mapped: abort words work
    bool use_board = false;                            // ... and the subtree roots travel through it
    uint64_t board_seq = 0;
    // settings (zk_shard_set_hash / _set_queries) and failure state
    int hash = ZK_HASH_SHA256;
    uint32_t queries = 1;
    double timeout_s = 120.0;                          // host-side waits on peers (ZK_SHARD_TIMEOUT_S)
    bool failed = false;                               // this rank left a collective phase with an error
    // per proof
    std::vector<std::vector<uint32_t>> tops;           // per tree: heap of 2G-1 digests (8 words each), root first
    bool have_trace = false;
    uint32_t first = 0, last = 0;
    zk_transcript_info info{};
    zk_shard_stats stats{};
    double device_bytes = 0;
};

namespace {

bool sharded(const zk_shard* s, uint32_t rho) { return rho < s->n_sharded; }
uint32_t* layer_ptr(zk_shard* s, uint32_t lid) { return s->d_layers + s->layer_off[lid]; }
uint32_t* tree_ptr(zk_shard* s, uint32_t t) { return s->d_trees + s->tree_off[t]; }
bool collectives(const zk_shard* s) { return s->G > 1 || s->force; }

// Host-side wait for stream work that depends on the peers (a collective).  Bounded (timeout_s), and ended at once by
// a peer that has posted an abort on the root board; the message names the phase, this rank and the peer.
int sync_peers(zk_shard* s, hipStream_t st, const char* what) {
    if (!collectives(s)) { HIPCHK(hipStreamSynchronize(st)); return ZK_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return ZK_OK;
        if (q != hipErrorNotReady) return fail(ZK_ERR_HIP, "rank %d of %d: device error while waiting for %s: %s", s->rank, s->G, what, hipGetErrorString(q));
        if ((++spins & 127) == 0) {
            if (s->have_board) {
                const int ab = s->board.aborted_peer();
                if (ab >= 0 && ab != s->rank)
                    return fail(ZK_ERR_HIP, "rank %d of %d: rank %d left the proof with error %d while this rank waited for %s", s->rank, s->G, ab,
                                -(int)s->board.bad_code, what);
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > s->timeout_s)
                return fail(ZK_ERR_HIP, "rank %d of %d: timed out after %.0f s waiting for %s (a peer never arrived; exchange #%llu)", s->rank, s->G, dt,
                            what, (unsigned long long)s->board_seq);
        }
        if (spins > 4096) sched_yield();
    }
}

// This rank leaves a collective phase with an error: tell the peers (root board) so that they stop waiting for it,
// remember it for zk_shard_destroy (ncclCommAbort instead of ncclCommDestroy), and say so on stderr.
int rank_failed(zk_shard* s, int rc) {
    if (!rc || s->failed) return rc;
    s->failed = true;
    if (s->have_board) s->board.post_abort((uint32_t)(-rc));
    if (s->peer) s->peer->post_abort((uint32_t)(-rc));
    if (collectives(s)) fprintf(stderr, "[zk_shard] rank %d of %d failed (%d): %s\n", s->rank, s->G, rc, last_error());
    return rc;
}

// ---- built-in transport: RCCL ---------------------------------------------------------------
#define NCCLCHK(s, expr)                                                                                  \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess)                                                                            \
            return fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (s)->rccl->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

int rccl_all_to_all(void* user, const uint32_t* const* send, uint32_t* const* recv, size_t words, void* stream) {
    zk_shard* s = static_cast<zk_shard*>(user);
    hipStream_t st = (hipStream_t)stream;
    const ncclComm_t comm = (st == s->xstream && s->xcomm) ? s->xcomm : s->comm;     // one communicator per stream
    // the all-to-all of the four-step transpose: every pair exchanges one piece, all 7 xGMI links busy at once.
    // A failed Send / Recv must not leave the thread's group open (every later RCCL call would be undefined):
    // the group is always ended, then the first error is reported.
    // This rank's own piece never leaves the GPU: a plain device copy in stream order (RCCL moves a send / receive pair of a
    // rank with itself through its transport kernel at ~0.14 TB/s: 1.4 ms of exchanges per 2^24 proof at one rank; at G ranks
    // it is 1/G of every exchange).  -DZK_SHARD_RCCL_SELF at build time keeps it inside the group.
    // With ONE rank (collectives forced: the rehearsal of this code path on a one-GPU box) the pair stays inside the group,
    // so that ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd are exercised by every one-GPU test and bench rehearsal.
#ifndef ZK_SHARD_RCCL_SELF
    const bool self_copy = s->G > 1;
#else
    const bool self_copy = false;
#endif
    if (self_copy && send[s->rank] != recv[s->rank])
        HIPCHK(hipMemcpyAsync(recv[s->rank], send[s->rank], words * 4, hipMemcpyDeviceToDevice, st));
    NCCLCHK(s, s->rccl->GroupStart());
    ncclResult_t first = ncclSuccess;
    int bad_peer = -1;
    for (int p = 0; p < s->G && first == ncclSuccess; ++p) {
        if (self_copy && p == s->rank) continue;
        ncclResult_t r = s->rccl->Send(send[p], words, ncclUint32, p, comm, st);
        if (r == ncclSuccess) r = s->rccl->Recv(recv[p], words, ncclUint32, p, comm, st);
        if (r != ncclSuccess) { first = r; bad_peer = p; }
    }
    const ncclResult_t end = s->rccl->GroupEnd();
    if (first != ncclSuccess)
        return fail(ZK_ERR_HIP, "rank %d: ncclSend/ncclRecv with peer %d failed: %s", s->rank, bad_peer, s->rccl->GetErrorString(first));
    if (end != ncclSuccess) return fail(ZK_ERR_HIP, "rank %d: ncclGroupEnd failed: %s", s->rank, s->rccl->GetErrorString(end));
    return ZK_OK;
}
int rccl_all_gather(void* user, const uint32_t* send, uint32_t* recv, size_t words, void* stream) {
    zk_shard* s = static_cast<zk_shard*>(user);
    const ncclComm_t comm = ((hipStream_t)stream == s->xstream && s->xcomm) ? s->xcomm : s->comm;
    NCCLCHK(s, s->rccl->AllGather(send, recv, words, ncclUint32, comm, (hipStream_t)stream));
    return ZK_OK;
}

// ---- built-in transport: peer copies through IPC handles (peer.hpp), no RCCL ------------------------------------
int peer_all_to_all(void* user, const uint32_t* const* send, uint32_t* const* recv, size_t words, void* stream) {
    zk_shard* s = static_cast<zk_shard*>(user);
    if (s->peer->exchange(send, nullptr, recv, nullptr, words, (hipStream_t)stream)) return fail(ZK_ERR_HIP, "%s", s->peer->error.c_str());
    return ZK_OK;
}
int peer_all_gather(void* user, const uint32_t* send, uint32_t* recv, size_t words, void* stream) {
    zk_shard* s = static_cast<zk_shard*>(user);
    if (s->peer->exchange(nullptr, send, nullptr, recv, words, (hipStream_t)stream)) return fail(ZK_ERR_HIP, "%s", s->peer->error.c_str());
    return ZK_OK;
}

// ---- collectives with the byte accounting of zk_shard_stats -----------------------------------
// A collective is enqueued on `st`: it first waits for the previous collective if that went to the other stream, so that
// the order in which collectives EXECUTE is the program order on every rank whatever the transport does internally.
int before_collective(zk_shard* s, hipStream_t st) {
    if (s->last_coll_stream && s->last_coll_stream != st) HIPCHK(hipStreamWaitEvent(st, s->ev_coll, 0));
    return ZK_OK;
}
int after_collective(zk_shard* s, hipStream_t st) {
    HIPCHK(hipEventRecord(s->ev_coll, st));
    s->last_coll_stream = st;
    return ZK_OK;
}
// ---- zk_shard_set_profiling: event pairs, resolved at the end of a call --------------------------------------
hipEvent_t timing_event(zk_shard* s) {
    if (!s->ev_pool.empty()) { hipEvent_t e = s->ev_pool.back(); s->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ScopedTimed {                                   // brackets what is enqueued on `st` during its lifetime
    zk_shard* s; hipStream_t st; int kind; hipEvent_t a = nullptr;
    ScopedTimed(zk_shard* s_, hipStream_t st_, int kind_) : s(s_), st(st_), kind(kind_) {
        if (s->profiling) { a = timing_event(s); if (a) (void)hipEventRecord(a, st); }
    }
    ~ScopedTimed() {
        if (!a) return;
        hipEvent_t b = timing_event(s);
        if (b) { (void)hipEventRecord(b, st); s->timed.push_back({a, b, kind}); } else s->ev_pool.push_back(a);
    }
};
void reset_timing(zk_shard* s) {
    for (auto& t : s->timed) { s->ev_pool.push_back(t.a); s->ev_pool.push_back(t.b); }
    s->timed.clear();
    s->tail_ms_acc = 0;
    s->decommit_ms_acc = 0;
    s->stats.exchange_ms = s->stats.exposed_exchange_ms = s->stats.tail_ms = s->stats.decommit_ms = 0;
    s->stats.exchanges = 0;
}
void collect_timing(zk_shard* s) {
    if (!s->profiling) return;
    for (auto& t : s->timed) {
        float ms = 0;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            if (t.kind != 2) { s->stats.exchange_ms += ms; s->stats.exchanges += 1; }
            if (t.kind != 0) s->stats.exposed_exchange_ms += ms;
        }
        s->ev_pool.push_back(t.a); s->ev_pool.push_back(t.b);
    }
    s->timed.clear();
    s->stats.tail_ms = s->tail_ms_acc;
    s->stats.decommit_ms = s->decommit_ms_acc;
}
int all_to_all(zk_shard* s, const uint32_t* const* send, uint32_t* const* recv, size_t words, hipStream_t st) {
    int rc = before_collective(s, st);
    if (rc) return rc;
    {
        ScopedTimed tm(s, st, st == s->xstream ? 0 : 1);
        rc = s->tp.all_to_all(s->tp.user, send, recv, words, (void*)st);
    }
    if (rc) return rc > 0 ? fail(ZK_ERR_HIP, "transport all_to_all failed (%d)", rc) : rc;
    const double sent = 4.0 * (double)words * (s->G - 1);
    s->stats.sent_bytes += sent;
    s->stats.all_to_all_bytes += sent;
    return after_collective(s, st);
}
int all_gather(zk_shard* s, const uint32_t* send, uint32_t* recv, size_t words, hipStream_t st) {
    int rc = before_collective(s, st);
    if (rc) return rc;
    {
        ScopedTimed tm(s, st, st == s->xstream ? 0 : 1);
        rc = s->tp.all_gather(s->tp.user, send, recv, words, (void*)st);
    }
    if (rc) return rc > 0 ? fail(ZK_ERR_HIP, "transport all_gather failed (%d)", rc) : rc;
    s->stats.sent_bytes += 4.0 * (double)words * (s->G - 1);
    return after_collective(s, st);
}
// min over the ranks of a host flag (setup only): one word through the transport
int agree(zk_shard* s, bool ok, bool* all_ok) {
    if (!collectives(s)) { *all_ok = ok; return ZK_OK; }
    s->h_small[0] = ok ? 1u : 0u;
    HIPCHK(hipMemcpyAsync(s->d_small, s->h_small, 4, hipMemcpyHostToDevice, s->stream));
    int rc = all_gather(s, s->d_small, s->d_small + 8, 1, s->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(s->h_small + 8, s->d_small + 8, 4 * (size_t)s->G, hipMemcpyDeviceToHost, s->stream));
    if ((rc = sync_peers(s, s->stream, "the setup agreement (all-gather of one word)"))) return rc;
    *all_ok = true;
    for (int q = 0; q < s->G; ++q) *all_ok = *all_ok && s->h_small[8 + q] == 1u;
    return ZK_OK;
}

// merkle.rs:38-47 over the G subtree roots (state words): heap of 2G-1 digests, root first
void host_merkle_top(const uint32_t* subroots, int G, std::vector<uint32_t>& heap, int hash) {
    heap.assign((size_t)(2 * G - 1) * 8, 0);
    memcpy(heap.data() + (size_t)(G - 1) * 8, subroots, (size_t)G * 32);
    for (int j = G - 2; j >= 0; --j) {
        const uint32_t *l = heap.data() + (size_t)(2 * j + 1) * 8, *r = heap.data() + (size_t)(2 * j + 2) * 8;
        if (hash == ZK_HASH_FIELD) {
            Digest dl, dr;
            memcpy(dl.w, l, 32); memcpy(dr.w, r, 32);
            const Digest d = fieldhash_inner(dl, dr, host_fieldhash_consts());
            memcpy(heap.data() + (size_t)j * 8, d.w, 32);
        } else {
            host_sha_inner(l, r, heap.data() + (size_t)j * 8);
        }
    }
}
// Consulted by the committer while it waits for a subtree's digests: the launch sits behind an all-to-all that never
// completes when a peer has left the proof.
int poll_peer_abort(void* user) {
    zk_shard* s = static_cast<zk_shard*>(user);
    if (!s->have_board) return 0;
    const int ab = s->board.aborted_peer();
    if (ab < 0 || ab == s->rank) return 0;
    return fail(ZK_ERR_HIP, "rank %d of %d: rank %d left the proof with error %d", s->rank, s->G, ab, -(int)s->board.bad_code);
}
// A commit launch whose digests never arrived: with collectives the usual cause is a peer that never sent its piece.
int commit_wait_failed(zk_shard* s, uint32_t lid, int rc) {
    if (!collectives(s)) return rc;
    const std::string inner = last_error();
    const int ab = s->have_board ? s->board.aborted_peer() : -1;
    if (ab >= 0 && ab != s->rank)
        return fail(rc, "rank %d of %d: rank %d left the proof with error %d; the commitment of layer %u could not complete (%s)", s->rank, s->G, ab,
                    -(int)s->board.bad_code, lid, inner.c_str());
    return fail(rc, "rank %d of %d: the commitment of layer %u (all-to-all + subtree, before root exchange #%llu) did not complete: %s", s->rank, s->G,
                lid, (unsigned long long)(s->board_seq + 1), inner.c_str());
}

// Cyclic layer `lid` (2^m_log values in total, 2^m_log / G here) -> subtree over this rank's block of leaves;
// root_out: the root of the whole tree (prover.rs:81, :176, :214 + what :85 / :180 / :224 feed the channel).
// the G subtree roots, on the host of every rank: through the shared page, else by all-gather; the top log2 G levels are
// hashed by every rank (merkle.rs:14-51 on G leaves that are digests)
int join_subtrees(zk_shard* s, uint32_t lid, uint32_t* nodes, const uint8_t mine_bytes[32], uint8_t root_out[32]) {
    const int G = s->G;
    int rc;
    uint32_t mine[8];
    Digest dm;
    bytes_to_digest(mine_bytes, dm);
    memcpy(mine, dm.w, 32);
    std::vector<uint32_t> sub((size_t)G * 8);
    if (!collectives(s)) {
        memcpy(sub.data(), mine, 32);
    } else if (s->use_board) {
        const RootBoard::Status bs = s->board.exchange(++s->board_seq, mine, sub.data(), s->timeout_s);
        if (bs == RootBoard::kPeerAborted)
            return fail(ZK_ERR_HIP, "rank %d of %d: rank %d left the proof with error %d (seen in root exchange #%llu, layer %u)", s->rank, G,
                        s->board.bad_peer, -(int)s->board.bad_code, (unsigned long long)s->board_seq, lid);
        if (bs != RootBoard::kOk)
            return fail(ZK_ERR_HIP, "rank %d of %d: rank %d did not post its subtree root of layer %u (root exchange #%llu timed out after %.0f s)", s->rank, G,
                        s->board.bad_peer, lid, (unsigned long long)s->board_seq, s->timeout_s);
    } else {
        if ((rc = committer_flush(s->committer, s->stream))) return rc;         // node 0 is host-built: on the device before it is gathered
        if ((rc = all_gather(s, nodes, s->d_small, 8, s->stream))) return rc;   // node 0 of every rank's subtree
        HIPCHK(hipMemcpyAsync(s->h_small, s->d_small, 32 * (size_t)G, hipMemcpyDeviceToHost, s->stream));
        if ((rc = sync_peers(s, s->stream, "the all-gather of the subtree roots"))) return rc;
        memcpy(sub.data(), s->h_small, 32 * (size_t)G);
    }
    host_merkle_top(sub.data(), G, s->tops[lid], s->hash);
    digest_words_to_bytes(s->tops[lid].data(), root_out);
    return ZK_OK;
}

// halo: this is the commitment of f inside a proof whose cp is computed from the received block (cp_from_f): behind the
// exchange(s) of the layer, in the same program order on every rank, goes the 2B-word all-gather of the positions after
// every block (packed from the cyclic shard by the caller).  It runs on the stream the LAST exchange ran on, so beside the
// hashing when the layer is exchanged in chunks; the main stream picks it up at ev_halo.
int commit_sharded(zk_shard* s, uint32_t lid, uint32_t m_log, uint8_t root_out[32], bool halo = false) {
    const int G = s->G;
    const uint32_t lg = s->lg;
    uint32_t* loc = layer_ptr(s, lid);
    const size_t cnt = s->layer_len[lid];
    uint32_t* nodes = tree_ptr(s, lid);
    uint8_t mine_bytes[32];
    int rc;
    if (collectives(s)) {
        const uint32_t log_cnt = m_log - 2 * lg;                  // words per (rank, peer) piece
        const size_t per = cnt >> lg;
        const uint32_t* send[64];
        uint32_t* recv[64];
        if ((s->chunked_mask >> lid) & 1u) {                      // zk_shard_plan: pieces of >= 2^overlap_min_log words
            // big layer: exchange and hash in K aligned chunks, so that hashing chunk c overlaps the exchange of
            // chunk c+1.  The exchanges are issued from a side stream that depends on the layer only.
            const uint32_t lk = zk_shard::kLogChunks, K = 1u << lk;
            const size_t cc = per >> lk;                          // words per (peer, chunk)
            HIPCHK(hipEventRecord(s->ev_layer, s->stream));
            HIPCHK(hipStreamWaitEvent(s->xstream, s->ev_layer, 0));
            // A chunk build is ONE launch of exactly one round of workgroups: back to back on one stream every launch pays
            // its own ramp-up and drain (4.61 instead of 4.08 ms of leaf hashing per proof at one rank).  Alternating
            // between two streams lets the next chunk's workgroups take the CUs as the previous chunk's leave them.
            // The build of chunk c is enqueued right behind ITS exchange (a host-staged transport blocks in the exchange:
            // this way chunk c is hashed while chunk c + 1 travels).
            const bool two = s->bstream != nullptr;
            for (uint32_t c = 0; c < K; ++c) {
                for (int g = 0; g < G; ++g) {
                    send[g] = loc + (size_t)g * per + (size_t)c * cc;
                    recv[g] = s->d_recv + ((size_t)c * G + g) * cc;
                }
                // Chunk 0 is the exposed one (nothing to hash beside it): its exchange goes to the MAIN stream, in order
                // between the kernel that produced the layer and the build of the chunk.  On the exchange stream it sat
                // behind two cross-queue event dependencies (layer -> exchange, exchange -> build), each of which the
                // hardware resolves in 20-35 us (kernel trace of a one-rank proof: 70 us from the end of the producer to the
                // start of the first build, of which the exchange itself is 17).  The other chunks keep the exchange stream:
                // their dependencies resolve while the previous chunk is being hashed.
                hipStream_t xs = c == 0 ? s->stream : s->xstream;
                hipStream_t bs = (two && (c & 1u)) ? s->bstream : s->stream;
                if ((rc = all_to_all(s, send, recv, cc, xs))) return rc;
                if (xs != bs) {
                    HIPCHK(hipEventRecord(s->ev_chunk[c], xs));
                    ScopedTimed stall(s, bs, 2);               // how long this hashing stream waits for chunk c's exchange
                    HIPCHK(hipStreamWaitEvent(bs, s->ev_chunk[c], 0));
                }
                if ((rc = zk_dev_merkle_build_chunk(s->d_recv + (size_t)c * G * cc, lg, log_cnt - lk, nodes, m_log - lg, c, bs, s->hash))) return rc;
                if (bs != s->stream) HIPCHK(hipEventRecord(s->ev_built[c], bs));
                if (halo && c + 1 == K) {
                    if ((rc = all_gather(s, s->d_halo_send, s->d_halo, 2 * s->B, xs))) return rc;
                    HIPCHK(hipEventRecord(s->ev_halo, xs));
                }
            }
            if (two)
                for (uint32_t c = 1; c < K; c += 2) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_built[c], 0));
            if ((rc = zk_dev_merkle_commit_finish(s->committer, nodes, m_log - lg, lk, s->stream, s->hash, mine_bytes))) return commit_wait_failed(s, lid, rc);
            s->stats.chunked_layers += 1;
        } else {
            for (int g = 0; g < G; ++g) { send[g] = loc + (size_t)g * per; recv[g] = s->d_recv + (size_t)g * per; }
            if ((rc = all_to_all(s, send, recv, per, s->stream))) return rc;   // piece q: rank q's share of my block
            if (halo) {
                if ((rc = all_gather(s, s->d_halo_send, s->d_halo, 2 * s->B, s->stream))) return rc;
                HIPCHK(hipEventRecord(s->ev_halo, s->stream));
            }
            if ((rc = zk_dev_merkle_commit(s->committer, s->d_recv, lg, log_cnt, nodes, s->stream, s->hash, mine_bytes))) return commit_wait_failed(s, lid, rc);
        }
    } else {
        if ((rc = zk_dev_merkle_commit(s->committer, loc, 0, m_log, nodes, s->stream, s->hash, mine_bytes))) return rc;
    }
    return join_subtrees(s, lid, nodes, mine_bytes, root_out);
}

// cp_from_f: the subtree over this rank's block of cp, computed from the block of f in the receive buffer (which the
// commitment of f left in all-to-all order, in chunks if that layer was exchanged in chunks) and the gathered halo
int commit_cp_from_f(zk_shard* s, const uint32_t alpha[3], uint8_t root_out[32]) {
    const uint32_t lg = s->lg, L = s->L;
    const bool chunked = (s->chunked_mask & 1u) != 0;
    ComposeBlockArgs g{};
    g.a.f = s->d_recv;
    g.a.inv_xm1 = s->d_inv_blk;
    g.halo = s->d_halo + (size_t)s->rank * ((2 * s->B) >> lg);
    g.lg = lg;
    g.log_cnt = L - 2 * lg - (chunked ? zk_shard::kLogChunks : 0u);
    g.log_m = L - lg;
    g.halo_stride = (uint32_t)(2 * s->B);
    g.e0 = (uint32_t)((size_t)s->rank << (L - lg));
    HIPCHK(hipStreamWaitEvent(s->stream, s->ev_halo, 0));
    uint8_t mine_bytes[32];
    uint32_t* nodes = tree_ptr(s, 1);
    // The folds need cp over this rank's CYCLIC shard (prover.rs:166-173 on the local coset): that sweep goes behind the hashing
    // on the same stream, before this thread waits for the digests, so it runs while the roots are exchanged.  (On the second
    // build stream, beside the hashing, it cost 0.1 ms per proof more; on a LOW-priority stream of its own 0.15-0.6 ms more:
    // measured, tools/ab_cp_from_f.py.)
    struct Ctx { zk_shard* s; const uint32_t* alpha; } cx{s, alpha};
    const int rc = dev_compose_block_commit(s->committer, s->dom_glob, g, s->first, s->last, alpha, nodes, s->stream, s->hash, mine_bytes,
                                            [](void* u) -> int {
                                                Ctx* c = static_cast<Ctx*>(u);
                                                return zk_dev_compose(c->s->dom_loc, layer_ptr(c->s, 0), layer_ptr(c->s, 1), c->s->first, c->s->last, c->alpha,
                                                                      c->s->stream);
                                            }, &cx);
    if (rc) return commit_wait_failed(s, 1, rc);
    return join_subtrees(s, 1, nodes, mine_bytes, root_out);
}

int do_lde(zk_shard* s) { return zk_dev_lde(s->dom_loc, s->d_trace, s->d_coef, layer_ptr(s, 0), s->stream); }

// prover.rs:266-289.  The slot list (4 + 2 rho0 openings: a value and the nodes of its path inside the owner's subtree) is
// the same on every rank; every rank fetches the slots it owns and the contributions are merged, so that every rank
// assembles the same bytes.  Two forms:
//   * ranks on one node (the shared page is mapped), or no collectives at all: ONE launch (fetch_kernel) reads this rank's
//     work list from host-mapped memory, writes its results there and raises a flag -- no copy command, no stream
//     synchronisation, as in the single-GPU prover (zkstark.hip: open_launch) --, the contributions travel through the page
//     (board.blob_post / blob_wait: one store and G polled loads), and the openings of the replicated tail are fetched by a
//     launch of their own that is in flight at the same time (tail_open_begin / _end);
//   * otherwise (another node, no_root_board, plain_collectives): two gathers, one all-gather, a device-to-host copy.
struct DecommitItem { int owner; uint64_t off; };
struct DecommitOpen { size_t nloc; uint32_t lid; size_t block; };
struct DecommitPlan {
    std::vector<DecommitItem> vit, dit;
    std::vector<DecommitOpen> opens;
};
void decommit_plan(zk_shard* s, size_t x, DecommitPlan& pl) {
    const size_t B = s->B, N = s->N;
    const uint32_t L = s->L, rho0 = s->n_sharded;
    const int G = s->G;
    std::vector<size_t> nodes;
    auto add_opening = [&](uint32_t lid, size_t leaf, uint32_t m_log) {
        const size_t blk = ((size_t)1 << m_log) >> s->lg;
        const size_t p = leaf / blk, lf = leaf % blk;
        pl.vit.push_back({(int)(leaf % (size_t)G), (uint64_t)(s->layer_off[lid] + leaf / (size_t)G)});
        nodes.clear();
        path_nodes(blk, lf, nodes);
        for (size_t nd : nodes) pl.dit.push_back({(int)p, (uint64_t)s->tree_off[lid] + (uint64_t)nd * 8});
        pl.opens.push_back({nodes.size(), lid, p});
    };
    add_opening(0, x, L); add_opening(0, x + B, L); add_opening(0, x + 2 * B, L); add_opening(1, x, L);   // prover.rs:266-277
    for (uint32_t i = 0; i < rho0; ++i) {                                                                   // prover.rs:280-289
        const size_t len = N >> i, xi = x % len, nx = (xi + len / 2) % len;
        add_opening(1 + i, xi, L - i);
        add_opening(1 + i, nx, L - i);
    }
}
// The transcript part, shared by both forms: val_of(i) = value of opening i, dig_of(j) = the 8 state words of path node j
template <typename ValOf, typename DigOf>
void decommit_commit(zk_shard* s, Channel& ch, const DecommitPlan& pl, ValOf val_of, DigOf dig_of, const uint32_t* tvals, const uint8_t* tpaths) {
    const uint32_t L = s->L, rho0 = s->n_sharded;
    const int G = s->G;
    std::vector<uint8_t> path;
    std::vector<size_t> nodes;
    size_t dpos = 0;
    auto path_of = [&](size_t k) -> const uint8_t* {          // opening k: the digests inside its owner's subtree, then the top path
        const DecommitOpen& o = pl.opens[k];
        path.resize(32 * (o.nloc + s->lg));
        for (size_t i = 0; i < o.nloc; ++i) digest_words_to_bytes(dig_of(dpos + i), path.data() + 32 * i);
        dpos += o.nloc;
        nodes.clear();
        if (G > 1) path_nodes((size_t)G, o.block, nodes);
        for (size_t i = 0; i < nodes.size(); ++i) digest_words_to_bytes(s->tops[o.lid].data() + nodes[i] * 8, path.data() + 32 * (o.nloc + i));
        return path.data();
    };
    for (size_t k = 0; k < 4; ++k) {                          // prover.rs:274-277
        const uint8_t* p = path_of(k);
        ch.commit_val_path(val_of(k), p, L);
    }
    std::vector<uint8_t> p0;
    for (uint32_t i = 0; i < rho0; ++i) {                     // prover.rs:288
        const size_t plen = L - i;
        const uint8_t* a = path_of(4 + 2 * i);
        p0.assign(a, a + 32 * plen);
        const uint8_t* b = path_of(5 + 2 * i);
        ch.commit_pair_paths(val_of(4 + 2 * i), val_of(5 + 2 * i), p0.data(), b, plen);
    }
    size_t tp = 0;
    for (uint32_t j = 0; j < s->tail_rounds; ++j) {           // the replicated layers, from the tail's own openings
        const size_t plen = L - rho0 - j;
        ch.commit_pair_paths(tvals[2 * j], tvals[2 * j + 1], tpaths + 32 * tp, tpaths + 32 * (tp + plen), plen);
        tp += 2 * plen;
    }
}

int decommit(zk_shard* s, Channel& ch, size_t x) {
    const int G = s->G, me = s->rank;
    const uint32_t L = s->L, rho0 = s->n_sharded;
    const double t_begin = now_us();
    DecommitPlan pl;
    decommit_plan(s, x, pl);
    const size_t nv = pl.vit.size(), nd = pl.dit.size(), row = nv + 8 * nd;
    if (nv + nd > s->gather_slots || row > s->gather_words) return fail(ZK_ERR_STATE, "zk_shard: gather capacity exceeded");
    std::vector<uint32_t> tvals(2 * (size_t)s->tail_rounds + 1);
    size_t tdig = 0;
    for (uint32_t j = 0; j < s->tail_rounds; ++j) tdig += 2 * (size_t)(L - rho0 - j);
    std::vector<uint8_t> tpaths(32 * tdig + 1);
    int rc;
    if (!collectives(s) || s->use_board) {
        // my slots, compact: values first, then digests (the order of the plan)
        size_t mv = 0, mdg = 0;
        for (size_t i = 0; i < nv; ++i) if (pl.vit[i].owner == me || !collectives(s)) s->h_fitems[mv++] = pl.vit[i].off;
        for (size_t i = 0; i < nd; ++i) if (pl.dit[i].owner == me || !collectives(s)) s->h_fitems[mv + mdg++] = pl.dit[i].off;
        const bool fetch = mv + mdg != 0;
        if (fetch) HIPCHK(launch_fetch(s->d_layers, s->d_trees, s->dm_fitems, (uint32_t)mv, (uint32_t)mdg, s->dm_fout, s->dm_fmail, ++s->fetch_seq,
                                       s->d_fcounter, s->stream, nullptr));
        if ((rc = tail_open_begin(s->tail, x))) return rc;                       // in flight beside it, on the tail's stream
        if (fetch && (rc = wait_flag(s->h_fmail, s->fetch_seq, s->stream, s->have_board ? poll_peer_abort : nullptr, s, s->timeout_s))) return rc;
        // fetch_kernel's result layout: the mdg digests (8 words each), then the mv values
        const uint32_t* mine = s->h_fout;
        std::vector<const uint32_t*> from(G, nullptr);
        std::vector<size_t> cnt_v(G, 0), cnt_d(G, 0);
        if (collectives(s)) {
            for (size_t i = 0; i < nv; ++i) cnt_v[pl.vit[i].owner] += 1;
            for (size_t i = 0; i < nd; ++i) cnt_d[pl.dit[i].owner] += 1;
            s->board.blob_post(++s->blob_seq, mine, 8 * mdg + mv);
            // one clock for the whole exchange, and the peers' contributions COPIED out of the shared page: in place they are only
            // valid until their owner posts the exchange after next (nothing enforced "consume before the next post")
            const auto t_wait0 = std::chrono::steady_clock::now();
            size_t total_words = 0;
            for (int q = 0; q < G; ++q) total_words += 8 * cnt_d[q] + cnt_v[q];
            s->blob_buf.resize(total_words);
            size_t at = 0;
            for (int q = 0; q < G; ++q) {
                if (q == me) { from[q] = mine; continue; }
                const RootBoard::Status bs = s->board.blob_wait(s->blob_seq, q, &from[q], s->timeout_s, t_wait0);
                if (bs == RootBoard::kPeerAborted)
                    return fail(ZK_ERR_HIP, "rank %d of %d: rank %d left the proof with error %d (seen in the decommitment exchange #%llu)", me, G,
                                s->board.bad_peer, -(int)s->board.bad_code, (unsigned long long)s->blob_seq);
                if (bs != RootBoard::kOk)
                    return fail(ZK_ERR_HIP, "rank %d of %d: rank %d did not post its openings (decommitment exchange #%llu timed out after %.0f s)", me, G,
                                s->board.bad_peer, (unsigned long long)s->blob_seq, s->timeout_s);
                const size_t words = 8 * cnt_d[q] + cnt_v[q];
                memcpy(s->blob_buf.data() + at, from[q], words * sizeof(uint32_t));
                from[q] = s->blob_buf.data() + at;
                at += words;
            }
        } else {
            from[0] = mine; cnt_v[0] = nv; cnt_d[0] = nd;
        }
        // position of slot i inside its owner's compact contribution
        std::vector<size_t> vpos(nv), dpos_in(nd), seen_v(G, 0), seen_d(G, 0);
        for (size_t i = 0; i < nv; ++i) { const int o = collectives(s) ? pl.vit[i].owner : 0; vpos[i] = seen_v[o]++; }
        for (size_t i = 0; i < nd; ++i) { const int o = collectives(s) ? pl.dit[i].owner : 0; dpos_in[i] = seen_d[o]++; }
        if ((rc = tail_open_end(s->tail, tvals.data(), tpaths.data()))) return rc;
        auto val_of = [&](size_t i) { const int o = collectives(s) ? pl.vit[i].owner : 0; return from[o][8 * cnt_d[o] + vpos[i]]; };
        auto dig_of = [&](size_t j) { const int o = collectives(s) ? pl.dit[j].owner : 0; return from[o] + 8 * dpos_in[j]; };
        decommit_commit(s, ch, pl, val_of, dig_of, tvals.data(), tpaths.data());
        s->decommit_ms_acc += (now_us() - t_begin) * 1e-3;
        return ZK_OK;
    }
    // collectives without the shared page: every rank gathers the slots it owns into a full row, one all-gather merges the rows
    for (size_t i = 0; i < nv; ++i) s->h_goff[i] = pl.vit[i].owner == me ? pl.vit[i].off : 0;
    for (size_t i = 0; i < nd; ++i) s->h_goff[nv + i] = pl.dit[i].owner == me ? pl.dit[i].off : 0;
    HIPCHK(hipMemcpyAsync(s->d_goff, s->h_goff, (nv + nd) * 8, hipMemcpyHostToDevice, s->stream));
    if ((rc = zk_dev_gather(s->d_layers, s->d_goff, (uint32_t)nv, 1, s->d_gout, s->stream))) return rc;
    if ((rc = zk_dev_gather(s->d_trees, s->d_goff + nv, (uint32_t)nd, 8, s->d_gout + nv, s->stream))) return rc;
    if ((rc = all_gather(s, s->d_gout, s->d_gall, row, s->stream))) return rc;
    HIPCHK(hipMemcpyAsync(s->h_gall, s->d_gall, row * 4 * (size_t)G, hipMemcpyDeviceToHost, s->stream));
    if ((rc = tail_open_begin(s->tail, x))) return rc;        // the tail layers' openings: every rank has them
    if ((rc = sync_peers(s, s->stream, "the all-gather of the decommitment"))) return rc;
    if ((rc = tail_open_end(s->tail, tvals.data(), tpaths.data()))) return rc;
    auto val_of = [&](size_t i) { return s->h_gall[(size_t)pl.vit[i].owner * row + i]; };
    auto dig_of = [&](size_t j) { return s->h_gall + (size_t)pl.dit[j].owner * row + nv + 8 * j; };
    decommit_commit(s, ch, pl, val_of, dig_of, tvals.data(), tpaths.data());
    s->decommit_ms_acc += (now_us() - t_begin) * 1e-3;
    return ZK_OK;
}

int prove(zk_shard* s, Channel& ch) {
    if (!s->have_trace) return fail(ZK_ERR_STATE, "zk_shard_prove: no trace uploaded");
    const uint32_t L = s->L, R = s->R, lg = s->lg, rho0 = s->n_sharded;
    int rc;
    uint8_t root[32];
    memset(&s->info, 0, sizeof s->info);
    s->info.public_last = s->last;
    s->stats.sent_bytes = s->stats.all_to_all_bytes = 0;
    s->stats.chunked_layers = 0;
    reset_timing(s);
    committer_drop_pending(s->committer);
    ch.data.reserve(ch.data.size() + proof_data_len(s->log_n, s->log_b, s->queries));
    if ((rc = do_lde(s))) return rc;                                              // prover.rs:60-70
    if (s->cp_from_f)                                                              // what every block's neighbour needs of my shard
        HIPCHK(launch_halo_pack(layer_ptr(s, 0), s->d_halo_send, L - 2 * lg, lg, (uint32_t)((2 * s->B) >> lg), s->stream));
    if ((rc = commit_sharded(s, 0, L, root, s->cp_from_f))) return rc;             // prover.rs:81
    ch.commit_hash(root);                                                          // prover.rs:85
    memcpy(s->info.roots[0], root, 32);
    uint32_t alpha[3];
    for (int i = 0; i < 3; ++i) alpha[i] = s->info.alpha_raw[i] = ch.get_u32();   // prover.rs:163-165
    if (s->cp_from_f) {
        // prover.rs:166-176 twice over: the commitment needs cp over this rank's BLOCK, the folds need it over its cyclic
        // shard.  The block of f is still in the receive buffer (nothing has been exchanged since), so the block form is
        // computed inside the leaf hashing; the cyclic form follows on the stream while the host waits for the root.
        if ((rc = commit_cp_from_f(s, alpha, root))) return rc;
    } else {
        if ((rc = zk_dev_compose(s->dom_loc, layer_ptr(s, 0), layer_ptr(s, 1), s->first, s->last, alpha, s->stream))) return rc;   // :166-173
        if ((rc = commit_sharded(s, 1, L, root))) return rc;                       // prover.rs:176
    }
    ch.commit_hash(root);                                                          // prover.rs:180
    memcpy(s->info.roots[1], root, 32);
    uint32_t free_term = 0;
    for (uint32_t rho = 0; rho < rho0; ++rho) {                                    // prover.rs:198-225, the distributed rounds
        const uint32_t beta = s->info.beta_raw[rho] = ch.get_u32();                // prover.rs:200
        const uint32_t m_log = L - rho;
        if (rho + 1 < rho0) {
            if ((rc = zk_dev_fri_fold(s->dom_loc, layer_ptr(s, 1 + rho), layer_ptr(s, 2 + rho), m_log - lg, rho, beta, s->stream))) return rc;
            if ((rc = commit_sharded(s, 2 + rho, m_log - 1, root))) return rc;
            ch.commit_hash(root);                                                  // prover.rs:224
            memcpy(s->info.roots[2 + rho], root, 32);
            continue;
        }
        // replication switch: fold locally, all-gather the G cyclic pieces, interleave to natural order, then the
        // commitment of that layer and every later round in one call on every rank (zk_tail_run)
        const size_t cnt = ((size_t)1 << (m_log - 1)) >> lg;
        const double t_tail = now_us();
        uint32_t* piece = s->d_recv;
        if ((rc = zk_dev_fri_fold(s->dom_loc, layer_ptr(s, 1 + rho), piece, m_log - lg, rho, beta, s->stream))) return rc;
        const uint32_t* handed = piece;
        if (collectives(s)) {
            if ((rc = all_gather(s, piece, s->d_gbuf, cnt, s->stream))) return rc;
            if ((rc = zk_dev_interleave(s->d_gbuf, s->d_repl, lg, m_log - 1 - lg, s->stream))) return rc;
            handed = s->d_repl;
            if ((rc = sync_peers(s, s->stream, "the all-gather of the first replicated layer"))) return rc;   // zk_tail_run waits on this stream
        }
        zk_channel chan_view;                                                      // zk_tail_run drives the caller's channel
        chan_view.ch = std::move(ch);
        std::vector<uint8_t> troots(32 * ((size_t)s->tail_rounds + 1));
        rc = zk_tail_run(s->tail, handed, s->stream, &chan_view, s->hash, s->info.beta_raw + rho0, troots.data(), &free_term);
        ch = std::move(chan_view.ch);
        if (rc) return rc;
        s->tail_ms_acc += (now_us() - t_tail) * 1e-3;
        for (uint32_t j = 0; j <= s->tail_rounds; ++j) memcpy(s->info.roots[1 + rho0 + j], troots.data() + 32 * j, 32);
    }
    (void)R;
    s->info.free_term = free_term;
    ch.commit_u32(free_term);                                                      // prover.rs:254
    uint32_t qraws[kMaxQueries];
    for (uint32_t k = 0; k < s->queries; ++k) qraws[k] = ch.get_u32();            // prover.rs:263 (x queries)
    s->info.query_raw = qraws[0];
    if ((rc = committer_flush(s->committer, s->stream))) return rc;               // the host-built tree tops: one launch, now that whole trees are read
    for (uint32_t k = 0; k < s->queries; ++k)                                      // prover.rs:266-289 per query
        if ((rc = decommit(s, ch, (size_t)qraws[k] % (s->N - 2 * s->B)))) return rc;
    collect_timing(s);
    return ZK_OK;
}

// ---- known-pattern exchange (zk_shard_create, zk_shard_self_test) -------------------------------------------
// Every rank fills the piece it sends to peer p with pattern(rank, p, j), the exchanges run exactly as a commitment runs
// them (plain on the main stream; in chunks on the exchange stream when the layout has chunked layers), and a kernel
// checks that word j from peer q is pattern(q, rank, j).  Then an all-gather of 16 words per rank, checked on the host.
int self_test(zk_shard* s) {
    s->stats.selftest_ok = 0;
    if (!collectives(s)) { s->stats.selftest_ok = 1; return ZK_OK; }
    const double t0 = now_us();
    const int G = s->G;
    const uint32_t lg = s->lg, log_per = s->L - 2 * lg;           // the (rank, peer) piece of layer 0: the largest
    const size_t per = (size_t)1 << log_per;
    uint32_t* loc = layer_ptr(s, 0);
    uint32_t* res = s->d_small + 2048 - 8;
    const uint32_t* send[64];
    uint32_t* recv[64];
    int rc;
    auto verdict = [&](const char* what, uint32_t log_chunks) -> int {
        HIPCHK(hipMemsetAsync(res, 0xff, 8, s->stream));
        HIPCHK(hipMemsetAsync(res, 0, 4, s->stream));
        HIPCHK(launch_pattern_check(s->d_recv, (uint32_t)s->rank, log_per, lg, log_chunks, res, s->stream));
        HIPCHK(hipMemcpyAsync(s->h_small, res, 8, hipMemcpyDeviceToHost, s->stream));
        int r2 = sync_peers(s, s->stream, what);
        if (r2) return r2;
        if (s->h_small[0]) {
            const uint32_t i = s->h_small[1], cc = (uint32_t)(per >> log_chunks);
            const uint32_t q = log_chunks ? (i / cc) % (uint32_t)G : i / (uint32_t)per;
            return fail(ZK_ERR_HIP, "rank %d of %d: self-test of %s failed: %u of %zu received words are wrong, the first at receive index %u (the piece of rank %u)",
                        s->rank, G, what, s->h_small[0], per * (size_t)G, i, q);
        }
        return ZK_OK;
    };
    // plain all-to-all on the main stream
    HIPCHK(launch_pattern_fill(loc, (uint32_t)s->rank, log_per, lg, s->stream));
    HIPCHK(hipMemsetAsync(s->d_recv, 0, per * (size_t)G * 4, s->stream));
    for (int g = 0; g < G; ++g) { send[g] = loc + (size_t)g * per; recv[g] = s->d_recv + (size_t)g * per; }
    if ((rc = all_to_all(s, send, recv, per, s->stream))) return rc;
    if ((rc = verdict("the all-to-all on the main stream", 0))) return rc;
    // the chunked form on the exchange stream, if this prover will use it
    if (s->chunked_mask) {
        const uint32_t lk = zk_shard::kLogChunks, K = 1u << lk;
        const size_t cc = per >> lk;
        HIPCHK(hipMemsetAsync(s->d_recv, 0, per * (size_t)G * 4, s->stream));
        HIPCHK(hipEventRecord(s->ev_layer, s->stream));
        HIPCHK(hipStreamWaitEvent(s->xstream, s->ev_layer, 0));
        for (uint32_t c = 0; c < K; ++c) {
            for (int g = 0; g < G; ++g) { send[g] = loc + (size_t)g * per + (size_t)c * cc; recv[g] = s->d_recv + ((size_t)c * G + g) * cc; }
            if ((rc = all_to_all(s, send, recv, cc, s->xstream))) return rc;
        }
        HIPCHK(hipEventRecord(s->ev_chunk[0], s->xstream));
        HIPCHK(hipStreamWaitEvent(s->stream, s->ev_chunk[0], 0));
        if ((rc = verdict("the chunked all-to-all on the exchange stream", lk))) return rc;
    }
    // all-gather: 16 words per rank, on the main stream and -- if the halo of cp will travel there (cp_from_f with f exchanged
    // in chunks) -- on the exchange stream with its communicator
    auto gather_check = [&](hipStream_t st, uint32_t tag, const char* what) -> int {
        for (uint32_t j = 0; j < 16; ++j) s->h_small[j] = shard_pattern((uint32_t)s->rank, tag, j);
        HIPCHK(hipMemcpyAsync(s->d_small, s->h_small, 64, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemsetAsync(s->d_small + 64, 0, 64 * (size_t)G, s->stream));
        if (st != s->stream) {
            HIPCHK(hipEventRecord(s->ev_layer, s->stream));
            HIPCHK(hipStreamWaitEvent(st, s->ev_layer, 0));
        }
        int r2 = all_gather(s, s->d_small, s->d_small + 64, 16, st);
        if (r2) return r2;
        if (st != s->stream) {
            HIPCHK(hipEventRecord(s->ev_chunk[0], st));
            HIPCHK(hipStreamWaitEvent(s->stream, s->ev_chunk[0], 0));
        }
        HIPCHK(hipMemcpyAsync(s->h_small + 64, s->d_small + 64, 64 * (size_t)G, hipMemcpyDeviceToHost, s->stream));
        if ((r2 = sync_peers(s, s->stream, what))) return r2;
        for (int q = 0; q < G; ++q)
            for (uint32_t j = 0; j < 16; ++j)
                if (s->h_small[64 + 16 * q + j] != shard_pattern((uint32_t)q, tag, j))
                    return fail(ZK_ERR_HIP, "rank %d of %d: self-test of %s failed: word %u of rank %d's part is wrong", s->rank, G, what, j, q);
        return ZK_OK;
    };
    if ((rc = gather_check(s->stream, 0xA6u, "the all-gather on the main stream"))) return rc;
    if (s->cp_from_f && (s->chunked_mask & 1u))
        if ((rc = gather_check(s->xstream, 0x6Au, "the all-gather on the exchange stream"))) return rc;
    s->stats.selftest_ok = 1;
    s->stats.selftest_ms = (now_us() - t0) * 1e-3;
    return ZK_OK;
}

template <typename T>
int dalloc(zk_shard* s, T** p, size_t bytes) {
    hipError_t e = hipMalloc((void**)p, bytes ? bytes : 4);
    if (e != hipSuccess) return fail(ZK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    s->device_bytes += (double)bytes;
    return ZK_OK;
}

}  // namespace

extern "C" {

int zk_shard_unique_id(uint8_t id_out[ZK_SHARD_ID_BYTES]) {
    if (!id_out) return fail(ZK_ERR_INVALID, "zk_shard_unique_id: null argument");
    const RcclApi* api = rccl_api();
    if (!api) return ZK_ERR_STATE;
    static_assert(sizeof(ncclUniqueId) == ZK_SHARD_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(ZK_ERR_HIP, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
    memcpy(id_out, &id, ZK_SHARD_ID_BYTES);
    return ZK_OK;
}

// The layout of a sharded proof as a pure function of (world, sizes, options): which FRI layers stay distributed,
// which of them are exchanged in chunks, and the bytes every rank sends to its peers.  zk_shard_create uses it, and so
// does the test mirror (tests/sharded_mirror.py), so the two cannot drift apart.  No GPU needed.
int zk_shard_plan(int world, uint32_t log_n, uint32_t log_b, const zk_shard_options* user_opt, zk_shard_plan_info* user_out) {
    if (!user_out) return fail(ZK_ERR_INVALID, "zk_shard_plan: out is null");
    if (!abi_bytes(user_out, "zk_shard_plan")) return ZK_ERR_INVALID;
    zk_shard_options opt_local;
    if (user_opt && abi_get(user_opt, &opt_local, "zk_shard_plan")) return ZK_ERR_INVALID;
    const zk_shard_options* opt = user_opt ? &opt_local : nullptr;
    zk_shard_plan_info plan_local{};
    zk_shard_plan_info* const out = &plan_local;
    if (int rc = check_proof_size("zk_shard_plan", log_n, log_b)) return rc;
    if (world < 1 || world > 32) return fail(ZK_ERR_INVALID, "zk_shard_plan: world size %d out of range (1 .. 32)", world);
    uint32_t lg = 0;
    while ((1 << lg) < world) ++lg;
    if ((1 << lg) != world || lg > log_b)
        return fail(ZK_ERR_INVALID, "zk_shard_plan: world size %d must be a power of two dividing the blow-up %u", world, 1u << log_b);
    // Defaults (round 5).  A sharded commitment costs ~45 us more than the same layer inside the fused replicated tail (one rank
    // through RCCL, tools/shard_min_layer.py: +43, +26, +10 us per extra sharded layer) plus an all-to-all of a few hundred KiB
    // per link (latency, ~20-30 us); replicating a layer costs every rank the whole layer's hashing, (1 - 1/G) of which sharding
    // saves: 2^21 leaves 235 us, 2^20 145 us, 2^19 100 us.  So 2^21 values pay at every G, 2^20 from G = 4 on.  (Rounds 1-4: 22,
    // from a fixed cost of ~250 us measured before the committer, the board and the one-launch decommitment existed.)
    uint32_t min_layer_log = world >= 4 ? 20 : 21, min_chunk_log = 14, overlap_min_log = 21;
    bool force = false;
    if (opt) {
        if (opt->min_layer_log) min_layer_log = opt->min_layer_log;
        if (opt->min_chunk_log) min_chunk_log = opt->min_chunk_log;
        if (opt->overlap_min_log) overlap_min_log = opt->overlap_min_log;
        force = opt->force_collectives != 0;
        if (opt->plain_collectives || opt->peer_copy) overlap_min_log = 99;   // plain collectives only: nothing is exchanged in chunks
    }
    const uint32_t L = log_n + log_b, R = log_n;
    if (L < 2 * lg + min_chunk_log)
        return fail(ZK_ERR_INVALID, "zk_shard_plan: domain 2^%u is too small to shard over %d ranks: use zk_prove", L, world);
    // FRI layer rho (2^(L-rho) values) stays distributed while it has >= 2^min_layer_log values and a (rank, peer)
    // piece has >= 2^min_chunk_log leaves; layer 0 (cp) is always distributed, like f; at least one round is left
    // to the replicated tail (the last layers are tiny)
    uint32_t ns = 0;
    for (uint32_t rho = 0; rho <= R; ++rho)
        if (L - rho >= min_layer_log && L - rho >= 2 * lg + min_chunk_log) ++ns;
    if (ns < 1) ns = 1;
    if (ns > R - 1) ns = R - 1;
    out->world = (uint32_t)world; out->log_world = lg;
    out->sharded_layers = ns;
    out->tail_rounds = R - ns;
    out->min_layer_log = min_layer_log; out->min_chunk_log = min_chunk_log; out->overlap_min_log = overlap_min_log;
    out->log_chunks = zk_shard::kLogChunks;
    const bool coll = world > 1 || force;
    // cp over a rank's block is a function of f over the block and the 2B positions after it: with the block of f already
    // there (the commitment of f), cp is committed without an exchange.  Needs blocks of >= 2B leaves made of pieces of
    // >= 2B / G words (n >= 2 G); `exchange_cp` keeps the exchange (rounds 1-4; A/B).
    out->cp_from_f = (coll && !(opt && opt->exchange_cp) && ((size_t)1 << log_n) >= 2 * (size_t)world) ? 1u : 0u;
    // committed distributed layers: id 0 = f, id 1 + rho = FRI layer rho < ns; layer id has 2^m_log values in total
    for (uint32_t lid = 0; lid <= ns && lid < 32; ++lid) {
        const uint32_t m_log = lid == 0 ? L : L - (lid - 1);
        const uint32_t log_cnt = m_log - 2 * lg;                     // words per (rank, peer) piece
        const bool chunked = coll && log_cnt >= overlap_min_log && log_cnt >= zk_shard::kLogChunks + 8;
        out->piece_log[lid] = log_cnt;
        if (lid == 1 && out->cp_from_f) continue;                    // cp is recomputed from the received block of f: no exchange
        if (chunked) { out->chunked_mask |= 1u << lid; out->chunked_layers += 1; }
        const double sent = coll ? 4.0 * (double)((size_t)1 << log_cnt) * (world - 1) : 0.0;   // to the world - 1 peers
        out->all_to_all_bytes += sent;
        if (lid == 0) out->lde_commit_bytes = sent;
    }
    return abi_put(user_out, plan_local, "zk_shard_plan");
}

int zk_shard_destroy(zk_shard* s) {
    if (!s) return ZK_OK;
    (void)hipSetDevice(s->device);
    // after an error a collective may still sit on the streams waiting for a peer: abort the communicators first
    if (s->rccl && s->failed) {
        if (s->xcomm) { (void)s->rccl->CommAbort(s->xcomm); s->xcomm = nullptr; }
        if (s->comm) { (void)s->rccl->CommAbort(s->comm); s->comm = nullptr; }
    }
    if (s->failed && !s->rccl && !s->peer && collectives(s)) {      // (the peer-copy transport is host-synchronous: nothing of it ever waits on a stream)
        // a caller's transport cannot be aborted from here: if one of its collectives is still pending, waiting for the
        // streams (or freeing device memory, which synchronises the device) would hang for ever
        bool pending = false;
        for (hipStream_t st : {s->stream, s->xstream, s->bstream})
            if (st && hipStreamQuery(st) == hipErrorNotReady) pending = true;
        if (pending) {
            fprintf(stderr, "[zk_shard] rank %d of %d: a collective of the caller's transport is still pending after a failure; "
                            "the prover's streams and device buffers are leaked instead of waited for\n", s->rank, s->G);
            s->board.close();
            delete s;
            return ZK_OK;
        }
    }
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->xstream) (void)hipStreamSynchronize(s->xstream);
    if (s->bstream) (void)hipStreamSynchronize(s->bstream);
    if (s->xcomm && s->rccl) (void)s->rccl->CommDestroy(s->xcomm);
    if (s->comm && s->rccl) (void)s->rccl->CommDestroy(s->comm);
    s->board.close();
    if (s->peer) { s->peer->close(); delete s->peer; s->peer = nullptr; }
    if (s->tail) zk_ctx_destroy(s->tail);
    if (s->committer) zk_committer_destroy(s->committer);
    if (s->dom_loc) zk_dom_destroy(s->dom_loc);
    if (s->dom_glob) zk_dom_destroy(s->dom_glob);
    for (void* p : {(void*)s->d_inv_blk, (void*)s->d_halo_send, (void*)s->d_halo})
        if (p) (void)hipFree(p);
    if (s->ev_halo) (void)hipEventDestroy(s->ev_halo);
    for (void* p : {(void*)s->d_trace, (void*)s->d_coef, (void*)s->d_layers, (void*)s->d_trees, (void*)s->d_recv, (void*)s->d_gbuf,
                    (void*)s->d_repl, (void*)s->d_small, (void*)s->d_goff, (void*)s->d_gout, (void*)s->d_gall})
        if (p) (void)hipFree(p);
    if (s->d_fcounter) (void)hipFree(s->d_fcounter);
    for (void* p : {(void*)s->h_goff, (void*)s->h_gall, (void*)s->h_small, (void*)s->h_fitems, (void*)s->h_fout, (void*)s->h_fmail})
        if (p) (void)hipHostFree(p);
    if (s->ev_layer) (void)hipEventDestroy(s->ev_layer);
    if (s->ev_coll) (void)hipEventDestroy(s->ev_coll);
    for (auto& t : s->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->ev_chunk) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->ev_built) if (e) (void)hipEventDestroy(e);
    if (s->xstream) (void)hipStreamDestroy(s->xstream);
    if (s->bstream) (void)hipStreamDestroy(s->bstream);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return ZK_OK;
}

int zk_shard_create(int device, int rank, int world, const uint8_t* id, const zk_shard_transport* transport, const zk_shard_options* user_opt,
                    uint32_t log_n, uint32_t log_b, zk_shard** out) {
    if (!out) return fail(ZK_ERR_INVALID, "zk_shard_create: out is null");
    *out = nullptr;
    zk_shard_options opt_local;
    if (user_opt && abi_get(user_opt, &opt_local, "zk_shard_create")) return ZK_ERR_INVALID;
    const zk_shard_options* opt = user_opt ? &opt_local : nullptr;
    auto t0 = std::chrono::steady_clock::now();
    if (int rc = check_proof_size("zk_shard_create", log_n, log_b)) return rc;
    if (world < 1 || world > 32) return fail(ZK_ERR_INVALID, "zk_shard_create: world size %d out of range (1 .. 32)", world);
    uint32_t lg = 0;
    while ((1 << lg) < world) ++lg;
    if ((1 << lg) != world || lg > log_b || rank < 0 || rank >= world)
        return fail(ZK_ERR_INVALID, "zk_shard_create: world size %d must be a power of two dividing the blow-up %u, 0 <= rank < world", world, 1u << log_b);
    if (transport && (!transport->all_to_all || !transport->all_gather)) return fail(ZK_ERR_INVALID, "zk_shard_create: incomplete transport");
    if (!transport && !id) return fail(ZK_ERR_INVALID, "zk_shard_create: the built-in transports need the shared 128 bytes (zk_shard_unique_id, or any bytes for peer_copy)");
    HIPCHK(hipSetDevice(device));
    zk_shard* s = new (std::nothrow) zk_shard();
    if (!s) return fail(ZK_ERR_NOMEM, "out of host memory");
    s->device = device; s->rank = rank; s->G = world; s->lg = lg;
    s->log_n = log_n; s->log_b = log_b; s->L = log_n + log_b; s->R = log_n;
    s->n = (size_t)1 << log_n; s->B = (size_t)1 << log_b; s->N = s->n << log_b;
    if (opt) {
        if (opt->min_layer_log) s->min_layer_log = opt->min_layer_log;
        if (opt->min_chunk_log) s->min_chunk_log = opt->min_chunk_log;
        if (opt->overlap_min_log) s->overlap_min_log = opt->overlap_min_log;
        s->force = opt->force_collectives != 0;
        s->plain = opt->plain_collectives != 0 || opt->peer_copy != 0;
        s->single_build_stream = opt->single_build_stream != 0;
        s->single_comm = opt->single_communicator != 0;
    }
    // bound of every host-side wait on a peer: the option, else the environment (documented in INTEGRATION.md), else 120 s
    if (opt && opt->timeout_s > 0) s->timeout_s = opt->timeout_s;
    else if (const char* e = getenv("ZK_SHARD_TIMEOUT_S")) { const double v = atof(e); if (v > 0) s->timeout_s = v; }
    int rc = ZK_OK;
    auto bail = [&](int code) { zk_shard_destroy(s); return code; };
#define HIPCHK_S(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            zk_shard_destroy(s);                                                              \
            return fail(ZK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                     \
    } while (0)
    const uint32_t L = s->L, R = s->R;
    zk_shard_plan_info plan;
    plan.struct_size = (uint32_t)sizeof plan;
    if ((rc = zk_shard_plan(world, log_n, log_b, opt, &plan))) return bail(rc);
    s->min_layer_log = plan.min_layer_log; s->min_chunk_log = plan.min_chunk_log; s->overlap_min_log = plan.overlap_min_log;
    const uint32_t ns = plan.sharded_layers;
    s->n_sharded = ns;
    s->tail_rounds = plan.tail_rounds;
    s->chunked_mask = plan.chunked_mask;
    s->cp_from_f = plan.cp_from_f != 0;
    (void)R;
    HIPCHK_S(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    {   // the exchanges run beside the hashing: their workgroups should win the CU slots the hashing frees
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        HIPCHK_S(hipStreamCreateWithPriority(&s->xstream, hipStreamNonBlocking, greatest));
    }
    HIPCHK_S(hipEventCreateWithFlags(&s->ev_layer, hipEventDisableTiming));
    HIPCHK_S(hipEventCreateWithFlags(&s->ev_coll, hipEventDisableTiming));
    for (hipEvent_t& e : s->ev_chunk) HIPCHK_S(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t& e : s->ev_built) HIPCHK_S(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!s->single_build_stream) HIPCHK_S(hipStreamCreateWithFlags(&s->bstream, hipStreamNonBlocking));
    // small device / pinned scratch (subtree roots, flags, the second communicator's id, self-test results): needed by the set-up below
    if ((rc = dalloc(s, &s->d_small, 8192))) return bail(rc);
    HIPCHK_S(hipHostMalloc((void**)&s->h_small, 8192));
    // transport
    if (transport) {
        s->tp = *transport;
    } else if (opt && opt->peer_copy) {
        // plain peer copies between the GPUs of this node (peer.hpp): the rung below RCCL.  Host-synchronous, so nothing is
        // exchanged in chunks beside the hashing (zk_shard_plan: peer_copy implies plain collectives).
        uint8_t dg[32];
        Sha256 hsh; hsh.update(id, ZK_SHARD_ID_BYTES); hsh.update(reinterpret_cast<const uint8_t*>("peer"), 4); hsh.finalize(dg);
        char pname[64];
        snprintf(pname, sizeof pname, "/zkstark_amd_p%02x%02x%02x%02x%02x%02x%02x%02x", dg[0], dg[1], dg[2], dg[3], dg[4], dg[5], dg[6], dg[7]);
        s->peer = new (std::nothrow) PeerTransport();
        if (!s->peer) return bail(fail(ZK_ERR_NOMEM, "out of host memory"));
        // staging: the largest collective is a layer-0 all-to-all, G pieces of N / G^2 words = this rank's whole shard of f
        const size_t stage_bytes = std::max<size_t>(((size_t)1 << (log_n + log_b - lg)) * 4, (size_t)1 << 20);
        if (!s->peer->open(pname, rank, world, s->timeout_s, stage_bytes))
            return bail(fail(ZK_ERR_HIP, "zk_shard_create: peer-copy transport (rank %d of %d): %s", rank, world, s->peer->error.c_str()));
        s->device_bytes += (double)s->peer->stage_bytes;
        s->tp.user = s;
        s->tp.all_to_all = peer_all_to_all;
        s->tp.all_gather = peer_all_gather;
        s->stats.peer_copy = 1;
    } else {
        s->rccl = rccl_api();
        if (!s->rccl) return bail(ZK_ERR_STATE);
        ncclUniqueId nid;
        memcpy(&nid, id, sizeof nid);
        ncclResult_t r = s->rccl->CommInitRank(&s->comm, world, nid, rank);
        if (r != ncclSuccess) { s->comm = nullptr; return bail(fail(ZK_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, s->rccl->GetErrorString(r))); }
        // did RCCL form the communicator this rank believes it is in?
        int nr = -1, ur = -1;
        ncclResult_t r1 = s->rccl->CommCount(s->comm, &nr), r2 = s->rccl->CommUserRank(s->comm, &ur);
        if (r1 != ncclSuccess || r2 != ncclSuccess || nr != world || ur != rank) {
            s->failed = true;                                    // zk_shard_destroy aborts the communicator
            return bail(fail(ZK_ERR_HIP, "zk_shard_create: RCCL communicator mismatch: ncclCommCount = %d (expected %d), ncclCommUserRank = %d (expected %d)",
                             nr, world, ur, rank));
        }
        s->stats.rccl_nranks = (uint32_t)nr;
        s->tp.user = s;
        s->tp.all_to_all = rccl_all_to_all;
        s->tp.all_gather = rccl_all_gather;
        s->stats.native_rccl = 1;
        s->stats.communicators = 1;
        // The chunked exchanges run on xstream beside the hashing: they get a communicator of their own, so that no
        // communicator is ever driven from two streams.  Rank 0 draws the second id, the first communicator carries it.
        if (plan.chunked_mask && !s->single_comm) {
            ncclUniqueId id2;
            memset(&id2, 0, sizeof id2);
            if (rank == 0) {
                r = s->rccl->GetUniqueId(&id2);
                if (r != ncclSuccess) { s->failed = true; return bail(fail(ZK_ERR_HIP, "ncclGetUniqueId (exchange communicator) failed: %s", s->rccl->GetErrorString(r))); }
            }
            memcpy(s->h_small, &id2, sizeof id2);
            HIPCHK_S(hipMemcpyAsync(s->d_small, s->h_small, sizeof id2, hipMemcpyHostToDevice, s->stream));
            if ((rc = all_gather(s, s->d_small, s->d_small + 32, 32, s->stream))) { s->failed = true; return bail(rc); }
            HIPCHK_S(hipMemcpyAsync(s->h_small + 32, s->d_small + 32, sizeof id2, hipMemcpyDeviceToHost, s->stream));   // rank 0's part
            if ((rc = sync_peers(s, s->stream, "the distribution of the exchange communicator's id"))) { s->failed = true; return bail(rc); }
            memcpy(&id2, s->h_small + 32, sizeof id2);
            r = s->rccl->CommInitRank(&s->xcomm, world, id2, rank);
            if (r != ncclSuccess) { s->xcomm = nullptr; s->failed = true; return bail(fail(ZK_ERR_HIP, "ncclCommInitRank(exchange communicator, rank %d of %d) failed: %s", rank, world, s->rccl->GetErrorString(r))); }
            r1 = s->rccl->CommCount(s->xcomm, &nr); r2 = s->rccl->CommUserRank(s->xcomm, &ur);
            if (r1 != ncclSuccess || r2 != ncclSuccess || nr != world || ur != rank) {
                s->failed = true;
                return bail(fail(ZK_ERR_HIP, "zk_shard_create: exchange communicator mismatch: ncclCommCount = %d (expected %d), ncclCommUserRank = %d (expected %d)", nr, world, ur, rank));
            }
            s->stats.communicators = 2;
        }
    }
    // this rank's coset: shift w h^rank, blow-up B/G; the replicated tail: layer ns of the proof is layer 0 of the
    // domain with n' = n >> ns and shift w^(2^ns)
    const uint32_t h = root_of_unity(L);
    const uint32_t shift = mulmod(GEN_W, powmod(h, (uint64_t)rank));
    if ((rc = zk_dom_create(device, log_n, log_b - lg, shift, 0, &s->dom_loc))) return bail(rc);
    if ((rc = zk_tail_create(device, s->tail_rounds, log_b, powmod(GEN_W, (uint64_t)1 << ns), &s->tail))) return bail(rc);
    if ((rc = zk_committer_create(device, &s->committer))) return bail(rc);
    if (s->cp_from_f) {
        // cp over this rank's block from the block of f it receives: the global domain's constants, 1/(x - 1) at the positions
        // of the block and of the 2B after it (x = w h^(rank N/G + t)), and the buffers of the halo all-gather
        const size_t M = s->N >> lg, blk = M + 2 * s->B;
        if ((rc = zk_dom_create(device, log_n, log_b, GEN_W, 1, &s->dom_glob))) return bail(rc);
        if ((rc = dalloc(s, &s->d_inv_blk, blk * 4)) || (rc = dalloc(s, &s->d_halo_send, 2 * s->B * 4)) ||
            (rc = dalloc(s, &s->d_halo, 2 * s->B * 4 * (size_t)world)))
            return bail(rc);
        HIPCHK_S(launch_build_inv_xm1_range(s->d_inv_blk, blk, (uint32_t)((size_t)rank * M), L, s->dom_glob->H.view(), s->dom_glob->shift_mont, s->stream));
        HIPCHK_S(hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming));
    }
    // layers: 0 = f, 1 + rho = FRI layer rho < ns (distributed); one allocation for layers, one for trees
    const size_t NL = s->N >> lg;
    size_t off = 0;
    for (uint32_t lid = 0; lid <= ns; ++lid) {
        const size_t len = lid == 0 ? NL : ((s->N >> (lid - 1)) >> lg);
        s->layer_off.push_back(off); s->layer_len.push_back(len); off += len;
    }
    const size_t layer_words = off;
    off = 0;
    for (uint32_t t = 0; t <= ns; ++t) { s->tree_off.push_back(off); s->tree_leaves.push_back(s->layer_len[t]); off += (2 * s->layer_len[t] - 1) * 8; }
    const size_t tree_words = off;
    const size_t repl = s->N >> ns;                            // the first replicated layer
    s->gather_slots = (size_t)(4 + 2 * ns) * (L + 1) + 64;
    s->gather_words = (size_t)(4 + 2 * ns) * (1 + 8 * (size_t)L) + 64;
    if ((rc = dalloc(s, &s->d_trace, s->n * 4)) || (rc = dalloc(s, &s->d_coef, 2 * s->n * 4)) || (rc = dalloc(s, &s->d_layers, layer_words * 4)) ||
        (rc = dalloc(s, &s->d_trees, tree_words * 4)) || (rc = dalloc(s, &s->d_recv, NL * 4)) || (rc = dalloc(s, &s->d_gbuf, repl * 4)) ||
        (rc = dalloc(s, &s->d_repl, repl * 4)) || (rc = dalloc(s, &s->d_goff, s->gather_slots * 8)) ||
        (rc = dalloc(s, &s->d_gout, s->gather_words * 4)) || (rc = dalloc(s, &s->d_gall, s->gather_words * 4 * (size_t)world)))
        return bail(rc);
    HIPCHK_S(hipHostMalloc((void**)&s->h_goff, s->gather_slots * 8));
    HIPCHK_S(hipHostMalloc((void**)&s->h_gall, s->gather_words * 4 * (size_t)world));
    HIPCHK_S(hipHostMalloc((void**)&s->h_fitems, s->gather_slots * 8, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_S(hipHostMalloc((void**)&s->h_fout, s->gather_words * 4, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK_S(hipHostMalloc((void**)&s->h_fmail, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(s->h_fmail, 0, 64);
    HIPCHK_S(hipHostGetDevicePointer((void**)&s->dm_fitems, s->h_fitems, 0));
    HIPCHK_S(hipHostGetDevicePointer((void**)&s->dm_fout, s->h_fout, 0));
    HIPCHK_S(hipHostGetDevicePointer((void**)&s->dm_fmail, s->h_fmail, 0));
    if ((rc = dalloc(s, &s->d_fcounter, 64))) return bail(rc);
    HIPCHK_S(hipMemsetAsync(s->d_fcounter, 0, 64, s->stream));
    // Tree tops built on the host reach the device with ONE launch per proof, before the decommitment, instead of one launch per
    // commitment in front of the next layer's kernels (4.7 us + a launch gap on the path of every commitment: one rank through RCCL
    // 6.02 -> 5.93 ms per 2^24 proof)
    committer_set_lazy(s->committer, s->d_trees);
    s->tops.resize(ns + 1);
    s->device_bytes += (double)zk_ctx_device_bytes(s->tail);
    // A shared-memory page when every rank can map the object (one node): the abort words always, and the subtree roots
    // unless the options ask for the all-gather (no_root_board, plain_collectives)
    if (collectives(s) && id) {
        uint8_t dg[32];
        Sha256 hsh; hsh.update(id, ZK_SHARD_ID_BYTES); hsh.finalize(dg);
        char name[64];
        snprintf(name, sizeof name, "/zkstark_amd_%02x%02x%02x%02x%02x%02x%02x%02x", dg[0], dg[1], dg[2], dg[3], dg[4], dg[5], dg[6], dg[7]);
        bool ok = true, all = false;
        if (rank == 0) ok = s->board.open_or_create(name, rank, world, true, s->gather_words);
        if ((rc = agree(s, ok, &all))) return bail(rc);        // the object exists (or rank 0 failed) before anybody opens it
        if (all && rank != 0) ok = s->board.open_or_create(name, rank, world, false, s->gather_words);
        bool mapped = false;
        if ((rc = agree(s, all && ok, &mapped))) return bail(rc);   // everybody has mapped it: the name can go
        if (rank == 0) shm_unlink(name);
        s->have_board = mapped;
        s->use_board = mapped && !(opt && opt->no_root_board) && !s->plain;
        if (!mapped) s->board.close();
    }
    HIPCHK_S(hipStreamSynchronize(s->stream));
#undef HIPCHK_S
    committer_set_poll(s->committer, s->have_board ? poll_peer_abort : nullptr, s, s->timeout_s);
    // known-pattern exchange through the transport, at the size of the largest piece: nothing reaches a proof through a
    // transport that permutes or drops data, and RCCL's lazily built connections are up before the first proof
    if ((rc = self_test(s))) { s->failed = true; if (s->have_board) s->board.post_abort((uint32_t)(-rc)); return bail(rc); }
    s->stats.sharded_layers = ns;
    s->stats.root_board = s->use_board ? 1 : 0;
    s->stats.device_bytes = s->device_bytes;
    s->stats.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = s;
    return ZK_OK;
}

int zk_shard_trace_upload(zk_shard* s, const uint32_t* trace, size_t count) {
    if (!s || !trace) return fail(ZK_ERR_INVALID, "zk_shard_trace_upload: null argument");
    if (count != s->n - 1) return fail(ZK_ERR_INVALID, "zk_shard_trace_upload: expected n-1 = %zu values, got %zu", s->n - 1, count);
    for (size_t i = 0; i < count; ++i)
        if (trace[i] >= P) return fail(ZK_ERR_INVALID, "zk_shard_trace_upload: trace[%zu] = %u is not a canonical residue", i, trace[i]);
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpyAsync(s->d_trace, trace, count * 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemsetAsync(s->d_trace + count, 0, 4, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->first = trace[0];
    s->last = trace[count - 1];
    s->have_trace = true;
    return ZK_OK;
}

int zk_shard_prove_channel(zk_shard* s, zk_channel* chan) {
    if (!s || !chan) return fail(ZK_ERR_INVALID, "zk_shard_prove_channel: null argument");
    HIPCHK(hipSetDevice(s->device));
    if (s->failed) return fail(ZK_ERR_STATE, "zk_shard_prove_channel: this rank left an earlier proof with an error; destroy the prover");
    return rank_failed(s, prove(s, chan->ch));
}

int zk_shard_prove(zk_shard* s, uint8_t* proof_out, size_t cap, size_t* proof_len, uint8_t state_out[32]) {
    if (!s || !proof_out || !state_out) return fail(ZK_ERR_INVALID, "zk_shard_prove: null argument");
    HIPCHK(hipSetDevice(s->device));
    if (s->failed) return fail(ZK_ERR_STATE, "zk_shard_prove: this rank left an earlier proof with an error; destroy the prover");
    Channel ch;                                                // main.rs:19
    int rc = rank_failed(s, prove(s, ch));
    if (rc) return rc;
    if (proof_len) *proof_len = ch.data.size();
    if (ch.data.size() > cap) return fail(ZK_ERR_BUFFER, "zk_shard_prove: proof needs %zu bytes, buffer has %zu", ch.data.size(), cap);
    memcpy(proof_out, ch.data.data(), ch.data.size());         // channel.rs:34-36
    memcpy(state_out, ch.state, 32);
    return ZK_OK;
}

int zk_shard_lde_commit(zk_shard* s, uint8_t root_out[32]) {
    if (!s || !root_out) return fail(ZK_ERR_INVALID, "zk_shard_lde_commit: null argument");
    if (!s->have_trace) return fail(ZK_ERR_STATE, "zk_shard_lde_commit: no trace uploaded");
    HIPCHK(hipSetDevice(s->device));
    s->stats.sent_bytes = s->stats.all_to_all_bytes = 0;
    s->stats.chunked_layers = 0;
    if (s->failed) return fail(ZK_ERR_STATE, "zk_shard_lde_commit: this rank left an earlier call with an error; destroy the prover");
    reset_timing(s);
    committer_drop_pending(s->committer);
    int rc = do_lde(s);                                         // prover.rs:60-70, each rank its cosets
    if (!rc) rc = commit_sharded(s, 0, s->L, root_out);         // the all-to-all transpose + prover.rs:81
    if (!rc) rc = committer_flush(s->committer, s->stream);     // the tree of f complete on the device
    if (!rc) collect_timing(s);
    return rank_failed(s, rc);
}

// Merkle hash and number of queries: like zk_ctx_set_hash / zk_ctx_set_queries, the same on every rank.
int zk_shard_set_hash(zk_shard* s, int hash_kind) {
    if (!s) return fail(ZK_ERR_INVALID, "null prover");
    if (hash_kind != ZK_HASH_SHA256 && hash_kind != ZK_HASH_FIELD) return fail(ZK_ERR_INVALID, "zk_shard_set_hash: unknown hash %d", hash_kind);
    s->hash = hash_kind;
    return ZK_OK;
}
int zk_shard_set_queries(zk_shard* s, uint32_t n_queries) {
    if (!s) return fail(ZK_ERR_INVALID, "null prover");
    if (n_queries < 1 || n_queries > kMaxQueries) return fail(ZK_ERR_INVALID, "zk_shard_set_queries: need 1 <= n_queries <= %u", kMaxQueries);
    s->queries = n_queries;
    return ZK_OK;
}
// Test hook: this rank behaves as if its next proof had failed before the first collective (peers must not hang).
int zk_shard_inject_failure(zk_shard* s, int code) {
    if (!s) return fail(ZK_ERR_INVALID, "null prover");
    return rank_failed(s, fail(code < 0 ? code : ZK_ERR_STATE, "injected failure on rank %d", s->rank));
}

int zk_shard_self_test(zk_shard* s) {
    if (!s) return fail(ZK_ERR_INVALID, "null prover");
    HIPCHK(hipSetDevice(s->device));
    if (s->failed) return fail(ZK_ERR_STATE, "zk_shard_self_test: this rank left an earlier call with an error; destroy the prover");
    return rank_failed(s, self_test(s));
}
int zk_shard_set_profiling(zk_shard* s, int on) {
    if (!s) return fail(ZK_ERR_INVALID, "null prover");
    s->profiling = on != 0;
    return ZK_OK;
}

int zk_shard_last_transcript(const zk_shard* s, zk_transcript_info* out) {
    if (!s || !out) return fail(ZK_ERR_INVALID, "zk_shard_last_transcript: null argument");
    return abi_put(out, s->info, "zk_shard_last_transcript");
}

int zk_shard_layer_read(zk_shard* s, uint32_t layer, size_t offset, size_t count, uint32_t* out) {
    if (!s || (!out && count)) return fail(ZK_ERR_INVALID, "zk_shard_layer_read: null argument");
    if (layer > s->n_sharded || offset + count > s->layer_len[layer]) return fail(ZK_ERR_INVALID, "zk_shard_layer_read: out of range");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpyAsync(out, layer_ptr(s, layer) + offset, count * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return ZK_OK;
}

int zk_shard_get_stats(const zk_shard* s, zk_shard_stats* out) {
    if (!s || !out) return fail(ZK_ERR_INVALID, "zk_shard_get_stats: null argument");
    return abi_put(out, s->stats, "zk_shard_get_stats");
}

}  // extern "C"
