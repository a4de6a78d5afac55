// peer.hpp -- the two collectives of zk_shard_transport as plain peer copies between the GPUs of ONE node: no RCCL.
//
// The fall-back rung of the sharded prover below "RCCL, plain collectives" (zk_shard_options.peer_copy; bench.py --gpus N
// tries it when RCCL cannot be brought up): every rank owns ONE staging buffer whose IPC handle (hipIpcGetMemHandle) it
// publishes on a POSIX shared-memory page at creation; the peers map it once (hipIpcOpenMemHandle).  Per collective a rank
// copies its pieces into its staging buffer, announces the collective on the page, and every rank PULLS its piece from each
// peer's staging buffer with a device-to-device copy on the caller's stream.  Host-synchronous by construction (a stream synchronisation and a meeting on the page before and after the
// copies), so nothing overlaps the hashing: it is there to yield a correct, measured line when a communicator cannot be
// formed, not to be fast.  Ranks that live in the same process (threads) hand each other raw pointers instead of handles.
//
// Protocol of collective k = 1, 2, ... (the same number on every rank):
//   1. my pieces are in my staging buffer (local copies, then hipStreamSynchronize); release-store posted = k;
//   2. for every peer q: wait posted[q] >= k, enqueue the copy of my piece out of q's staging buffer;
//   3. hipStreamSynchronize; release-store done = k; wait done[q] >= k for every q (nobody refills its staging buffer earlier).
#pragma once
#include <fcntl.h>
#ifndef ZK_PEER_NO_HIP
#include <hip/hip_runtime.h>
#endif
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

// Diagnostic build (ZK_BUILD_DEFS="-DZK_PEER_DEBUG=1"): every step of every collective on stderr with a time stamp.
#ifdef ZK_PEER_DEBUG
#define PEER_DBG(...) do { fprintf(stderr, "[peer %d %.6f] ", rank, now_s()); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } while (0)
#else
#define PEER_DBG(...) do { } while (0)
#endif

namespace zk {
namespace impl {

// ThreadSanitizer follows synchronisation by ADDRESS, and every rank maps the shared page at an address of its own: under TSan (the
// ranks are threads of one test process) the release / acquire pairs on the page are therefore mirrored on process-wide keys, one per
// (rank, word), so that the tool sees the happens-before the shared physical page provides.  No code in a normal build.
#if defined(__SANITIZE_THREAD__)
extern "C" void __tsan_acquire(void* addr);
extern "C" void __tsan_release(void* addr);
inline char* peer_tsan_key(int r, int word) { static char k[32][4]; return &k[r & 31][word & 3]; }
#define PEER_TSAN_RELEASE(r, w) __tsan_release(::zk::impl::peer_tsan_key((r), (w)))
#define PEER_TSAN_ACQUIRE(r, w) __tsan_acquire(::zk::impl::peer_tsan_key((r), (w)))
#else
#define PEER_TSAN_RELEASE(r, w) do { } while (0)
#define PEER_TSAN_ACQUIRE(r, w) do { } while (0)
#endif

// What the transport needs from the device runtime, as a policy: the protocol on the shared page (everything below) is plain host
// code and runs under ThreadSanitizer with host memory standing in for the GPU (tests/peer_check.cpp: -DZK_PEER_NO_HIP).
#ifndef ZK_PEER_NO_HIP
struct PeerHipOps {
    using stream_t = hipStream_t;
    static bool alloc(void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; }
    static void release(void* p) { (void)hipFree(p); }
    static bool export_handle(void* p, uint8_t out[64], std::string& err) {
        hipIpcMemHandle_t h;
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
        if (hipIpcGetMemHandle(&h, p) != hipSuccess) { err = std::string("hipIpcGetMemHandle failed: ") + hipGetErrorString(hipGetLastError()); return false; }
        memcpy(out, &h, 64);
        return true;
    }
    static void* import_handle(const uint8_t in[64], std::string& err) {
        hipIpcMemHandle_t h;
        memcpy(&h, in, 64);
        void* base = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { err = std::string("hipIpcOpenMemHandle failed: ") + hipGetErrorString(e); return nullptr; }
        return base;
    }
    static void unimport(void* p) { (void)hipIpcCloseMemHandle(p); }
    static bool copy(void* dst, const void* src, size_t bytes, stream_t st) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess; }
    static bool sync(stream_t st) { return hipStreamSynchronize(st) == hipSuccess; }
    static const char* last_error() { return hipGetErrorString(hipGetLastError()); }
};
#endif

template <class Ops>
struct PeerTransportT {
    using stream_t = typename Ops::stream_t;
    static constexpr int kMaxWorld = 32;
    static constexpr uint64_t kMagic = 0x7a6b706565723032ull;      // "zkpeer02"
    struct Slot {
        uint64_t posted, done;                                     // collective numbers (release / acquire)
        uint64_t words;                                            // of the collective `posted` announces (consistency check)
        uint32_t abort_code, ready;                                // ready: the staging buffer below is published
        uint8_t handle[64];                                        // hipIpcMemHandle_t of this rank's staging buffer
        uint64_t stage_bytes, raw, pid;                            // its size, its address in the owner's process (same-process peers), the owner
        uint64_t pad[2];
    };
    struct Page {
        uint64_t magic;
        uint32_t mapped;                                           // ranks that have mapped the page
        uint32_t pad[13];
        Slot slot[kMaxWorld];
    };
    Page* page = nullptr;
    int rank = 0, G = 1;
    double timeout_s = 120.0;
    uint64_t seq = 0;
    std::string name, error;
    // Every piece travels through ONE staging buffer per rank, allocated here and mapped by the peers once, at creation: the
    // caller's own allocations are never exported.  (The first version published a handle of whatever allocation a send pointer
    // lay in; mapping the 3.2 GB tree array of a 2^24-element shard for an 8-word all-gather never returned from
    // hipIpcOpenMemHandle, in both processes at once -- docs/LOG.md, round 6 item 5.  The extra local copy is ~10 us per 32 MB.)
    uint32_t* stage = nullptr;
    size_t stage_bytes = 0;
    const char* peer_stage[kMaxWorld] = {};                        // peers' staging buffers as mapped here (own: stage)
    bool opened[kMaxWorld] = {};
    int bad_peer = -1;

    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // Rank 0 creates the page, the others wait for it (bounded); the name goes once every rank has mapped it.  Then every rank
    // allocates and publishes its staging buffer (bytes: the largest collective, G pieces of the largest piece) and maps its peers'.
    bool open(const char* shm_name, int rank_, int world, double timeout, size_t bytes) {
        rank = rank_; G = world; timeout_s = timeout; name = shm_name;
        if (world > kMaxWorld) { error = "world size above 32"; return false; }
        const double t0 = now_s();
        int fd = -1;
        if (rank == 0) {
            shm_unlink(shm_name);
            fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)sizeof(Page)) != 0) { error = "cannot create the shared page"; if (fd >= 0) ::close(fd); return false; }
        } else {
            for (;;) {
                fd = shm_open(shm_name, O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == sizeof(Page)) break;
                if (fd >= 0) { ::close(fd); fd = -1; }
                if (now_s() - t0 > timeout_s) { error = "rank 0 never created the shared page of the peer-copy transport"; return false; }
                struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr);
            }
        }
        void* p = mmap(nullptr, sizeof(Page), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) { error = "mmap of the shared page failed"; return false; }
        page = static_cast<Page*>(p);
        if (rank == 0) __atomic_store_n(&page->magic, kMagic, __ATOMIC_RELEASE);       // a fresh object is zero-filled
        while (__atomic_load_n(&page->magic, __ATOMIC_ACQUIRE) != kMagic) {
            if (now_s() - t0 > timeout_s) { error = "the shared page of the peer-copy transport was never initialised"; return false; }
            sched_yield();
        }
        __atomic_fetch_add(&page->mapped, 1u, __ATOMIC_ACQ_REL);
        bool all = true;
        while (__atomic_load_n(&page->mapped, __ATOMIC_ACQUIRE) < (uint32_t)G) {
            if (now_s() - t0 > timeout_s) {
                error = "peer-copy transport: only " + std::to_string(__atomic_load_n(&page->mapped, __ATOMIC_ACQUIRE)) + " of " + std::to_string(G) +
                        " ranks mapped the shared page within " + std::to_string((int)timeout_s) + " s";
                all = false;
                break;
            }
            sched_yield();
        }
        if (rank == 0) shm_unlink(shm_name);                        // the name goes in every case: nothing is left in /dev/shm
        if (!all) return false;
        // the staging buffer: allocate, publish, map the peers'
        stage_bytes = (bytes + 255) & ~(size_t)255;
        if (!Ops::alloc((void**)&stage, stage_bytes)) { error = "allocation of the staging buffer failed"; stage = nullptr; return false; }
        Slot& my = page->slot[rank];
        if (!Ops::export_handle(stage, my.handle, error)) return false;
        my.stage_bytes = stage_bytes; my.raw = (uint64_t)(uintptr_t)stage; my.pid = (uint64_t)getpid();
        PEER_TSAN_RELEASE(rank, 2);
        __atomic_store_n(&my.ready, 1u, __ATOMIC_RELEASE);
        for (int q = 0; q < G; ++q) {
            Slot& sl = page->slot[q];
            while (!__atomic_load_n(&sl.ready, __ATOMIC_ACQUIRE)) {
                if (now_s() - t0 > timeout_s) { error = "rank " + std::to_string(q) + " never published its staging buffer (peer-copy transport)"; return false; }
                sched_yield();
            }
            PEER_TSAN_ACQUIRE(q, 2);
            if (q == rank || sl.pid == (uint64_t)getpid()) { peer_stage[q] = (const char*)(uintptr_t)sl.raw; continue; }   // a thread of this process
            const double t_open = now_s();
            PEER_DBG("mapping the staging buffer of rank %d (%llu bytes)", q, (unsigned long long)sl.stage_bytes);
            std::string why;
            void* base = Ops::import_handle(sl.handle, why);
            if (!base) { error = "mapping the staging buffer of rank " + std::to_string(q) + ": " + why; return false; }
            if (now_s() - t_open > 0.25)                           // an anomaly worth a line: normally milliseconds
                fprintf(stderr, "[zk_shard] rank %d: peer-copy transport: mapping the staging buffer of rank %d took %.2f s\n", rank, q, now_s() - t_open);
            peer_stage[q] = (const char*)base;
            opened[q] = true;
        }
        return true;
    }
    void close() {
        for (int q = 0; q < kMaxWorld; ++q)
            if (opened[q]) { Ops::unimport((void*)peer_stage[q]); opened[q] = false; }
        if (stage) { Ops::release(stage); stage = nullptr; }
        if (page) munmap(page, sizeof(Page));
        page = nullptr;
    }
    void post_abort(uint32_t code) {
        if (page) __atomic_store_n(&page->slot[rank].abort_code, code ? code : 1u, __ATOMIC_RELEASE);
    }

    // 0, or 1 with `error` set
    int wait_word(const uint64_t* word, uint64_t want, int q, const char* what) {
        const double t0 = now_s();
        uint64_t spins = 0;
        while (__atomic_load_n(word, __ATOMIC_ACQUIRE) < want) {
            if ((++spins & 255) == 0) {
                for (int r = 0; r < G; ++r) {
                    const uint32_t c = __atomic_load_n(&page->slot[r].abort_code, __ATOMIC_ACQUIRE);
                    if (c && r != rank) { bad_peer = r; error = "rank " + std::to_string(r) + " left the proof with error " + std::to_string(c) + " (peer-copy transport, " + what + ")"; return 1; }
                }
                if (now_s() - t0 > timeout_s) {
                    bad_peer = q;
                    error = "rank " + std::to_string(rank) + ": timed out after " + std::to_string((int)timeout_s) + " s waiting for rank " + std::to_string(q) + " (" + what +
                            " of peer-copy collective #" + std::to_string(want) + ")";
                    return 1;
                }
            }
            if (spins > 4096) sched_yield();
        }
        return 0;
    }
    // all-to-all: send[p] (words) goes to rank p, recv[q] comes from rank q; all-gather: send_all to everybody, recv_all[q * words ..] from q
    int exchange(const uint32_t* const* send, const uint32_t* send_all, uint32_t* const* recv, uint32_t* recv_all, size_t words, stream_t st) {
        if (!page || !stage) { error = "peer-copy transport is closed"; return 1; }
        const size_t need = (send ? (size_t)G : (size_t)1) * words * 4;
        if (need > stage_bytes) { error = "peer-copy transport: a collective of " + std::to_string(need) + " bytes exceeds the staging buffer (" + std::to_string(stage_bytes) + ")"; return 1; }
        const uint64_t k = ++seq;
        PEER_DBG("collective #%llu (%s, %zu words): staging", (unsigned long long)k, send ? "all-to-all" : "all-gather", words);
        // 1. my pieces into my staging buffer (stream-ordered behind their producers), then the stream drained
        for (int p = 0; p < (send ? G : 1); ++p) {
            const void* src = send ? (const void*)send[p] : (const void*)send_all;
            if (!Ops::copy((char*)stage + (size_t)p * words * 4, src, words * 4, st)) { error = "staging copy failed"; return 1; }
        }
        if (!Ops::sync(st)) { error = std::string("stream synchronisation before a peer-copy collective failed: ") + Ops::last_error(); return 1; }
        Slot& my = page->slot[rank];
        my.words = words;
        PEER_TSAN_RELEASE(rank, 0);
        __atomic_store_n(&my.posted, k, __ATOMIC_RELEASE);
        PEER_DBG("#%llu: posted", (unsigned long long)k);
        // 2. pull my piece from every peer's staging buffer
        for (int i = 0; i < G; ++i) {
            const int q = (rank + i) % G;                          // every rank starts with a different peer
            if (wait_word(&page->slot[q].posted, k, q, "the pieces")) return 1;
            PEER_TSAN_ACQUIRE(q, 0);
            if (page->slot[q].words != words) { error = "peer-copy transport: rank " + std::to_string(q) + " runs a collective of another size (the ranks diverged)"; return 1; }
            const char* src = peer_stage[q] + (send ? (size_t)rank * words * 4 : (size_t)0);
            void* dst = recv ? (void*)recv[q] : (void*)(recv_all + (size_t)q * words);
            if (!Ops::copy(dst, src, words * 4, st)) { error = "peer copy failed"; return 1; }
        }
        PEER_DBG("#%llu: copies enqueued, synchronising", (unsigned long long)k);
        if (!Ops::sync(st)) { error = std::string("peer copies failed: ") + Ops::last_error(); return 1; }
        // 3. nobody overwrites a staging buffer before every peer has read it
        PEER_TSAN_RELEASE(rank, 1);
        __atomic_store_n(&my.done, k, __ATOMIC_RELEASE);
        for (int q = 0; q < G; ++q) {
            if (wait_word(&page->slot[q].done, k, q, "the end of the copies")) return 1;
            PEER_TSAN_ACQUIRE(q, 1);
        }
        PEER_DBG("#%llu: complete", (unsigned long long)k);
        return 0;
    }
};

#ifndef ZK_PEER_NO_HIP
using PeerTransport = PeerTransportT<PeerHipOps>;
#endif

}  // namespace impl
}  // namespace zk
