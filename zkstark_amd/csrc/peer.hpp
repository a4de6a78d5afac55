// peer.hpp -- the two collectives of zk_shard_transport as plain peer copies between the GPUs of ONE node: no RCCL.
//
// The fall-back rung of the sharded prover below "RCCL, plain collectives" (zk_shard_options.peer_copy; bench.py --gpus N
// tries it when RCCL cannot be brought up): every rank publishes, per collective, where the piece for each peer lies -- an
// IPC handle of the allocation (hipIpcGetMemHandle) plus a byte offset -- on a POSIX shared-memory page, the receiver maps
// the allocation once (hipIpcOpenMemHandle, cached) and PULLS its piece with a device-to-device copy on the caller's
// stream.  Host-synchronous by construction (a stream synchronisation and a meeting on the page before and after the
// copies), so nothing overlaps the hashing: it is there to yield a correct, measured line when a communicator cannot be
// formed, not to be fast.  Ranks that live in the same process (threads) hand each other raw pointers instead of handles.
//
// Protocol of collective k = 1, 2, ... (the same number on every rank):
//   1. my send pieces are complete (hipStreamSynchronize); write my G entries; release-store posted = k;
//   2. for every peer q: wait posted[q] >= k, read q's entry for me, enqueue the copy;
//   3. hipStreamSynchronize; release-store done = k; wait done[q] >= k for every q (nobody reuses a send buffer earlier).
// One message buffer per rank suffices: a rank writes the entries of k + 1 only after step 3 of k.
#pragma once
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <vector>

namespace zk {
namespace impl {

struct PeerTransport {
    static constexpr int kMaxWorld = 32;
    static constexpr uint64_t kMagic = 0x7a6b706565723031ull;      // "zkpeer01"
    struct Entry {                                                 // where the piece for one destination lies
        uint8_t handle[64];                                        // hipIpcMemHandle_t of the allocation
        uint64_t offset;                                           // byte offset of the piece inside it
        uint64_t raw;                                              // the sender's own pointer (same-process peers)
        uint64_t pid;
    };
    struct Slot {
        uint64_t posted, done;
        uint32_t abort_code, pad[11];
        Entry to[kMaxWorld];
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
    struct Opened { uint8_t handle[64]; void* base; };
    std::vector<Opened> opened[kMaxWorld];                         // peer allocations this rank has mapped
    struct Mine { void* base; hipIpcMemHandle_t handle; };
    std::vector<Mine> mine;                                        // this rank's allocations with their handles
    int bad_peer = -1;

    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // Rank 0 creates the page, the others wait for it (bounded); the name goes once every rank has mapped it.
    bool open(const char* shm_name, int rank_, int world, double timeout) {
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
        return all;
    }
    void close() {
        for (int q = 0; q < kMaxWorld; ++q) {
            for (auto& o : opened[q]) (void)hipIpcCloseMemHandle(o.base);
            opened[q].clear();
        }
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
    bool fill(Entry& e, const void* ptr) {
        memset(&e, 0, sizeof e);
        e.raw = (uint64_t)(uintptr_t)ptr;
        e.pid = (uint64_t)getpid();
        if (!ptr) return true;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) != hipSuccess) { error = "hipMemGetAddressRange failed for a send buffer"; return false; }
        e.offset = (uint64_t)((const char*)ptr - (const char*)base);
        for (auto& m : mine)
            if (m.base == base) { memcpy(e.handle, &m.handle, 64); return true; }
        Mine m{base, {}};
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
        if (hipIpcGetMemHandle(&m.handle, base) != hipSuccess) { error = std::string("hipIpcGetMemHandle failed: ") + hipGetErrorString(hipGetLastError()); return false; }
        mine.push_back(m);
        memcpy(e.handle, &m.handle, 64);
        return true;
    }
    const void* resolve(int q, const Entry& e) {
        if (e.pid == (uint64_t)getpid()) return (const void*)(uintptr_t)e.raw;        // a thread of this process
        for (auto& o : opened[q])
            if (!memcmp(o.handle, e.handle, 64)) return (const char*)o.base + e.offset;
        hipIpcMemHandle_t h;
        memcpy(&h, e.handle, 64);
        void* base = nullptr;
        const hipError_t err = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
        if (err != hipSuccess) { error = std::string("hipIpcOpenMemHandle(rank ") + std::to_string(q) + ") failed: " + hipGetErrorString(err); return nullptr; }
        Opened o;
        memcpy(o.handle, e.handle, 64);
        o.base = base;
        opened[q].push_back(o);
        return (const char*)base + e.offset;
    }
    // send[p] != nullptr for every p (all-to-all) or one pointer for everybody (all-gather: send_all)
    int exchange(const uint32_t* const* send, const uint32_t* send_all, uint32_t* const* recv, uint32_t* recv_all, size_t words, hipStream_t st) {
        if (!page) { error = "peer-copy transport is closed"; return 1; }
        if (hipStreamSynchronize(st) != hipSuccess) { error = "hipStreamSynchronize before a peer-copy collective failed"; return 1; }
        const uint64_t k = ++seq;
        Slot& my = page->slot[rank];
        for (int p = 0; p < G; ++p)
            if (!fill(my.to[p], send ? (const void*)send[p] : (const void*)send_all)) return 1;
        __atomic_store_n(&my.posted, k, __ATOMIC_RELEASE);
        for (int i = 0; i < G; ++i) {
            const int q = (rank + i) % G;                          // every rank starts with a different peer
            if (wait_word(&page->slot[q].posted, k, q, "the send pointers")) return 1;
            Entry e;
            memcpy(&e, &page->slot[q].to[rank], sizeof e);
            const void* src = q == rank ? (const void*)(uintptr_t)e.raw : resolve(q, e);
            if (!src) return 1;
            void* dst = recv ? (void*)recv[q] : (void*)(recv_all + (size_t)q * words);
            if (dst != src && hipMemcpyAsync(dst, src, words * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { error = "peer copy failed"; return 1; }
        }
        if (hipStreamSynchronize(st) != hipSuccess) { error = std::string("peer copies failed: ") + hipGetErrorString(hipGetLastError()); return 1; }
        __atomic_store_n(&my.done, k, __ATOMIC_RELEASE);
        for (int q = 0; q < G; ++q)
            if (wait_word(&page->slot[q].done, k, q, "the end of the copies")) return 1;
        return 0;
    }
};

}  // namespace impl
}  // namespace zk
