// board.hpp -- exchange of the G subtree roots of a commitment between the ranks of ONE node through POSIX
// shared memory (host only, header-only so that tests/board_check.cpp can run it under ThreadSanitizer).
//
// A commitment of the sharded prover ends with every rank needing all G roots on the host, to hash the top of the
// tree and feed the channel (prover.rs:85).  As a device collective that is an all-gather plus a device-to-host
// read, ~100 us of fixed cost for 256 bytes; through a shared page it is one store and G polled loads.
//
// The page also carries one ABORT word per rank: a rank that leaves a proof with an error posts its code there, and
// every rank waiting in an exchange sees it within microseconds instead of waiting out the timeout.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <chrono>

namespace zk {
namespace impl {

struct RootBoard {
    static constexpr int kRing = 4;        // a rank cannot run more than one exchange ahead of the slowest one
    static constexpr int kSlotWords = 16;  // 8 digest words, words 8-9 = the 64-bit sequence number (8-byte aligned)
    static constexpr int kBlobRing = 2;    // blobs: a rank posts exchange k + 1 only after it has read every peer's blob of exchange k
    enum Status { kOk = 0, kTimeout = 1, kPeerAborted = 2 };
    uint32_t* slots = nullptr;             // [kRing][G][kSlotWords], then [G] abort words, then the blob area
    size_t bytes = 0;
    size_t blob_words = 0;                 // capacity of one blob (0: no blob area); a blob is followed by 16 header words (sequence number)
    int G = 0, rank = 0;
    // filled by a failed exchange: the rank that was waited for (or that aborted), and its abort code
    int bad_peer = -1;
    uint32_t bad_code = 0;

    uint32_t* abort_words() const { return slots + (size_t)kRing * G * kSlotWords; }
    // Blob area: the decommitment's contributions (prover.rs:266-289: the values and path nodes each rank owns) travel like
    // the roots do, as one store and G polled loads instead of an all-gather plus a device-to-host copy.
    uint32_t* blob(uint64_t seq, int q) const {
        return abort_words() + (size_t)((G + 15) & ~15) + ((size_t)(seq % kBlobRing) * G + (size_t)q) * (blob_words + 16);
    }

    bool open_or_create(const char* name, int rank_, int world, bool create, size_t blob_words_ = 0) {
        G = world; rank = rank_;
        blob_words = (blob_words_ + 15) & ~(size_t)15;
        bytes = ((size_t)kRing * world * kSlotWords + (size_t)((world + 15) & ~15) + (blob_words ? (size_t)kBlobRing * world * (blob_words + 16) : 0)) * sizeof(uint32_t);
        int fd;
        if (create) {
            shm_unlink(name);                                       // a stale object of a crashed run
            fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) return false;
            if (ftruncate(fd, (off_t)bytes) != 0) { ::close(fd); shm_unlink(name); return false; }
        } else {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) != 0 || (size_t)st.st_size != bytes) { ::close(fd); return false; }
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        slots = static_cast<uint32_t*>(p);                          // a fresh object is zero-filled: sequence numbers start at 1
        return true;
    }
    void close() {
        if (slots) munmap(slots, bytes);
        slots = nullptr;
    }
    // This rank is leaving the protocol with an error: every peer's pending and future exchange fails at once.
    void post_abort(uint32_t code) {
        if (slots) __atomic_store_n(abort_words() + rank, code ? code : 1u, __ATOMIC_RELEASE);
    }
    // -1, or the lowest rank that has posted an abort (its code in bad_code)
    int aborted_peer() {
        if (!slots) return -1;
        for (int q = 0; q < G; ++q) {
            const uint32_t c = __atomic_load_n(abort_words() + q, __ATOMIC_ACQUIRE);
            if (c) { bad_code = c; return q; }
        }
        return -1;
    }
    // Exchange number seq = 1, 2, ... (the same on every rank; 64 bits: it never wraps into the zero-filled state).
    // mine: this rank's root as 8 state words; all: [G][8] out.  kTimeout: rank bad_peer never posted exchange seq
    // (it died or diverged); kPeerAborted: rank bad_peer left with error bad_code.
    Status exchange(uint64_t seq, const uint32_t mine[8], uint32_t* all, double timeout_s = 120.0) {
        uint32_t* row = slots + (size_t)(seq % kRing) * G * kSlotWords;
        uint32_t* my = row + (size_t)rank * kSlotWords;
        for (int i = 0; i < 8; ++i) __atomic_store_n(my + i, mine[i], __ATOMIC_RELAXED);
        __atomic_store_n(reinterpret_cast<uint64_t*>(my + 8), seq, __ATOMIC_RELEASE);   // after the digest
        const auto t0 = std::chrono::steady_clock::now();
        for (int q = 0; q < G; ++q) {
            const uint32_t* src = row + (size_t)q * kSlotWords;
            const Status st = wait_seq(reinterpret_cast<const uint64_t*>(src + 8), seq, q, t0, timeout_s);
            if (st != kOk) return st;
            for (int i = 0; i < 8; ++i) all[(size_t)q * 8 + i] = __atomic_load_n(src + i, __ATOMIC_RELAXED);
        }
        return kOk;
    }
    // Blob exchange number seq = 1, 2, ... (its own sequence, the same on every rank).  blob_post publishes this rank's
    // `words` words (<= blob_words); blob_wait returns rank q's blob of that exchange in place (valid until this rank has
    // posted exchange seq + 2: a caller that keeps the data beyond its next blob_post copies it out, as shard.hip's decommit does).
    void blob_post(uint64_t seq, const uint32_t* data, size_t words) {
        uint32_t* my = blob(seq, rank);
        memcpy(my, data, words * sizeof(uint32_t));
        __atomic_store_n(reinterpret_cast<uint64_t*>(my + blob_words), seq, __ATOMIC_RELEASE);   // after the data
    }
    // t0: when the caller started waiting for THIS exchange (one clock for all peers, as exchange() has: G waits of timeout_s
    // each would add up to G * timeout_s)
    Status blob_wait(uint64_t seq, int q, const uint32_t** data, double timeout_s = 120.0,
                     std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now()) {
        const uint32_t* src = blob(seq, q);
        const Status st = wait_seq(reinterpret_cast<const uint64_t*>(src + blob_words), seq, q, t0, timeout_s);
        if (st == kOk) *data = src;
        return st;
    }

  private:
    Status wait_seq(const uint64_t* word, uint64_t seq, int q, std::chrono::steady_clock::time_point t0, double timeout_s) {
        uint64_t spins = 0;
        while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {
            ++spins;
            if (spins < 4096) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
                continue;
            }
            // a peer is late by more than a few microseconds: stop burning the core it may need (ranks as threads
            // under a CPU quota, RCCL proxy threads), look for aborts, and watch the clock
            if ((spins & 63) == 0) {
                const int ab = aborted_peer();
                if (ab >= 0) { bad_peer = ab; return kPeerAborted; }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { bad_peer = q; return kTimeout; }
            }
            if (spins < 65536) sched_yield();
            else { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }
        }
        return kOk;
    }
};

}  // namespace impl
}  // namespace zk
