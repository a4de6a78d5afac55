// board.hpp -- exchange of the G subtree roots of a commitment between the ranks of ONE node through POSIX
// shared memory (host only, header-only so that tests/board_check.cpp can run it under ThreadSanitizer).
//
// A commitment of the sharded prover ends with every rank needing all G roots on the host, to hash the top of the
// tree and feed the channel (prover.rs:85).  As a device collective that is an all-gather plus a device-to-host
// read, ~100 us of fixed cost for 256 bytes; through a shared page it is one store and G polled loads.
#pragma once
#include <fcntl.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>

namespace zk {
namespace impl {

struct RootBoard {
    static constexpr int kRing = 4;        // a rank cannot run more than one exchange ahead of the slowest one
    uint32_t* slots = nullptr;             // [kRing][G][16]: 8 digest words, word 8 = sequence number
    size_t bytes = 0;
    int G = 0, rank = 0;

    bool open_or_create(const char* name, int rank_, int world, bool create) {
        G = world; rank = rank_;
        bytes = (size_t)kRing * world * 16 * sizeof(uint32_t);
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
    // Exchange number seq = 1, 2, ... (the same on every rank).  mine: this rank's root as 8 state words;
    // all: [G][8] out.  Returns false on timeout (a rank died or diverged).
    bool exchange(uint32_t seq, const uint32_t mine[8], uint32_t* all, double timeout_s = 120.0) {
        uint32_t* row = slots + (size_t)(seq % kRing) * G * 16;
        uint32_t* my = row + (size_t)rank * 16;
        for (int i = 0; i < 8; ++i) __atomic_store_n(my + i, mine[i], __ATOMIC_RELAXED);
        __atomic_store_n(my + 8, seq, __ATOMIC_RELEASE);            // after the digest
        auto t0 = std::chrono::steady_clock::now();
        for (int q = 0; q < G; ++q) {
            const uint32_t* src = row + (size_t)q * 16;
            uint64_t spins = 0;
            while (__atomic_load_n(src + 8, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
                if ((++spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
            }
            for (int i = 0; i < 8; ++i) all[(size_t)q * 8 + i] = __atomic_load_n(src + i, __ATOMIC_RELAXED);
        }
        return true;
    }
};

}  // namespace impl
}  // namespace zk
