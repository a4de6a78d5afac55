// host_sha.hpp -- SHA-256 of Merkle nodes on the host CPU (x86 SHA extensions).
//
// Why the host hashes at all: the top of a Merkle tree is a chain of dependent hashes.  One wave
// needs ~4.4 us per level (2 293 dependent-issue instructions), a CPU core with SHA-NI ~55 ns per
// node.  The one-call prover therefore lets the device build each tree down to 2^H nodes and
// finishes the 2^H - 1 nodes above on the thread that runs the Fiat-Shamir channel anyway
// (merkle.rs:40-46 is the definition either way).  Digests are handled as the eight big-endian
// state words, the form the device stores them in, so no byte swapping happens on this path.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zk {

// true when the CPU has the SHA extensions (checked once)
bool host_sha_available();
// on = false forces the portable code path (tests compare the two); on = true re-enables the extensions if present
void host_sha_use_extensions(bool on);
// true when levels of >= 16 nodes are hashed sixteen at a time on 512-bit registers (AVX-512F, needs the SHA extensions
// too: the narrow levels stay on them); host_sha_use_wide(false) keeps everything on the SHA unit (tests, A/B)
bool host_sha_wide_available();
void host_sha_use_wide(bool on);
// one compression of a 64-byte block given as sixteen big-endian-decoded words (transcript hashing)
void host_sha_compress(uint32_t state[8], const uint32_t block[16]);
// `blocks` consecutive 64-byte blocks of a byte stream (transcript hashing: the state stays in registers from block to
// block and the big-endian decoding is one byte shuffle per 16 bytes)
void host_sha_blocks(uint32_t state[8], const uint8_t* data, size_t blocks);
// out = SHA256(be32(v)) as state words (merkle.rs:30-34)
void host_sha_leaf(uint32_t v, uint32_t out[8]);
// out[8 i ..] = SHA256(be32(vals[i])) for i < n
void host_sha_leaves(const uint32_t* vals, size_t n, uint32_t* out);
// out = SHA256(left || right), all as state words (merkle.rs:42-45)
void host_sha_inner(const uint32_t left[8], const uint32_t right[8], uint32_t out[8]);
// cnt consecutive nodes of one level from their 2 cnt children (contiguous, 8 words each)
void host_sha_inner_run(const uint32_t* children, uint32_t* out, size_t cnt);
// Heap levels above a full level: nodes[(2^depth - 1) ..] holds the 2^depth digests of level `depth`
// (8 words each, heap order as merkle.rs:14-51); fills levels depth-1 .. 0 in place.
void host_sha_reduce(uint32_t* nodes, uint32_t depth);
// The same for one sub-tree only: levels depth-1 .. top below node `sub` of level `top` (sub < 2^top).  A team of
// threads takes one sub-tree each, then one thread finishes with host_sha_reduce(nodes, top).
void host_sha_reduce_sub(uint32_t* nodes, uint32_t depth, uint32_t top, size_t sub);

}  // namespace zk
