// shard.hpp -- internals of the sharded prover (shard.hip): one proof across the GPUs of a node.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "internal.hpp"

namespace zk {
namespace impl {

// RCCL, loaded at run time (librccl.so.1): a process that never shards needs no RCCL, and a process that has
// PyTorch loaded gets PyTorch's copy (same SONAME), i.e. the one that matches the HIP runtime already in use.
struct RcclApi;
const RcclApi* rccl_api();                 // nullptr + fail() recorded when the library cannot be loaded

// zkstark.hip: consulted by a committer while it waits for posted digests (non-zero return ends the wait)
// timeout_s > 0: also the bound of that wait
void committer_set_poll(zk_committer* k, int (*poll)(void*), void* user, double timeout_s = 0.0);

// zkstark.hip: lazy committer (the sharded prover).  trees_base: the device array every d_nodes of a later commit points into;
// the host-built tree tops then stay in the committer's staging buffer until committer_flush copies them into the array with
// ONE launch on `s` (before anything on the device reads the upper levels of a tree: the decommitment).  nullptr: eager again.
void committer_set_lazy(zk_committer* k, uint32_t* trees_base);
int committer_flush(zk_committer* k, hipStream_t s);
void committer_drop_pending(zk_committer* k);       // a proof that starts over forgets what an aborted one left behind

// zkstark.hip: cp over one rank's block from the block of f it received, and the subtree over it (ComposeBlockArgs)
int dev_compose_block_commit(zk_committer* k, const zk_dom* glob, ComposeBlockArgs geom, uint32_t first, uint32_t last, const uint32_t alpha_raw[3],
                             uint32_t* d_nodes, hipStream_t s, int hash_kind, uint8_t root_out[32], int (*enqueued)(void*) = nullptr, void* user = nullptr);

}  // namespace impl
}  // namespace zk
