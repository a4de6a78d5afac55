// version.cpp -- the only translation unit that sees the source hash (build.py passes -DZK_SRC_HASH), so an
// edit anywhere else recompiles that file and this stub, not the whole library.
#include "../../include/zkstark_amd.h"

#ifndef ZK_SRC_HASH
#define ZK_SRC_HASH "unknown"
#endif

// The hash sits behind a marker so that build.py can read it from the file without loading the library.
static const char kBuildTag[] = "zkstark_amd.build_hash=" ZK_SRC_HASH;

extern "C" {
const char* zk_version(void) { return "zkstark_amd 0.4 (gfx950)"; }
uint32_t zk_abi_version(void) { return ZK_ABI_VERSION; }
const char* zk_build_hash(void) { return kBuildTag + sizeof("zkstark_amd.build_hash=") - 1; }
}
