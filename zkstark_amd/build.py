"""Builds libzkstark_amd.so (HIP kernels + host prover + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.

Every translation unit is compiled to its own object (in parallel, cached under build/ by a hash of
the source, every header and the flags), then linked.  The library carries a hash of all sources and
headers (`zk_build_hash()`, compiled into version.cpp only, so an edit recompiles one file plus that
stub); `_lib.load()` compares it with the tree and rebuilds or refuses a stale binary.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libzkstark_amd.so")
HASHFILE = LIB + ".hash"
SOURCES = ["kernels.hip", "ntt_fast.hip", "domain.hip", "zkstark.hip", "batch.hip", "shard.hip", "host_sha.cpp", "version.cpp"]
HEADERS = ["field.hpp", "sha256.hpp", "sha256_quad.hpp", "fieldhash.hpp", "fieldhash_f64.hpp", "kernels.hpp", "transcript.hpp", "host_sha.hpp", "internal.hpp", "pool.hpp",
           "shard.hpp", "board.hpp", "peer.hpp", os.path.join("..", "..", "include", "zkstark_amd.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function"]
# build-time experiments (e.g. ZK_BUILD_DEFS="-DZK_MONT_VARIANT=4"): part of the flags, hence of the source hash
FLAGS += os.environ.get("ZK_BUILD_DEFS", "").split()


def _read(rel):
    with open(os.path.join(CSRC, rel), "rb") as f:
        return f.read()


def source_hash():
    """Hash of everything the library is built from (sources, headers, flags, this file's lists)."""
    h = hashlib.sha256()
    for rel in SOURCES + HEADERS:
        h.update(os.path.basename(rel).encode() + b"\0")
        h.update(_read(rel))
        h.update(b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:24]


def built_hash():
    """Hash the existing library was built from, or None.  The authority is the string compiled into the binary
    (version.cpp: "zkstark_amd.build_hash=<hash>", also what zk_build_hash() returns); the git-ignored sidecar file is
    only a shortcut and is not needed."""
    if not os.path.exists(LIB):
        return None
    try:
        with open(HASHFILE) as f:
            side = f.read().strip()
        if side and os.path.getmtime(HASHFILE) >= os.path.getmtime(LIB):
            return side
    except OSError:
        pass
    try:
        with open(LIB, "rb") as f:
            blob = f.read()
        i = blob.find(b"zkstark_amd.build_hash=")
        if i < 0:
            return None
        tail = blob[i + len(b"zkstark_amd.build_hash="):i + len(b"zkstark_amd.build_hash=") + 64]
        return tail.split(b"\0", 1)[0].decode("ascii", "replace") or None
    except OSError:
        return None


def is_stale():
    return built_hash() != source_hash()


def build(force=False, verbose=False, if_missing_only=False):
    """Rebuilds when the library is missing or was built from other sources.  force: always relink (objects
    still come from the cache when their inputs are unchanged).  if_missing_only: kept for callers that must
    never compile (several ranks importing at once still serialise on the lock)."""
    if not force and os.path.exists(LIB) and (if_missing_only or not is_stale()):
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:           # one builder at a time across processes
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB) and (if_missing_only or not is_stale()):
            return LIB
        return _build(verbose)


def _compile(hipcc, src, want, headers_blob, verbose):
    """Object for `src`: from the cache when source, headers and flags are unchanged."""
    extra = [f'-DZK_SRC_HASH="{want}"'] if src == "version.cpp" else []
    key = hashlib.sha256(_read(src) + b"\0" + (want.encode() if extra else headers_blob) + " ".join(FLAGS).encode()).hexdigest()[:20]
    obj = os.path.join(OBJDIR, f"{os.path.splitext(src)[0]}.{key}.o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(OBJDIR):                    # drop earlier versions of this object
        if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".o"):
            os.unlink(os.path.join(OBJDIR, old))
    tmp = obj + f".tmp{os.getpid()}"
    cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, obj)
    return obj


def _build(verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libzkstark_amd.so")
    os.makedirs(OBJDIR, exist_ok=True)
    want = source_hash()
    headers_blob = b"\0".join(_read(h) for h in HEADERS)
    workers = max(1, min(len(SOURCES), (os.cpu_count() or 2) - 1, 6))
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        objs = list(ex.map(lambda s: _compile(hipcc, s, want, headers_blob, verbose), SOURCES))
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl", "-lpthread", "-lrt"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)                              # atomic: a concurrent loader never sees a partial file
    with open(HASHFILE + ".tmp", "w") as f:
        f.write(want + "\n")
    os.replace(HASHFILE + ".tmp", HASHFILE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
