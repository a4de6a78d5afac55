"""Builds libzkstark_amd.so (HIP kernels + host prover + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzkstark_amd.so")
SOURCES = ["kernels.hip", "ntt_fast.hip", "domain.hip", "zkstark.hip", "batch.hip", "host_sha.cpp"]
HEADERS = ["field.hpp", "sha256.hpp", "fieldhash.hpp", "kernels.hpp", "transcript.hpp", "host_sha.hpp", "internal.hpp", "pool.hpp", os.path.join("..", "..", "include", "zkstark_amd.h")]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, if_missing_only=False):
    """force: always rebuild.  if_missing_only: never rebuild an existing library (what import-time
    loading uses: several ranks may import at once, and a snapshot copy does not preserve mtimes)."""
    if not force and os.path.exists(LIB) and (if_missing_only or not _stale()):
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:           # one builder at a time across processes
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB) and (if_missing_only or not _stale()):
            return LIB
        return _build(verbose)


def _build(verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libzkstark_amd.so")
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)                              # atomic: a concurrent loader never sees a partial file
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
