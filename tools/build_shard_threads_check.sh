#!/bin/bash
# Builds tests/shard_threads_check.c against the library of THIS tree, stamped with that library's build hash, into the path
# given as $1 (default: a fresh file under /tmp).  The binary refuses to run against any other build of the library
# (docs/LOG.md, round 6 item 1: a harness binary must never outlive the library it was compiled against), so session scripts
# call this first instead of keeping a binary under tools/.  Prints the path.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$(mktemp /tmp/shard_threads_check.XXXXXX)}"
HASH="$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); from zkstark_amd import build; print(build.built_hash() or 'unknown')")"
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -DZK_EXPECT_BUILD_HASH="\"$HASH\"" -I/opt/rocm/include -I"$ROOT/include" "$ROOT/tests/shard_threads_check.c" \
    -L"$ROOT/zkstark_amd" -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,"$ROOT/zkstark_amd" -Wl,-rpath,/opt/rocm/lib -o "$OUT"
echo "$OUT"
