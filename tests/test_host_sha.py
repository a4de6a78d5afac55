"""csrc/host_sha.cpp (the host thread's share of the Merkle trees: tops and small FRI layers, plus the
transcript compressions) against hashlib, through every code path: portable, x86 SHA extensions (one or two nodes at a
time), and sixteen nodes at a time on AVX-512 registers for levels that wide."""
import hashlib
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hsha") / "host_sha_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "host_sha_check.cpp"),
                           os.path.join(ROOT, "zkstark_amd", "csrc", "host_sha.cpp"), "-o", exe])
    return exe


def _expected(depth, seed):
    m, x = 1 << depth, seed
    vals = []
    for i in range(m):
        x = (x * 1664525 + 1013904223) & 0xffffffff
        vals.append(0 if i == 0 else 0xffffffff if i == 1 else 3221225472 if i == 2 else x)
    nodes = [None] * (2 * m - 1)
    for i, v in enumerate(vals):
        nodes[m - 1 + i] = hashlib.sha256(struct.pack(">I", v)).digest()          # merkle.rs:30-34
    for j in range(m - 2, -1, -1):
        nodes[j] = hashlib.sha256(nodes[2 * j + 1] + nodes[2 * j + 2]).digest()    # merkle.rs:40-46
    msg = b""
    for _ in range(40):                                                            # 2 full blocks + 8 words
        x = (x * 1664525 + 1013904223) & 0xffffffff
        msg += struct.pack(">I", x)
    # the checker's third block is words 32..39 of the stream followed by standard padding for 160 bytes
    return [n.hex() for n in nodes], hashlib.sha256(msg).hexdigest()


@pytest.mark.parametrize("ext", [0, 1, 2])
@pytest.mark.parametrize("depth,seed", [(0, 1), (1, 2), (3, 7), (4, 5), (5, 3), (8, 12345), (10, 99)])
def test_host_sha_matches_hashlib(checker, ext, depth, seed):
    out = subprocess.run([checker, str(ext), str(depth), str(seed)], capture_output=True, text=True, check=True).stdout.split()
    have_ext, have_wide = out[0] == "1", out[1] == "1"
    if ext and not have_ext:
        pytest.skip("CPU without SHA extensions: only the portable path exists here")
    if ext == 2 and not have_wide:
        pytest.skip("CPU without AVX-512F: no sixteen-at-a-time path here")
    assert have_ext == bool(ext) and have_wide == (ext == 2)
    nodes, chain = _expected(depth, seed)
    assert out[2:-1] == nodes
    assert out[-1] == chain
