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
    for _ in range(8):                                                             # (the checker drew 16 words for that block too)
        x = (x * 1664525 + 1013904223) & 0xffffffff
    stream = bytearray()
    for _ in range(5 * 64):                                                        # host_sha_blocks: five raw blocks
        x = (x * 1664525 + 1013904223) & 0xffffffff
        stream.append(x >> 24)
    return [n.hex() for n in nodes], hashlib.sha256(msg).hexdigest(), _compress_only(bytes(stream)).hex()


def _compress_only(data):
    """SHA-256 state after whole blocks, no padding (what the byte-stream entry point leaves in `state`)."""
    K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
         0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
         0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
         0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
         0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
         0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
         0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
    M = 0xffffffff
    ror = lambda v, n: ((v >> n) | (v << (32 - n))) & M
    st = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    assert len(data) % 64 == 0
    for off in range(0, len(data), 64):
        w = list(struct.unpack(">16I", data[off:off + 64]))
        for t in range(16, 64):
            s0 = ror(w[t - 15], 7) ^ ror(w[t - 15], 18) ^ (w[t - 15] >> 3)
            s1 = ror(w[t - 2], 17) ^ ror(w[t - 2], 19) ^ (w[t - 2] >> 10)
            w.append((w[t - 16] + s0 + w[t - 7] + s1) & M)
        a, b, c, d, e, f, g, h = st
        for t in range(64):
            t1 = (h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & M & g)) + K[t] + w[t]) & M
            t2 = ((ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & M
            a, b, c, d, e, f, g, h = (t1 + t2) & M, a, b, c, (d + t1) & M, e, f, g
        st = [(x + y) & M for x, y in zip(st, (a, b, c, d, e, f, g, h))]
    return struct.pack(">8I", *st)


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
    nodes, chain, stream = _expected(depth, seed)
    assert out[2:-2] == nodes
    assert out[-2] == chain
    assert out[-1] == stream
