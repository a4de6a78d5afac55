"""The product's host-only code (transcript.hpp: channel, wire format, verifier; fieldhash.hpp on the
host) compiled with g++ -fsanitize=address,undefined and run on valid, truncated, bit-flipped and random
proofs.  CPU only (GPU sanitizers are not available on the pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "host_sanitizer_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "host_sanitizer_check.cpp"),
                           os.path.join(ROOT, "zkstark_amd", "csrc", "host_sha.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("log_n,log_b,hash_kind", [(6, 2, 0), (10, 3, 0), (5, 1, 1)])
def test_verifier_memory_safety(orc, checker, tmp_path, log_n, log_b, hash_kind):
    try:
        orc.set_hash(hash_kind)
        r = orc.prove(log_n, log_b, want_vectors=False)
    finally:
        orc.set_hash(0)
    assert r.rc == 0
    p = tmp_path / "proof.bin"
    p.write_bytes(r.proof)
    out = subprocess.run([checker, str(p), str(log_n), str(log_b), str(r.public_last), str(hash_kind)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok: valid accepted" in out.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_batch_pool_under_sanitizers(tmp_path, san):
    """The fork-join pool of the batched prover (csrc/pool.hpp): lock-free job hand-off with spinning and blocking
    workers, run under ThreadSanitizer and ASan/UBSan."""
    exe = str(tmp_path / "pool_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", f"-fsanitize={san}", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "pool_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pool ok" in out.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_root_board_under_sanitizers(tmp_path, san):
    """The shared-memory exchange of subtree roots between the ranks of the sharded prover (csrc/board.hpp):
    1-8 ranks, thousands of exchanges with skewed timing, size mismatch, a dead rank (timeout)."""
    exe = str(tmp_path / "board_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", f"-fsanitize={san}", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "board_check.cpp"), "-o", exe, "-lrt"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "board ok" in out.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_peer_copy_protocol_under_sanitizers(tmp_path, san):
    """The peer-copy transport of the sharded prover (csrc/peer.hpp, round 6) with host memory standing in for the GPU (the device
    runtime is a policy of the transport; -DZK_PEER_NO_HIP): 1-8 ranks as threads, thousands of all-to-alls and all-gathers of
    changing sizes with skewed timing and every word checked, a collective larger than the staging buffer, a rank that leaves
    (its peers return at once, naming it), a rank that never comes (bounded; nothing left in /dev/shm).  Under ThreadSanitizer the
    release / acquire pairs on the shared page are mirrored on process-wide keys (every rank maps the page at its own address);
    dropping the second meeting of a collective makes this test fail with data races (checked by hand in round 6)."""
    exe = str(tmp_path / "peer_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", f"-fsanitize={san}", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "peer_check.cpp"), "-o", exe, "-lrt"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr[-4000:]
    assert "peer transport protocol ok" in out.stdout
