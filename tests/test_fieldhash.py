"""BASELINE.json configs[4]: the Merkle hash swapped for the build's field-native Poseidon2-style hash
(zkstark_amd/csrc/fieldhash.hpp).  "Parity: self-defined" -- the reference has only SHA-256 -- so the
checker is the oracle's independent plain-residue implementation of the same spec."""
import hashlib

import numpy as np
import pytest

P = 3221225473


@pytest.fixture()
def field_oracle(orc):
    orc.set_hash(orc.HASH_FIELD)
    yield orc
    orc.set_hash(orc.HASH_SHA256)


# ---- CPU: host-side verifier of the product against the oracle ------------------------------
def test_host_fieldhash_verifier_accepts_oracle_proofs(zk, field_oracle):
    for log_n, log_b in ((4, 1), (6, 2), (9, 3)):
        r = field_oracle.prove(log_n, log_b, want_vectors=False)
        assert r.rc == 0
        assert field_oracle.verify(r.proof, log_n, log_b, r.public_last) == 0
        zk.Proof(r.state, r.proof, log_n, log_b, r.public_last, hash="field").verify()
        with pytest.raises(zk.ZkError):           # the SHA-256 verifier must reject a field-hash proof
            zk.Proof(r.state, r.proof, log_n, log_b, r.public_last, hash="sha256").verify()


def test_host_fieldhash_path(zk, field_oracle):
    vals = np.arange(1, 17, dtype=np.uint32) * 1234567 % P
    nodes = field_oracle.merkle_build(vals)
    for leaf in (0, 5, 15):
        path = [bytes(h) for h in field_oracle.merkle_trace(nodes, leaf)]
        assert zk.compute_root_from_path(int(vals[leaf]), leaf, path, hash="field") == bytes(nodes[0])
        assert zk.compute_root_from_path(int(vals[leaf]), leaf, path, hash="sha256") != bytes(nodes[0])


def test_permutation_is_a_bijection_on_samples(field_oracle):
    """distinct inputs -> distinct outputs; permuting zero is not zero (constants are live)."""
    outs = set()
    for k in range(64):
        s = np.zeros(16, dtype=np.uint32)
        s[k % 16] = k + 1
        outs.add(field_oracle.fieldhash_permute(s).tobytes())
    assert len(outs) == 64
    assert field_oracle.fieldhash_permute(np.zeros(16, dtype=np.uint32)).any()


# ---- GPU ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("log_m", [0, 1, 4, 9, 12, 16, 19])
def test_fieldhash_merkle_matches_oracle(zk, field_oracle, log_m):
    rng = np.random.default_rng(900 + log_m)
    vals = rng.integers(0, P, size=1 << log_m, dtype=np.uint64).astype(np.uint32)
    vals[0] = 0
    got = zk.Merkle.new(1 << log_m, vals, hash="field").nodes
    want = field_oracle.merkle_build(vals)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(4, 3), (10, 3), (13, 3), (16, 3)])
def test_fieldhash_prover_matches_oracle(zk, field_oracle, log_n, log_b):
    want = field_oracle.prove(log_n, log_b, want_vectors=False)
    with zk.Context(log_n, log_b, hash="field") as ctx:
        proof = ctx.prove(zk.trace_fibsq((1 << log_n) - 1))
    assert proof.data == want.proof and proof.state == want.state
    proof.verify()
    assert field_oracle.verify(proof.data, log_n, log_b, want.public_last) == 0


@pytest.mark.gpu
def test_fieldhash_full_size_properties(zk):
    """2^20 leaves: the root is node(left root, right root); a path opens to the root."""
    rng = np.random.default_rng(77)
    vals = rng.integers(0, P, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    m = zk.Merkle.new(1 << 20, vals, hash="field")
    half = zk.Merkle.new(1 << 19, vals[:1 << 19], hash="field")
    assert half[0] == m[1]
    assert zk.compute_root_from_path(int(vals[777]), 777, m.trace(777), hash="field") == m[0]
    assert all(int.from_bytes(m[0][4 * i:4 * i + 4], "big") < P for i in range(8))
