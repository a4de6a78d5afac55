#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the CPU oracle (oracle/).

The reference (a Rust bin crate) cannot be built or run in this environment, so the
fixtures are produced by the oracle in BOTH of its modes (literal polynomial.rs
arithmetic and the NTT restatement, which must agree) after the oracle itself has
been pinned to the reference's own known answers (tests/test_oracle_reference.py).
Rows that depend only on quantities the reference pins are tagged "pinned"; rows that
also depend on the bincode transcript encoding are tagged "derived" (SURVEY.md App. C).

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

import oracle  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


def canonical():
    r = oracle.prove(10, 3, mode=oracle.MODE_NTT)
    rn = oracle.prove(10, 3, mode=oracle.MODE_NAIVE)
    assert r.rc == 0 and rn.rc == 0 and r.proof == rn.proof and r.state == rn.state
    assert np.array_equal(r.f_eval, rn.f_eval)
    x = r.query_raw % (8192 - 16)
    return {
        "params": {"log_n": 10, "log_blowup": 3, "a0": 1, "a1": 3141592},
        "pinned": {
            "generator": oracle.generator(),
            "trace_last": int(r.trace[1022]),
            "virtual_point": oracle.virtual_point(r.trace, 10),
            "f_eval_head": [int(v) for v in r.f_eval[:3]],
            "f_eval_tail": [int(v) for v in r.f_eval[-3:]],
            "f_eval_sha256": sha(r.f_eval),
            "f_eval_root": bytes(r.roots[0]).hex(),
            "cp_degree": int(rn.cp_degree),
        },
        "derived": {
            "alpha_raw": r.alpha_raw,
            "cp_eval_sha256": sha(r.cp_layers[0]),
            "roots": [bytes(x_).hex() for x_ in r.roots],
            "beta_raw": r.beta_raw,
            "layer_sha256": [sha(l) for l in r.cp_layers],
            "free_term": r.free_term,
            "query_raw": r.query_raw,
            "query_index": x,
            "opened": [int(r.f_eval[x]), int(r.f_eval[x + 8]), int(r.f_eval[x + 16]), int(r.cp_layers[0][x])],
            "proof_len": len(r.proof),
            "proof_sha256": hashlib.sha256(r.proof).hexdigest(),
            "final_state": r.state.hex(),
            "proof_hex": r.proof.hex(),
        },
    }


def other_sizes():
    rows = []
    for log_n, log_b, a1 in [(2, 1, 3141592), (4, 3, 7), (5, 2, 11), (6, 2, 3141592), (8, 3, 123456789),
                             (12, 3, 99), (14, 1, 3141592), (15, 3, 3141592)]:
        r = oracle.prove(log_n, log_b, 1, a1)
        assert r.rc == 0
        if log_n <= 8:
            rn = oracle.prove(log_n, log_b, 1, a1, mode=oracle.MODE_NAIVE)
            assert rn.rc == 0 and rn.proof == r.proof, (log_n, log_b)
        rows.append({"log_n": log_n, "log_blowup": log_b, "a1": a1, "public_last": r.public_last,
                     "f_eval_sha256": sha(r.f_eval), "f_eval_root": bytes(r.roots[0]).hex(),
                     "last_root": bytes(r.roots[-1]).hex(), "free_term": r.free_term,
                     "proof_len": len(r.proof), "proof_sha256": hashlib.sha256(r.proof).hexdigest(),
                     "final_state": r.state.hex()})
    return rows


def merkle_vectors():
    rows = []
    rng = np.random.default_rng(7)
    for log_m in (0, 1, 3, 6, 11, 12, 15):
        vals = rng.integers(0, oracle.P, size=1 << log_m, dtype=np.uint64).astype(np.uint32)
        nodes = oracle.merkle_build(vals)
        rows.append({"log_m": log_m, "seed": 7, "values_sha256": sha(vals), "root": bytes(nodes[0]).hex(),
                     "nodes_sha256": hashlib.sha256(nodes.tobytes()).hexdigest()})
    return rows


if __name__ == "__main__":
    with open(os.path.join(HERE, "stark101_canonical.json"), "w") as f:
        json.dump(canonical(), f, indent=1)
    with open(os.path.join(HERE, "prover_sizes.json"), "w") as f:
        json.dump(other_sizes(), f, indent=1)
    with open(os.path.join(HERE, "merkle_random.json"), "w") as f:
        json.dump(merkle_vectors(), f, indent=1)
    print("golden fixtures written")
