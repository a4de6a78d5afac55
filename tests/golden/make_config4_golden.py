#!/usr/bin/env python3
"""tests/golden/config4_2e26.json: BASELINE.json configs[3] on the CPU oracle -- the Merkle root (SHA-256 tree,
merkle.rs:14-51) of the 2^26-point LDE (prover.rs:60-70 at trace group 2^23, blow-up 8) of the canonical Fibonacci-square
trace, the first 64 values and the SHA-256 of the vector.  About a minute on 8 cores; the field evaluations and the tree
are quantities the reference pins (unique interpolant + SHA-256), so the row is "pinned", not "derived".

    python tests/golden/make_config4_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

import oracle  # noqa: E402

log_n, log_b = 23, 3
a = oracle.trace_fibsq((1 << log_n) - 1)
f = oracle.lde(a, log_n, log_b)
root = bytes(oracle.merkle_build(f)[0])
out = {"params": {"log_n": log_n, "log_blowup": log_b, "a0": 1, "a1": 3141592, "domain": 1 << (log_n + log_b)},
       "pinned": {"f_eval_root": root.hex(), "f_eval_head": [int(v) for v in f[:64]], "trace_last": int(a[-1]),
                  "f_eval_sha256": hashlib.sha256(np.ascontiguousarray(f, dtype="<u4").tobytes()).hexdigest()}}
with open(os.path.join(HERE, "config4_2e26.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(out["pinned"]["f_eval_root"])
