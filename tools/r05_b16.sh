#!/bin/bash
set -o pipefail
O=gpurun_out/r05b16; mkdir -p $O
for lb in 4 3; do
  timeout -k 10 400 python bench.py --steps 20 --no-cpu-baseline --soak-seconds 0 --in-flight 1 --batch-log $lb > $O/bench_lb$lb.json 2> $O/bench_lb$lb.err || { tail -5 $O/bench_lb$lb.err; exit 1; }
  python3 - $O/bench_lb$lb.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); b = d["batched_2e24"]
print(b["proofs"], "proofs:", round(b["ms_per_proof"], 3), "ms per proof; two batches in flight", round(b["two_batches_in_flight"]["ms_per_proof"], 3), "equal", b["every_proof_equals_zk_prove"], "GB", round(b["device_bytes"] / 1e9, 1), {k: v for k, v in b.items() if "floor" in k or "frac" in k})
P
done
