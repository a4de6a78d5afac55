#!/bin/bash
# A/B of the quad form (one field hash on four lanes, levels of 33 .. 128 nodes per workgroup; ZK_FIELD_QUAD_MAX_NODES: 0 = never)
O=${1:-gpurun_out/ab_fquad}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "-DZK_FIELD_QUAD_MAX_NODES=0" "-DZK_FIELD_QUAD_MAX_NODES=64" "-DZK_FIELD_QUAD_MAX_NODES=128" "-DZK_FIELD_QUAD_MAX_NODES=0" "-DZK_FIELD_QUAD_MAX_NODES=64" "-DZK_FIELD_QUAD_MAX_NODES=128"; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 300 python -m pytest tests/test_fieldhash.py -m gpu -x -q > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; tail -5 $O/pytest_$tag.log; continue; }
    timeout -k 10 300 python bench.py --hash field --steps 20 --warmup 3 --no-secondary --soak-seconds 0 > $O/field_$tag.json 2> $O/field_$tag.err
    python3 - "$v" $O/field_$tag.json >> $O/summary.txt <<'PY'
import json, sys
v, f = sys.argv[1:3]
b = json.load(open(f))
top = [(x["launches"], x["ms"]) for x in b["stages"] if x["kernel"] == "merkle_top"][0]
print(f"{v:34s} field-hash 2^24 proof {b['ms_per_step']:.3f} ms, merkle_top {top}, parity {b.get('parity_checked')}")
PY
    tail -1 $O/summary.txt
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
