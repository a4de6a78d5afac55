#!/bin/bash
# round 6, session 5: where the latency phase of a field-hash tree should switch between its forms, re-swept with the quad form
# present (round 5 swept ROW_MAX before the quad form existed): each variant built on the box and parity-checked.
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "" "-DZK_FIELD_ROW_MAX_NODES=16" "-DZK_FIELD_ROW_MAX_NODES=16 -DZK_FIELD_QUAD_MAX_NODES=128" "" "-DZK_FIELD_ROW_MAX_NODES=16"; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 400 python -m pytest tests/test_fieldhash.py tests/test_gpu_kernels.py -m gpu -x -q -k "fieldhash or config5 or field" > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; tail -20 $O/pytest_$tag.log; continue; }
    timeout -k 10 300 python bench.py --hash field --steps 30 --warmup 3 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/field_$tag.json 2> $O/field_$tag.err
    python3 - "$v" $O $tag >> $O/summary.txt <<'PY'
import json, sys
v, O, tag = sys.argv[1:4]
f = json.load(open(f"{O}/field_{tag}.json"))
top = [(x["launches"], round(x["ms"], 4)) for x in f["stages"] if x["kernel"] == "merkle_top"][0]
print(f"{v or '(default: row <= 32, quad <= 64)':60s} field {f['ms_per_step']:.3f} ms, merkle_top {top}")
PY
    tail -1 $O/summary.txt
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
