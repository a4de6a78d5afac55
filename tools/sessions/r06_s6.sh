#!/bin/bash
# round 6, session 6: the row form for few LEAVES of the field hash (ZK_FIELD_ROW_LEAF_MAX) against leaves one lane per hash, with
# the re-swept level thresholds (row <= 16, quad <= 64) as the default: each variant built on the box and parity-checked.
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "-DZK_FIELD_ROW_LEAF_MAX=0" "" "-DZK_FIELD_ROW_LEAF_MAX=16" "-DZK_FIELD_ROW_LEAF_MAX=0" ""; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 400 python -m pytest tests/test_fieldhash.py tests/test_gpu_kernels.py tests/test_gpu_shard_native.py -m gpu -x -q -k "fieldhash or config5 or field" > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; tail -20 $O/pytest_$tag.log; continue; }
    timeout -k 10 300 python bench.py --hash field --steps 30 --warmup 3 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/field_$tag.json 2> $O/field_$tag.err
    python3 - "$v" $O $tag >> $O/summary.txt <<'PY'
import json, sys
v, O, tag = sys.argv[1:4]
f = json.load(open(f"{O}/field_{tag}.json"))
top = [(x["launches"], round(x["ms"], 4)) for x in f["stages"] if x["kernel"] == "merkle_top"][0]
print(f"{v or '(default: leaves in the row form up to 32 per workgroup)':60s} field {f['ms_per_step']:.3f} ms, merkle_top {top}")
PY
    tail -1 $O/summary.txt
done
export ZK_BUILD_DEFS="-DZK_WG_TRACE=1"
python -m zkstark_amd.build > $O/build_trace.log 2>&1
export ZK_WG_TRACE_FILE=$PWD/$O/wg_21_field.raw
timeout -k 10 200 python tools/wg_trace.py run 21 field > $O/wg_run_21_field.log 2>&1; echo "trace run rc=$?"
python tools/wg_trace.py report $ZK_WG_TRACE_FILE > $O/wg_report_21_field.txt 2>&1
tail -5 $O/wg_report_21_field.txt
unset ZK_BUILD_DEFS ZK_WG_TRACE_FILE
python -m zkstark_amd.build > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; echo "full pytest rc=$?"; tail -3 $O/pytest_full.log
echo done
