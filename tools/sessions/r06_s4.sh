#!/bin/bash
# round 6, session 4: A/B of the write-through hand-over of a workgroup's node (ZK_WG_RELAXED_PUBLISH, kernels.hip) in the real
# pipeline: each variant built on the box and parity-checked, then the headline, device-only, configs[1], the 2^20 proof and the
# field hash from short bench runs; then the latency-phase trace of the default (new) variant.
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "-DZK_WG_RELAXED_PUBLISH=0" "" "-DZK_WG_RELAXED_PUBLISH=0" ""; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_fieldhash.py -m gpu -x -q > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; tail -20 $O/pytest_$tag.log; continue; }
    timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --soak-seconds 0 --in-flight 1 --batch-log 0 --no-fieldhash-leg > $O/sha_$tag.json 2> $O/sha_$tag.err
    timeout -k 10 300 python bench.py --hash field --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/field_$tag.json 2> $O/field_$tag.err
    python3 - "$v" $O $tag >> $O/summary.txt <<'PY'
import json, sys
v, O, tag = sys.argv[1:4]
s = json.load(open(f"{O}/sha_{tag}.json")); f = json.load(open(f"{O}/field_{tag}.json"))
top = lambda b: [(x["launches"], round(x["ms"], 4)) for x in b["stages"] if x["kernel"] == "merkle_top"][0]
c1 = s["lde_commit_2e20"]; f20 = s["full_2e20"]
print(f"{v or '(default: write-through hand-over)':36s} sha256 {s['ms_per_step']:.3f} ms (device-only {s['ms_per_step_device_only']:.3f}), merkle_top {top(s)} | configs[1] {c1['us']:.1f} / {c1['us_sustained']:.1f} / root-only {c1['us_root_only']:.1f} us | "
      f"2^20 proof {f20['ms']*1e3:.0f} us (latency launches {f20['latency_launches']['ms']*1e3:.0f} us) | field {f['ms_per_step']:.3f} ms, merkle_top {top(f)}")
PY
    tail -1 $O/summary.txt
done
export ZK_BUILD_DEFS="-DZK_WG_TRACE=1"
python -m zkstark_amd.build > $O/build_trace.log 2>&1
for cfg in "21 sha256" "17 sha256" "21 field"; do
    set -- $cfg
    export ZK_WG_TRACE_FILE=$PWD/$O/wg_$1_$2.raw
    timeout -k 10 200 python tools/wg_trace.py run $1 $2 > $O/wg_run_$1_$2.log 2>&1; echo "trace run $cfg rc=$?"
    python tools/wg_trace.py report $ZK_WG_TRACE_FILE > $O/wg_report_$1_$2.txt 2>&1
    tail -5 $O/wg_report_$1_$2.txt
done
unset ZK_BUILD_DEFS ZK_WG_TRACE_FILE
python -m zkstark_amd.build > /dev/null 2>&1
cat $O/summary.txt
echo done
