#!/bin/bash
# A/B of the one-launch latency phase (merkle_wg_kernel's continuation, ZK_MERKLE_CONTINUATION) in the real pipeline, one session:
# each variant is built on the GPU box and checked for parity; then the field-hash proof, the device-only SHA-256 proof and the
# default proof are timed, and the latency launches of one field-hash proof are listed from a kernel trace.
# Usage: bash tools/ab_continuation.sh OUTDIR
O=${1:-gpurun_out/ab_cont}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "-DZK_MERKLE_CONTINUATION=0" "" "-DZK_MERKLE_CONTINUATION=0" ""; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_fieldhash.py -m gpu -x -q -k "merkle or fieldhash or prover" > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; continue; }
    timeout -k 10 300 python bench.py --hash field --steps 20 --warmup 3 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/field_$tag.json 2> $O/field_$tag.err
    timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/sha_$tag.json 2> $O/sha_$tag.err
    rm -rf $O/prof_$tag
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$tag -- python3 bench.py --hash field --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_$tag.log 2>&1
    python3 - "$v" $O $tag >> $O/summary.txt <<'PY'
import csv, glob, json, os, sys
v, O, tag = sys.argv[1:4]
f = json.load(open(f"{O}/field_{tag}.json")); s = json.load(open(f"{O}/sha_{tag}.json"))
top = lambda b: [(x["launches"], x["ms"]) for x in b["stages"] if x["kernel"] == "merkle_top"][0]
print(f"{v or '(default: continuation)':32s} field {f['ms_per_step']:.3f} ms, merkle_top {top(f)} | sha256 {s['ms_per_step']:.3f} ms, device-only {s['ms_per_step_device_only']:.3f} ms, merkle_top {top(s)}")
t = max(glob.glob(f"{O}/prof_{tag}/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = [r for r in csv.DictReader(open(t)) if "merkle_wg_kernel" in r["Kernel_Name"]]
rows = rows[-23 if "=0" not in v else -38:]
print("    last proof's latency launches (workgroups: us): " + " ".join(f"{int(r['Grid_Size']) // 256}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows))
PY
    tail -2 $O/summary.txt
    find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
