#!/bin/bash
O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 ./tools/fh64_probe > $O/fh64_probe.txt 2>&1; echo "probe rc=$?"; grep -E "equality|VGPR|chain" $O/fh64_probe.txt
timeout -k 10 600 python -m pytest tests/test_fieldhash.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest_f.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_f.log
timeout -k 10 400 python bench.py --hash field --steps 20 --warmup 3 > $O/bench_field.json 2> $O/bench_field.err; echo "bench field rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_field -- python3 bench.py --hash field --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_field.log 2>&1
python3 - $O <<'PY'
import json, sys, glob, os
O = sys.argv[1]
b = json.load(open(f"{O}/bench_field.json"))
print("field", round(b["ms_per_step"], 3), "ms parity", b["parity_checked"], "frac", round(b["roofline"]["frac"], 3), "chain", b["roofline"]["valu"]["chain_ns_per_instr"], "hashing", b["roofline"].get("hashing"))
for s in b["stages"]: print("  ", s["kernel"], s["launches"], s["ms"])
print("pipelined", b.get("pipelined"))
f = max(glob.glob(f"{O}/prof_field/**/*kernel_stats.csv", recursive=True), key=os.path.getsize)
print(open(f).read()[:3000])
PY
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
echo done
