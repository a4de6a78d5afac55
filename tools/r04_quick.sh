#!/bin/bash
# quick check of a kernel change: the kernel / prover parity tests, a short bench line, configs[1] under rocprofv3
# Usage: bash tools/r04_quick.sh OUTDIR
O=${1:-gpurun_out/r04q}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_fieldhash.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --soak-seconds 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 20 > $O/prof_cfg2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_bench.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
python3 - $O <<'PY'
import json, sys, glob, csv
O = sys.argv[1]
b = json.load(open(O + "/bench.json"))
st = {x["kernel"]: x for x in b["stages"]}
print("proof", round(b["ms_per_step"], 3), "ms  device_only", round(b.get("device_only", {}).get("ms_per_step", 0), 3), " pipelined", round(b.get("pipelined", {}).get("ms_per_proof", 0), 3),
      " config2", round(b.get("lde_commit_2e20", {}).get("us", 0), 1), "us  parity", b.get("parity_checked"), " frac", b["roofline"]["frac"])
print("stages:", {k: round(v["ms"], 3) for k, v in st.items()})
for d in ("prof_cfg2", "prof_bench"):
    for f in glob.glob(f"{O}/{d}/**/*kernel_stats.csv", recursive=True):
        print(d)
        for r in csv.DictReader(open(f)):
            print("  %-90s calls %5s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
echo done
