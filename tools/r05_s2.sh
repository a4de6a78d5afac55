#!/bin/bash
# round-5 session 2: the sharded prover's one-launch decommitment and the restructured bench line
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout -k 10 900 python -m pytest tests/test_gpu_shard_native.py tests/test_gpu_sharded.py tests/test_bench_cli.py tests/test_cabi.py -m gpu -x -q > $O/pytest_shard.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_shard.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
ZK_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank.json 2> $O/bench_sharded_1rank.err; echo "sharded 1 rank rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo "rehearsal 2 rc=$?"
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
b = json.load(open(f"{O}/bench.json"))
print("bench", round(b["ms_per_step"], 3), "ms; device_only", b.get("ms_per_step_device_only"), "; batched", b.get("batched_2e24"), "; cfg2", {k: round(v, 1) for k, v in b["lde_commit_2e20"].items() if k.startswith("us")}, "parity", b["parity_checked"], "frac", round(b["roofline"]["frac"], 3))
print([s for s in b["stages"] if s["kernel"] == "ntt"])
for name in ("bench_sharded_1rank", "bench_rehearsal_n2"):
    try:
        r = json.load(open(f"{O}/{name}.json"))
        print(name, round(r["ms_per_step"], 3), "ms parity", r["parity_checked"], "decommit_ms", r["shard"].get("decommit_ms"), "tail", r["shard"]["tail_ms"], [ (k, round(v["ms"], 3), v["parity"], v.get("speedup_over_single_gpu")) for k, v in r.items() if k.startswith("strong_")], r.get("legs_skipped"))
    except Exception as e:
        print(name, "unreadable:", e)
PY
echo done
