#!/bin/bash
O=gpurun_out/r05e; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 400 python bench.py --hash field --steps 20 --warmup 3 --no-secondary > $O/bench_field.json 2> $O/bench_field.err; echo "bench field rc=$?"
timeout -k 10 400 python bench.py --steps 50 --no-secondary > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for n in ("bench_field", "bench"):
    b = json.load(open(f"{O}/{n}.json"))
    print(n, round(b["ms_per_step"], 3), "ms parity", b["parity_checked"], "device_only", b.get("ms_per_step_device_only"), "frac", round(b["roofline"]["frac"], 3))
    for s in b["stages"]: print("  ", s["kernel"], s["launches"], s["ms"])
PY
echo done
