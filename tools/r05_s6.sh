#!/bin/bash
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "benchmark_domain or batch" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 -c "
import json; b = json.load(open('$O/bench.json'))
print(round(b['ms_per_step'], 3), b['ms_per_step_device_only'], b['parity_checked'])
print(json.dumps(b['batched_2e24'], indent=1))
print(b['lde_commit_2e20'])"
echo done
