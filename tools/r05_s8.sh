#!/bin/bash
O=gpurun_out/r05k; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_shard_native.py tests/test_gpu_sharded.py tests/test_bench_cli.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo done
