#!/bin/bash
bash tools/r04_quick.sh gpurun_out/r04q5 || exit 1
timeout -k 10 600 python -m pytest tests/test_cabi.py tests/test_gpu_sharded.py tests/test_gpu_shard_native.py -m gpu -x -q 2>&1 | tail -3
ZK_HOST_TIMING=1 timeout -k 10 120 python tools/host_timing.py 21 2>&1 | tail -6
