#!/bin/bash
# round 6, session 11: early launch (zk_ctx_set_early_launch): parity, the failure path, and the A/B inside one build
O=gpurun_out/r06l; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "early_launch or prover or config" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python tools/ab_early_launch.py 21 17 10 > $O/ab_early_launch.txt 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/ab_early_launch.txt
