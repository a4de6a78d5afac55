#!/bin/bash
O=gpurun_out/r06m; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "prove_many or early" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
