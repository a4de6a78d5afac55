#!/bin/bash
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 ./tools/fh64_probe > $O/fh64_probe.txt 2>&1; echo "probe rc=$?"; cat $O/fh64_probe.txt
timeout -k 10 300 python tools/batch_inflight.py 21 > $O/batch_inflight.txt 2>&1; echo "batch rc=$?"; grep threads $O/batch_inflight.txt
timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pending or merkle or config2" > $O/pytest_k.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_k.log
for r in 50 500 50; do timeout -k 10 120 python tools/config2_only.py 17 $r 2>&1 | grep "LDE + Merkle"; done
echo done
