#!/bin/bash
O=gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -12 $O/pytest.log
