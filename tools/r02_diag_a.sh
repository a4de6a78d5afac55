#!/bin/bash
# round-2 GPU session A: full GPU test suite, then the diagnostics the perf items need
O=gpurun_out/r02a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 120 ./tools/valu_microbench > $O/valu_microbench.txt 2>&1; tail -3 $O/valu_microbench.txt
ZK_HOST_TIMING=1 timeout -k 10 200 python tools/host_timing.py 21 > $O/laps_2e24.txt 2>&1
timeout -k 10 200 python tools/config2_lde_commit.py > $O/config2.txt 2>&1; cat $O/config2.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 50 > $O/prof_cfg2.log 2>&1
ZK_HOST_TIMING=1 timeout -k 10 300 python tools/batch_laps.py > $O/batch_laps.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2e24 -- python3 tools/host_timing.py 21 > $O/prof_2e24.log 2>&1
echo done
