#!/bin/bash
O=gpurun_out/r02i; mkdir -p $O
timeout -k 10 600 python tools/shard_soak.py 2 14 200 > $O/shard_soak.txt 2>&1; tail -2 $O/shard_soak.txt
timeout -k 10 600 python tools/shard_soak.py 4 15 100 >> $O/shard_soak.txt 2>&1; tail -1 $O/shard_soak.txt
ZK_HOST_TIMING=1 timeout -k 10 200 python tools/config2_only.py 17 20 > $O/config2_laps.txt 2>&1; tail -5 $O/config2_laps.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -40 $O/pytest.log
echo done
