#!/bin/bash
set -o pipefail
O=gpurun_out/r05lz2; mkdir -p $O
timeout -k 10 200 python tools/ab_cp_from_f.py 21 > $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
grep "^cp" $O/ab.txt | cut -c1-150
timeout -k 10 700 python -m pytest tests/test_gpu_shard_native.py tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
