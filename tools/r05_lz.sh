#!/bin/bash
set -o pipefail
O=gpurun_out/r05lz; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_shard_native.py tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/ab_cp_from_f.py 21 > $O/ab.txt 2>&1 || { tail $O/ab.txt; exit 1; }
grep "^cp" $O/ab.txt | cut -c1-150
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/stc || exit 1
for cfg in "8 24 3 0 0 0 3" "8 21 3 0 0 0 5" "4 21 3 0 0 0 5" "8 21 3 0 0 0 5" "4 21 3 0 0 0 5"; do timeout -k 5 90 /tmp/stc $cfg 2>&1 | grep -E "timing|rank"; done
