#!/bin/bash
set -o pipefail
O=gpurun_out/r05cp4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_shard_native.py tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 200 python tools/ab_cp_from_f.py 21 > $O/ab_2e24.txt 2>&1 || { tail $O/ab_2e24.txt; exit 1; }
grep "^cp" $O/ab_2e24.txt
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/stc || exit 1
for cfg in "4 23 3 0 0 0 3" "8 24 3 0 0 0 3" "4 21 3 0 0 0 5"; do
  for mode in new old new old new old; do
    if [ $mode = old ]; then export ZK_HARNESS_EXCHANGE_CP=1; else unset ZK_HARNESS_EXCHANGE_CP; fi
    echo "== $cfg cp: $mode" >> $O/threads.txt
    timeout -k 5 90 /tmp/stc $cfg 2>&1 | grep -E "timing|rank" >> $O/threads.txt || { echo FAILED >> $O/threads.txt; tail -5 $O/threads.txt; exit 1; }
  done
done
unset ZK_HARNESS_EXCHANGE_CP
cat $O/threads.txt
