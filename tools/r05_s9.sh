#!/bin/bash
O=gpurun_out/r05p; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o tools/shard_threads_check || exit 1
for rep in 1 2 3; do for a in 0 1; do for w in "8 24" "4 23" "2 22"; do ZK_HARNESS_ASYNC=$a timeout -k 5 90 ./tools/shard_threads_check $w 3 0 0 0 3 2>&1 | grep -E "timing|rank" >> $O/weak.txt; done; done; done
for rep in 1 2; do for a in 0 1; do for w in 8 4 2; do ZK_HARNESS_ASYNC=$a timeout -k 5 60 ./tools/shard_threads_check $w 21 3 0 0 0 5 2>&1 | grep -E "timing|rank" >> $O/strong.txt; done; done; done
cut -c1-60,105-200 $O/weak.txt; echo; cut -c1-60,105-200 $O/strong.txt
echo done
