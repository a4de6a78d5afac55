#!/bin/bash
O=gpurun_out/r05j; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o tools/shard_threads_check || exit 1
for w in "8 24" "4 23" "2 22"; do for ml in 0 21 20; do echo "# weak shape: world/log_n $w, min_layer_log=$ml" >> $O/threads_weak_ml.txt; timeout -k 5 60 ./tools/shard_threads_check $w 3 $ml 0 0 3 2>&1 | grep -E "timing|threads ok|rank" >> $O/threads_weak_ml.txt; done; done
for w in 8 4 2; do for ml in 0 21 20; do echo "# strong shape: world $w, min_layer_log=$ml" >> $O/threads_strong_ml.txt; timeout -k 5 40 ./tools/shard_threads_check $w 21 3 $ml 0 0 5 2>&1 | grep -E "timing|threads ok|rank" >> $O/threads_strong_ml.txt; done; done
cat $O/threads_weak_ml.txt $O/threads_strong_ml.txt | grep -E "^#|timing"
echo done
