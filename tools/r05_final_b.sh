#!/bin/bash
# round-5 final GPU session, part B: the N > 1 path on one GPU (forced one-rank RCCL, staged rehearsals with the strong leg, watchdog
# and failure rehearsals), the ranks-as-threads timings (weak and strong shapes), soaks, then the whole GPU parity suite.
O=gpurun_out/r05z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ZK_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank.json 2> $O/bench_sharded_1rank.err; echo "sharded 1 rank rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_TRANSPORT=torch timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank_torch.json 2> $O/bench_sharded_1rank_torch.err; echo "sharded 1 rank torch rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_SIMULATE_NATIVE_FAILURE=hang timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-secondary --log-n 18 > $O/bench_rehearsal_hang.json 2> $O/bench_rehearsal_hang.err; echo "hang rehearsal rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo "rehearsal 2 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 4 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_n4.json 2> $O/bench_rehearsal_n4.err; echo "rehearsal 4 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_torchrun_n2.json 2> $O/bench_rehearsal_torchrun_n2.err; echo "torchrun rehearsal rc=$?"
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o tools/shard_threads_check
for rep in 1 2 3; do for w in "8 24" "4 23" "2 22"; do timeout -k 10 300 ./tools/shard_threads_check $w 3 0 0 0 3 2>&1 | grep -E "timing|threads ok" >> $O/shard_threads_timing.txt; done; done
for w in 8 4 2; do for ml in 0 20 23; do echo "# strong shape: a 2^24 proof over $w ranks (threads of one process, one GPU), min_layer_log=$ml (0 = the default: 20 from 4 ranks on, else 21)" >> $O/shard_threads_strong.txt; timeout -k 10 120 ./tools/shard_threads_check $w 21 3 $ml 0 0 5 2>&1 | grep -E "timing|threads ok|rank" >> $O/shard_threads_strong.txt; done; done
timeout -k 10 200 python tools/ab_cp_from_f.py 21 > $O/ab_cp_from_f.txt 2>&1; echo "ab cp_from_f rc=$?"
for rep in 1 2; do for m in new old; do if [ $m = old ]; then export ZK_HARNESS_EXCHANGE_CP=1; else unset ZK_HARNESS_EXCHANGE_CP; fi; for w in "8 24" "4 23" "2 22"; do echo "== world/log_n $w, cp: $m" >> $O/ab_cp_threads.txt; timeout -k 10 300 ./tools/shard_threads_check $w 3 0 0 0 3 2>&1 | grep -E "timing" >> $O/ab_cp_threads.txt; done; done; done; unset ZK_HARNESS_EXCHANGE_CP
timeout -k 10 200 python tools/soak.py 40 > $O/soak.txt 2>&1; echo "soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 300 python tools/shard_soak.py 2 14 200 >> $O/soak.txt 2>&1; echo "shard soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 300 python tools/shard_rccl_soak.py 16 300 >> $O/soak.txt 2>&1; echo "rccl soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo done B
