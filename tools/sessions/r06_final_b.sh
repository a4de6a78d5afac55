#!/bin/bash
# round-6 final GPU session, part B: the N > 1 path on one GPU -- one rank forced through RCCL / torch's communicator / the peer-copy
# transport, the watchdog rehearsal, staged rehearsals with 2 and 4 ranks (and under torchrun), two ranks SHARING the GPU on the real
# ladder (RCCL refuses, the peer-copy rung takes over), the ranks-as-threads timings (weak and strong shapes) -- then soaks and the
# whole GPU parity suite.  The harness is built by the one recipe (stamped with the library's build); nothing is filtered.
O=gpurun_out/r06z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ZK_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank.json 2> $O/bench_sharded_1rank.err; echo "sharded 1 rank rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_TRANSPORT=torch timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank_torch.json 2> $O/bench_sharded_1rank_torch.err; echo "sharded 1 rank torch rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_TRANSPORT=peer timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank_peer.json 2> $O/bench_sharded_1rank_peer.err; echo "sharded 1 rank peer rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_SIMULATE_NATIVE_FAILURE=hang timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-secondary --log-n 18 > $O/bench_rehearsal_hang.json 2> $O/bench_rehearsal_hang.err; echo "hang rehearsal rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo "rehearsal 2 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 4 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_n4.json 2> $O/bench_rehearsal_n4.err; echo "rehearsal 4 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_torchrun_n2.json 2> $O/bench_rehearsal_torchrun_n2.err; echo "torchrun rehearsal rc=$?"
ZK_BENCH_SHARE_GPU=1 ZK_BENCH_RUNG_BUDGET_S=40,30,60,60 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_peer_n2.json 2> $O/bench_rehearsal_peer_n2.err; echo "peer-copy rung, 2 ranks sharing the GPU rc=$?"
ZK_BENCH_SHARE_GPU=1 ZK_BENCH_TRANSPORT=peer timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_peer_n4.json 2> $O/bench_rehearsal_peer_n4.err; echo "peer-copy rung, 4 ranks sharing the GPU rc=$?"
STC=$(bash tools/build_shard_threads_check.sh /tmp/shard_threads_check) || { echo "harness build failed"; exit 1; }
for rep in 1 2 3; do for w in "8 24" "4 23" "2 22"; do timeout -k 10 300 $STC $w 3 0 0 0 3 >> $O/shard_threads_timing.txt 2>&1; done; done
for w in 8 4 2; do for ml in 0 20 23; do echo "# strong shape: a 2^24 proof over $w ranks (threads of one process, one GPU), min_layer_log=$ml (0 = the default: 20 from 4 ranks on, else 21)" >> $O/shard_threads_strong.txt; timeout -k 10 120 $STC $w 21 3 $ml 0 0 5 >> $O/shard_threads_strong.txt 2>&1; done; done
timeout -k 10 200 python tools/soak.py 40 > $O/soak.txt 2>&1; echo "soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 300 python tools/shard_soak.py 2 14 200 >> $O/soak.txt 2>&1; echo "shard soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 300 python tools/shard_rccl_soak.py 16 300 >> $O/soak.txt 2>&1; echo "rccl soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo done B
