#!/bin/bash
# round 6, session 10: the peer-copy transport rebuilt around one staging buffer per rank: big sizes with ranks sharing the GPU, its
# tests, and the two bench rehearsals that did not finish before
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
sed -n '/^cat > \/tmp\/peer_big.py/,/^PY$/p' tools/sessions/r06_s8.sh | sed '1d;$d' > /tmp/peer_big.py
for cfg in "2 20" "2 22" "4 23"; do timeout -k 10 200 python /tmp/peer_big.py $cfg > $O/peer_$(echo $cfg | tr ' ' '_').txt 2>&1; echo "$cfg rc=$?"; grep -v "amdgpu.ids" $O/peer_$(echo $cfg | tr ' ' '_').txt | tail -3 | cut -c1-300; done
timeout -k 10 600 python -m pytest tests/test_gpu_shard_native.py -m gpu -x -q -k "peer or plain_c" > $O/pytest_peer.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_peer.log
ZK_BENCH_SHARE_GPU=1 ZK_BENCH_RUNG_BUDGET_S=40,30,60,60 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_peer_n2.json 2> $O/bench_rehearsal_peer_n2.err; echo "peer-copy rung, 2 ranks sharing the GPU rc=$?"
ZK_BENCH_SHARE_GPU=1 ZK_BENCH_TRANSPORT=peer timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --log-n 21 > $O/bench_rehearsal_peer_n4.json 2> $O/bench_rehearsal_peer_n4.err; echo "peer-copy rung, 4 ranks sharing the GPU rc=$?"
grep "\[bench\]" $O/bench_rehearsal_peer_n2.err | cut -c1-250 | tail; grep "\[bench\]" $O/bench_rehearsal_peer_n4.err | cut -c1-250 | tail -5
timeout -k 10 300 python -m pytest tests/test_bench_cli.py -m gpu -x -q -k "peer or transports_one_rank" > $O/pytest_bench.log 2>&1; echo "pytest bench rc=$?"; tail -3 $O/pytest_bench.log
