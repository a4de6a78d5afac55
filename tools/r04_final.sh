#!/bin/bash
# round-4 final GPU session: the artefacts that go into profiles/ (all from ONE build), then the whole parity suite.
O=gpurun_out/r04z; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --hash field --steps 20 --warmup 3 > $O/bench_field.json 2> $O/bench_field.err; echo "bench field rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_field -- python3 bench.py --hash field --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_field.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_staged -- python3 bench.py --staged-only > $O/prof_staged.log 2>&1
ZK_HOST_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 20 > $O/prof_cfg2.log 2>&1
ZK_HOST_TIMING=1 timeout -k 10 120 python tools/config2_only.py 17 20 > $O/config2_laps.txt 2>&1
ZK_HOST_TIMING=1 timeout -k 10 120 python tools/host_timing.py 21 > $O/proof_laps.txt 2>&1
for r in 50 500 5000 50; do timeout -k 10 120 python tools/config2_only.py 17 $r >> $O/config2_warmup.txt 2>&1; done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_staged -- python3 bench.py --staged-only > $O/pmc_fetch_staged.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_staged -- python3 bench.py --staged-only > $O/pmc_write_staged.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
# traffic.json from THIS build's PMC passes, then the default line again so that roofline.traffic carries a matching stamp
python tools/pmc_traffic.py $O/pmc_fetch,$O/pmc_fetch_staged $O/pmc_write,$O/pmc_write_staged profiles/traffic.json "${ZK_COMMIT:-final}" > /dev/null && cp profiles/traffic.json $O/traffic.json
python tools/pmc_valu.py $O/pmc_sq profiles/valu_utilization.json "${ZK_COMMIT:-final}" > /dev/null && cp profiles/valu_utilization.json $O/valu_utilization.json
timeout -k 10 600 python bench.py > $O/bench_stamped.json 2> $O/bench_stamped.err; echo "bench (stamped) rc=$?"
ZK_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank.json 2> $O/bench_sharded_1rank.err; echo "sharded 1 rank rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_TRANSPORT=torch timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank_torch.json 2> $O/bench_sharded_1rank_torch.err; echo "sharded 1 rank torch rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_SIMULATE_NATIVE_FAILURE=hang timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-secondary --log-n 18 > $O/bench_rehearsal_hang.json 2> $O/bench_rehearsal_hang.err; echo "hang rehearsal rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_SIMULATE_NATIVE_FAILURE=id timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-secondary --log-n 18 > $O/bench_rehearsal_idfail.json 2> $O/bench_rehearsal_idfail.err; echo "id failure rehearsal rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 20 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo "rehearsal 2 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 4 --steps 3 --warmup 1 --log-n 19 > $O/bench_rehearsal_n4.json 2> $O/bench_rehearsal_n4.err; echo "rehearsal 4 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 18 > $O/bench_rehearsal_torchrun_n2.json 2> $O/bench_rehearsal_torchrun_n2.err; echo "torchrun rehearsal rc=$?"
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o tools/shard_threads_check
for w in "8 24" "4 23" "2 22"; do timeout -k 10 300 ./tools/shard_threads_check $w 3 0 0 0 3 2>&1 | grep -E "timing|threads ok" >> $O/shard_threads_timing.txt; done
for s in "10 3" "14 3" "17 3"; do timeout -k 10 300 python tools/batch_bench.py $s >> $O/batch_sizes.txt 2>&1; done
timeout -k 10 200 python tools/soak.py 40 > $O/soak.txt 2>&1; echo "soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 300 python tools/shard_soak.py 2 14 200 >> $O/soak.txt 2>&1; echo "shard soak rc=$?" | tee -a $O/soak.txt
timeout -k 10 300 python tools/shard_rccl_soak.py 16 300 >> $O/soak.txt 2>&1; echo "rccl soak rc=$?" | tee -a $O/soak.txt
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
timeout -k 10 1150 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo done
