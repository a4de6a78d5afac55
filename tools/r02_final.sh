#!/bin/bash
# round-2 GPU session D: the artefacts that go into profiles/ (kernel stats, PMC passes, bench lines, probes)
O=gpurun_out/r02h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0"
timeout -k 10 120 ./tools/valu_microbench > $O/valu_microbench.txt 2>&1; tail -4 $O/valu_microbench.txt
timeout -k 10 120 ./tools/sha_latency_probe > $O/sha_latency_probe.txt 2>&1; timeout -k 10 120 ./tools/valu_mix_probe > $O/valu_mix_probe.txt 2>&1; timeout -k 10 120 ./tools/valu_mix_probe2 >> $O/valu_mix_probe.txt 2>&1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 > $O/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_staged -- python3 bench.py --staged-only > $O/prof_staged.log 2>&1
ZK_HOST_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 20 > $O/prof_cfg2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_staged -- python3 bench.py --staged-only > $O/pmc_fetch_staged.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_staged -- python3 bench.py --staged-only > $O/pmc_write_staged.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_staged -- python3 bench.py --staged-only > $O/pmc_sq_staged.log 2>&1
for s in "10 3" "14 3" "17 3"; do timeout -k 10 300 python tools/batch_bench.py $s >> $O/batch_sizes.txt 2>&1; done
ZK_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 20 --no-secondary > $O/bench_sharded_1rank.json 2> $O/bench_sharded_1rank.err; echo "sharded 1 rank rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 20 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo "rehearsal 2 rc=$?"
ZK_BENCH_STAGED=1 timeout -k 10 300 python bench.py --gpus 4 --steps 3 --warmup 1 --log-n 19 > $O/bench_rehearsal_n4.json 2> $O/bench_rehearsal_n4.err; echo "rehearsal 4 rc=$?"
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
echo done
