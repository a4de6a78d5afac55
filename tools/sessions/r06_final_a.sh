#!/bin/bash
# round-6 final GPU session, part A: the lines and profiles that go into profiles/ (all from ONE build): smoke, the default line,
# the field-hash line, rocprofv3 kernel stats of both / of the stage-by-stage API / of configs[1] / of one 2^20 proof, host laps, the
# PMC passes (traffic of the SHA-256 build AND of the field-hash build, VALU), then the default line again so that roofline.traffic
# and fieldhash_2e24.roofline.traffic carry this build's stamp.  Nothing is filtered; every step has its own timeout.
O=gpurun_out/r06z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 600 python bench.py --hash field --steps 20 --warmup 3 > $O/bench_field.json 2> $O/bench_field.err; echo "bench field rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_field -- python3 bench.py --hash field --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_field.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_staged -- python3 bench.py --staged-only > $O/prof_staged.log 2>&1
ZK_HOST_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 20 > $O/prof_cfg2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2e20 -- python3 tools/prof_single.py 17 > $O/prof_2e20.log 2>&1
ZK_HOST_TIMING=1 timeout -k 10 120 python tools/config2_only.py 17 20 > $O/config2_laps.txt 2>&1
ZK_HOST_TIMING=1 timeout -k 10 120 python tools/host_timing.py 21 > $O/proof_laps.txt 2>&1
for r in 50 500 5000 50; do timeout -k 10 120 python tools/config2_only.py 17 $r >> $O/config2_warmup.txt 2>&1; done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_staged -- python3 bench.py --staged-only > $O/pmc_fetch_staged.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_staged -- python3 bench.py --staged-only > $O/pmc_write_staged.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_field -- $B --hash field > $O/pmc_fetch_field.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_field -- $B --hash field > $O/pmc_write_field.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_field -- $B --hash field > $O/pmc_sq_field.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch,$O/pmc_fetch_staged $O/pmc_write,$O/pmc_write_staged profiles/traffic.json "${ZK_COMMIT:-final}" > $O/pmc_traffic.txt && cp profiles/traffic.json $O/traffic.json
python tools/pmc_traffic.py $O/pmc_fetch_field $O/pmc_write_field profiles/traffic_fieldhash.json "${ZK_COMMIT:-final}" > $O/pmc_traffic_field.txt && cp profiles/traffic_fieldhash.json $O/traffic_fieldhash.json
python tools/pmc_valu.py $O/pmc_sq profiles/valu_utilization.json "${ZK_COMMIT:-final}" > /dev/null && cp profiles/valu_utilization.json $O/valu_utilization.json
timeout -k 10 600 python bench.py > $O/bench_stamped.json 2> $O/bench_stamped.err; echo "bench (stamped) rc=$?"
timeout -k 10 200 python tools/batch_inflight.py 21 > $O/batch_inflight.txt 2>&1; echo "batch in flight rc=$?"
for s in "10 3" "14 3" "17 3"; do timeout -k 10 300 python tools/batch_bench.py $s >> $O/batch_sizes.txt 2>&1; done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
echo done A
