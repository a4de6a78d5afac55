#!/bin/bash
# round-3 GPU session 1 (ran at commit f4c0d91, when the subtree kernel still had its heap variant: ZK_MERKLE_HEAP and
# ZK_MERKLE_WG_WAVES no longer exist, see profiles/r03_ab_subtree_heap.txt): probes, parity of the new subtree kernel,
# A/B of its switches, configs[4] (field hash) on the record
O=gpurun_out/r03a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 ./tools/montmul_probe > $O/montmul_probe.txt 2>&1; echo "montmul rc=$?"; head -12 $O/montmul_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python tools/ab_env.py $O/ab_merkle.txt "lds_k4(r02):ZK_MERKLE_HEAP=0" "heap_k4:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=4" \
    "heap_k5:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=5" "heap_k6:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=6" "heap_k7:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=7" \
    "heap_k4_wg1:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=4,ZK_MERKLE_WG_WAVES=1" "heap_k4_wg2:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=4,ZK_MERKLE_WG_WAVES=2" \
    "lds_k4_wg1:ZK_MERKLE_HEAP=0,ZK_MERKLE_WG_WAVES=1" "heap_k5_lat18:ZK_MERKLE_HEAP=1,ZK_MERKLE_MAX_K=5,ZK_MERKLE_LATENCY_LOG=18" 2>&1 | tail -12
timeout -k 10 600 python bench.py --hash field --steps 10 --warmup 2 > $O/bench_field.json 2> $O/bench_field.err; echo "bench field rc=$?"
B="python3 bench.py --hash field --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_field -- python3 bench.py --hash field --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_field.log 2>&1; echo "prof field rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_field -- $B > $O/pmc_sq_field.log 2>&1; echo "pmc sq field rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_field -- $B > $O/pmc_fetch_field.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_field -- $B > $O/pmc_write_field.log 2>&1
S="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_sha -- $S > $O/pmc_sq_sha.log 2>&1; echo "pmc sq sha rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_stall_sha -- $S > $O/pmc_stall_sha.log 2>&1; echo "pmc stall sha rc=$?"
timeout -k 10 120 rocprofv3 --list-avail > $O/avail.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
echo done
