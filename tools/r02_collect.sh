#!/bin/bash
# Copies the artefacts of tools/r02_final.sh (merged back under gpurun_out/r02h) into profiles/ and derives
# traffic.json / valu_utilization.json from the PMC passes.  Run in the build container after the GPU call.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r02h; P=profiles   # (clear $O before tools/r02_final.sh: gpurun merges new files next to old ones)
cp $O/valu_microbench.txt $P/r02_valu_microbench.txt; cp $O/sha_latency_probe.txt $P/r02_sha_latency_probe.txt; cp $O/valu_mix_probe.txt $P/r02_valu_mix_probe.txt
cp $O/bench.json $P/r02_bench_2e24.json; cp $O/bench_sharded_1rank.json $P/r02_bench_sharded_1rank.json
cp $O/bench_rehearsal_n2.json $P/r02_bench_rehearsal_n2.json; cp $O/bench_rehearsal_n4.json $P/r02_bench_rehearsal_n4.json
grep -v "amdgpu.ids" $O/batch_sizes.txt > $P/r02_batch_sizes.txt
cp $(find $O/prof_bench -name "*kernel_stats.csv") $P/r02_bench_2e24_kernel_stats.csv
cp $(find $O/prof_staged -name "*kernel_stats.csv") $P/r02_staged_2e24_kernel_stats.csv
cp $(find $O/prof_cfg2 -name "*kernel_stats.csv") $P/r02_config2_2e20_kernel_stats.csv
mkdir -p $P/r02_pmc
cp $(find $O/pmc_fetch -name "*counter_collection.csv") $P/r02_pmc/fetch_size_counter_collection.csv
cp $(find $O/pmc_write -name "*counter_collection.csv") $P/r02_pmc/write_size_counter_collection.csv
cp $(find $O/pmc_fetch_staged -name "*counter_collection.csv") $P/r02_pmc/fetch_size_staged_counter_collection.csv
cp $(find $O/pmc_write_staged -name "*counter_collection.csv") $P/r02_pmc/write_size_staged_counter_collection.csv
cp $(find $O/pmc_sq -name "*counter_collection.csv") $P/r02_pmc/sq_counter_collection.csv
cp $(find $O/pmc_sq_staged -name "*counter_collection.csv") $P/r02_pmc/sq_staged_counter_collection.csv
python tools/pmc_traffic.py $O/pmc_fetch,$O/pmc_fetch_staged $O/pmc_write,$O/pmc_write_staged $P/traffic.json "$(git rev-parse --short HEAD)" > /dev/null
python tools/pmc_valu.py $O/pmc_sq $P/valu_utilization.json > /dev/null
python3 -c "
import json
d = json.load(open('$P/traffic.json')); print('traffic.json:', d['commit'], d['build_hash'], round(d['merkle_leaf_bytes_per_launch'] / 1e6, 1), 'MB per leaf launch')
b = json.load(open('$P/r02_bench_2e24.json')); print('bench:', round(b['ms_per_step'], 3), 'ms per proof, parity_checked', b['parity_checked'], 'build', b['build_hash'])"
