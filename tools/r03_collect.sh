#!/bin/bash
# Copies the artefacts of tools/r03_final.sh (merged back under gpurun_out/r03f) into profiles/ and derives
# traffic.json / valu_utilization.json from the PMC passes.  Run in the build container after the GPU call.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03f; P=profiles
cp $O/bench_stamped.json $P/r03_bench_2e24.json; cp $O/bench_field.json $P/r03_bench_2e24_fieldhash.json
cp $O/sha_latency_probe.txt $P/r03_sha_latency_probe.txt; cp $O/montmul_probe.txt $P/r03_montmul_probe.txt
cp $O/sha_quad_probe.txt $P/r03_sha_quad_probe.txt; cp $O/host_sha.txt $P/r03_host_sha.txt; cp $O/levels_report.txt $P/r03_levels_quad.txt
cp $O/bench_sharded_1rank.json $P/r03_bench_sharded_1rank.json; cp $O/bench_sharded_1rank_torch.json $P/r03_bench_sharded_1rank_torch_transport.json
cp $O/bench_rehearsal_n2.json $P/r03_bench_rehearsal_n2.json; cp $O/bench_rehearsal_n4.json $P/r03_bench_rehearsal_n4.json
cp $O/bench_rehearsal_n2_strong.json $P/r03_bench_rehearsal_n2_strong.json
grep -v "amdgpu.ids" $O/batch_sizes.txt > $P/r03_batch_sizes.txt
grep -v "amdgpu.ids" $O/config2_laps.txt > $P/r03_config2_laps.txt; grep -v "amdgpu.ids" $O/config2_switches.txt > $P/r03_config2_switches.txt
biggest() { ls -S $(find $1 -name "$2") | head -1; }      # a run may leave one file per process: the benchmark's is the large one
cp $(biggest $O/prof_bench "*kernel_stats.csv") $P/r03_bench_2e24_kernel_stats.csv
cp $(biggest $O/prof_field "*kernel_stats.csv") $P/r03_bench_2e24_fieldhash_kernel_stats.csv
cp $(biggest $O/prof_staged "*kernel_stats.csv") $P/r03_staged_2e24_kernel_stats.csv
cp $(biggest $O/prof_cfg2 "*kernel_stats.csv") $P/r03_config2_2e20_kernel_stats.csv
mkdir -p $P/r03_pmc
for n in fetch write fetch_staged write_staged sq sq_staged stall fetch_field write_field sq_field; do
    cp $(biggest $O/pmc_$n "*counter_collection.csv") $P/r03_pmc/${n}_counter_collection.csv
done
cp $O/traffic.json $P/traffic.json      # made on the GPU box from the same PMC passes (tools/r03_final.sh), stamped there
python tools/pmc_traffic.py $O/pmc_fetch_field $O/pmc_write_field $P/r03_traffic_fieldhash.json "$(git rev-parse --short HEAD)" > /dev/null
python tools/pmc_valu.py $O/pmc_sq $P/valu_utilization.json > /dev/null
python3 -c "
import json
d = json.load(open('$P/traffic.json')); print('traffic.json:', d['commit'], d['build_hash'], round(d['merkle_leaf_bytes_per_launch'] / 1e6, 1), 'MB per leaf launch')
b = json.load(open('$P/r03_bench_2e24.json')); print('bench:', round(b['ms_per_step'], 3), 'ms per proof, parity_checked', b['parity_checked'], 'build', b['build_hash'])
f = json.load(open('$P/r03_bench_2e24_fieldhash.json')); print('field:', round(f['ms_per_step'], 2), 'ms per proof, parity_checked', f['parity_checked'], 'build', f['build_hash'])"
