#!/bin/bash
# Copies the artefacts of tools/r04_final.sh (merged back under gpurun_out/r04z) into profiles/.  Run in the build container after the GPU call.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r04z; P=profiles
cp $O/bench_stamped.json $P/r04_bench_2e24.json; cp $O/bench_field.json $P/r04_bench_2e24_fieldhash.json
cp $O/bench_sharded_1rank.json $P/r04_bench_sharded_1rank.json; cp $O/bench_sharded_1rank_torch.json $P/r04_bench_sharded_1rank_torch_transport.json
cp $O/bench_rehearsal_n2.json $P/r04_bench_rehearsal_n2.json; cp $O/bench_rehearsal_n4.json $P/r04_bench_rehearsal_n4.json
cp $O/bench_rehearsal_torchrun_n2.json $P/r04_bench_rehearsal_torchrun_n2.json
cp $O/bench_rehearsal_hang.json $P/r04_bench_rehearsal_hang.json; grep -E "^\[bench\]" $O/bench_rehearsal_hang.err > $P/r04_bench_rehearsal_hang.log || true
cp $O/bench_rehearsal_idfail.json $P/r04_bench_rehearsal_idfail.json; grep -E "^\[bench\]" $O/bench_rehearsal_idfail.err > $P/r04_bench_rehearsal_idfail.log || true
grep -v "amdgpu.ids" $O/batch_sizes.txt > $P/r04_batch_sizes.txt
grep -v "amdgpu.ids" $O/config2_laps.txt > $P/r04_config2_laps.txt
grep "zk timing" $O/proof_laps.txt | tail -12 > $P/r04_proof_laps.txt
grep "LDE + Merkle commit" $O/config2_warmup.txt > $P/r04_config2_warmup_final.txt
cp $O/shard_threads_timing.txt $P/r04_shard_threads_timing.txt
grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/soak.txt > $P/r04_soak.txt
# gpurun MERGES a session's files into the local directory, which may still hold those of an earlier session (same names
# apart from the process id): of the files matching, take the biggest one among those written within 3 minutes of the newest
# (a run may leave one file per process: the benchmark's is the large one)
biggest() { python3 - "$1" "$2" <<'PY'
import fnmatch, os, sys
files = [os.path.join(d, f) for d, _, fs in os.walk(sys.argv[1]) for f in fs if fnmatch.fnmatch(f, sys.argv[2])]
newest = max(os.path.getmtime(f) for f in files)
print(max((f for f in files if os.path.getmtime(f) >= newest - 180), key=os.path.getsize))
PY
}
cp $(biggest $O/prof_bench "*kernel_stats.csv") $P/r04_bench_2e24_kernel_stats.csv
cp $(biggest $O/prof_field "*kernel_stats.csv") $P/r04_bench_2e24_fieldhash_kernel_stats.csv
cp $(biggest $O/prof_staged "*kernel_stats.csv") $P/r04_staged_2e24_kernel_stats.csv
cp $(biggest $O/prof_cfg2 "*kernel_stats.csv") $P/r04_config2_2e20_kernel_stats.csv
mkdir -p $P/r04_pmc
for n in fetch write fetch_staged write_staged sq; do
    cp $(biggest $O/pmc_$n "*counter_collection.csv") $P/r04_pmc/${n}_counter_collection.csv
done
cp $O/traffic.json $P/traffic.json      # made on the GPU box from the same PMC passes (tools/r04_final.sh), stamped there
cp $O/valu_utilization.json $P/valu_utilization.json
python3 -c "
import json
d = json.load(open('$P/traffic.json')); print('traffic.json:', d['commit'], d['build_hash'], round(d['merkle_leaf_bytes_per_launch'] / 1e6, 1), 'MB per leaf launch')
b = json.load(open('$P/r04_bench_2e24.json')); print('bench:', round(b['ms_per_step'], 3), 'ms per proof, parity_checked', b['parity_checked'], 'build', b['build_hash'])
f = json.load(open('$P/r04_bench_2e24_fieldhash.json')); print('field:', round(f['ms_per_step'], 2), 'ms per proof, parity_checked', f['parity_checked'], 'build', f['build_hash'])"
