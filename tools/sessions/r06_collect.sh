#!/bin/bash
# Copies the artefacts of tools/sessions/r06_final_a.sh / r06_final_b.sh (merged back under gpurun_out/r06z) into profiles/.  Run in the
# build container after the GPU calls.
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out/r06z; P=profiles
cp $O/bench_stamped.json $P/r06_bench_2e24.json; cp $O/bench_field.json $P/r06_bench_2e24_fieldhash.json
grep -v "amdgpu.ids" $O/batch_sizes.txt > $P/r06_batch_sizes.txt
grep "threads" $O/batch_inflight.txt > $P/r06_batch_inflight_2e24.txt
grep -v "amdgpu.ids" $O/config2_laps.txt > $P/r06_config2_laps.txt
grep "zk timing" $O/proof_laps.txt | tail -12 > $P/r06_proof_laps.txt
grep "LDE + Merkle commit" $O/config2_warmup.txt > $P/r06_config2_warmup.txt
if [ -f $O/bench_sharded_1rank.json ]; then
    cp $O/bench_sharded_1rank.json $P/r06_bench_sharded_1rank.json; cp $O/bench_sharded_1rank_torch.json $P/r06_bench_sharded_1rank_torch_transport.json
    cp $O/bench_sharded_1rank_peer.json $P/r06_bench_sharded_1rank_peer_transport.json
    cp $O/bench_rehearsal_n2.json $P/r06_bench_rehearsal_n2.json; cp $O/bench_rehearsal_n4.json $P/r06_bench_rehearsal_n4.json
    cp $O/bench_rehearsal_torchrun_n2.json $P/r06_bench_rehearsal_torchrun_n2.json
    cp $O/bench_rehearsal_peer_n2.json $P/r06_bench_rehearsal_peer_n2.json; grep -E "^\[bench\]|\[zk_shard\]" $O/bench_rehearsal_peer_n2.err > $P/r06_bench_rehearsal_peer_n2.log || true
    cp $O/bench_rehearsal_peer_n4.json $P/r06_bench_rehearsal_peer_n4.json
    cp $O/bench_rehearsal_hang.json $P/r06_bench_rehearsal_hang.json; grep -E "^\[bench\]" $O/bench_rehearsal_hang.err > $P/r06_bench_rehearsal_hang.log || true
    { echo "# tests/shard_threads_check.c <world> <log_n> 3 0 0 0 3, three repetitions: the native sharded prover with the ranks as THREADS of one process on ONE MI355X"
      echo "# (device-to-device transport standing in for xGMI; the GPU is shared, so a figure is the device work of all ranks together).  Weak-scaling sizes, 2^24 elements per rank."
      echo "# Harness built by tools/build_shard_threads_check.sh (stamped with the library build; every wait bounded at 20 s); stdout and stderr unfiltered."
      cat $O/shard_threads_timing.txt; } > $P/r06_shard_threads_timing.txt
    { echo "# the STRONG shape on the same harness: one 2^24 proof over 8 / 4 / 2 ranks (threads of one process sharing one GPU), min_layer_log swept."
      echo "# 'per rank' = total / G is a LOWER bound of a rank's time on its own GPU: the replicated parts (the tail below min_layer_log, the size-n iNTT, the decommitment) are in it G times and do not shrink with G."
      cat $O/shard_threads_strong.txt; } > $P/r06_shard_threads_strong.txt
    grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/soak.txt > $P/r06_soak.txt
fi
biggest() { python3 - "$1" "$2" <<'PY'
import fnmatch, os, sys
files = [os.path.join(d, f) for d, _, fs in os.walk(sys.argv[1]) for f in fs if fnmatch.fnmatch(f, sys.argv[2])]
newest = max(os.path.getmtime(f) for f in files)
print(max((f for f in files if os.path.getmtime(f) >= newest - 180), key=os.path.getsize))
PY
}
cp $(biggest $O/prof_bench "*kernel_stats.csv") $P/r06_bench_2e24_kernel_stats.csv
cp $(biggest $O/prof_field "*kernel_stats.csv") $P/r06_bench_2e24_fieldhash_kernel_stats.csv
cp $(biggest $O/prof_staged "*kernel_stats.csv") $P/r06_staged_2e24_kernel_stats.csv
cp $(biggest $O/prof_cfg2 "*kernel_stats.csv") $P/r06_config2_2e20_kernel_stats.csv
cp $(biggest $O/prof_2e20 "*kernel_stats.csv") $P/r06_full_2e20_kernel_stats.csv
mkdir -p $P/r06_pmc
for n in fetch write fetch_staged write_staged fetch_field write_field sq sq_field; do
    cp $(biggest $O/pmc_$n "*counter_collection.csv") $P/r06_pmc/${n}_counter_collection.csv
done
cp $O/traffic.json $P/traffic.json      # made on the GPU box from the same PMC passes (tools/sessions/r06_final_a.sh), stamped there
cp $O/traffic_fieldhash.json $P/traffic_fieldhash.json
cp $O/valu_utilization.json $P/valu_utilization.json
python3 -c "
import json
d = json.load(open('$P/traffic.json')); print('traffic.json:', d['commit'], d['build_hash'], round(d['merkle_leaf_bytes_per_launch'] / 1e6, 1), 'MB per leaf launch')
d = json.load(open('$P/traffic_fieldhash.json')); print('traffic_fieldhash.json:', d['commit'], d['build_hash'], round(d['merkle_leaf_bytes_per_launch'] / 1e6, 1), 'MB per leaf launch')
b = json.load(open('$P/r06_bench_2e24.json')); print('bench:', round(b['ms_per_step'], 3), 'ms per proof, device-only', round(b['ms_per_step_device_only'], 3), 'parity_checked', b['parity_checked'], 'build', b['build_hash'], 'traffic from this build', b['roofline']['traffic_from_this_build'])
print('  full_2e20', round(b['full_2e20']['ms'], 4), 'fieldhash_2e24', round(b['fieldhash_2e24']['ms'], 3), b['fieldhash_2e24']['parity']['equal'], 'traffic from this build', b['fieldhash_2e24']['roofline']['traffic_from_this_build'])
lb = b['batched_2e24'].get('larger_batches', {})
print('  larger batches:', lb.get('proofs'), round(lb.get('ms_per_proof', 0), 3), round(lb.get('two_batches_in_flight', {}).get('ms_per_proof', 0), 3), lb.get('two_batches_in_flight', {}).get('frac_of_hashing_floor'))
print('  batched_2e24', round(b['batched_2e24']['ms_per_proof'], 3), round(b['batched_2e24']['two_batches_in_flight']['ms_per_proof'], 3), 'cfg2', round(b['lde_commit_2e20']['us'], 1), round(b['lde_commit_2e20']['us_sustained'], 1), round(b['lde_commit_2e20']['us_root_only'], 1), 'frac', round(b['roofline']['frac'], 3))
f = json.load(open('$P/r06_bench_2e24_fieldhash.json')); print('field:', round(f['ms_per_step'], 2), 'ms per proof, parity_checked', f['parity_checked'], 'build', f['build_hash'])"
