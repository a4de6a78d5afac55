#!/bin/bash
# Copies the artefacts of tools/r05_final_a.sh / r05_final_b.sh (merged back under gpurun_out/r05z) into profiles/.  Run in the build container after the GPU calls.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r05z; P=profiles
cp $O/bench_stamped.json $P/r05_bench_2e24.json; cp $O/bench_field.json $P/r05_bench_2e24_fieldhash.json
grep -v "amdgpu.ids" $O/batch_sizes.txt > $P/r05_batch_sizes.txt
grep "threads" $O/batch_inflight.txt > $P/r05_batch_inflight_2e24.txt
grep -v "amdgpu.ids" $O/config2_laps.txt > $P/r05_config2_laps.txt
grep "zk timing" $O/proof_laps.txt | tail -12 > $P/r05_proof_laps.txt
grep "LDE + Merkle commit" $O/config2_warmup.txt > $P/r05_config2_warmup.txt
if [ -f $O/bench_sharded_1rank.json ]; then
    cp $O/bench_sharded_1rank.json $P/r05_bench_sharded_1rank.json; cp $O/bench_sharded_1rank_torch.json $P/r05_bench_sharded_1rank_torch_transport.json
    cp $O/bench_rehearsal_n2.json $P/r05_bench_rehearsal_n2.json; cp $O/bench_rehearsal_n4.json $P/r05_bench_rehearsal_n4.json
    cp $O/bench_rehearsal_torchrun_n2.json $P/r05_bench_rehearsal_torchrun_n2.json
    cp $O/bench_rehearsal_hang.json $P/r05_bench_rehearsal_hang.json; grep -E "^\[bench\]" $O/bench_rehearsal_hang.err > $P/r05_bench_rehearsal_hang.log || true
    { echo "# tests/shard_threads_check.c <world> <log_n> 3 0 0 0 3, three repetitions: the native sharded prover with the ranks as THREADS of one process on ONE MI355X"
      echo "# (device-to-device transport standing in for xGMI; the GPU is shared, so a figure is the device work of all ranks together).  Weak-scaling sizes, 2^24 elements per rank."
      echo "# Round 3: 48.03 / 24.15 / 13.09 ms; round 4: 51.05 / 23.45 / 12.00 ms (the 8-rank figure did not reproduce: docs/LOG.md, round 5 item 2).  Final build of round 5: layers of >= 2^21 values distributed (>= 2^20 from 4 ranks on)."
      cat $O/shard_threads_timing.txt; } > $P/r05_shard_threads_timing.txt
    { echo "# the STRONG shape on the same harness: one 2^24 proof over 8 / 4 / 2 ranks (threads of one process sharing one GPU), min_layer_log swept."
      echo "# 'per rank' = total / G is a LOWER bound of a rank's time on its own GPU: the replicated parts (the tail below min_layer_log, the size-n iNTT, the decommitment) are in it G times and do not shrink with G."
      cat $O/shard_threads_strong.txt; } > $P/r05_shard_threads_strong.txt
    { echo "# tools/ab_cp_from_f.py 21 (one MI355X; one rank, collectives forced through RCCL; 2^24-point proof; two repetitions, alternating)."
      echo "# cp 'from f': recomputed over the rank's block from the received block of f (default, round 5); 'exchanged': zk_shard_options.exchange_cp (rounds 1-4)."
      echo "# At ONE rank the exchange of a rank with itself is RCCL's transport kernel at ~0.14 TB/s: exchange_ms is what drops.  Kernel timeline: r05_cp_from_f_timeline.txt."
      grep "^cp" $O/ab_cp_from_f.txt
      echo "# ranks as threads of one process sharing the GPU (tests/shard_threads_check.c <world> <log_n> 3 0 0 0 3; 'old' = ZK_HARNESS_EXCHANGE_CP=1):"
      cat $O/ab_cp_threads.txt; } > $P/r05_ab_cp_from_f.txt
    grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/soak.txt > $P/r05_soak.txt
fi
biggest() { python3 - "$1" "$2" <<'PY'
import fnmatch, os, sys
files = [os.path.join(d, f) for d, _, fs in os.walk(sys.argv[1]) for f in fs if fnmatch.fnmatch(f, sys.argv[2])]
newest = max(os.path.getmtime(f) for f in files)
print(max((f for f in files if os.path.getmtime(f) >= newest - 180), key=os.path.getsize))
PY
}
cp $(biggest $O/prof_bench "*kernel_stats.csv") $P/r05_bench_2e24_kernel_stats.csv
cp $(biggest $O/prof_field "*kernel_stats.csv") $P/r05_bench_2e24_fieldhash_kernel_stats.csv
cp $(biggest $O/prof_staged "*kernel_stats.csv") $P/r05_staged_2e24_kernel_stats.csv
cp $(biggest $O/prof_cfg2 "*kernel_stats.csv") $P/r05_config2_2e20_kernel_stats.csv
mkdir -p $P/r05_pmc
for n in fetch write fetch_staged write_staged sq sq_field; do
    cp $(biggest $O/pmc_$n "*counter_collection.csv") $P/r05_pmc/${n}_counter_collection.csv
done
cp $O/traffic.json $P/traffic.json      # made on the GPU box from the same PMC passes (tools/r05_final_a.sh), stamped there
cp $O/valu_utilization.json $P/valu_utilization.json
python3 -c "
import json
d = json.load(open('$P/traffic.json')); print('traffic.json:', d['commit'], d['build_hash'], round(d['merkle_leaf_bytes_per_launch'] / 1e6, 1), 'MB per leaf launch')
b = json.load(open('$P/r05_bench_2e24.json')); print('bench:', round(b['ms_per_step'], 3), 'ms per proof, device-only', round(b['ms_per_step_device_only'], 3), 'parity_checked', b['parity_checked'], 'build', b['build_hash'], 'traffic from this build', b['roofline']['traffic_from_this_build'])
lb = b['batched_2e24'].get('larger_batches', {})
print('  larger batches:', lb.get('proofs'), round(lb.get('ms_per_proof', 0), 3), round(lb.get('two_batches_in_flight', {}).get('ms_per_proof', 0), 3), lb.get('two_batches_in_flight', {}).get('frac_of_hashing_floor'))
print('  batched_2e24', round(b['batched_2e24']['ms_per_proof'], 3), round(b['batched_2e24']['two_batches_in_flight']['ms_per_proof'], 3), 'cfg2', round(b['lde_commit_2e20']['us'], 1), round(b['lde_commit_2e20']['us_sustained'], 1), 'frac', round(b['roofline']['frac'], 3))
f = json.load(open('$P/r05_bench_2e24_fieldhash.json')); print('field:', round(f['ms_per_step'], 2), 'ms per proof, parity_checked', f['parity_checked'], 'build', f['build_hash'])"
