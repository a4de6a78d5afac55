#!/bin/bash
# round-5 diagnostic session 1: the 8-rank total-work figure with and without chunking of the 2^21-word pieces, the strong shape
# (a 2^24 proof over 8 / 4 / 2 ranks as threads of one process) with min_layer_log swept, a 2^24 batch, the field-hash line.
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o tools/shard_threads_check || exit 1
T=$O/threads.txt
for ov in 0 22; do echo "# weak 8 ranks, overlap_min_log=$ov (0 = default 21)" >> $T; timeout -k 10 200 ./tools/shard_threads_check 8 24 3 0 0 $ov 3 2>&1 | grep -E "timing|threads ok" >> $T; done
for w in 8 4 2; do for ml in 0 20 21 23 24; do echo "# strong shape: world $w, log_n 21, min_layer_log=$ml (0 = default 22)" >> $T; timeout -k 10 120 ./tools/shard_threads_check $w 21 3 $ml 0 0 5 2>&1 | grep -E "timing|threads ok|rank" >> $T; done; done
echo "threads done"
timeout -k 10 200 python - > $O/batch24.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
for lb in (0, 1, 2):
    batch = 1 << lb
    with zk.BatchContext(21, 3, lb) as bc:
        bc.gen_fibsq([1] * batch, [3141592 + p for p in range(batch)])
        bc.prove_raw()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            bc.prove_raw()
        dt = (time.perf_counter() - t0) / reps
        print("batch %d x 2^24: %.3f ms per batch, %.3f ms per proof, device bytes %.2f GB" % (batch, dt * 1e3, dt * 1e3 / batch, bc.device_bytes / 1e9), flush=True)
PY
echo "batch done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace8_def -- ./tools/shard_threads_check 8 24 3 0 0 0 1 > $O/trace8_def.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace8_ov22 -- ./tools/shard_threads_check 8 24 3 0 0 22 1 > $O/trace8_ov22.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections, os
O = sys.argv[1]
for name in ("trace8_def", "trace8_ov22"):
    files = glob.glob(f"{O}/{name}/**/*kernel_trace.csv", recursive=True)
    if not files: print(name, "no trace"); continue
    f = max(files, key=os.path.getsize)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(f"{O}/{name}_summary.txt", "w") as out:
        for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            out.write(f"{us:12.1f} us {n:6d} launches  {k}\n")
PY
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
echo done
