#!/bin/bash
# round-4 GPU session 1: NTT passes with buffer addressing / mid tile / in-pass coefficient preparation
O=gpurun_out/r04a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "ntt or lde or config2 or config3 or canonical or other_sizes or batch_prover_matches or shard_domain" > $O/pytest_ntt.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_ntt.log; tail -5 $O/pytest_ntt.log
timeout -k 10 120 python tools/config2_lde_commit.py > $O/config2.txt 2>&1; cat $O/config2.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --soak-seconds 0 --in-flight 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 20 > $O/prof_cfg2.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
python - <<'PY'
import json,glob
try:
    b=json.load(open('gpurun_out/r04a/bench.json'))
    print('ms_per_step',b['ms_per_step'],'cfg2',b.get('lde_commit_2e20',{}).get('us'),'devonly',b.get('device_only',{}).get('ms_per_step'))
    for s in b['stages']: print(s['kernel'],s['launches'],s['ms'],s['hbm_frac'])
except Exception as e: print('bench parse failed',e)
for f in glob.glob('gpurun_out/r04a/prof_*/**/*kernel_stats.csv',recursive=True):
    print(f)
    for l in open(f).read().splitlines()[:14]: print('  ',l[:200])
PY
echo done
