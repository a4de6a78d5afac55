#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
ZK_HOST_TIMING=1 timeout -k 10 120 python tools/config2_only.py 17 20 > $O/config2_laps.txt 2>&1
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04e/bench.json'))
print('ms_per_step',b['ms_per_step'],'cfg2',b.get('lde_commit_2e20',{}).get('us'),'devonly',b.get('device_only',{}).get('ms_per_step'),'pipe',b.get('pipelined',{}).get('ms_per_proof'),'parity',b.get('parity_checked'))
for s in b['stages']: print(s['kernel'],s['launches'],s['ms'],s['hbm_frac'])
PY
echo done
