#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 600 python -m pytest tests/test_fieldhash.py -m gpu -q > $O/pytest_field.log 2>&1; echo "pytest field rc=$?"; tail -4 $O/pytest_field.log
timeout -k 10 300 python bench.py --hash field --steps 8 --warmup 2 --soak-seconds 0 --in-flight 1 --no-secondary > $O/bench_field.json 2> $O/bench_field.err; echo "bench field rc=$?"
python -c "
import json; b=json.load(open('$O/bench_field.json')); print('field ms', b['ms_per_step'], b.get('parity'), b.get('cpu_baseline'))"
bash tools/ab_ntt_tiles.sh $O/ab_ntt > $O/ab_ntt.log 2>&1; cat $O/ab_ntt/summary.txt
echo done
