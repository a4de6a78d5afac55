#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
for k in 4 3 2 1; do
  ZK_MERKLE_MAX_K=$k timeout -k 10 300 python bench.py --steps 20 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/bench_k$k.json 2> $O/bench_k$k.err
  python3 -c "
import json
d=json.load(open('$O/bench_k$k.json'))
print('max_k', $k, 'ms_per_step', round(d['ms_per_step'],3), [(s['kernel'], s['launches'], s['ms']) for s in d['stages']])"
done
echo done
