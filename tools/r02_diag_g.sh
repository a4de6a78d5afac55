#!/bin/bash
O=gpurun_out/r02g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --deselect tests/test_gpu_kernels.py::test_maximum_domain_2e30 -k "merkle or config or prover_matches or host_levels or batch" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
for k in 4 3 2; do
  ZK_MERKLE_MAX_K=$k timeout -k 10 300 python bench.py --steps 30 --no-secondary --no-cpu-baseline --soak-seconds 0 > $O/bench_k$k.json 2> $O/bench_k$k.err
  python3 -c "
import json
d=json.load(open('$O/bench_k$k.json'))
print('max_k', $k, 'ms_per_step', round(d['ms_per_step'],3), [(s['kernel'], s['launches'], s['ms']) for s in d['stages']])"
done
echo done
