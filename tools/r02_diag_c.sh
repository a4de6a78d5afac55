#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shard_native.py -m gpu -q -x --deselect tests/test_gpu_kernels.py::test_maximum_domain_2e30 -k "host_levels or shard_from_plain_c or config2 or lde or batch or c_abi" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
timeout -k 10 600 python tools/host_levels_sweep.py > $O/host_levels.txt 2>&1; cat $O/host_levels.txt
timeout -k 10 200 python tools/config2_lde_commit.py > $O/config2.txt 2>&1; cat $O/config2.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 50 > $O/prof_cfg2.log 2>&1
echo done
