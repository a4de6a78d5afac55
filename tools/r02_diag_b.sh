#!/bin/bash
# round-2 GPU session B: NTT rework + batch fixes: tests, probes, config 2, batches, the bench line
O=gpurun_out/r02b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_kernels.py::test_maximum_domain_2e30 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
timeout -k 10 120 ./tools/valu_microbench > $O/valu_microbench.txt 2>&1
timeout -k 10 120 ./tools/sha_latency_probe > $O/sha_latency_probe.txt 2>&1; cat $O/sha_latency_probe.txt
timeout -k 10 200 python tools/config2_lde_commit.py > $O/config2.txt 2>&1; cat $O/config2.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 tools/config2_only.py 17 50 > $O/prof_cfg2.log 2>&1
timeout -k 10 300 python tools/batch_laps.py > $O/batch_laps.txt 2>&1; grep "^==" $O/batch_laps.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2e24 -- python3 tools/host_timing.py 21 > $O/prof_2e24.log 2>&1
echo done
