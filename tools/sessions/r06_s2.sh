#!/bin/bash
# round 6, session 2: the rest of the GPU suite after the fixed test, the launch-gate probe, the latency-phase itemisation
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests/test_fieldhash.py tests/test_gpu_shard_native.py tests/test_gpu_sharded.py tests/test_host_sha.py tests/test_kernel_descriptors.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_gate_probe tools/launch_gate_probe.hip > $O/probe_build.log 2>&1 && timeout -k 10 120 /tmp/launch_gate_probe > $O/launch_gate_probe.txt 2>&1; echo "probe rc=$?"; cat $O/launch_gate_probe.txt
export ZK_BUILD_DEFS="-DZK_WG_TRACE=1"
python -m zkstark_amd.build > $O/build_trace.log 2>&1 || { echo "trace build failed"; tail $O/build_trace.log; }
for cfg in "21 sha256" "17 sha256" "21 field"; do
    set -- $cfg
    export ZK_WG_TRACE_FILE=$O/wg_$1_$2.raw
    timeout -k 10 200 python tools/wg_trace.py run $1 $2 > $O/wg_run_$1_$2.log 2>&1; echo "trace run $cfg rc=$?"
    python tools/wg_trace.py report $ZK_WG_TRACE_FILE > $O/wg_report_$1_$2.txt 2>&1
    tail -8 $O/wg_report_$1_$2.txt
done
unset ZK_BUILD_DEFS ZK_WG_TRACE_FILE
python -m zkstark_amd.build > /dev/null 2>&1
echo done
