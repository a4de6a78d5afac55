#!/bin/bash
# round 6, session 3: the latency-phase itemisation (ZK_WG_TRACE build), then the default build again
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export ZK_BUILD_DEFS="-DZK_WG_TRACE=1"
python -m zkstark_amd.build > $O/build_trace.log 2>&1 || { echo "trace build failed"; tail $O/build_trace.log; }
for cfg in "21 sha256" "17 sha256" "21 field"; do
    set -- $cfg
    export ZK_WG_TRACE_FILE=$PWD/$O/wg_$1_$2.raw
    timeout -k 10 200 python tools/wg_trace.py run $1 $2 > $O/wg_run_$1_$2.log 2>&1; echo "trace run $cfg rc=$?"
    python tools/wg_trace.py report $ZK_WG_TRACE_FILE > $O/wg_report_$1_$2.txt 2>&1
    tail -12 $O/wg_report_$1_$2.txt
done
unset ZK_BUILD_DEFS ZK_WG_TRACE_FILE
python -m zkstark_amd.build > /dev/null 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pending or caller_allocated or merkle" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo done
