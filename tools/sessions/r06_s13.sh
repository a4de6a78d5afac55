#!/bin/bash
O=gpurun_out/r06n; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 python tools/soak_early_launch.py 12 > $O/soak_early_launch.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/soak_early_launch.txt
