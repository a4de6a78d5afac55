#!/bin/bash
O=gpurun_out/r06h; rm -rf $O; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -o /tmp/launch_gate_probe tools/launch_gate_probe.hip > $O/probe_build.log 2>&1 && timeout -k 10 200 /tmp/launch_gate_probe > $O/launch_gate_probe.txt 2>&1; echo "probe rc=$?"; cat $O/launch_gate_probe.txt
