#!/bin/bash
# round 6, session 14: the largest domains on the final build (2^27 ... 2^30 on one GPU), with and without the early-launch option
O=gpurun_out/r06o; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 500 python tools/big_domain_check.py 24 25 26 27 > $O/big_domains.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/big_domains.txt
