#!/bin/bash
set -o pipefail
O=gpurun_out/r05cp2; mkdir -p $O
timeout -k 10 300 python tools/ab_cp_from_f.py 21 > $O/ab_2e24.txt 2>&1 || { tail -20 $O/ab_2e24.txt; exit 1; }
cat $O/ab_2e24.txt
