#!/bin/bash
# HBM traffic of the ComposeBlockSrc leaf launch (cp over a rank's block from the received block of f) from two PMC passes
set -o pipefail
O=$PWD/gpurun_out/r05cp8; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/prof_cp_from_f.py 21 new > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/prof_cp_from_f.py 21 new > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_sharded.json cp8 > $O/traffic_sharded.txt
grep -i "ComposeBlock\|InterleaveSrc\|compose_kernel4" $O/traffic_sharded.txt
find $O -name "*.db" -delete
