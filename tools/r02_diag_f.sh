#!/bin/bash
O=gpurun_out/r02f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters.txt | sort -u | tr '\n' ' ' > $O/sq_names.txt
for k in 4 2; do
  ZK_MERKLE_MAX_K=$k timeout -k 10 200 python tools/merkle_only.py 21 5 > $O/plain_k$k.txt 2>&1; cat $O/plain_k$k.txt
  ZK_MERKLE_MAX_K=$k timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LEVEL_WAVES --output-format csv -d $O/pmc1_k$k -- python3 tools/merkle_only.py 21 2 > $O/pmc1_k$k.log 2>&1
  ZK_MERKLE_MAX_K=$k timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc2_k$k -- python3 tools/merkle_only.py 21 2 > $O/pmc2_k$k.log 2>&1
done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
echo done
