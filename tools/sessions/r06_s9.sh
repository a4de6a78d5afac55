#!/bin/bash
O=gpurun_out/r06j; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export ZK_BUILD_DEFS="-DZK_PEER_DEBUG=1"
python -m zkstark_amd.build > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
sed -n '/^cat > \/tmp\/peer_big.py/,/^PY$/p' tools/sessions/r06_s8.sh | sed '1d;$d' > /tmp/peer_big.py
timeout -k 10 120 python /tmp/peer_big.py 2 22 > $O/peer_2_22.txt 2>&1; echo "rc=$?"
grep "^\[peer" $O/peer_2_22.txt | tail -40
unset ZK_BUILD_DEFS; python -m zkstark_amd.build > /dev/null 2>&1
