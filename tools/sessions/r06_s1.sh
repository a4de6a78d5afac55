#!/bin/bash
# round 6, session 1: the ABI change, the peer-copy transport, the three-file bench with its two new legs.  Nothing is filtered.
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json; tail -5 $O/bench.err
