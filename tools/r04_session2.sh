#!/bin/bash
# round-4 GPU session 2: full parity suite on the new build (NTT addressing, shard options / second communicator / self-test,
# bench supervisor), sharded bench lines on one GPU, NTT tile A/B.
O=gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -6 $O/pytest.log
ZK_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-secondary > $O/bench_sharded_1rank.json 2> $O/bench_sharded_1rank.err; echo "sharded 1 rank rc=$?"
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_SIMULATE_NATIVE_FAILURE=hang timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-secondary --log-n 18 > $O/bench_hang.json 2> $O/bench_hang.err; echo "hang rehearsal rc=$?"; tail -5 $O/bench_hang.err
ZK_BENCH_FORCE_SHARDED=1 ZK_BENCH_SIMULATE_NATIVE_FAILURE=id timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-secondary --log-n 18 > $O/bench_idfail.json 2> $O/bench_idfail.err; echo "id failure rehearsal rc=$?"; tail -3 $O/bench_idfail.err
ZK_BENCH_STAGED=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --log-n 20 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo "rehearsal 2 rc=$?"
python - <<'PY'
import json
for f in ("bench_sharded_1rank","bench_hang","bench_idfail","bench_rehearsal_n2"):
    try:
        b=json.loads(open(f"gpurun_out/r04b/{f}.json").read().strip().splitlines()[-1])
        sh=b["shard"]
        print(f, "ms",round(b["ms_per_step"],3),"transport",b["transport"],"note",b.get("transport_note"),"ladder",b.get("ladder",{}).get("rung"),b.get("ladder",{}).get("seconds_since_supervisor_start"),
              "exch",sh.get("exchange_ms"),"exposed",sh.get("exposed_exchange_ms"),"tail",sh.get("tail_ms"),"comms",sh.get("communicators"),"selftest",sh.get("selftest_ok"),"parity",b.get("parity_checked"))
    except Exception as e: print(f,"parse failed",e)
PY
echo done
