#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "" "-DZK_NTT_MID_MAX_LOG=30"; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_$tag.log 2>&1
    echo "== $v"; grep ntt_pass $O/prof_$tag/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/.*ntt_pass_fast_kernel//'
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
ZK_BENCH_STAGED=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 18 > $O/bench_torchrun_n2.json 2> $O/bench_torchrun_n2.err; echo "torchrun rehearsal rc=$?"; tail -c 600 $O/bench_torchrun_n2.json | head -c 300; echo
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
echo done
