#!/bin/bash
# A/B of the radix-512 passes for transforms of 2^17 words (configs[1]: 6 -> 4 NTT launches per LDE), one session: each variant is
# built on the GPU box, checked for parity (NTT / LDE / configs[1] / prover tests), then configs[1] is timed (first 50 iterations
# after 5, and 2000 sustained) and its NTT kernels are summed from a rocprofv3 kernel trace.
# Usage: bash tools/ab_ntt_radix512.sh OUTDIR
O=${1:-gpurun_out/ab_r512}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "" "-DZK_NTT_RADIX512=1" "" "-DZK_NTT_RADIX512=1"; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lde or ntt or config2 or prover or batch" > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; tail -5 $O/pytest_$tag.log; continue; }
    rm -rf $O/prof_$tag
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 tools/config2_only.py 17 20 > $O/prof_$tag.log 2>&1
    for r in 50 2000 50; do timeout -k 10 120 python tools/config2_only.py 17 $r 2>&1 | grep "LDE + Merkle" >> $O/cfg2_$tag.txt; done
    python3 - "$v" $O/prof_$tag $O/cfg2_$tag.txt >> $O/summary.txt <<'PY'
import csv, glob, sys
v, d, c2 = sys.argv[1:4]
f = max(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=lambda p: len(open(p).read()))
rows = [r for r in csv.DictReader(open(f)) if "ntt_pass" in r["Name"]]
iters = 25
tot = sum(float(r["TotalDurationNs"]) for r in rows) / iters / 1e3
parts = " ".join("%s x%d:%.1f" % (r["Name"].split("ntt_pass_fast_kernel")[1][:22].replace(" ", ""), int(r["Calls"]) // iters, float(r["AverageNs"]) / 1e3) for r in rows)
t = [l.split("commit ")[1].split(" us")[0] for l in open(c2).read().splitlines()[-3:]]
print(f"{v or '(default: radix 64 x 64 x 32)':30s} NTT launches of one LDE {sum(int(r['Calls']) for r in rows) // iters}, {tot:5.1f} us | configs[1] {t[0]} / {t[1]} / {t[2]} us (50 / 2000 / 50 iterations) | {parts}")
PY
    tail -1 $O/summary.txt
    find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
