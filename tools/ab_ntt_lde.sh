#!/bin/bash
# A/B of the LDE's NTT passes in the real pipeline, one session (boxes differ by +-3 %): each variant is built on the GPU box, checked for
# parity (LDE / NTT tests), then the NTT kernels of a 2^24 proof are summed from a rocprofv3 kernel trace and configs[1] is timed.
# Usage: bash tools/ab_ntt_lde.sh OUTDIR
O=${1:-gpurun_out/ab_ntt_lde}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
: > $O/summary.txt
for v in "" "-DZK_NTT_SMALL_MAX_LOG=21" "-DZK_NTT_LDE_DIRECT=1" "-DZK_NTT_SMALL_MAX_LOG=21 -DZK_NTT_LDE_DIRECT=1" ""; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lde or ntt or config2 or config3" > $O/pytest_$tag.log 2>&1 || { echo "$v: PARITY FAILED" | tee -a $O/summary.txt; continue; }
    rm -rf $O/prof_$tag
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/prof_$tag.log 2>&1
    timeout -k 10 120 python tools/config2_only.py 17 2000 > $O/cfg2_$tag.txt 2>&1
    python3 - "$v" $O/prof_$tag $O/cfg2_$tag.txt >> $O/summary.txt <<'PY'
import csv, glob, sys
v, d, c2 = sys.argv[1:4]
f = max(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=lambda p: len(open(p).read()))
rows = [r for r in csv.DictReader(open(f)) if "ntt_pass" in r["Name"]]
big = [r for r in rows if "12," in r["Name"]]
proofs = min(int(r["Calls"]) for r in big)
tot = sum(float(r["TotalDurationNs"]) for r in big) / proofs / 1e3
parts = " ".join("%s:%.1f" % (r["Name"].split("ntt_pass_fast_kernel")[1][:24].replace(" ", ""), float(r["AverageNs"]) / 1e3) for r in big)
print(f"{v or '(default)':52s} LDE {tot:6.1f} us/proof ({394.26 / tot / 8e6 * 1e6 * 100:.1f} % of 8 TB/s) | {parts} | {open(c2).read().strip().splitlines()[-1][:60]}")
PY
    tail -1 $O/summary.txt
    find $O -name "*.db" -delete
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
