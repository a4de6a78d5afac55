#!/bin/bash
# A/B of the NTT workgroup tile thresholds (kernels.hpp: ZK_NTT_SMALL_MAX_LOG, ZK_NTT_MID_MAX_LOG) in the real pipeline:
# each variant is built on the GPU box, then the NTT stage of a 2^24 proof and configs[1] (2^20) are timed.
# Usage: bash tools/ab_ntt_tiles.sh OUTDIR
O=${1:-gpurun_out/ab_ntt}; mkdir -p $O
: > $O/summary.txt
for v in "" "-DZK_NTT_MID_MAX_LOG=30" "-DZK_NTT_SMALL_MAX_LOG=16 -DZK_NTT_MID_MAX_LOG=22" "-DZK_NTT_SMALL_MAX_LOG=16 -DZK_NTT_MID_MAX_LOG=30" "-DZK_NTT_MID_MAX_LOG=20"; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "d$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --soak-seconds 0 --in-flight 1 > $O/sha_$tag.json 2> $O/sha_$tag.err
    python3 - "$v" $O/sha_$tag.json >> $O/summary.txt <<'PY'
import json, sys
v, a = sys.argv[1:3]
try:
    s = json.load(open(a))
    st = {x["kernel"]: x for x in s["stages"]}
    print(f"{v or '(default)':52s} proof {s['ms_per_step']:.3f} ms | ntt {st['ntt']['ms'] * 1e3:6.1f} us ({st['ntt']['launches']} launches, hbm {st['ntt']['hbm_frac']:.3f})  "
          f"config2 {s.get('lde_commit_2e20', {}).get('us', 0):6.1f} us  staged ntt {s.get('staged', {}).get('ntt', {}).get('ms', 0) * 1e3:6.1f} us")
except Exception as e:
    print(f"{v}: {e}")
PY
    tail -1 $O/summary.txt
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
