#!/bin/bash
# A/B of the build-time variants of the field arithmetic (field.hpp: ZK_MONT_VARIANT, ZK_ADDSUB_CMP) in the REAL kernels:
# each variant is built on the GPU box (hipcc, ~1 min), then the NTT stage, the stand-alone compose / fold kernels and the
# field-hash prover are timed.  Usage: bash tools/ab_field_variants.sh OUTDIR
O=${1:-gpurun_out/ab_field}; mkdir -p $O
: > $O/summary.txt
for v in "-DZK_MONT_VARIANT=2" "-DZK_MONT_VARIANT=4" "-DZK_MONT_VARIANT=4 -DZK_ADDSUB_CMP=1" "-DZK_MONT_VARIANT=2 -DZK_ADDSUB_CMP=1" "-DZK_MONT_VARIANT=0"; do
    export ZK_BUILD_DEFS="$v"
    tag=$(echo "$v" | tr -d ' ' | tr -c 'A-Za-z0-9=_\n' '_')
    python -m zkstark_amd.build > $O/build_$tag.log 2>&1 || { echo "$v: build failed" | tee -a $O/summary.txt; continue; }
    timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --soak-seconds 0 --in-flight 1 > $O/sha_$tag.json 2> $O/sha_$tag.err
    timeout -k 10 300 python bench.py --hash field --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --soak-seconds 0 --in-flight 1 > $O/field_$tag.json 2> $O/field_$tag.err
    python3 - "$v" $O/sha_$tag.json $O/field_$tag.json >> $O/summary.txt <<'PY'
import json, sys
v, a, b = sys.argv[1:4]
try:
    s = json.load(open(a)); f = json.load(open(b))
    st = {x["kernel"]: x["ms"] for x in s["stages"]}
    ft = {x["kernel"]: x["ms"] for x in f["stages"]}
    print(f"{v:42s} sha proof {s['ms_per_step']:.3f} ms | ntt {st.get('ntt', 0) * 1e3:6.1f} us  compose {st.get('compose (stand-alone)', 0) * 1e3:5.1f} us  "
          f"fold x4 {st.get('fri_fold (stand-alone)', 0) * 1e3:5.1f} us  config2 {s.get('lde_commit_2e20', {}).get('us', 0):6.1f} us | "
          f"field proof {f['ms_per_step']:.2f} ms (leaf {ft.get('merkle_leaf', 0):.2f}, top {ft.get('merkle_top', 0):.2f})")
except Exception as e:
    print(f"{v}: {e}")
PY
    tail -1 $O/summary.txt
done
unset ZK_BUILD_DEFS
python -m zkstark_amd.build > /dev/null 2>&1
echo done
