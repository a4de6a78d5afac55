#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05cp5; mkdir -p $O
export TMPDIR=/tmp
for m in new old; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$m -o t -- python3 tools/prof_cp_from_f.py 21 $m > $O/prof_$m.log 2>&1 || { tail -20 $O/prof_$m.log; exit 1; }
  f=$(find $O/prof_$m -name "*kernel_trace.csv" | head -1)
  python3 - "$f" > $O/timeline_$m.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last proof: from the last-but-one "ntt_pass_fast_kernel<1u" group start
starts = [i for i, r in enumerate(rows) if "ntt_pass_fast_kernel<1u" in r["Kernel_Name"]]
# 3 inverse passes per proof: take the third-from-last as the start of the last proof
i0 = starts[-3] if len(starts) >= 3 else 0
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("zk::", "")[:70]
    print("%9.1f us  +gap %7.1f  dur %8.1f  q%s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
P
  rm -rf $O/prof_$m
done
head -70 $O/timeline_new.txt
