#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05tl; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python3 tools/prof_single.py 21 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/timeline_single.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "ntt_pass_fast_kernel<1u" in r["Kernel_Name"]]
i0 = starts[-3]
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("zk::", "")[:64]
    print("%9.1f us  +gap %7.1f  dur %8.1f  q%s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
P
rm -rf $O/prof
cat $O/timeline_single.txt
