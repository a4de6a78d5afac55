#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05cp3; mkdir -p $O
export TMPDIR=/tmp
for m in new old; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o t -- python3 tools/prof_cp_from_f.py 21 $m > $O/prof_$m.log 2>&1 || { tail -20 $O/prof_$m.log; exit 1; }
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m" >> $O/summary.txt
  python3 - "$f" >> $O/summary.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-150s calls %5s avg %9.1f us total %9.1f us" % (r["Name"][:150], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
P
  cp "$f" $O/kernel_stats_$m.csv; rm -rf $O/prof_$m
done
cat $O/summary.txt


