import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "merkle" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0].replace("void zk::", "")
        agg[(name, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    v = sorted(v)
    print(f"{name:60s} grid {grid:8d} ({grid // wg:5d} workgroups) launches {len(v):3d}  median {v[len(v) // 2] / 1e3:8.2f} us  min {v[0] / 1e3:8.2f} us")
