#!/usr/bin/env python3
"""VALU utilisation of the Merkle and NTT kernels from hardware counters (one rocprofv3 --pmc pass with
SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES): writes profiles/valu_utilization.json.

  issue slots  = 1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) cycles / 4 cycles per wave64 VALU instruction
  utilisation  = SQ_INSTS_VALU / issue slots
  clock        = GRBM_GUI_ACTIVE / 8 / kernel duration   (MI355X_MICROARCH.md, DVFS section)

    python tools/pmc_valu.py gpurun_out/pmc_sq profiles/valu_utilization.json
"""
import collections
import csv
import glob
import json
import sys


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {"_method": __doc__.strip().split("\n\n")[0], "kernels": []}
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -max(kv[1]["_dur_ns"])):
        if ("merkle" not in name and "ntt" not in name and "coef_prepare" not in name) or "SQ_INSTS_VALU" not in v:
            continue
        mean = lambda c: sum(v[c]) / len(v[c])
        cycles = mean("GRBM_GUI_ACTIVE") / 8.0
        slots = 1024.0 * cycles / 4.0
        res["kernels"].append({
            "kernel": name[:name.rfind("(")] if name.endswith(")") else name, "grid_threads": int(grid), "launches_seen": len(v["SQ_INSTS_VALU"]),
            "valu_wave_instructions": mean("SQ_INSTS_VALU"), "valu_instructions_per_wave": mean("SQ_INSTS_VALU") / mean("SQ_WAVES"),
            "duration_us": mean("_dur_ns") / 1e3, "clock_ghz": cycles / mean("_dur_ns"),
            "valu_utilization": mean("SQ_INSTS_VALU") / slots})
    json.dump(res, open(out, "w"), indent=1)
    for k in res["kernels"][:16]:
        print(f"{k['valu_utilization']*100:6.1f} %  {k['clock_ghz']:.2f} GHz  {k['duration_us']:9.1f} us  {k['valu_instructions_per_wave']:9.0f} instr/wave  {k['kernel'][:70]} grid {k['grid_threads']}")


if __name__ == "__main__":
    main()
