#!/usr/bin/env python3
"""VALU work of the Merkle and NTT kernels from hardware counters (one rocprofv3 --pmc pass with
SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE): writes profiles/valu_utilization.json.

Two clock-free figures per kernel, both from SQ_INSTS_VALU and the dispatch duration alone:

  valu_instr_per_wave     = SQ_INSTS_VALU / SQ_WAVES            (checks the static instruction counts of DESIGN.md)
  issue_slots_at_nominal  = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * duration * 2.4 GHz)
                            = fraction of the 64 lanes/clk/CU roofline at the NOMINAL clock, i.e. the same number
                            bench.py reports as roofline.valu.frac from its own timing

GRBM_GUI_ACTIVE / 8 / duration estimates the clock the chip held, but only for dispatches of >= 0.3 ms
(MI355X_MICROARCH.md, DVFS section: the counter window is wider than a short kernel and the quotient reads
high, up to 3.6 "GHz" on a 2.4 GHz part); for shorter dispatches `clock_ghz` is null, and no utilisation
"at the measured clock" is derived for them.

    python tools/pmc_valu.py gpurun_out/pmc_sq profiles/valu_utilization.json
"""
import collections
import csv
import glob
import json
import sys

NOMINAL_GHZ, SIMDS, CYCLES_PER_WAVE64_OP = 2.4, 1024, 4.0


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    import os, subprocess, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = sys.argv[3] if len(sys.argv) > 3 else (subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None)
    sys.path.insert(0, root)
    try:
        from zkstark_amd import build as zbuild
        build_hash = zbuild.source_hash()
    except Exception:                      # noqa: BLE001
        build_hash = None
    res = {"_method": __doc__.strip(), "commit": commit, "build_hash": build_hash, "collected": time.strftime("%Y-%m-%d"), "kernels": []}
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -max(kv[1]["_dur_ns"])):
        if ("merkle" not in name and "ntt" not in name and "coef_prepare" not in name and "compose" not in name and "fri_fold" not in name) \
                or "SQ_INSTS_VALU" not in v:
            continue
        mean = lambda c: sum(v[c]) / len(v[c])
        dur = mean("_dur_ns")
        clock = None
        if "GRBM_GUI_ACTIVE" in v and dur >= 300e3:
            clock = mean("GRBM_GUI_ACTIVE") / 8.0 / dur
        row = {"kernel": name[:name.rfind("(")] if name.endswith(")") else name, "grid_threads": int(grid),
               "launches_seen": len(v["SQ_INSTS_VALU"]), "valu_wave_instructions": mean("SQ_INSTS_VALU"),
               "valu_instr_per_wave": mean("SQ_INSTS_VALU") / mean("SQ_WAVES") if "SQ_WAVES" in v else None,
               "duration_us": dur / 1e3,
               "issue_slots_at_nominal": mean("SQ_INSTS_VALU") * CYCLES_PER_WAVE64_OP / (SIMDS * dur * NOMINAL_GHZ),
               "clock_ghz": clock}
        if clock:
            row["issue_slots_at_measured_clock"] = mean("SQ_INSTS_VALU") * CYCLES_PER_WAVE64_OP / (SIMDS * dur * clock)
        res["kernels"].append(row)
    json.dump(res, open(out, "w"), indent=1)
    for k in res["kernels"][:20]:
        clk = f"{k['clock_ghz']:.2f} GHz" if k["clock_ghz"] else "   -    "
        print(f"{k['issue_slots_at_nominal'] * 100:6.1f} % of nominal  {clk}  {k['duration_us']:9.1f} us  "
              f"{(k['valu_instr_per_wave'] or 0):9.0f} instr/wave  {k['kernel'][:70]} grid {k['grid_threads']}")


if __name__ == "__main__":
    main()
