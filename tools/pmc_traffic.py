#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as gpurun requires)
into profiles/traffic.json: HBM bytes per launch of each kernel.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read, so it is
doubled (checked here on ntt_pass_kernel<1>, which reads 4 * 2^24 B and reports 32 MiB);
WRITE_SIZE is exact for streaming stores.

The output is stamped with the commit and the library build hash it was collected from (bench.py prints the
stamp next to roofline.traffic, so a stale file is visible in the line itself).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 profiles/traffic.json [commit]
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time


def load(dirs, counter):
    """dirs: one directory or several separated by commas (e.g. the bench run and the --staged-only run)."""
    agg = collections.defaultdict(list)
    for d in dirs.split(","):
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    if commit is None:
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            commit = None
    sys.path.insert(0, root)
    try:
        from zkstark_amd import build as zbuild
        build_hash = zbuild.source_hash()
    except Exception:                      # noqa: BLE001
        build_hash = None
    F, W = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    res = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 1 --warmup 1`; "
                      "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over all launches of the kernel",
           "commit": commit, "build_hash": build_hash, "collected": time.strftime("%Y-%m-%d"), "kernels": {}}
    for k in sorted(set(F) | set(W)):
        if not k.startswith(("void zk::", "zk::")):
            continue
        nf, nw = len(F.get(k, [])), len(W.get(k, []))
        fb = 2.0 * sum(F.get(k, [])) * 1024 / max(nf, 1)
        wb = sum(W.get(k, [])) * 1024 / max(nw, 1)
        res["kernels"][k] = {"launches_seen": nf, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                             "hbm_bytes_per_launch": fb + wb}
    # the dominant kernel class of bench.py: every leaf-mode instantiation of merkle_subtree_kernel
    # (plain, fused-fold and fused-compose sources), averaged over all of their launches
    # ... of the FIRST directory of each list only (the `bench.py --steps 1 --warmup 1` run): the same launch population
    # as the bench line's algorithmic_bytes_per_launch; the --staged-only run has a different mix of leaf launches
    F1, W1 = load(fetch.split(",")[0], "FETCH_SIZE"), load(write.split(",")[0], "WRITE_SIZE")
    leaf = [k for k in set(F1) | set(W1) if "merkle_subtree_kernel<" in k and ", true," in k]
    nl = sum(len(W1.get(k, [])) for k in leaf)
    if nl:
        tot = sum(2.0 * sum(F1.get(k, [])) + sum(W1.get(k, [])) for k in leaf) * 1024
        res["merkle_leaf_bytes_per_launch"] = tot / nl
        res["merkle_leaf_launches_seen"] = nl
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  {k[:90]}")


if __name__ == "__main__":
    main()
