#!/usr/bin/env python3
"""A/B runs of bench.py under different tuning environments (one child process per variant: the switches are read once).

    python tools/ab_env.py OUT.txt "NAME:K=V,K=V" "NAME2:..." [-- extra bench.py arguments]

Prints ms per proof and the per-stage milliseconds of every variant; each child also checks its proof with the verifier.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    out_path, variants = args[0], args[1:]
    lines = []
    for v in variants:
        name, _, kv = v.partition(":")
        env = dict(os.environ)
        for item in filter(None, kv.split(",")):
            k, _, val = item.partition("=")
            env[k] = val
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-secondary", "--no-cpu-baseline",
               "--soak-seconds", "0", "--in-flight", "1"] + extra
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            lines.append(f"{name:28s} FAILED rc={p.returncode}: {p.stderr.strip().splitlines()[-1] if p.stderr.strip() else ''}")
            print(lines[-1], flush=True)
            continue
        r = json.loads(p.stdout.strip().splitlines()[-1])
        st = {s["kernel"]: s["ms"] for s in r["stages"]}
        dev = r.get("device_only", {}).get("ms_per_step")
        lines.append(f"{name:28s} {r['ms_per_step']:7.3f} ms/proof | leaf {st.get('merkle_leaf', 0):6.3f} inner {st.get('merkle_inner', 0):6.3f} "
                     f"top {st.get('merkle_top', 0):6.3f} ntt {st.get('ntt', 0):6.3f} | kernel ns/instr {r['roofline']['valu']['kernel_ns_per_instr']:.4f} "
                     f"chain {r['roofline']['valu']['chain_ns_per_instr']} | device-only {dev}")
        print(lines[-1], flush=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
