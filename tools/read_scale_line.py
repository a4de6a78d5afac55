#!/usr/bin/env python3
"""Reads a `bench.py --gpus N` line (the driver's SCALE_rNN.json entries, or a rehearsal under profiles/) against the prediction written
down beforehand (profiles/r06_plan_only.json, from `bench.py --plan-only`): which rung produced the line, measured against estimated
ms per proof, where the difference sits (exchange time on the streams, the part of it the hashing waited for, the replicated tail, the
decommitment -- per rank), the bytes on the links against the plan, and the strong-scaling leg against its estimate.

    python tools/read_scale_line.py LINE.json [PLAN.json]        # LINE.json: one JSON object (or a file whose last line is one)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_line(path):
    text = open(path).read().strip()
    try:
        d = json.loads(text)
    except ValueError:
        d = json.loads(text.splitlines()[-1])
    return d.get("parsed", d)                                  # the driver wraps the line in {"parsed": ...}


def main():
    line = load_line(sys.argv[1])
    plan = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_plan_only.json")))
    n = line["n_gpus"]
    print(f"{n} GPU(s), transport {line.get('transport')} (rung {(line.get('ladder') or {}).get('rung')}, worker {(line.get('ladder') or {}).get('worker')}), "
          f"scaling {line['scaling']}, domain 2^{line['config']['log_n'] + line['config']['log_blowup']}: {line['ms_per_step']:.3f} ms per proof = "
          f"{line['value']:.3e} {line['unit']}; parity {line.get('parity_checked')}")
    if line.get("transport_note"):
        print("  NOTE:", line["transport_note"])
    if n == 1 and not line.get("shard"):
        print("  (a single-GPU line: nothing to compare)")
        return
    plain = bool((line.get("ladder") or {}).get("plain_collectives"))          # rungs 1-3, or a rehearsal started with --plain-collectives
    runs = {(r["world"], r["shape"], "plain" in r["transport"]): r for r in plan["runs"] if "estimate" in r}

    def compare(tag, rec_ms, shard, shape):
        r = runs.get((n, shape, bool(plain)))
        if not r or not r.get("estimate"):
            print(f"  {tag}: no estimate for world {n}, {shape}, {'plain' if plain else 'chunked'}")
            return
        e = r["estimate"]
        print(f"  {tag}: measured {rec_ms:.3f} ms against an ESTIMATE of {e['ms_per_proof_ESTIMATE']:.3f} ms "
              f"(compute >= {e['compute_ms_per_rank_lower_bound']:.2f}, links if exposed {e['link_ms_if_fully_exposed']:.3f}, "
              f"{r['collectives_per_proof']} collectives x ~25 us); single GPU {e['single_gpu_ms']:.2f} ms")
        if shard:
            a2a = shard.get("all_to_all_bytes")
            print(f"    bytes a rank puts on its links in the all-to-alls: {a2a:.0f} measured, {r['plan']['all_to_all_bytes']:.0f} planned"
                  f"{'' if a2a == r['plan']['all_to_all_bytes'] else '   <-- DIFFERS'}; sharded layers {shard.get('sharded_layers')} (plan {r['plan']['sharded_layers']}), "
                  f"chunked {shard.get('chunked_layers')} (plan {r['plan']['chunked_layers']})")
            for pr in shard.get("per_rank") or []:
                print(f"    rank {pr['rank']}: {pr['ms_per_step_local']:.3f} ms; exchanges {pr['exchange_ms']:.3f} ms on the streams, {pr['exposed_exchange_ms']:.3f} exposed; "
                      f"tail {pr['tail_ms']:.3f}, decommitment {pr['decommit_ms']:.3f}; self-test {'ok' if pr['selftest_ok'] else 'FAILED'}")

    shape = "weak" if line["scaling"] == "weak" else "strong"
    compare("headline", line["ms_per_step"], line.get("shard"), shape)
    for k, v in line.items():
        if k.startswith("strong_2e") and isinstance(v, dict) and "ms" in v:
            compare(k, v["ms"], v.get("shard"), "strong")
            if v.get("speedup_over_single_gpu"):
                print(f"    speed-up over the single-GPU prover in the same run: {v['speedup_over_single_gpu']:.2f} x ({v['single_gpu_ms']:.2f} ms -> {v['ms']:.2f} ms), parity {v['parity']['equal']}")
    for k in ("lde_commit_sharded", "config4_2e26"):
        if isinstance(line.get(k), dict) and "ms" in line[k]:
            extra = f", chunked / plain = {line[k]['plain_ab']['chunked_over_plain']:.2f}" if isinstance(line[k].get("plain_ab"), dict) and "chunked_over_plain" in line[k]["plain_ab"] else ""
            print(f"  {k}: {line[k]['ms']:.3f} ms{extra}" + (f", root matches golden: {line[k].get('root_matches_golden')}" if k == "config4_2e26" else ""))
    if line.get("legs_skipped"):
        print("  legs skipped:", line["legs_skipped"])


if __name__ == "__main__":
    main()
