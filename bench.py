#!/usr/bin/env python3
"""bench.py -- STARK-101 prover throughput on MI355X.

Metric (BASELINE.json): field-elements/s through LDE + Merkle + FRI = N / t, N = evaluation
domain size, t = wall time from "trace values resident on the device" to "proof bytes on the
host" (context / twiddle setup excluded, reported separately).  Default workload: the full
prover at domain 2^24 (BASELINE.json configs[2]; trace group 2^21, blow-up 8) on synthetic
Fibonacci-square traces.  With --gpus N > 1: ONE proof at domain 2^24 * N sharded over the N
GPUs (weak scaling) by the native sharded prover (zk_shard_*: RCCL all-to-all per commitment).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 21] [--log-blowup 3]
    python bench.py --gpus N --plan-only      # the sharded layout and a written-down estimate: no GPU, no timing

`--gpus N` without a launcher starts the N ranks itself (one child process per GPU, before anything
touches the GPU); under `torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel
(merkle_subtree_kernel<leaf>), timed live with HIP events on the launch stream inside the
timed region; `cpu_baseline` is the CPU oracle (oracle/, a port of the reference algorithm
with O(N log N) transforms) on a bounded sample, rank 0 at N=1 only; `parity_checked` says the
timed proof's bytes were compared with the oracle's proof of the same trace.

Three files since round 6: this one (arguments, the N = 1 headline, the one line), bench_legs.py (the secondary legs of
the N = 1 line: configs[1], the full proof at 2^20, the field hash at 2^24, batches, ...), bench_multi.py (N > 1: launcher,
supervisors, transport ladder, the sharded run, --plan-only).
"""
import argparse
import json
import math
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench_legs import HASH_MODEL, HBM_PEAK_GBS, SIMDS, VALU_PEAK_4CYC_TOPS, mix_peak_tops   # noqa: E402  (no GPU, no torch)

PROFILE_TRAFFIC = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch (stamped with its commit)
SECONDARY_BUDGET_S = 420.0     # N = 1: a secondary leg is not STARTED once the run is this old (the line names it in legs_skipped)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=21, help="log2 of the trace group size n (per GPU)")
    ap.add_argument("--log-blowup", type=int, default=3)
    ap.add_argument("--hash", choices=("sha256", "field"), default="sha256",
                    help="Merkle hash: the reference's SHA-256 (the benchmark), or the field-native hash of configs[4]")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (configs[1], proofs in flight, batches, staged stages): profiling runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, help="also report throughput with this many proofs in flight (1 = skip)")
    ap.add_argument("--batch-log", type=int, default=3, help="batched_2e24 leg: 2^this proofs of the benchmark's domain in lockstep (8 x 2^24: 28 GB)")
    ap.add_argument("--soak-seconds", type=float, default=5.0,
                    help="after the timed region: keep proving for this long (untimed by the metric; steady-state figure)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=None, help="oracle sample: domain 2^(this+blowup); default: the benchmark's own size")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = domain 2^(log_n + blowup) * N, per-GPU work fixed (default); strong = the single-GPU domain split over N GPUs")
    ap.add_argument("--plain-collectives", action="store_true", help="N > 1: no chunked exchange, no shared-memory root board")
    ap.add_argument("--staged-only", action="store_true", help="run only the stage-by-stage leg (rocprofv3 of compose / fold kernels)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print zk_shard_plan for N = 2, 4, 8 (or --gpus N) at the weak and strong shapes with the estimated per-rank "
                         "critical path (one-GPU proxies under profiles/), as one JSON document; needs no GPU")
    ap.add_argument("--no-fieldhash-leg", action="store_true", help="N = 1: skip the fieldhash_2e24 leg (configs[4]; ~10 s incl. its oracle check)")
    return ap.parse_args()


def host_cores():
    """Cores this process may use: the affinity mask (what `nproc` prints), capped by a cgroup CPU quota."""
    import oracle
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = max(1, math.ceil(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    return oracle.usable_cores(), aff, quota


def cpu_baseline(sample_log_n, log_b):
    """Times the CPU oracle (kind 'port': the reference is a Rust crate that cannot be built here) on a bounded
    sample of the same workload, on every core this process may use.  Returns (record, oracle proof result)."""
    import oracle
    cores, nproc, quota = host_cores()
    N = 1 << (sample_log_n + log_b)
    oracle.set_threads(cores)
    oracle.prove(max(sample_log_n - 3, 4), log_b, want_vectors=False)        # page in, spin up threads
    t0 = time.perf_counter()
    r = oracle.prove(sample_log_n, log_b, want_vectors=False)
    dt_all = time.perf_counter() - t0
    assert r.rc == 0
    oracle.set_threads(1)
    t0 = time.perf_counter()
    oracle.prove(sample_log_n - 3, log_b, want_vectors=False)
    dt_one = time.perf_counter() - t0
    # BASELINE.md plan item 1: the reference's own algorithm (naive Lagrange, per-point solve, schoolbook
    # division; single thread like the reference) at the only size the reference supports, configs[0]
    t0 = time.perf_counter()
    rn = oracle.prove(10, 3, mode=oracle.MODE_NAIVE, want_vectors=False)
    dt_naive = time.perf_counter() - t0
    assert rn.rc == 0
    oracle.set_threads(cores)
    rec = {
        "reference_algorithm": {"workload": "configs[0]: trace 1023, domain 8192, literal polynomial.rs arithmetic (O(n^3)), 1 thread",
                                "seconds": dt_naive, "value": 8192 / dt_naive, "unit": "field-elements/s"},
        "value": N / dt_all, "unit": "field-elements/s", "cores": cores, "kind": "port",
        "nproc": nproc, "host_logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
        "sample": f"oracle full prover (NTT mode), domain 2^{sample_log_n + log_b}, {cores} OpenMP threads = every core this process "
                  f"may use (nproc = {nproc}, host logical CPUs = {os.cpu_count()}, cgroup CPU quota = {quota if quota else 'none'}), {dt_all:.2f} s",
        "single_thread_value": (N // 8) / dt_one,
        "single_thread_sample": f"domain 2^{sample_log_n + log_b - 3}, 1 thread, {dt_one:.2f} s",
    }
    return rec, r


def traffic_record():
    if not os.path.exists(PROFILE_TRAFFIC):
        return None, None
    with open(PROFILE_TRAFFIC) as f:
        t = json.load(f)
    return t.get("merkle_leaf_bytes_per_launch"), {k: t.get(k) for k in ("commit", "build_hash", "collected") if k in t}


def kernel_clock_record(hash_name):
    """Clock the dominant kernel held in the PMC pass (GRBM_GUI_ACTIVE over its >= 0.3 ms launches; profiles/valu_utilization.json,
    stamped like traffic.json): the chain probe's short launches run at a higher clock than a 1.5 ms hashing launch does."""
    path = os.path.join(ROOT, "profiles", "valu_utilization.json")
    if hash_name != "sha256" or not os.path.exists(path):
        return None
    with open(path) as f:
        v = json.load(f)
    rows = [k for k in v.get("kernels", []) if "merkle_subtree_kernel<zk::PlainSrc, true, 0>" in k["kernel"] and k.get("clock_ghz")]
    if not rows:
        return None
    k = max(rows, key=lambda r: r["grid_threads"])
    return {"clock_ghz": k["clock_ghz"], "valu_instr_per_wave": k["valu_instr_per_wave"], "launch_us_under_pmc": k["duration_us"],
            "stamp": {x: v.get(x) for x in ("commit", "build_hash", "collected")}}


def emit_line(R, result):
    """Rank 0: builds and prints THE one JSON line from what `result` holds (at the end of the run, or -- N > 1 -- from the
    watchdog when a secondary leg hangs after the headline was measured)."""
    if R.emitted:
        return
    R.emitted.append(True)
    args, world, sharded_run, _lib, json_fd = R.args, R.world, R.sharded_run, R._lib, R.json_fd
    log_n, log_b = result.get("log_n", R.log_n), R.log_b          # N > 1, weak scaling: the sharded proof's own size
    dt = result["dt"]                                # N > 1: already the slowest rank's time (measure())
    proof = result.get("proof")
    N = 1 << (log_n + log_b)
    ms_per_step = dt / args.steps * 1e3
    value = result["units"] / dt
    dom = result["dom"]
    ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    traffic, traffic_stamp = traffic_record()
    hm = HASH_MODEL[args.hash]
    valu_ach = dom["ops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0            # T lane-ops/s
    mix_peak = mix_peak_tops(args.hash)
    # the kernel's own issue rate: wave-instructions per SIMD per ns (ops are lane-ops: / 64 lanes / 1024 SIMDs)
    kernel_ns_per_instr = dom["ms"] * 1e6 / (dom["ops"] / 64 / SIMDS) if dom["ops"] else None
    chain = result.get("chain") or []
    best_chain = min((c["ns_per_instr"] for c in chain), default=None)
    kclk = kernel_clock_record(args.hash) if not sharded_run else None
    valu = {"achieved": valu_ach, "unit": "T lane-ops/s (32-bit)",
            "peak_mix_weighted": mix_peak, "frac_of_mix_peak": valu_ach / mix_peak,
            "peak_all_4_cycle": VALU_PEAK_4CYC_TOPS, "frac_of_4_cycle_peak": valu_ach / VALU_PEAK_4CYC_TOPS,
            # the nominal figure of MI355X_MICROARCH.md: every op at 2 cycles per wave64 instruction (no kernel here can reach
            # it: 57 % of SHA-256's instructions are 4-cycle ops)
            "peak_nominal": 2 * VALU_PEAK_4CYC_TOPS, "frac_of_nominal_peak": valu_ach / (2 * VALU_PEAK_4CYC_TOPS),
            "ops_per_leaf_hash": hm["leaf_ops"], "ops_per_inner_hash": hm["inner_ops"], "four_cycle_share": round(hm["four_cycle_share"], 4),
            "kernel_ns_per_instr": kernel_ns_per_instr,
            # the same in cycles at the clock the kernel held under the PMC pass: the chain probe's short launches
            # hold a HIGHER clock (chain[].clock_ghz), so ns compare wall time, cycles compare issue efficiency
            "kernel_clock_pmc": kclk,
            "kernel_cycles_per_instr": (kernel_ns_per_instr * kclk["clock_ghz"]) if (kclk and kernel_ns_per_instr) else None,
            "kernel_clock_from_this_build": bool(kclk) and kclk["stamp"].get("build_hash") == _lib.build_hash(),
            "chain": chain, "chain_ns_per_instr": best_chain,
            "frac_of_chain": (best_chain / kernel_ns_per_instr) if (best_chain and kernel_ns_per_instr) else None,
            "peak_basis": "mix-weighted: 1024 SIMDs x 64 lanes x 2.4 GHz / (4 f4 + 2 (1 - f4)) cycles, f4 = share of 4-cycle ops in the hash "
                          "(profiles/r02_valu_microbench.txt); chain: zk_probe_hash_chain, the compiled inner hash in a dependent chain, "
                          "12 launches back to back, clock read from s_memtime / s_memrealtime"}
    roofline = {
        "kernel": "merkle_subtree_kernel<leaf>" if args.hash == "sha256" else "merkle_subtree_fh_kernel<leaf>",
        "bound": "valu",
        "achieved": valu_ach, "peak": mix_peak, "unit": "T lane-ops/s", "frac": valu_ach / mix_peak,
        "traffic": traffic if (args.hash == "sha256" and not sharded_run) else None, "traffic_stamp": traffic_stamp,
        # True when the PMC passes behind `traffic` were collected from the very build that ran this line
        "traffic_from_this_build": bool(traffic_stamp) and traffic_stamp.get("build_hash") == _lib.build_hash(),
        "launches": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
        "note": "integer-VALU bound (SURVEY.md 8d): frac is against the mix-weighted issue peak at the nominal clock; valu.frac_of_chain is "
                "against the measured steady-state rate of the compiled hash; hbm{} is the same launches against the HBM roofline; "
                "stages[] lists the HBM-bound kernels",
        "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1)},
        "valu": valu,
    }
    if sharded_run and (result.get("shard") or {}).get("chunked_layers"):
        # chunk builds of a layer alternate between two streams (csrc/shard.hip): a launch's HIP-event duration then
        # includes the time it shares the chip with its neighbour, so achieved / frac are LOWER bounds on this line
        roofline["note"] += ("; sharded run with chunked layers: the chunk launches of a layer run two at a time on two streams, the per-launch "
                             "durations overlap and achieved / frac are lower bounds (ZK_SHARD_ONE_BUILD_STREAM=1 gives unshared launches)")
    # the hashing of one proof against its floor: every Merkle launch of the per-stage proof, and the time the same
    # instruction count needs at the chain rate (a floor the kernels cannot beat by construction)
    pk = result["per_kernel"]
    hash_ms = sum(pk[k]["ms"] for k in ("merkle_leaf", "merkle_inner") if k in pk)
    hash_ops = sum(pk[k]["ops"] for k in ("merkle_leaf", "merkle_inner") if k in pk)
    if best_chain and hash_ops:
        floor_ms = hash_ops / 64 / SIMDS * best_chain * 1e-6
        roofline["hashing"] = {"ms_per_proof": hash_ms, "floor_ms_at_chain_rate": floor_ms, "frac": floor_ms / hash_ms if hash_ms else None,
                               "wave_instructions_per_simd": hash_ops / 64 / SIMDS}
    stages = []
    def add_stage(name, st, note=None):
        if st["launches"]:
            gbs = st["bytes"] / (st["ms"] * 1e-3) / 1e9 if st["ms"] > 0 else 0.0
            row = {"kernel": name, "launches": st["launches"], "ms": round(st["ms"], 4),
                   "algorithmic_GB": round(st["bytes"] / 1e9, 4), "GBps": round(gbs, 1),
                   "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                   "valu_frac_of_mix_peak": round(st["ops"] / (st["ms"] * 1e-3) / 1e12 / mix_peak, 4) if st["ms"] > 0 and st["ops"] and "merkle" in name else None}
            if name == "ntt":
                # algorithmic_GB counts every PASS (three-pass transforms: 6 launches read and write their arrays once each);
                # SURVEY 8d's compulsory figure for the same stage is 1 N (iNTT) + 4.5 N (coset NTT): input once, output once
                comp = 5.5 * N
                row["compulsory_GB"] = round(comp / 1e9, 4)
                row["hbm_frac_of_compulsory"] = round(comp / (st["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if st["ms"] > 0 else None
                row["note"] = "algorithmic_GB = per-pass traffic of the 6 launches; compulsory_GB = 5.5 N (SURVEY 8d): a three-pass transform moves 4.3 x the compulsory bytes"
            if note:
                row["note"] = note
            stages.append(row)
    for name, st in result["per_kernel"].items():
        add_stage(name, st)
    for name, st in (result.get("staged") or {}).items():
        if name != "ntt":
            add_stage(name + " (stand-alone)", st, "stage-by-stage API: this kernel is fused into leaf hashing in the timed path")
    out = {
        "metric": "field-elements/s through LDE+Merkle+FRI (full STARK-101 prover)",
        "value": value, "unit": "field-elements/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step,
        # the same proofs with nothing on the host thread (tree tops and FRI tail on the device): what travels to another host
        "ms_per_step_device_only": (result.get("device_only") or {}).get("ms_per_step"),
        "higher_is_better": True,
        "scaling": result["scaling"], "vs_baseline": None,
        "dtype": "u32 (mod 3*2^30+1) + SHA-256" if args.hash == "sha256" else "u32 (mod 3*2^30+1), field-native Merkle hash",
        "data": "synthetic Fibonacci-square trace (a0=1, a1=3141592), deterministic",
        "config": {"workload": f"full prover: LDE + compose + FRI + Merkle, domain 2^{log_n + log_b} "
                               f"(trace group 2^{log_n}, blow-up {1 << log_b})" + (f" per proof; {result['parallelism']}" if world > 1 else ""),
                   "log_n": log_n, "log_blowup": log_b, "domain": 1 << (log_n + log_b), "fri_rounds": log_n, "merkle_hash": args.hash,
                   "parallelism": result["parallelism"],
                   # host thread's share of the latency-bound end: [tree-top levels, log2 of the largest host-side FRI layer]
                   "host_levels": result.get("host_levels"), "host_hashing": result.get("host_hashing")},
        "roofline": roofline,
        "stages": stages,
        "setup_ms": round(result["setup_ms"], 1), "device_bytes": result["device_bytes"],
        "proof_bytes": result["proof_bytes"], "build_hash": _lib.build_hash(),
    }
    for k in ["device_only", "pipelined", "soak", "lde_commit_2e20", "full_2e20", "fieldhash_2e24", "reference_size_2e13", "batched_2e13", "lde_commit_sharded",
              "config4_2e26", "shard", "transport", "transport_note", "ladder"] + sorted(k_ for k_ in result if k_.startswith(("strong_2e", "batched_2e2"))):
        if result.get(k) is not None:
            out[k] = result[k]
    if result.get("legs_skipped"):
        out["legs_skipped"] = result["legs_skipped"]
    if sharded_run:
        out["transport"], out["transport_note"] = result["transport"], result["transport_note"]
        out["parity_checked"] = bool(result["parity"] and result["parity"].get("equal"))
        out["parity"] = result["parity"]
    if world == 1 and not sharded_run and not args.no_cpu_baseline and args.hash == "sha256":
        sample = args.cpu_sample_log_n if args.cpu_sample_log_n is not None else log_n
        out["cpu_baseline"], oracle_proof = cpu_baseline(sample, log_b)
        if sample == log_n:
            # the timed proof against the oracle's proof of the same trace: every byte, and the final channel state
            ok = proof.data == oracle_proof.proof and proof.state == oracle_proof.state
            out["parity_checked"] = bool(ok)
            out["parity"] = {"against": f"CPU oracle, full proof bytes + channel state at domain 2^{log_n + log_b}", "equal": bool(ok)}
            if not ok:
                print("[bench] PARITY FAILURE: the timed proof differs from the CPU oracle's", file=sys.stderr, flush=True)
        else:
            out["parity_checked"] = False
    if world == 1 and not sharded_run and not args.no_cpu_baseline and args.hash == "field":
        # configs[4]: the hash is the build's own definition; the checker is the oracle's independent implementation (plain
        # residues; eight hashes at a time in exact double arithmetic, itself pinned on its scalar form by the CPU tests).
        # Round 4: fast enough for the benchmark's own size -- every byte of the TIMED proof and the final channel state.
        import oracle
        oracle.set_hash(oracle.HASH_FIELD)
        oracle.set_threads(host_cores()[0])
        t0 = time.perf_counter()
        want = oracle.prove(log_n, log_b, want_vectors=False)
        dt_o = time.perf_counter() - t0
        oracle.set_hash(oracle.HASH_SHA256)
        ok = want.rc == 0 and proof.data == want.proof and proof.state == want.state
        out["parity_checked"] = bool(ok)
        out["parity"] = {"against": f"CPU oracle (field hash, independent implementation): full proof bytes + channel state at domain 2^{log_n + log_b} "
                                    f"(the timed proof; oracle {dt_o:.1f} s on {host_cores()[0]} threads)", "equal": bool(ok)}
        out["cpu_baseline"] = {"value": N / dt_o, "unit": "field-elements/s", "cores": host_cores()[0], "kind": "port",
                               "sample": f"oracle full prover with the field hash, domain 2^{log_n + log_b}, {dt_o:.2f} s"}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if out.get("parity_checked") is False and (out.get("parity") or {}).get("equal") is False:
        result["exit_code"] = 4

def run_single(R):
    """N = 1: the headline (args.steps timed proofs at domain 2^(log_n + log_blowup), trace resident -> proof bytes on the host),
    then the secondary legs (bench_legs.py), then the line."""
    import bench_legs as legs
    args, zk, barrier = R.args, R.zk, R.barrier
    log_n, log_b, local_rank = R.log_n, R.log_b, R.local_rank
    t_run = time.time()
    N = 1 << (log_n + log_b)
    ctx = zk.Context(log_n, log_b, device=local_rank, hash=args.hash)
    trace = zk.trace_fibsq((1 << log_n) - 1)
    ctx.trace_upload(trace)                      # resident before the timed region
    for _ in range(args.warmup):
        proof = ctx.prove()
    ctx.set_profiling(("merkle_leaf",))          # events around the dominant kernel only
    ctx.kernel_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = ctx.prove()
    barrier()
    dt = time.perf_counter() - t0
    dom = ctx.kernel_stats(reset=True)["merkle_leaf"]
    proof.verify()
    # one extra untimed step with every kernel class bracketed: per-stage table
    ctx.set_profiling("all")
    ctx.prove()
    per_kernel = ctx.kernel_stats(reset=True)
    ctx.set_profiling(())
    result = {"dt": dt, "dom": dom, "per_kernel": per_kernel, "setup_ms": ctx.setup_ms,
              "device_bytes": ctx.device_bytes, "proof_bytes": len(proof.data), "scaling": "weak",
              "units": N * args.steps, "parallelism": "single-gpu", "host_levels": list(ctx.host_levels),
              "host_hashing": zk.host_hash_mode(), "proof": proof, "legs_skipped": []}
    sha = args.hash == "sha256"
    secondary = not args.no_secondary

    def leg(key, fn, *a):
        """One secondary leg: not started when the run is older than SECONDARY_BUDGET_S; an error is recorded, never fatal."""
        if time.time() - t_run > SECONDARY_BUDGET_S:
            result["legs_skipped"].append(f"{key}: not started, the run was {time.time() - t_run:.0f} s old")
            return None
        try:
            result[key] = fn(*a)
        except zk.ZkError as e:
            result[key] = {"error": str(e)}
        return result[key]

    # the same proofs with everything on the device (host_levels (0, 0): no tree tops, no FRI tail on the host thread)
    if sha and tuple(ctx.host_levels) != (0, 0):
        leg("device_only", legs.device_only, R, ctx, proof)
    result["chain"] = legs.chain_probe(R, args.hash)      # roofline probe: the compiled hash in a dependent chain
    floor_ms = legs.hashing_floor_ms(per_kernel, result["chain"])
    if args.soak_seconds > 0:                             # keep the device busy for a few seconds: steady-state figure
        leg("soak", legs.soak, R, ctx, N)
    if sha and secondary:
        leg("staged", legs.staged, R, log_n, log_b)
        leg("lde_commit_2e20", legs.lde_commit_2e20, R)                    # BASELINE configs[1]
        leg("full_2e20", legs.full_2e20, R, result["chain"])               # the metric's other domain, full prover
        if time.time() - t_run <= SECONDARY_BUDGET_S:
            try:
                result["reference_size_2e13"], result["batched_2e13"] = legs.reference_size(R)   # BASELINE configs[0]
            except zk.ZkError as e:
                result["reference_size_2e13"] = {"error": str(e)}
        # throughput mode at the metric's own domain: 2^batch_log proofs in lockstep, every proof compared byte for byte with
        # zk_prove of the same trace; and the same with batches twice as large (`larger_batches`: 16 x 2^24, 55 GB each)
        if log_n + log_b <= 24:
            key = f"batched_2e{log_n + log_b}"
            rec_b = leg(key, legs.batched, R, ctx, proof, trace, log_n, log_b, args.batch_log, True, floor_ms)
            if rec_b and "error" not in rec_b and args.batch_log + 1 + log_n + log_b <= 28 and time.time() - t_run <= SECONDARY_BUDGET_S:
                try:
                    rec_b["larger_batches"] = legs.batched(R, ctx, proof, trace, log_n, log_b, args.batch_log + 1, False, floor_ms)
                except zk.ZkError as e:
                    rec_b["larger_batches"] = {"error": str(e)}
    if args.in_flight > 1 and secondary:
        leg("pipelined", legs.pipelined, R, ctx, trace, log_n, log_b, args.hash)
    ctx.close()
    # BASELINE configs[4] in the same line (round 6): a few timed field-hash proofs at the headline's domain, byte-compared with
    # the oracle; its own context, after the headline's has been closed (device memory)
    if sha and secondary and not args.no_fieldhash_leg and not args.no_cpu_baseline and log_n + log_b <= 24:
        try:
            chain_f = legs.chain_probe(R, "field")
        except zk.ZkError:
            chain_f = None
        leg("fieldhash_2e24" if log_n + log_b == 24 else f"fieldhash_2e{log_n + log_b}", legs.fieldhash_2e24, R, log_n, log_b, trace, chain_f)
    if not result["legs_skipped"]:
        del result["legs_skipped"]
    emit_line(R, result)
    if result.get("exit_code"):
        sys.exit(result["exit_code"])


def main():
    args = parse()
    if args.plan_only:
        import bench_multi
        sys.exit(bench_multi.plan_only(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench_multi
        sys.exit(bench_multi.spawn_ranks(args))             # nothing here has touched torch or the GPU yet
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("ZK_BENCH_FORCE_SHARDED") == "1"
    if multi and not args.staged_only and os.environ.get("ZK_BENCH_WORKER") != "1":
        import bench_multi
        sys.exit(bench_multi.supervise())                   # this process stays clear of torch and the GPU; the work runs in a child
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("ZK_BENCH_TEST_WORKER_SLEEP"):         # tests/test_bench_cli.py: a worker that is busy when SIGTERM arrives above it
        time.sleep(float(os.environ["ZK_BENCH_TEST_WORKER_SLEEP"]))
    if os.environ.get("ZK_BENCH_TEST_WORKER_SCRIPT") and os.environ.get("ZK_BENCH_WORKER") == "1":
        # tests/test_bench_cli.py: the supervisor's state machine without a GPU -- the k-th worker of this rank exits with the k-th
        # scripted code ("7@1" first reports rung 1 as reached, as a real worker does before it dies there)
        script = os.environ["ZK_BENCH_TEST_WORKER_SCRIPT"].split(",")
        step = script[min(int(os.environ.get("ZK_BENCH_ATTEMPT", "0")) + int(os.environ.get("ZK_BENCH_TEST_STALE_SEEN", "0")), len(script) - 1)]
        code, _, reached = step.partition("@")
        print(f"[bench-test] worker: rung {os.environ.get('ZK_BENCH_RUNG')}, attempt {os.environ.get('ZK_BENCH_ATTEMPT')}, generation "
              f"{os.environ.get('ZK_BENCH_GEN')}: exiting with {step}", file=sys.stderr, flush=True)
        if reached and os.environ.get("ZK_BENCH_STATUS"):
            with open(os.environ["ZK_BENCH_STATUS"], "w") as f:
                f.write(reached)
        sys.exit(int(code))
    if world != args.gpus:
        args.gpus = world
    # before anything initialises HIP / HSA: dmabuf IPC for RCCL's peer-to-peer buffers, loopback for the gloo control plane
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and os.path.isdir("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: the container's hostname may not resolve

    # stdout carries exactly one JSON line: native libraries (RCCL prints a banner when a communicator is
    # created) write to file descriptor 1 directly, so it is pointed at stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import zkstark_amd as zk
    from zkstark_amd import _lib

    # ZK_BENCH_STAGED=1 rehearses the N > 1 path on a one-GPU box: every rank uses cuda:0 and the collectives
    # are staged through host memory (gloo).  Never a measurement configuration.
    staged = os.environ.get("ZK_BENCH_STAGED") == "1"
    force_sharded = os.environ.get("ZK_BENCH_FORCE_SHARDED") == "1"      # the N > 1 code path (RCCL) with one rank
    sharded_run = world > 1 or force_sharded
    if staged or os.environ.get("ZK_BENCH_SHARE_GPU") == "1":     # rehearsals on a one-GPU box: every rank on cuda:0 (with SHARE_GPU
        local_rank = 0                                            # the REAL transport ladder runs: RCCL refuses, the peer-copy rung works)
    ndev = torch.cuda.device_count()
    if ndev <= local_rank:
        print(f"[bench] rank {rank}: --gpus {world} needs {world} GPUs on this node, {ndev} visible", file=sys.stderr, flush=True)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    lib = _lib.load()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def dev_stats():
        arr = _lib.kernel_stat_array()
        _lib.check(lib.zk_dev_kernel_stats(arr, len(arr), 1))
        return {name: {"launches": int(a.launches), "ms": a.ms, "bytes": a.bytes, "ops": a.ops} for name, a in zip(_lib.KERNEL_CLASSES, arr)}

    R = types.SimpleNamespace(args=args, rank=rank, local_rank=local_rank, world=world, zk=zk, _lib=_lib, lib=lib, torch=torch, dist=dist,
                              log_n=args.log_n, log_b=args.log_blowup, staged=staged, force_sharded=force_sharded, sharded_run=sharded_run,
                              barrier=barrier, dev_stats=dev_stats, json_fd=json_fd, emitted=[], wd=None)
    R.emit_line = lambda result: emit_line(R, result)
    if args.staged_only:
        import bench_legs
        os.write(json_fd, (json.dumps({"staged": bench_legs.staged(R, R.log_n, R.log_b)}) + "\n").encode())
        return
    if sharded_run:
        import bench_multi
        bench_multi.worker_rendezvous(R)                    # watchdog, generation, gloo control plane
        bench_multi.run_sharded(R)
    else:
        run_single(R)


if __name__ == "__main__":
    main()
