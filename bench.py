#!/usr/bin/env python3
"""bench.py -- STARK-101 prover throughput on MI355X.

Metric (BASELINE.json): field-elements/s through LDE + Merkle + FRI = N / t, N = evaluation
domain size, t = wall time from "trace values resident on the device" to "proof bytes on the
host" (context / twiddle setup excluded, reported separately).  Default workload: the full
prover at domain 2^24 (BASELINE.json configs[2]; trace group 2^21, blow-up 8) on synthetic
Fibonacci-square traces.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 21] [--log-blowup 3]

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel
(merkle_subtree_kernel<leaf>), timed live with HIP events on the context stream inside the
timed region; `cpu_baseline` is the CPU oracle (oracle/, a port of the reference algorithm
with O(N log N) transforms) on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
# 32-bit VALU: 64 lanes/clk/CU measured for every op SHA-256 uses (tools/valu_microbench.hip):
# 256 CUs x 64 lanes x 2.4 GHz = 39.3 T lane-ops/s nominal; 35 T measured at the clock the chip holds.
VALU_PEAK_TOPS = 256 * 64 * 2.4e9 / 1e12
PROFILE_TRAFFIC = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch
PROFILE_VALU = os.path.join(ROOT, "profiles", "valu_utilization.json")   # PMC-derived VALU issue utilisation


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=21, help="log2 of the trace group size n")
    ap.add_argument("--log-blowup", type=int, default=3)
    ap.add_argument("--hash", choices=("sha256", "field"), default="sha256",
                    help="Merkle hash: the reference's SHA-256 (the benchmark), or the field-native hash of configs[4]")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (configs[1] and proofs in flight): profiling runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, help="also report throughput with this many proofs in flight (1 = skip)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=21, help="oracle sample: domain 2^(this+blowup)")
    return ap.parse_args()


def cpu_baseline(sample_log_n, log_b):
    """Times the CPU oracle (kind 'port': the reference is a Rust crate that cannot be built
    here) on a bounded sample of the same workload: the full prover at a smaller domain."""
    import oracle
    cores = min(os.cpu_count() or 1, 16)
    N = 1 << (sample_log_n + log_b)
    oracle.set_threads(cores)
    oracle.prove(sample_log_n - 3, log_b, want_vectors=False)        # page in, spin up threads
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
    return {
        "reference_algorithm": {"workload": "configs[0]: trace 1023, domain 8192, literal polynomial.rs arithmetic (O(n^3)), 1 thread",
                                "seconds": dt_naive, "value": 8192 / dt_naive, "unit": "field-elements/s"},
        "value": N / dt_all, "unit": "field-elements/s", "cores": cores, "kind": "port",
        "sample": f"oracle full prover (NTT mode), domain 2^{sample_log_n + log_b}, {cores} OpenMP threads, {dt_all:.2f} s",
        "single_thread_value": (N // 8) / dt_one,
        "single_thread_sample": f"domain 2^{sample_log_n + log_b - 3}, 1 thread, {dt_one:.2f} s",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # stdout carries exactly one JSON line: native libraries (RCCL prints a banner when a communicator is
    # created) write to file descriptor 1 directly, so it is pointed at stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import zkstark_amd as zk

    # ZK_BENCH_STAGED=1 rehearses the N > 1 path on a one-GPU box: every rank uses cuda:0 and the
    # collectives go through gloo (host-staged).  Never a measurement configuration.
    staged = os.environ.get("ZK_BENCH_STAGED") == "1"
    if staged:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    log_n, log_b = args.log_n, args.log_blowup
    N = 1 << (log_n + log_b)

    # ZK_BENCH_FORCE_SHARDED=1: run the N > 1 code path (sharded prover + RCCL collectives) with one rank
    force_sharded = os.environ.get("ZK_BENCH_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        from zkstark_amd import sharded
        if world == 1:
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        result = sharded.bench(args, rank, local_rank, world, barrier, staged=staged, force=force_sharded)
        log_n = result["log_n"]
        N = 1 << (log_n + log_b)
    else:
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
                  "units": N * args.steps, "parallelism": "single-gpu", "host_levels": list(ctx.host_levels)}
        # secondary figure: BASELINE.json configs[1], domain 2^20 LDE + Merkle commit (trace resident -> root on host)
        if args.hash == "sha256" and not args.no_secondary:
            with zk.Context(17, 3, device=local_rank) as c2:
                c2.trace_upload(zk.trace_fibsq((1 << 17) - 1))
                for _ in range(3):
                    c2.lde(); c2.merkle_commit(0)
                t0 = time.perf_counter()
                for _ in range(20):
                    c2.lde(); c2.merkle_commit(0)
                dt2 = (time.perf_counter() - t0) / 20
            result["lde_commit_2e20"] = {"workload": "configs[1]: domain 2^20 LDE + Merkle commit", "us": dt2 * 1e6,
                                         "value": (1 << 20) / dt2, "unit": "field-elements/s"}
        if args.hash == "sha256" and not args.no_secondary:
            # configs[0] on the GPU path: the reference's own size (trace 1023, domain 8192)
            with zk.Context(10, 3, device=local_rank) as c0:
                c0.trace_upload(zk.trace_fibsq(1023))
                for _ in range(3):
                    c0.prove()
                t0 = time.perf_counter()
                for _ in range(20):
                    c0.prove()
                dt0 = (time.perf_counter() - t0) / 20
            result["reference_size_2e13"] = {"workload": "configs[0]: full prover, trace 1023, domain 8192", "us": dt0 * 1e6,
                                             "value": 8192 / dt0, "unit": "field-elements/s"}
            # the same size, 1024 independent proofs in lockstep (zk_batch_*): traces resident -> all proof bytes on host
            with zk.BatchContext(10, 3, 10, device=local_rank) as bc:
                bc.gen_fibsq([1] * 1024, [3141592 + p for p in range(1024)])
                bc.prove_raw()
                t0 = time.perf_counter()
                for _ in range(5):
                    bc.prove_raw()
                dtb = (time.perf_counter() - t0) / 5
            result["batched_2e13"] = {"workload": "configs[0] x 1024: batch of 1024 proofs, trace 1023, domain 8192 each",
                                      "ms_per_batch": dtb * 1e3, "us_per_proof": dtb * 1e6 / 1024,
                                      "value": 1024 * 8192 / dtb, "unit": "field-elements/s"}
        if args.in_flight > 1 and not args.no_secondary:
            # secondary figure: several independent proofs in flight on one GPU (one context, stream and
            # host thread each), so one proof's latency-bound tree tops overlap another's hashing
            import threading
            ctxs = [ctx] + [zk.Context(log_n, log_b, device=local_rank, hash=args.hash) for _ in range(args.in_flight - 1)]
            for c in ctxs[1:]:
                c.trace_upload(trace)
                c.prove()
            def work(c):
                for _ in range(args.steps):
                    c.prove()
            barrier()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
            [t.start() for t in th]
            [t.join() for t in th]
            barrier()
            dtp = time.perf_counter() - t0
            result["pipelined"] = {"proofs_in_flight": args.in_flight, "value": args.in_flight * N * args.steps / dtp,
                                   "unit": "field-elements/s", "ms_per_proof": dtp / (args.in_flight * args.steps) * 1e3}
            for c in ctxs[1:]:
                c.close()
        ctx.close()

    dt = result["dt"]
    if world > 1 or force_sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if staged else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = result["units"] / dt
        dom = result["dom"]
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
        traffic = None
        if os.path.exists(PROFILE_TRAFFIC):
            with open(PROFILE_TRAFFIC) as f:
                traffic = json.load(f).get("merkle_leaf_bytes_per_launch")
        hw_valu = None
        if os.path.exists(PROFILE_VALU):
            with open(PROFILE_VALU) as f:
                ks = [k for k in json.load(f)["kernels"] if "merkle_subtree_kernel" in k["kernel"] and ", true," in k["kernel"]]
            if ks:   # the largest leaf launch: issue-slot utilisation and the clock the chip held (hardware counters)
                big = max(ks, key=lambda k: k["duration_us"])
                hw_valu = {"utilization": big["valu_utilization"], "clock_ghz": big["clock_ghz"], "kernel": big["kernel"],
                           "source": "rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE (profiles/valu_utilization.json)"}
        # SHA-256 work of the dominant kernel in 32-bit lane-ops (1 leaf + inner hashes)
        roofline = {
            "kernel": "merkle_subtree_kernel<leaf>", "bound": "hbm",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,
            "launches": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
            "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
            "note": "SHA-256 is integer-VALU bound, not HBM bound (SURVEY.md 8d): see valu{}; stages[] lists the HBM-bound kernels",
            "valu": {"achieved": dom["ops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0, "peak": VALU_PEAK_TOPS,
                     "unit": "T lane-ops/s (32-bit)",
                     "frac": (dom["ops"] / (dom["ms"] * 1e-3) / 1e12 / VALU_PEAK_TOPS) if dom["ms"] > 0 else 0.0,
                     "ops_per_leaf_hash": 1259 if args.hash == "sha256" else 10200,
                     "ops_per_inner_hash": 2293 if args.hash == "sha256" else 10300,
                     "hw_counters": hw_valu if args.hash == "sha256" else None},
        }
        stages = []
        for name, st in result["per_kernel"].items():
            if st["launches"]:
                gbs = st["bytes"] / (st["ms"] * 1e-3) / 1e9 if st["ms"] > 0 else 0.0
                stages.append({"kernel": name, "launches": st["launches"], "ms": round(st["ms"], 4),
                               "algorithmic_GB": round(st["bytes"] / 1e9, 4), "GBps": round(gbs, 1),
                               "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                               "valu_frac": round(st["ops"] / (st["ms"] * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4) if st["ms"] > 0 and st["ops"] else None})
        out = {
            "metric": "field-elements/s through LDE+Merkle+FRI (full STARK-101 prover)",
            "value": value, "unit": "field-elements/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": result["scaling"], "vs_baseline": None, "dtype": "u32 (mod 3*2^30+1) + SHA-256",
            "data": "synthetic Fibonacci-square trace (a0=1, a1=3141592), deterministic",
            "config": {"workload": f"full prover: LDE + compose + FRI + Merkle, domain 2^{log_n + log_b} "
                                   f"(trace group 2^{log_n}, blow-up {1 << log_b})" + (f" per proof; {result['parallelism']}" if world > 1 else ""),
                       "log_n": log_n, "log_blowup": log_b, "domain": N, "fri_rounds": log_n, "merkle_hash": args.hash,
                       "parallelism": result["parallelism"],
                       # host thread's share of the latency-bound end: [tree-top levels, log2 of the largest host-side FRI layer]
                       "host_levels": result.get("host_levels")},
            "roofline": roofline,
            "stages": stages,
            "setup_ms": round(result["setup_ms"], 1), "device_bytes": result["device_bytes"],
            "proof_bytes": result["proof_bytes"],
        }
        if "pipelined" in result:
            out["pipelined"] = result["pipelined"]
        for k in ("lde_commit_2e20", "reference_size_2e13", "batched_2e13", "lde_commit_sharded"):
            if result.get(k) is not None:
                out[k] = result[k]
        if world == 1 and not args.no_cpu_baseline and args.hash == "sha256":
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_log_n, log_b)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
