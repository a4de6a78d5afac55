#!/usr/bin/env python3
"""bench_legs.py -- the secondary legs of bench.py's N = 1 line: everything measured AFTER the headline (full prover at domain
2^24) has been timed.  Each leg takes the run namespace R (bench.main), the headline's context where it needs it, and returns the
record that goes into the line under its own key.  Nothing here touches the oracle except `fieldhash_2e24`, which uses it as the
CHECKER of the timed field-hash proof (the oracle is test infrastructure: oracle/__init__.py).

Legs (keys of the line):
  device_only        the headline with nothing on the host thread (host_levels (0, 0))
  chain              zk_probe_hash_chain: the compiled hash in a dependent chain (the roofline's measured floor)
  soak               a few seconds of back-to-back proofs
  staged             the stand-alone compose / fold kernels (fused away in the timed path)
  lde_commit_2e20    BASELINE configs[1]: domain 2^20 LDE + Merkle commit
  full_2e20          one FULL proof at the metric's other domain, 2^20 (round 6), with its floor and its latency-phase share
  fieldhash_2e24     BASELINE configs[4]: a few timed proofs with the field-native hash, byte-compared with the oracle (round 6)
  reference_size_2e13, batched_2e13, batched_2e24, pipelined
"""
import threading
import time

HBM_PEAK_GBS = 8000.0
NOMINAL_GHZ, SIMDS = 2.4, 256 * 4
VALU_PEAK_4CYC_TOPS = SIMDS * 64 * NOMINAL_GHZ * 1e9 / 4 / 1e12
# Per hash: VALU instructions (ISA count, tools/kernel_descriptors.py; tests/test_kernel_descriptors.py pins them against
# the built code object) and the 4-cycle share of the mix, which gives the mix-weighted issue peak.
HASH_MODEL = {
    "sha256": {"leaf_ops": 1259, "inner_ops": 2293, "probe_ops": 2246, "four_cycle_share": (940 + 365) / 2262.0},
    # field hash (double precision since round 5, csrc/fieldhash_f64.hpp): ISA loop counts (straight-line part + 8 trips of the
    # full-round loops + 10 trips of the two-partial-round loop; tests/test_kernel_descriptors.py re-counts them from the built
    # code object).  Every instruction is a double-precision op: the 4-cycle class (measured 4.1 - 5.5 cycles, tools/fh64_probe.hip)
    "field": {"leaf_ops": 5015, "inner_ops": 5072, "probe_ops": 5046, "four_cycle_share": 1.0},
}


def mix_peak_tops(hash_name):
    """Issue peak for this hash's instruction mix at the nominal clock: lanes / (mean cycles per instruction)."""
    f4 = HASH_MODEL[hash_name]["four_cycle_share"]
    return SIMDS * 64 * NOMINAL_GHZ * 1e9 / (4 * f4 + 2 * (1 - f4)) / 1e12


def hashing_floor_ms(per_kernel, chain, include_top=False):
    """Time the Merkle launches of one proof need at the chain rate of the compiled hash (a floor by construction).  The throughput
    launches only, as every round reported it at 2^24 (there the latency launches hold 0.2 % of the hashes); include_top adds the
    latency launches' hashes (a 2^20 proof: most trees are small enough to be ONE latency launch, leaves included)."""
    keys = ("merkle_leaf", "merkle_inner") + (("merkle_top",) if include_top else ())
    ops = sum(per_kernel[k]["ops"] for k in keys if k in per_kernel)
    best = min((c["ns_per_instr"] for c in chain), default=None)
    return (ops / 64 / SIMDS * best * 1e-6) if (best and ops) else None


def device_only(R, ctx, proof):
    keep = tuple(ctx.host_levels)
    ctx.set_host_levels(0, 0)
    for _ in range(2):
        dproof = ctx.prove()
    reps = min(R.args.steps, 20)
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dproof = ctx.prove()
    R.barrier()
    rec = {"ms_per_step": (time.perf_counter() - t0) / reps * 1e3, "steps": reps, "host_levels": [0, 0],
           "same_proof": dproof.data == proof.data and dproof.state == proof.state}
    ctx.set_host_levels(*keep)
    return rec


def chain_probe(R, hash_name):
    """The compiled inner hash in a dependent chain, >= 10 launches back to back, at the residency of the subtree kernels and at
    twice it; the clock is read, not assumed."""
    hm = HASH_MODEL[hash_name]
    chain = []
    for wps in (4, 8):
        pr = R.zk.probe_hash_chain(hash_name, waves_per_simd=wps, hashes=16 if hash_name == "sha256" else 4, launches=12, device=R.local_rank)
        chain.append({"waves_per_simd": wps, "ns_per_instr": pr["ns_per_hash_per_simd"] / hm["probe_ops"], "clock_ghz": round(pr["clock_ghz"], 3),
                      "cycles_per_instr": pr["ns_per_hash_per_simd"] / hm["probe_ops"] * pr["clock_ghz"], "launches": pr["launches"],
                      "launch_ms": pr["ms"] / pr["launches"]})
    return chain


def soak(R, ctx, N):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < R.args.soak_seconds:
        ctx.prove(); k += 1
    ds = time.perf_counter() - t0
    return {"seconds": round(ds, 2), "proofs": k, "ms_per_proof": ds / k * 1e3, "value": N * k / ds, "unit": "field-elements/s"}


def staged(R, log_n, log_b):
    """The stage-by-stage API once (zk_lde, zk_merkle_commit, zk_compose, zk_fri_fold): the stand-alone
    compose_kernel and fri_fold_kernel, which the one-call prover fuses into leaf hashing, timed with HIP events."""
    zk = R.zk
    with zk.Context(log_n, log_b, device=R.local_rank) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        for rep in range(2):                                # second pass is the measured one
            c.set_profiling("all" if rep else ())
            c.kernel_stats(reset=True)
            c.lde()
            c.merkle_commit(0)
            c.compose([361545003, 3235878091, 2708123352])
            c.merkle_commit(1)
            for r in range(4):
                c.fri_fold(r, 4195595581 + r)
            c.sync()
        st = c.kernel_stats(reset=True)
        c.set_profiling(())
    return {k: st[k] for k in ("compose", "fri_fold", "ntt")}


def lde_commit_2e20(R):
    """BASELINE configs[1]: domain 2^20 LDE + Merkle commit, trace resident -> root on the host AND the whole tree on the device.
    Merkle::new (merkle.rs:14-51) returns a whole tree, so every timed iteration orders the device copy of the host-built tree top
    on the stream (`Context.stream`: zk_ctx_stream settles it; enqueued, not waited for -- what rounds 1-4 did inside the commit).
    `us_root_only` beside it is round 5's figure: the commit returns with the root and leaves that 7 us copy launch to the next
    reader of the device arrays, which a loop of commits of the same layer never has."""
    zk = R.zk
    with zk.Context(17, 3, device=R.local_rank) as c2:
        c2.trace_upload(zk.trace_fibsq((1 << 17) - 1))

        def loop(n, settle):
            t0 = time.perf_counter()
            for _ in range(n):
                c2.lde(); c2.merkle_commit(0)
                if settle:
                    c2.stream                                # noqa: B018  (property: orders the pending tree-top copy on the stream)
            return (time.perf_counter() - t0) / n
        # 0.2 ms of work per iteration: the first ~50 iterations after the idle time of the context setup run 4-5 % slower than
        # the sustained rate (profiles/r04_config2_warmup.txt), so both are reported: `us` = first 50 after 5, as rounds 1-5
        loop(5, True)
        dt_cold = loop(50, True)
        loop(150, True)
        dt_sust = loop(500, True)
        dt_root = loop(500, False)
        c2.sync()
    hm = HASH_MODEL["sha256"]
    floor_us = ((1 << 20) * hm["leaf_ops"] + ((1 << 20) - 1) * hm["inner_ops"]) / (VALU_PEAK_4CYC_TOPS * 1e12) * 1e6
    return {"workload": "configs[1]: domain 2^20 LDE + Merkle commit (root on the host, whole tree on the device)", "us": dt_cold * 1e6,
            "iterations": 50, "warmup_iterations": 5,
            "us_sustained": dt_sust * 1e6, "sustained_iterations": 500, "sustained_warmup_iterations": 205,
            "us_root_only": dt_root * 1e6,
            "us_root_only_note": "sustained, WITHOUT ordering the device copy of the host-built tree top per iteration (round 5's `us_sustained`)",
            "value": (1 << 20) / dt_cold, "value_sustained": (1 << 20) / dt_sust, "unit": "field-elements/s",
            "valu_floor_us": floor_us, "frac_of_valu_floor": floor_us / (dt_cold * 1e6),
            "hbm_floor_us": 73.5 * (1 << 20) / (HBM_PEAK_GBS * 1e9) * 1e6}


def full_2e20(R, chain):
    """The metric's OTHER domain (BASELINE: "at domain 2^20 / 2^24"): one full proof at domain 2^20 (trace group 2^17), trace
    resident -> proof bytes on the host, with what bounds it: the hashing floor at the chain rate, and the latency-bound share --
    prover.rs:198-225 is 17 dependent rounds, each a commitment (a latency launch + a host turn) before the next fold."""
    zk = R.zk
    log_n, log_b = 17, 3
    N = 1 << (log_n + log_b)
    with zk.Context(log_n, log_b, device=R.local_rank) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        for _ in range(5):
            proof = c.prove()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            proof = c.prove()
        dt = (time.perf_counter() - t0) / reps
        proof.verify(strict=True)
        # the same proofs with the next round's launches enqueued before the current commitment is waited for (zk_ctx_set_early_launch:
        # an option, off by default; the bytes must not change)
        early = None
        if c.set_early_launch(True):
            for _ in range(5):
                p2 = c.prove()
            t0 = time.perf_counter()
            for _ in range(reps):
                p2 = c.prove()
            early = {"ms": (time.perf_counter() - t0) / reps * 1e3, "same_proof": p2.data == proof.data and p2.state == proof.state}
        c.set_early_launch(False)
        c.set_profiling("all")
        c.kernel_stats(reset=True)
        c.prove()
        pk = c.kernel_stats(reset=True)
        c.set_profiling(())
        host_levels = list(c.host_levels)
    floor = hashing_floor_ms(pk, chain, include_top=True)
    dev_ms = sum(v["ms"] for v in pk.values())
    top = pk["merkle_top"]
    # one host turn per commitment the device posts: f, cp and every FRI layer above the host tail (2^9 values)
    host_turns = 2 + max(0, (log_n + log_b - 1) - host_levels[1]) if host_levels[0] else 0
    rec = {"workload": f"full prover, domain 2^{log_n + log_b} (trace group 2^{log_n}, blow-up 8): trace resident -> proof bytes on host",
           "ms": dt * 1e3, "iterations": reps, "value": N / dt, "unit": "field-elements/s", "verifies_strict": True,
           "hashing_floor_ms_at_chain_rate": floor, "frac_of_hashing_floor": (floor / (dt * 1e3)) if floor else None,
           "device_ms_all_kernels": dev_ms,
           "latency_launches": {"count": int(top["launches"]), "ms": top["ms"]},
           "throughput_launches": {"count": int(pk["merkle_leaf"]["launches"] + pk["merkle_inner"]["launches"]),
                                   "ms": pk["merkle_leaf"]["ms"] + pk["merkle_inner"]["ms"]},
           "ntt": {"count": int(pk["ntt"]["launches"]), "ms": pk["ntt"]["ms"]},
           "host_turns": {"count": host_turns, "ms": max(0.0, dt * 1e3 - dev_ms),
                          "note": "wall time of a proof minus the summed kernel durations of a profiled proof: PCIe post -> tree top on the host "
                                  "thread -> transcript -> next launch, per commitment, plus the host-side FRI tail and the decommitment"},
           "early_launch": early,
           "note": "latency-bound: 17 dependent rounds (prover.rs:198-225); the device hashes for less than a third of the proof"}
    return rec


def fieldhash_2e24(R, log_n, log_b, trace, chain_field):
    """BASELINE configs[4] in the driver-run line: a few timed proofs at domain 2^24 with the field-native Merkle hash
    (zk_ctx_set_hash(ZK_HASH_FIELD)), the timed proof compared byte for byte (and final channel state) with the oracle's
    independent implementation of the same self-defined hash."""
    import os
    import oracle
    zk = R.zk
    N = 1 << (log_n + log_b)
    with zk.Context(log_n, log_b, device=R.local_rank, hash="field") as c:
        c.trace_upload(trace)
        for _ in range(2):
            proof = c.prove()
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            proof = c.prove()
        dt = (time.perf_counter() - t0) / reps
        c.set_profiling("all")
        c.kernel_stats(reset=True)
        c.prove()
        pk = c.kernel_stats(reset=True)
        c.set_profiling(())
    oracle.set_hash(oracle.HASH_FIELD)
    oracle.set_threads(oracle.usable_cores())
    try:
        t0 = time.perf_counter()
        want = oracle.prove(log_n, log_b, want_vectors=False)
        dt_o = time.perf_counter() - t0
    finally:
        oracle.set_hash(oracle.HASH_SHA256)
    ok = want.rc == 0 and proof.data == want.proof and proof.state == want.state
    traffic = None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_fieldhash.json")
    if os.path.exists(tpath):
        import json
        with open(tpath) as f:
            t = json.load(f)
        traffic = {"merkle_leaf_bytes_per_launch": t.get("merkle_leaf_bytes_per_launch"), "algorithmic_bytes_per_launch": t.get("algorithmic_bytes_per_launch"),
                   "stamp": {k: t.get(k) for k in ("commit", "build_hash", "collected") if k in t},
                   "from_this_build": t.get("build_hash") == R._lib.build_hash()}
    leaf = pk["merkle_leaf"]
    mixp = mix_peak_tops("field")
    return {"workload": f"configs[4]: full prover, domain 2^{log_n + log_b}, Merkle hash = the field-native hash (self-defined; csrc/fieldhash_f64.hpp)",
            "ms": dt * 1e3, "iterations": reps, "value": N / dt, "unit": "field-elements/s",
            "parity": {"against": f"CPU oracle (field hash, independent implementation): every proof byte + channel state of the timed proof "
                                  f"(oracle {dt_o:.1f} s on {oracle.usable_cores()} threads)", "equal": bool(ok)},
            # (no "hashing floor at the chain rate" here: the double-precision hash issues FASTER in the tree kernels, six waves per
            # SIMD, than in the probe's dependent chain at four or eight, so the chain rate is not a floor for this hash)
            "latency_launches": {"count": int(pk["merkle_top"]["launches"]), "ms": pk["merkle_top"]["ms"]},
            "throughput_launches": {"count": int(leaf["launches"] + pk["merkle_inner"]["launches"]), "ms": leaf["ms"] + pk["merkle_inner"]["ms"]},
            "roofline": {"kernel": "merkle_subtree_kernel<leaf, field>", "bound": "valu", "unit": "T lane-ops/s (double precision)",
                         "achieved": (leaf["ops"] / (leaf["ms"] * 1e-3) / 1e12) if leaf["ms"] else None, "peak": mixp,
                         "frac": (leaf["ops"] / (leaf["ms"] * 1e-3) / 1e12 / mixp) if leaf["ms"] else None,
                         "launches": int(leaf["launches"]), "avg_launch_ms": leaf["ms"] / max(leaf["launches"], 1),
                         "algorithmic_bytes_per_launch": leaf["bytes"] / max(leaf["launches"], 1),
                         # HBM bytes per leaf launch from the PMC passes of the FIELD build (profiles/traffic_fieldhash.json, tools/pmc_traffic.py)
                         "traffic": traffic["merkle_leaf_bytes_per_launch"] if traffic else None,
                         "traffic_stamp": traffic["stamp"] if traffic else None,
                         "traffic_from_this_build": bool(traffic and traffic["from_this_build"])},
            "chain": chain_field}


def reference_size(R):
    """configs[0] on the GPU path: the reference's own size (trace 1023, domain 8192), one proof and 1024 in lockstep."""
    zk = R.zk
    with zk.Context(10, 3, device=R.local_rank) as c0:
        c0.trace_upload(zk.trace_fibsq(1023))
        for _ in range(3):
            c0.prove()
        t0 = time.perf_counter()
        for _ in range(20):
            c0.prove()
        dt0 = (time.perf_counter() - t0) / 20
        early_us = None
        if c0.set_early_launch(True):                       # the option of zk_ctx_set_early_launch at the reference's own size
            for _ in range(3):
                c0.prove()
            t0 = time.perf_counter()
            for _ in range(20):
                c0.prove()
            early_us = (time.perf_counter() - t0) / 20 * 1e6
    one = {"workload": "configs[0]: full prover, trace 1023, domain 8192", "us": dt0 * 1e6, "value": 8192 / dt0, "unit": "field-elements/s",
           "us_early_launch": early_us}
    with zk.BatchContext(10, 3, 10, device=R.local_rank) as bc:
        bc.gen_fibsq([1] * 1024, [3141592 + p for p in range(1024)])
        bc.prove_raw()
        t0 = time.perf_counter()
        for _ in range(5):
            bc.prove_raw()
        dtb = (time.perf_counter() - t0) / 5
    batch = {"workload": "configs[0] x 1024: batch of 1024 proofs, trace 1023, domain 8192 each",
             "ms_per_batch": dtb * 1e3, "us_per_proof": dtb * 1e6 / 1024, "value": 1024 * 8192 / dtb, "unit": "field-elements/s"}
    return one, batch


def batched(R, ctx, proof, trace, log_n, log_b, lbt, compare, floor_ms):
    """2^lbt proofs of the benchmark's domain in lockstep (zk_batch_*), then two such batches in flight (one host thread each): the
    latency-bound phases of one batch (16 commitments that wait for the host's challenge, the small FRI layers) overlap the
    hashing of the other.  compare: every proof against zk_prove of its trace (the headline prover, itself oracle-compared)."""
    zk = R.zk
    N = 1 << (log_n + log_b)
    nb = 1 << lbt
    seeds = [3141592 + p_ for p_ in range(nb)]
    reps = 5
    with zk.BatchContext(log_n, log_b, lbt, device=R.local_rank) as bc:
        bc.gen_fibsq([1] * nb, seeds)
        bdata, bstates = bc.prove_raw()
        t0 = time.perf_counter()
        for _ in range(reps):
            bc.prove_raw()
        dtb = (time.perf_counter() - t0) / reps
        bbytes = bc.device_bytes
    rec = {"workload": f"{nb} independent proofs of domain 2^{log_n + log_b} in lockstep (zk_batch_*): traces resident -> all proof bytes on host",
           "proofs": nb, "ms_per_batch": dtb * 1e3, "ms_per_proof": dtb * 1e3 / nb, "value": nb * N / dtb, "unit": "field-elements/s",
           "device_bytes": int(bbytes)}
    if compare:
        same = True
        for p_ in range(nb):
            one = proof if p_ == 0 else ctx.prove(zk.trace_fibsq((1 << log_n) - 1, 1, seeds[p_]))
            same = same and bdata[p_].tobytes() == one.data and bstates[p_].tobytes() == one.state
        ctx.trace_upload(trace)                               # the context goes on with the benchmark's trace
        rec["every_proof_equals_zk_prove"] = bool(same)
    bcs = []
    try:
        for t_ in range(2):
            bc2 = zk.BatchContext(log_n, log_b, lbt, device=R.local_rank)
            bc2.gen_fibsq([1] * nb, [s_ + 16 * t_ for s_ in seeds])
            bc2.prove_raw()
            bcs.append(bc2)

        def work_b(bc_):
            for _ in range(reps):
                bc_.prove_raw()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work_b, args=(bc_,)) for bc_ in bcs]
        [t_.start() for t_ in th]
        [t_.join() for t_ in th]
        dt2b = time.perf_counter() - t0
        rec["two_batches_in_flight"] = {"proofs": 2 * nb, "ms_per_proof": dt2b * 1e3 / (2 * nb * reps), "value": 2 * nb * reps * N / dt2b,
                                        "unit": "field-elements/s", "device_bytes": int(sum(b_.device_bytes for b_ in bcs))}
    finally:
        for bc_ in bcs:
            bc_.close()
    if floor_ms:
        rec["hashing_floor_ms_per_proof_at_chain_rate"] = floor_ms
        rec["frac_of_hashing_floor"] = floor_ms / rec["ms_per_proof"]
        rec["two_batches_in_flight"]["frac_of_hashing_floor"] = floor_ms / rec["two_batches_in_flight"]["ms_per_proof"]
    return rec


def pipelined(R, ctx, trace, log_n, log_b, hash_name):
    """Several independent proofs in flight on one GPU (one context, stream and host thread each), so one proof's
    latency-bound tree tops overlap another's hashing."""
    zk, n = R.zk, R.args.in_flight
    N = 1 << (log_n + log_b)
    ctxs = [ctx] + [zk.Context(log_n, log_b, device=R.local_rank, hash=hash_name) for _ in range(n - 1)]
    for c in ctxs[1:]:
        c.trace_upload(trace)
        c.prove()
    reps = min(R.args.steps, 20)

    def work(c):
        for _ in range(reps):
            c.prove()
    R.barrier()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    [t.start() for t in th]
    [t.join() for t in th]
    R.barrier()
    dtp = time.perf_counter() - t0
    for c in ctxs[1:]:
        c.close()
    return {"proofs_in_flight": n, "value": n * N * reps / dtp, "unit": "field-elements/s", "ms_per_proof": dtp / (n * reps) * 1e3}
