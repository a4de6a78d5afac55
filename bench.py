#!/usr/bin/env python3
"""bench.py -- STARK-101 prover throughput on MI355X.

Metric (BASELINE.json): field-elements/s through LDE + Merkle + FRI = N / t, N = evaluation
domain size, t = wall time from "trace values resident on the device" to "proof bytes on the
host" (context / twiddle setup excluded, reported separately).  Default workload: the full
prover at domain 2^24 (BASELINE.json configs[2]; trace group 2^21, blow-up 8) on synthetic
Fibonacci-square traces.  With --gpus N > 1: ONE proof at domain 2^24 * N sharded over the N
GPUs (weak scaling) by the native sharded prover (zk_shard_*: RCCL all-to-all per commitment).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 21] [--log-blowup 3]

`--gpus N` without a launcher starts the N ranks itself (one child process per GPU, before anything
touches the GPU); under `torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel
(merkle_subtree_kernel<leaf>), timed live with HIP events on the launch stream inside the
timed region; `cpu_baseline` is the CPU oracle (oracle/, a port of the reference algorithm
with O(N log N) transforms) on a bounded sample, rank 0 at N=1 only; `parity_checked` says the
timed proof's bytes were compared with the oracle's proof of the same trace.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
# 32-bit VALU issue model, measured per op at 1/2/4/8 waves per SIMD (tools/valu_microbench.hip,
# profiles/r02_valu_microbench.txt): a wave64 instruction occupies its SIMD for 4 cycles (v_alignbit, v_add3, v_mul_*,
# v_lshl_add) or 2 cycles (v_bitop3, v_add_u32, v_xor, shifts, v_cndmask, v_sub_co; only when several waves share the SIMD).
NOMINAL_GHZ, SIMDS = 2.4, 256 * 4
VALU_PEAK_4CYC_TOPS = SIMDS * 64 * NOMINAL_GHZ * 1e9 / 4 / 1e12          # 39.3 T lane-ops/s if every op took 4 cycles
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
PROFILE_TRAFFIC = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch (stamped with its commit)


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
    return ap.parse_args()


# ---- N > 1 without a launcher: one child per GPU, started before this process touches torch or the GPU ----------
class _Terminated(Exception):
    """SIGTERM / SIGINT reached this process (an outer `timeout`, the launcher stopping the other ranks)."""

    def __init__(self, signum):
        super().__init__(f"signal {signum}")
        self.signum = signum


def _raise_on_signals():
    """SIGTERM and SIGINT raise _Terminated in the main thread, so that `finally` blocks run and children are ended."""
    import signal

    def handler(signum, frame):
        raise _Terminated(signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, handler)


def _end_child(p, grace_s=5.0):
    """Ends exactly the child process `p` (and nothing else): SIGTERM, a grace period, SIGKILL."""
    if p is None or p.poll() is not None:
        return
    p.terminate()
    try:
        p.wait(timeout=grace_s)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()


def spawn_ranks(args):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    _raise_on_signals()                                     # an outer `timeout` ends the ranks too, not only this launcher
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # N ranks on one host: torch's CPU ops (the control plane; the host-staged rehearsal transport) must not start one
            # OpenMP thread per logical CPU each -- torchrun sets 1 for the same reason (rehearsal n2: 3.4 s per step without, 0.09 s with)
            env.setdefault("OMP_NUM_THREADS", "4")
            # rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes to stderr
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))
        rc = 0
        live = set(range(args.gpus))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
                    for q in live:
                        procs[q].terminate()               # exactly the children started above: each is a supervisor that
                                                           # ends its worker and removes its files on SIGTERM
            time.sleep(0.05)
        return rc
    except _Terminated as e:
        print(f"[bench] launcher: {e}: stopping the ranks", file=sys.stderr, flush=True)
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            _end_child(p_, 8.0)
        return 128 + e.signum


# ---- N > 1: every rank is a SUPERVISOR that runs the real work in a fresh child process -----------------------------
# The multi-GPU run is the driver's (one shot, 600 s limit), and two of the things that can go wrong in it cannot be
# handled inside a process: ncclCommInitRank that never returns, and a collective that waits for a peer for ever.
# A hung RCCL call cannot be cancelled, so the worker that runs it is killed by its own watchdog (os._exit(7): a plain
# exit, never an exec) and the supervisor -- which never touches torch or the GPU -- starts a FRESH worker on the next
# rung of the transport ladder.  Budget: every rung has its own deadline, the whole ladder prints a line inside ~300 s.
LADDER = (("native", False), ("native", True), ("torch", True))      # (transport, plain collectives)
RUNG_BUDGET_S = (50.0, 40.0, 40.0)     # rendezvous + communicator(s) + self-test + first verified proof, per rung
if os.environ.get("ZK_BENCH_RUNG_BUDGET_S"):                 # rehearsals shorten the deadlines (tests/test_bench_cli.py)
    RUNG_BUDGET_S = tuple(float(x) for x in os.environ["ZK_BENCH_RUNG_BUDGET_S"].split(","))
RUN_BUDGET_S = 150.0                   # the timed proofs of the headline (after the first proof)
LEG_BUDGET_S = 60.0                    # every secondary leg after the headline (parity, lde_commit, strong leg, configs[3]): a SOFT
                                       # deadline -- the line is printed without a leg that hangs
if os.environ.get("ZK_BENCH_LEG_BUDGET_S"):                  # rehearsals shorten it (tests/test_bench_cli.py)
    LEG_BUDGET_S = float(os.environ["ZK_BENCH_LEG_BUDGET_S"])
RENDEZVOUS_BUDGET_S = 150.0            # gloo rendezvous of the workers: no RCCL in it, but the ranks' first `import torch` on a fresh
                                       # box can finish a minute apart, and a rank that gives up early would split the generations
SHARD_TIMEOUT_S = 20.0                 # zk_shard_options.timeout_s: every host-side wait on a peer inside the library


def _rendezvous_tag():
    return f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"


# Worker generations.  Every supervisor starts a fresh worker whenever its own worker exits, so the ranks must AGREE on
# which generation of workers is meeting (one gloo rendezvous file per generation).  Counting deaths locally is not
# enough -- a rank whose worker dies once more than its peers' (a crash inside the rendezvous) would wait in a store the
# others never open -- so the generation lives in one shared file per run: a supervisor that starts a worker takes
# max(shared, its own last + 1) under a lock, and a worker waiting for its peers leaves (exit code 8) as soon as the
# shared number has moved past its own.  Whoever opens a new generation removes what an earlier, killed run may have
# left under the same names.
STALE_GENERATION = 8


def _gen_paths(tag, gen, world):
    return f"/tmp/zkbench_store_{tag}_{gen}", [f"/tmp/zkbench_join_{tag}_{gen}_{r}" for r in range(world)]


def _bump_generation(tag, at_least, world):
    import fcntl
    path = f"/tmp/zkbench_gen_{tag}"
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            with open(path) as f:
                cur = int(f.read().strip())
        except (OSError, ValueError):
            cur = -1
        if at_least > cur:                                   # this supervisor opens a new generation
            store, joins = _gen_paths(tag, at_least, world)
            for stale in [store] + joins:
                try:
                    os.unlink(stale)
                except OSError:
                    pass
            with open(path + ".tmp", "w") as f:
                f.write(str(at_least))
            os.replace(path + ".tmp", path)
            cur = at_least
        return cur


def _current_generation(tag):
    try:
        with open(f"/tmp/zkbench_gen_{tag}") as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def join_generation(tag, gen, rank, world, budget_s):
    """Worker side, before the gloo rendezvous: wait until every rank's worker of THIS generation is here.  Returns
    'ok', 'stale' (the run has moved on to a later generation: leave at once) or 'timeout'."""
    _, joins = _gen_paths(tag, gen, world)
    with open(joins[rank], "w") as f:
        f.write(str(os.getpid()))
    t0 = time.time()
    while True:
        if all(os.path.exists(j) for j in joins):
            return "ok"
        if _current_generation(tag) > gen:
            return "stale"
        if time.time() - t0 > budget_s:
            return "timeout"
        time.sleep(0.05)


def supervise():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    tag = _rendezvous_tag()
    status = f"/tmp/zkbench_status_{tag}_{rank}"
    first = os.environ.get("ZK_BENCH_TRANSPORT", "native")
    rung = {"native": 0, "torch": 2}.get(first, 0)
    if os.environ.get("ZK_BENCH_STAGED") == "1":
        rung = 0                                             # one rung only: the host-staged rehearsal transport
    t_start = time.time()
    attempt = 0                                              # workers of this rank that ran a rung (ladder progress)
    stale_restarts = 0
    gen = 0
    gens_used = set()
    code = 1
    p = None
    _raise_on_signals()                                      # SIGTERM (the launcher, an outer `timeout`) ends the worker too
    try:
        while attempt < len(LADDER) + 1:
            gen = _bump_generation(tag, gen, world)
            gens_used.add(gen)
            store, _ = _gen_paths(tag, gen, world)
            env = dict(os.environ, ZK_BENCH_WORKER="1", ZK_BENCH_RUNG=str(rung), ZK_BENCH_ATTEMPT=str(attempt), ZK_BENCH_STATUS=status,
                       ZK_BENCH_STORE=store, ZK_BENCH_GEN=str(gen), ZK_BENCH_TAG=tag, ZK_BENCH_T0=repr(t_start))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            try:
                os.unlink(status)
            except OSError:
                pass
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)   # stdout inherited: the worker prints the line
            code = p.wait()
            p = None
            if code == 0:
                return 0
            if code == STALE_GENERATION and stale_restarts < 8:
                # the peers had already moved on when this worker arrived: join them, the ladder does not advance
                stale_restarts += 1
                gen = max(gen + 1, _current_generation(tag))
                continue
            try:
                with open(status) as f:
                    last = int(f.read().strip())
            except (OSError, ValueError):
                last = rung - 1                               # died before it could say anything: the same rung again
            # 3: not enough GPUs, 4: a proof that differs (parity): nothing another transport would change
            if code in (3, 4) or os.environ.get("ZK_BENCH_STAGED") == "1" or last + 1 >= len(LADDER):
                return code if code > 0 else 1
            where = f"rung {last} ({LADDER[last][0]}{' + plain' if LADDER[last][1] else ''})" if last >= 0 else "the rendezvous"
            print(f"[bench] rank {rank}: worker exited with {code} on {where} after "
                  f"{time.time() - t_start:.0f} s; starting a fresh worker on rung {last + 1}", file=sys.stderr, flush=True)
            rung = last + 1
            attempt += 1
            gen += 1
        return code if code > 0 else 1
    except _Terminated as e:
        print(f"[bench] rank {rank}: supervisor: {e}: ending the worker", file=sys.stderr, flush=True)
        return 128 + e.signum
    finally:
        _end_child(p)                                        # never leave the process that holds the GPU behind
        mine = [status, status + ".tmp"]
        for g in gens_used:
            store, joins = _gen_paths(tag, g, world)
            mine.append(joins[rank])
            if rank == 0:
                mine.append(store)
        if rank == 0:
            mine += [f"/tmp/zkbench_gen_{tag}", f"/tmp/zkbench_gen_{tag}.lock", f"/tmp/zkbench_gen_{tag}.tmp"]
        for path in mine:
            try:
                os.unlink(path)
            except OSError:
                pass


class Watchdog:
    """Ends the process when an armed deadline passes: the only way out of an RCCL call that never returns.  A HARD deadline
    (the transport ladder: communicators, self-test, first proof) exits with code 7 and the supervisor starts a fresh worker
    on the next rung.  A SOFT deadline guards a secondary leg that runs after the headline has been measured: `on_late` prints
    the line with what has been measured so far (rank 0), then every rank exits with code 0 -- a leg that hangs costs that
    leg, never the measurement."""

    def __init__(self, rank):
        import threading
        self.rank, self.deadline, self.what, self.on_late = rank, None, "", None
        self._lock = threading.Lock()
        threading.Thread(target=self._run, daemon=True).start()

    def arm(self, seconds, what, on_late=None):
        with self._lock:
            self.deadline, self.what, self.on_late = time.time() + seconds, what, on_late

    def disarm(self):
        with self._lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                late = self.deadline is not None and time.time() > self.deadline
                what, on_late = self.what, self.on_late
            if late and on_late is not None:
                print(f"[bench] rank {self.rank}: WATCHDOG: secondary leg '{what}' did not finish in time; the line is printed without it",
                      file=sys.stderr, flush=True)
                code = 0
                try:
                    code = on_late(what) or 0             # e.g. 4 when the headline proof had already failed its parity check
                finally:
                    os._exit(code)
            if late:
                print(f"[bench] rank {self.rank}: WATCHDOG: '{what}' did not finish in time; this worker exits (7) and the supervisor "
                      f"starts a fresh one on the next rung", file=sys.stderr, flush=True)
                os._exit(7)


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


def staged_leg(zk, log_n, log_b, device):
    """The stage-by-stage API once (zk_lde, zk_merkle_commit, zk_compose, zk_fri_fold): the stand-alone
    compose_kernel and fri_fold_kernel, which the one-call prover fuses into leaf hashing, timed with HIP events."""
    with zk.Context(log_n, log_b, device=device) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        ch = zk.Channel()
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


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                         # nothing here has touched torch or the GPU yet
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("ZK_BENCH_FORCE_SHARDED") == "1"
    if multi and not args.staged_only and os.environ.get("ZK_BENCH_WORKER") != "1":
        sys.exit(supervise())                               # this process stays clear of torch and the GPU; the work runs in a child
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("ZK_BENCH_TEST_WORKER_SLEEP"):         # tests/test_bench_cli.py: a worker that is busy when SIGTERM arrives above it
        time.sleep(float(os.environ["ZK_BENCH_TEST_WORKER_SLEEP"]))
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
    if staged:
        local_rank = 0
    ndev = torch.cuda.device_count()
    if ndev <= local_rank:
        print(f"[bench] rank {rank}: --gpus {world} needs {world} GPUs on this node, {ndev} visible", file=sys.stderr, flush=True)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    wd = None
    if sharded_run:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        wd = Watchdog(rank)
        start_rung = int(os.environ.get("ZK_BENCH_RUNG", "0"))
        if os.environ.get("ZK_BENCH_STATUS"):                 # a worker that dies before its first rung is retried on the SAME rung
            with open(os.environ["ZK_BENCH_STATUS"], "w") as f:
                f.write(str(start_rung - 1))
        wd.arm(RENDEZVOUS_BUDGET_S, "rendezvous of the control plane (gloo)")
        # control plane only (unique id broadcast, agreement rounds, max over ranks of the time): gloo on the host, one
        # rendezvous file per worker generation (no port to collide with a previous generation's).  The data path is RCCL
        # inside the library (zk_shard_*: grouped ncclSend/ncclRecv all-to-all, ncclAllGather).
        store = os.environ.get("ZK_BENCH_STORE")
        if store:
            # every rank's worker of THIS generation is here before the rendezvous file is touched; a worker whose peers have
            # moved on to a later generation leaves at once and its supervisor joins them (supervise())
            how = join_generation(os.environ["ZK_BENCH_TAG"], int(os.environ.get("ZK_BENCH_GEN", "0")), rank, world, RENDEZVOUS_BUDGET_S - 5.0)
            if how == "stale":
                print(f"[bench] rank {rank}: the other ranks are already in a later worker generation; rejoining", file=sys.stderr, flush=True)
                os._exit(STALE_GENERATION)
            if how == "timeout":
                print(f"[bench] rank {rank}: the other ranks' workers did not arrive within {RENDEZVOUS_BUDGET_S:.0f} s", file=sys.stderr, flush=True)
                os._exit(7)
            dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=RENDEZVOUS_BUDGET_S))
        else:
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=90))
        dist.barrier()                                        # every rank is here: the rung deadlines start together

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    log_n, log_b = args.log_n, args.log_blowup
    lib = _lib.load()

    def dev_stats():
        arr = _lib.kernel_stat_array()
        _lib.check(lib.zk_dev_kernel_stats(arr, len(arr), 1))
        return {name: {"launches": int(a.launches), "ms": a.ms, "bytes": a.bytes, "ops": a.ops} for name, a in zip(_lib.KERNEL_CLASSES, arr)}

    result = {}
    oracle_proof = None
    if args.staged_only:
        out = {"staged": staged_leg(zk, log_n, log_b, local_rank)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        return
    emitted = []

    def emit_line(result):
        """Rank 0: builds and prints THE one JSON line from what `result` holds (at the end of the run, or -- N > 1 -- from the
        watchdog when a secondary leg hangs after the headline was measured)."""
        if emitted:
            return
        emitted.append(True)
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
        for k in ["device_only", "pipelined", "soak", "lde_commit_2e20", "reference_size_2e13", "batched_2e13", "lde_commit_sharded", "config4_2e26", "shard",
                  "transport", "transport_note", "ladder"] + sorted(k_ for k_ in result if k_.startswith(("strong_2e", "batched_2e2"))):
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

    if sharded_run:
        lg = world.bit_length() - 1
        # weak scaling: per-GPU work equals the single-GPU workload; strong: the single-GPU domain over all GPUs
        log_n = args.log_n + (lg if args.scaling == "weak" else 0)
        from zkstark_amd import sharded
        trace = zk.trace_fibsq((1 << log_n) - 1)

        def all_ok(ok):                                       # do all ranks agree that a step worked? (gloo, host)
            t_ = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(t_, op=dist.ReduceOp.MIN)
            return bool(t_.item())

        def shared_from_rank0(make):
            """One object made on rank 0 and handed to every rank; a failure on rank 0 reaches every rank as the SAME ZkError
            (round 3 let rank 0 raise before the broadcast, leaving the others blocked in it)."""
            box = [None]
            if rank == 0:
                try:
                    box[0] = ("ok", make())
                except zk.ZkError as e:
                    box[0] = ("err", e.code, str(e))
            dist.broadcast_object_list(box, src=0)
            if box[0][0] == "err":
                raise zk.ZkError(box[0][1], f"rank 0: {box[0][2]}")
            return box[0][1]

        def shard_ctx(kind, plain, log_n_, transport_):
            if kind == "native" and os.environ.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "id":      # rehearsal: rank 0 cannot load RCCL
                uid_ = shared_from_rank0(lambda: (_ for _ in ()).throw(zk.ZkError(-4, "simulated: RCCL is not available (ZK_BENCH_SIMULATE_NATIVE_FAILURE=id)")))
            elif kind == "native":
                uid_ = shared_from_rank0(zk.shard_unique_id)
            else:
                uid_ = shared_from_rank0(lambda: os.urandom(128))    # names the shared-memory root board only
            return zk.ShardContext(log_n_, log_b, rank, world, uid_, device=local_rank, transport=transport_, force_collectives=force_sharded,
                                   plain_collectives=plain, timeout_s=SHARD_TIMEOUT_S)

        def make_prover(kind, plain):
            """kind: 'native' = RCCL loaded by the library (ncclCommInitRank inside zk_shard_create); 'torch' = the same
            collectives through torch.distributed's own RCCL communicator; 'staged' = host-staged (rehearsal on one GPU).
            zk_shard_create ends with the known-pattern self-test of the transport (all-to-all on every stream in use, all-gather)."""
            tp_ = None
            if kind == "staged":
                tp_ = sharded.staged_transport()
            elif kind == "torch":
                tp_ = sharded.device_transport(dist.new_group(backend="nccl"))
            sp_ = shard_ctx(kind, plain, log_n, tp_)
            sp_.trace_upload(trace)
            if kind == "native" and os.environ.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "1":   # rehearsal of the fallback
                sp_.inject_failure()
                sp_.close()
                raise zk.ZkError(-2, "simulated failure of the native transport (ZK_BENCH_SIMULATE_NATIVE_FAILURE)")
            if kind == "native" and os.environ.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "hang":  # rehearsal of the watchdog
                time.sleep(3600)
            return sp_, tp_, sp_.prove()                       # the first proof is part of "does this transport work"

        # The first proof must come into being AND be a valid proof on every rank (strict verifier: transcript replay +
        # every opening).  If not, the line says what failed and the run goes down a fixed ladder -- never silently:
        #   native RCCL, chunked exchange + root board  ->  native RCCL, plain collectives (one all-to-all per layer on the
        #   main stream, subtree roots by all-gather)  ->  the same plain collectives through torch.distributed's own RCCL
        #   communicator (sharded.device_transport): a second, independent way to the same wire.
        # An ERROR moves to the next rung inside this process; a HANG is ended by the watchdog and the supervisor starts a
        # fresh worker on the next rung (every rung has its own deadline, RUNG_BUDGET_S).
        # One multi-GPU run is all this code gets (the driver's); everything a one-GPU box can rehearse of it is rehearsed.
        def proof_valid(p_):
            try:
                p_.verify(strict=True)
                return True
            except zk.ZkError as e:
                print(f"[bench] rank {rank}: the first proof does not verify: {e}", file=sys.stderr, flush=True)
                return False

        status_path = os.environ.get("ZK_BENCH_STATUS")

        def note_rung(i):
            if status_path:
                with open(status_path + ".tmp", "w") as f:
                    f.write(str(i))
                os.replace(status_path + ".tmp", status_path)

        if staged:
            ladder = [("staged", bool(args.plain_collectives))]
            first_rung = 0
        else:
            ladder = list(LADDER)
            if args.plain_collectives:
                ladder = [(k, True) for k, _ in ladder]
            # every rank starts where the furthest rank starts (a supervisor that saw its worker die later than the others)
            t_ = torch.tensor([int(os.environ.get("ZK_BENCH_RUNG", "0"))], dtype=torch.int32)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            first_rung = int(t_.item())
        attempt_no = int(os.environ.get("ZK_BENCH_ATTEMPT", "0"))
        notes = [f"fresh worker #{attempt_no}: an earlier worker hung or died on a rung before {first_rung}"] if attempt_no else []
        sp = transport = proof = None
        kind, plain = ladder[min(first_rung, len(ladder) - 1)]
        rung_used = None
        for ri in range(first_rung, len(ladder)):
            kind, plain = ladder[ri]
            note_rung(ri)
            wd.arm(RUNG_BUDGET_S[min(ri, len(RUNG_BUDGET_S) - 1)], f"rung {ri}: {kind}{' + plain collectives' if plain else ''} "
                   "(communicators, self-test, first verified proof)")
            err = None
            try:
                sp, transport, proof = make_prover(kind, plain)
                if not proof_valid(proof):
                    err = "the first proof does not verify"
            except zk.ZkError as e:
                err = str(e)
                print(f"[bench] rank {rank}: {kind} transport{' (plain collectives)' if plain else ''} failed: {err}", file=sys.stderr, flush=True)
            if all_ok(err is None):
                rung_used = ri
                break
            notes.append(f"{kind}{' + plain collectives' if plain else ''} failed ({err or 'on another rank'})")
            if sp is not None:
                sp.inject_failure()                           # abort, do not destroy, a communicator that may be half-formed
                sp.close()
            sp = transport = proof = None
        if sp is None:
            sys.exit(5)
        args.plain_collectives = plain
        wd.arm(RUN_BUDGET_S, "the timed proofs of the headline")
        transport_note = ("FALLBACK: " + "; ".join(notes) + f"; running on {kind}{' + plain collectives' if plain else ''}") if notes else None

        def measure(sp_, log_n_, steps_, warm_):
            """`steps_` timed proofs on prover `sp_` (max over the ranks), then two untimed ones: every kernel class bracketed
            with HIP events, and every exchange (zk_shard_set_profiling) -- how long the collectives take on their streams, how
            much of that the hashing streams wait for, and the replicated tail, per rank, so that a bad scaling figure from the
            one multi-GPU run can be read: links, overlap or tail."""
            p_ = None
            for _ in range(warm_):
                p_ = sp_.prove()
            _lib.check(lib.zk_dev_set_profiling(1 << _lib.KERNEL_CLASSES.index("merkle_leaf")))   # dominant kernel only
            dev_stats()
            barrier()
            t0_ = time.perf_counter()
            for _ in range(steps_):
                p_ = sp_.prove()
            dt_local_ = time.perf_counter() - t0_             # this rank's own time (before the closing barrier)
            barrier()
            t_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)         # the slowest rank's time
            dom_ = dev_stats()["merkle_leaf"]
            st_ = sp_.stats()
            _lib.check(lib.zk_dev_set_profiling((1 << len(_lib.KERNEL_CLASSES)) - 1))
            sp_.prove()
            per_kernel_ = dev_stats()
            _lib.check(lib.zk_dev_set_profiling(0))
            sp_.set_profiling(True)
            sp_.prove()
            stx = sp_.stats()
            sp_.set_profiling(False)
            mine = {"rank": rank, "ms_per_step_local": dt_local_ / steps_ * 1e3, "sent_bytes": stx["sent_bytes"], "all_to_all_bytes": stx["all_to_all_bytes"],
                    "exchange_ms": stx["exchange_ms"], "exposed_exchange_ms": stx["exposed_exchange_ms"], "tail_ms": stx["tail_ms"],
                    "decommit_ms": stx["decommit_ms"],
                    "exchanges": stx["exchanges"], "chunked_layers": stx["chunked_layers"], "rccl_nranks": stx["rccl_nranks"],
                    "communicators": stx["communicators"], "selftest_ok": stx["selftest_ok"], "selftest_ms": stx["selftest_ms"]}
            per_rank_ = [None] * world
            dist.all_gather_object(per_rank_, mine)
            plan_ = zk.shard_plan(world, log_n_, log_b, force_collectives=force_sharded, plain_collectives=plain)
            gathered_ = [None] * world
            dist.all_gather_object(gathered_, p_.data[:64] + p_.state)
            return {"proof": p_, "dt": float(t_.item()), "steps": steps_, "dom": dom_, "per_kernel": per_kernel_, "st": st_, "per_rank": per_rank_,
                    "plan": {k: plan_[k] for k in ("sharded_layers", "tail_rounds", "chunked_layers", "chunked_mask", "log_chunks", "overlap_min_log",
                                                   "min_layer_log", "min_chunk_log", "piece_log", "all_to_all_bytes", "cp_from_f")},
                    "chunked_mask": plan_["chunked_mask"], "ranks_agree": all(g == gathered_[0] for g in gathered_)}

        def shard_record(m_, n_elems):
            st_, per_rank_ = m_["st"], m_["per_rank"]
            return {**st_, "sent_bytes_per_proof_per_rank": st_["sent_bytes"], "ranks_agree": m_["ranks_agree"],
                    "exchanged_bytes_per_element": st_["all_to_all_bytes"] * world / n_elems if world > 1 else 0.0,
                    # from the profiled proof (max over ranks; per_rank has every rank's own figures)
                    "exchange_ms": max(r["exchange_ms"] for r in per_rank_), "exposed_exchange_ms": max(r["exposed_exchange_ms"] for r in per_rank_),
                    "tail_ms": max(r["tail_ms"] for r in per_rank_), "decommit_ms": max(r["decommit_ms"] for r in per_rank_),
                    "selftest_ok": all(r["selftest_ok"] for r in per_rank_),
                    "per_rank": per_rank_, "timeout_s": SHARD_TIMEOUT_S, "plan": m_["plan"]}

        def single_gpu_parity(log_n_, trace_, proof_, time_it=0):
            """Rank 0: the same trace on the single-GPU prover (itself pinned on the CPU oracle by the tests and by the N = 1
            line): every byte and the final channel state must be equal.  time_it > 0: also that prover's ms per proof."""
            if rank != 0:
                return None
            try:
                with zk.Context(log_n_, log_b, device=local_rank) as c1:
                    one = c1.prove(trace_)
                    rec = {"against": f"single-GPU prover at domain 2^{log_n_ + log_b} (oracle-pinned)", "equal": one.data == proof_.data and one.state == proof_.state}
                    if time_it:
                        t0_ = time.perf_counter()
                        for _ in range(time_it):
                            c1.prove()
                        rec["single_gpu_ms"] = (time.perf_counter() - t0_) / time_it * 1e3
                return rec
            except zk.ZkError as e:
                return {"against": "single-GPU prover", "equal": None, "skipped": str(e)}

        m = measure(sp, log_n, args.steps, max(args.warmup - 1, 0))
        proof = m["proof"]
        N = 1 << (log_n + log_b)
        st = m["st"]
        result = {"dt": m["dt"], "dom": m["dom"], "per_kernel": m["per_kernel"], "setup_ms": st["setup_ms"], "device_bytes": int(st["device_bytes"]),
                  "proof_bytes": len(proof.data), "scaling": args.scaling, "units": N * args.steps,
                  "parallelism": {"native": f"one proof sharded over {world} GPUs (cyclic domain; native RCCL all-to-all per commitment)",
                                  "torch": f"one proof sharded over {world} GPUs (cyclic domain; RCCL all-to-all per commitment through torch.distributed)",
                                  "staged": f"REHEARSAL: {world} ranks on one GPU, host-staged collectives"}[kind],
                  "transport": kind, "transport_note": transport_note,
                  "shard": shard_record(m, N),
                  "ladder": {"rung": rung_used, "transport": kind, "plain_collectives": plain, "worker": attempt_no, "notes": notes,
                             "seconds_since_supervisor_start": (time.time() - float(os.environ["ZK_BENCH_T0"])) if os.environ.get("ZK_BENCH_T0") else None,
                             "rung_budget_s": list(RUNG_BUDGET_S), "run_budget_s": RUN_BUDGET_S, "leg_budget_s": LEG_BUDGET_S},
                  "parity": None, "legs_skipped": [], "proof": proof}
        # From here on the headline exists.  Every further leg runs under a SOFT deadline: if it hangs (a collective that
        # never returns), rank 0 prints the line with what has been measured and every rank exits 0.
        def soft(what):
            def late(_what):
                result["legs_skipped"].append(f"{what}: did not finish within {LEG_BUDGET_S:.0f} s (watchdog); this leg and the later ones are missing")
                if rank == 0:
                    emit_line(result)
                return result.get("exit_code") or 0
            wd.arm(LEG_BUDGET_S, what, on_late=late)

        # parity: every rank's bytes must equal the single-GPU prover's
        soft("parity of the headline proof against the single-GPU prover")
        if rank == 0:
            proof.verify(strict=True)
        result["parity"] = single_gpu_parity(log_n, trace, proof)
        barrier()
        def time_lde_commit(ctx_, reps=10):
            root0 = ctx_.lde_commit()
            barrier()
            t0_ = time.perf_counter()
            for _ in range(reps):
                root1 = ctx_.lde_commit()
            barrier()
            dtl_ = torch.tensor([(time.perf_counter() - t0_) / reps], dtype=torch.float64)
            dist.all_reduce(dtl_, op=dist.ReduceOp.MAX)
            return float(dtl_.item()), root0 == root1, ctx_.stats()["all_to_all_bytes"], root1
        if not args.no_secondary:                            # BASELINE.json configs[3] shape at the prover's own domain
            soft("lde_commit_sharded")
            try:
                dtl, stable, a2a, root_c = time_lde_commit(sp)
            except zk.ZkError as e:                           # recorded; the prover is spent, the later legs make their own
                dtl = None
                result["lde_commit_sharded"] = {"error": str(e)}
        if not args.no_secondary and dtl is not None:
            lde_commit = {"workload": f"configs[3] shape: sharded LDE + all-to-all transpose + Merkle commit, domain 2^{log_n + log_b} over {world} GPUs",
                          "ms": dtl * 1e3, "value": N / dtl, "unit": "field-elements/s", "root_stable": stable,
                          "all_to_all_bytes_per_rank": a2a, "chunked": bool(m["chunked_mask"] & 1)}
            if m["chunked_mask"] & 1:
                # the same commitment with PLAIN collectives (one all-to-all on the main stream, no overlap with the hashing): the
                # A/B of the chunked exchange on the very links of this run, not inferred from one-GPU rehearsals
                try:
                    with shard_ctx(kind, True, log_n, transport) as spp:
                        spp.trace_upload(trace)
                        dtp, stable_p, _, root_p = time_lde_commit(spp)
                    lde_commit["plain_ab"] = {"ms": dtp * 1e3, "root_equal": root_p == root_c, "chunked_over_plain": dtl / dtp}
                except zk.ZkError as e:
                    lde_commit["plain_ab"] = {"error": str(e)}
            result["lde_commit_sharded"] = lde_commit
        sp.close()
        # STRONG scaling at the metric's own domain (BASELINE: "at domain 2^20 / 2^24; 1/2/4/8-GPU scaling"): the single-GPU
        # workload -- one 2^(log_n + blow-up) proof -- sharded over the N ranks of this run, beside the weak-scaling headline
        strong_leg = (world > 1 or os.environ.get("ZK_BENCH_STRONG_LEG") == "1") and args.scaling == "weak" and not args.no_secondary
        if strong_leg:
            sl = args.log_n
            key = f"strong_2e{sl + log_b}"
            soft(key)
            if os.environ.get("ZK_BENCH_SIMULATE_LEG_HANG") == "strong":     # rehearsal: a collective of this leg never returns
                time.sleep(3600)
            try:
                tr_s = trace if sl == log_n else zk.trace_fibsq((1 << sl) - 1)
                with shard_ctx(kind, plain, sl, transport) as sps:
                    sps.trace_upload(tr_s)
                    first_s = sps.prove()
                    ok_first = True
                    try:
                        first_s.verify(strict=True)
                    except zk.ZkError:
                        ok_first = False
                    ms_ = measure(sps, sl, min(args.steps, 20), 2)
                par_s = single_gpu_parity(sl, tr_s, ms_["proof"], time_it=5)
                Ns = 1 << (sl + log_b)
                rec = {"workload": f"full prover, domain 2^{sl + log_b} (the single-GPU workload) sharded over {world} GPUs: strong scaling",
                       "ms": ms_["dt"] / ms_["steps"] * 1e3, "steps": ms_["steps"], "value": Ns * ms_["steps"] / ms_["dt"], "unit": "field-elements/s",
                       "first_proof_verifies": ok_first, "parity": par_s, "ranks_agree": ms_["ranks_agree"],
                       "shard": shard_record(ms_, Ns)}
                if par_s and par_s.get("single_gpu_ms"):
                    rec["single_gpu_ms"] = par_s["single_gpu_ms"]
                    rec["speedup_over_single_gpu"] = par_s["single_gpu_ms"] / rec["ms"]
                result[key] = rec
                if par_s and par_s.get("equal") is False:
                    result["parity"] = {**(result["parity"] or {}), "equal": False, "strong_leg_differs": True}
            except zk.ZkError as e:
                result[key] = {"error": str(e)}
            barrier()
        if not args.no_secondary and world in (2, 4, 8) and log_b == 3:
            # BASELINE.json configs[3] at EXACTLY its size: domain 2^26 (trace group 2^23) over the N GPUs of this run
            soft("config4_2e26")
            try:                                              # an ERROR in a secondary leg is recorded, never fatal to the line
                with shard_ctx(kind, plain, 23, transport) as sp4:
                    sp4.trace_upload(zk.trace_fibsq((1 << 23) - 1))
                    dtl, stable, a2a, root4 = time_lde_commit(sp4)
                    st4 = sp4.stats()
                golden = None
                try:                                          # tests/golden/config4_2e26.json: the CPU oracle's root (orc.lde + orc.merkle_build)
                    with open(os.path.join(ROOT, "tests", "golden", "config4_2e26.json")) as f:
                        golden = json.load(f)["pinned"]["f_eval_root"]
                except (OSError, KeyError, ValueError):
                    pass
                result["config4_2e26"] = {"workload": f"configs[3]: domain 2^26 NTT (LDE) sharded over {world} GPUs, all-to-all transpose, Merkle commit",
                                          "ms": dtl * 1e3, "value": (1 << 26) / dtl, "unit": "field-elements/s", "root_stable": stable,
                                          "all_to_all_bytes_per_rank": a2a, "rccl_nranks": st4["rccl_nranks"], "chunked_layers": st4["chunked_layers"],
                                          "root": root4.hex(), "root_matches_golden": (root4.hex() == golden) if golden else None}
            except zk.ZkError as e:
                result["config4_2e26"] = {"error": str(e)}
        soft("closing barrier")
        barrier()
        wd.disarm()                                           # the last step that waits for a peer: from here on the line WILL be printed once
        parity = result["parity"]
        if parity and parity.get("equal") is False:
            print("[bench] sharded proof differs from the single-GPU prover", file=sys.stderr, flush=True)
            sys.exit(4)
        if not result["shard"]["ranks_agree"]:
            print("[bench] the ranks disagree on the proof", file=sys.stderr, flush=True)
            sys.exit(4)
    else:
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
                  "host_hashing": zk.host_hash_mode(), "proof": proof}
        # the same proofs with everything on the device (host_levels (0, 0): no tree tops, no FRI tail on the host thread)
        if args.hash == "sha256" and tuple(ctx.host_levels) != (0, 0):
            keep = tuple(ctx.host_levels)
            ctx.set_host_levels(0, 0)
            for _ in range(2):
                dproof = ctx.prove()
            reps = min(args.steps, 20)
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                dproof = ctx.prove()
            barrier()
            result["device_only"] = {"ms_per_step": (time.perf_counter() - t0) / reps * 1e3, "steps": reps, "host_levels": [0, 0],
                                     "same_proof": dproof.data == proof.data and dproof.state == proof.state}
            ctx.set_host_levels(*keep)
        # roofline probe: the compiled inner hash in a dependent chain, >= 10 launches back to back, at the residency of
        # the subtree kernels (SHA-256: 8 waves per SIMD) and at half of it; the clock is read, not assumed
        hm = HASH_MODEL[args.hash]
        chain = []
        for wps in (4, 8):
            pr = zk.probe_hash_chain(args.hash, waves_per_simd=wps, hashes=16 if args.hash == "sha256" else 4, launches=12, device=local_rank)
            chain.append({"waves_per_simd": wps, "ns_per_instr": pr["ns_per_hash_per_simd"] / hm["probe_ops"], "clock_ghz": round(pr["clock_ghz"], 3),
                          "cycles_per_instr": pr["ns_per_hash_per_simd"] / hm["probe_ops"] * pr["clock_ghz"], "launches": pr["launches"],
                          "launch_ms": pr["ms"] / pr["launches"]})
        result["chain"] = chain
        # soak: keep the device busy for a few seconds (driver-side sampling sees it; steady-state figure)
        if args.soak_seconds > 0:
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < args.soak_seconds:
                ctx.prove(); k += 1
            ds = time.perf_counter() - t0
            result["soak"] = {"seconds": round(ds, 2), "proofs": k, "ms_per_proof": ds / k * 1e3, "value": N * k / ds, "unit": "field-elements/s"}
        if args.hash == "sha256" and not args.no_secondary:
            # stand-alone compose / fold kernels (fused into leaf hashing in the timed path)
            result["staged"] = staged_leg(zk, log_n, log_b, local_rank)
            # secondary figure: BASELINE.json configs[1], domain 2^20 LDE + Merkle commit (trace resident -> root on host)
            with zk.Context(17, 3, device=local_rank) as c2:
                c2.trace_upload(zk.trace_fibsq((1 << 17) - 1))
                # 0.2 ms of work per iteration: the first ~50 iterations after the idle time of the context setup run 4-5 %
                # slower than the sustained rate (195 against 187 us, profiles/r04_config2_warmup.txt), so both are reported
                for _ in range(5):
                    c2.lde(); c2.merkle_commit(0)
                t0 = time.perf_counter()
                for _ in range(50):
                    c2.lde(); c2.merkle_commit(0)
                dt2_cold = (time.perf_counter() - t0) / 50
                for _ in range(150):
                    c2.lde(); c2.merkle_commit(0)
                t0 = time.perf_counter()
                for _ in range(500):
                    c2.lde(); c2.merkle_commit(0)
                dt2 = (time.perf_counter() - t0) / 500
            floor_us = ((1 << 20) * HASH_MODEL['sha256']['leaf_ops'] + ((1 << 20) - 1) * HASH_MODEL['sha256']['inner_ops']) / (VALU_PEAK_4CYC_TOPS * 1e12) * 1e6
            # ONE method from round 5 on: `us` / `value` are the first 50 iterations after 5 warm-up ones (what rounds 1-3 reported,
            # and what a caller that commits once in a while sees); the sustained rate (500 iterations after 205) is beside it
            result["lde_commit_2e20"] = {"workload": "configs[1]: domain 2^20 LDE + Merkle commit", "us": dt2_cold * 1e6,
                                         "iterations": 50, "warmup_iterations": 5,
                                         "us_sustained": dt2 * 1e6, "sustained_iterations": 500, "sustained_warmup_iterations": 205,
                                         "value": (1 << 20) / dt2_cold, "value_sustained": (1 << 20) / dt2, "unit": "field-elements/s",
                                         "valu_floor_us": floor_us, "frac_of_valu_floor": floor_us / (dt2_cold * 1e6),
                                         "hbm_floor_us": 73.5 * (1 << 20) / (HBM_PEAK_GBS * 1e9) * 1e6}
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
            # throughput mode at the metric's own domain: 2^log_batch independent 2^24 proofs in lockstep (zk_batch_*), every proof
            # compared byte for byte with zk_prove of the same trace (the single prover above, itself compared with the oracle)
            if log_n + log_b <= 24:
                import threading

                def batched_leg(lbt, compare):
                    """2^lbt proofs in lockstep, then two such batches in flight (one host thread each): the latency-bound phases of one
                    batch (16 commitments that wait for the host's challenge, the small FRI layers) overlap the hashing of the other."""
                    nb = 1 << lbt
                    seeds = [3141592 + p_ for p_ in range(nb)]
                    reps = 5
                    with zk.BatchContext(log_n, log_b, lbt, device=local_rank) as bc:
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
                    if compare:                               # proof p against the single prover on trace p
                        same = True
                        for p_ in range(nb):
                            one = proof if p_ == 0 else ctx.prove(zk.trace_fibsq((1 << log_n) - 1, 1, seeds[p_]))
                            same = same and bdata[p_].tobytes() == one.data and bstates[p_].tobytes() == one.state
                        ctx.trace_upload(trace)               # the context goes on with the benchmark's trace
                        rec["every_proof_equals_zk_prove"] = bool(same)
                    bcs = []
                    try:
                        for t_ in range(2):
                            bc2 = zk.BatchContext(log_n, log_b, lbt, device=local_rank)
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
                    if result.get("chain"):
                        pk_ = result["per_kernel"]
                        ops_ = sum(pk_[k_]["ops"] for k_ in ("merkle_leaf", "merkle_inner") if k_ in pk_)
                        floor_ms = ops_ / 64 / SIMDS * min(c_["ns_per_instr"] for c_ in result["chain"]) * 1e-6
                        rec["hashing_floor_ms_per_proof_at_chain_rate"] = floor_ms
                        rec["frac_of_hashing_floor"] = floor_ms / rec["ms_per_proof"]
                        rec["two_batches_in_flight"]["frac_of_hashing_floor"] = floor_ms / rec["two_batches_in_flight"]["ms_per_proof"]
                    return rec

                # throughput mode at the metric's own domain: 2^batch_log proofs in lockstep (default 8: round 4's review), every proof
                # compared byte for byte with zk_prove of the same trace; and the same with batches twice as large, where the two
                # batches in flight come within a few per cent of what the device can hash (`larger_batches`: 16 x 2^24, 55 GB each)
                try:
                    rec_b = batched_leg(args.batch_log, True)
                    if args.batch_log + 1 + log_n + log_b <= 28:
                        try:
                            rec_b["larger_batches"] = batched_leg(args.batch_log + 1, False)
                        except zk.ZkError as e:
                            rec_b["larger_batches"] = {"error": str(e)}
                    result[f"batched_2e{log_n + log_b}"] = rec_b
                except zk.ZkError as e:
                    result[f"batched_2e{log_n + log_b}"] = {"error": str(e)}
        if args.in_flight > 1 and not args.no_secondary:
            # secondary figure: several independent proofs in flight on one GPU (one context, stream and
            # host thread each), so one proof's latency-bound tree tops overlap another's hashing
            import threading
            ctxs = [ctx] + [zk.Context(log_n, log_b, device=local_rank, hash=args.hash) for _ in range(args.in_flight - 1)]
            for c in ctxs[1:]:
                c.trace_upload(trace)
                c.prove()
            reps = min(args.steps, 20)
            def work(c):
                for _ in range(reps):
                    c.prove()
            barrier()
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
            [t.start() for t in th]
            [t.join() for t in th]
            barrier()
            dtp = time.perf_counter() - t0
            result["pipelined"] = {"proofs_in_flight": args.in_flight, "value": args.in_flight * N * reps / dtp,
                                   "unit": "field-elements/s", "ms_per_proof": dtp / (args.in_flight * reps) * 1e3}
            for c in ctxs[1:]:
                c.close()
        ctx.close()

    if rank == 0:
        emit_line(result)
        if result.get("exit_code"):
            sys.exit(result["exit_code"])
    if sharded_run:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
